/*
 * rgbid_oracle_tracker.c -- CPU restatement of the host driver (VisodoTracker) of the reference:
 * SE(3) helpers (src/util_funcs.cpp:31-155), the Gauss-Newton loop (src/visodo.cpp:944-1479),
 * keyframe logic and fusion (src/visodo.cpp:826-893,1481-1764,1967-2247).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see rgbid_oracle.h).
 *
 * Eigen (LLT, inverse, JacobiSVD) is an un-vendored, version-unpinned dependency of the reference
 * (CMakeLists.txt:44); its operations are restated here from their definitions in double:
 *   A.llt().solve(b)            -> Cholesky LL^T forward/back substitution
 *   M.inverse() (3x3)           -> cofactor inverse;  (6x6 dynamic) -> partial-pivot Gauss-Jordan
 *   JacobiSVD U*V^T             -> orthogonal polar factor (Newton iteration X <- (X + X^-T)/2)
 */
#include "rgbid_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ 3x3 helpers (double, row-major) */
static void m3_mul(const double A[9], const double B[9], double C[9]) {
  double T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  memcpy(C, T, sizeof(T));
}
static void m3_mulv(const double A[9], const double v[3], double r[3]) {
  double t[3];
  for (int i = 0; i < 3; ++i) t[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
  memcpy(r, t, sizeof(t));
}
static void m3_T(const double A[9], double T[9]) {
  double t[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[i * 3 + j] = A[j * 3 + i];
  memcpy(T, t, sizeof(t));
}
static void m3_inv(const double A[9], double I[9]) {
  /* cofactor inverse (Eigen's fixed-size 3x3 inverse is cofactor based) */
  double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  double id = 1.0 / det;
  double t[9];
  t[0] = c00 * id; t[1] = (A[2] * A[7] - A[1] * A[8]) * id; t[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  t[3] = c01 * id; t[4] = (A[0] * A[8] - A[2] * A[6]) * id; t[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  t[6] = c02 * id; t[7] = (A[1] * A[6] - A[0] * A[7]) * id; t[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  memcpy(I, t, sizeof(t));
}
static void m3_id(double A[9]) { memset(A, 0, 9 * sizeof(double)); A[0] = A[4] = A[8] = 1.0; }
static void skew3(const double w[3], double S[9]) {
  /* include/util_funcs.h:50-58 */
  S[0] = 0; S[1] = -w[2]; S[2] = w[1];
  S[3] = w[2]; S[4] = 0; S[5] = -w[0];
  S[6] = -w[1]; S[7] = w[0]; S[8] = 0;
}

void orc_force_orthogonal(const double M[9], double R[9]) {
  /* forceOrthogonalisation util_funcs.cpp:150-155: U*V^T of the SVD == orthogonal polar factor */
  double X[9];
  memcpy(X, M, sizeof(X));
  for (int it = 0; it < 60; ++it) {
    double Xi[9], XiT[9], N[9];
    m3_inv(X, Xi);
    m3_T(Xi, XiT);
    double d = 0;
    for (int i = 0; i < 9; ++i) { N[i] = 0.5 * (X[i] + XiT[i]); d += fabs(N[i] - X[i]); }
    memcpy(X, N, sizeof(X));
    if (d < 1e-17) break;
  }
  memcpy(R, X, sizeof(X));
}

void orc_expmap_rot(const double w[3], double R[9]) {
  /* expMapRot util_funcs.cpp:124-148 */
  double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double O[9], O2[9], Rr[9];
  skew3(w, O);
  m3_mul(O, O, O2);
  double a, b;
  if (theta < 0.00001) { a = 1.0; b = 0.5; }
  else { a = sin(theta) / theta; b = (1 - cos(theta)) / (theta * theta); }
  m3_id(Rr);
  for (int i = 0; i < 9; ++i) Rr[i] += a * O[i] + b * O2[i];
  orc_force_orthogonal(Rr, R);
}

void orc_expmap(const double w[3], const double v[3], double R[9], double t[3]) {
  /* expMap util_funcs.cpp:86-122 */
  double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double O[9], O2[9], Rr[9], Q[9];
  skew3(w, O);
  m3_mul(O, O, O2);
  double a, b, qa, qb;
  if (theta < 0.00001) { a = 1.0; b = 0.5; qa = 0.5; qb = 1.0 / 6.0; }
  else {
    a = sin(theta) / theta; b = (1 - cos(theta)) / (theta * theta);
    qa = (1 - cos(theta)) / (theta * theta); qb = (1 - (sin(theta) / theta)) / (theta * theta);
  }
  m3_id(Rr); m3_id(Q);
  for (int i = 0; i < 9; ++i) { Rr[i] += a * O[i] + b * O2[i]; Q[i] += qa * O[i] + qb * O2[i]; }
  orc_force_orthogonal(Rr, R);
  m3_mulv(Q, v, t);
}

void orc_logmap(const double M[9], const double trans[3], double twist[6]) {
  /* logMap util_funcs.cpp:31-83 */
  double R[9];
  orc_force_orthogonal(M, R);
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = acos(c), theta2 = theta * theta, th_by_sinth;
  if (s < 1e-5) th_by_sinth = 1.0 + (1.0 / 6.0) * theta2 + (7.0 / 360.0) * theta2 * theta2;
  else th_by_sinth = theta / s;
  double vth = th_by_sinth / 2.0;
  rx *= vth; ry *= vth; rz *= vth;
  double om[3] = { rx, ry, rz }, O[9], O2[9], Q[9], Qi[9];
  skew3(om, O);
  m3_mul(O, O, O2);
  double th = sqrt(rx * rx + ry * ry + rz * rz);
  m3_id(Q);
  if (th < 0.00001) for (int i = 0; i < 9; ++i) Q[i] += 0.5 * O[i] + (1.0 / 6.0) * O2[i];
  else for (int i = 0; i < 9; ++i) Q[i] += (1 - cos(theta)) / (theta * theta) * O[i] + (1 - (sin(theta) / theta)) / (theta * theta) * O2[i];
  m3_inv(Q, Qi);
  double v[3];
  m3_mulv(Qi, trans, v);
  twist[0] = v[0]; twist[1] = v[1]; twist[2] = v[2];
  twist[3] = rx; twist[4] = ry; twist[5] = rz;
}

int orc_llt_solve6(const double A[36], const double b[6], double x[6]) {
  /* Eigen: A.llt().solve(b) (visodo.cpp:1249).  A non-PD matrix yields NaNs, as in Eigen. */
  double L[36];
  memset(L, 0, sizeof(L));
  for (int j = 0; j < 6; ++j) {
    double d = A[j * 6 + j];
    for (int k = 0; k < j; ++k) d -= L[j * 6 + k] * L[j * 6 + k];
    double ljj = sqrt(d);
    L[j * 6 + j] = ljj;
    for (int i = j + 1; i < 6; ++i) {
      double s = A[i * 6 + j];
      for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = s / ljj;
    }
  }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k];
    y[i] = s / L[i * 6 + i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
    x[i] = s / L[i * 6 + i];
  }
  for (int i = 0; i < 6; ++i) if (x[i] != x[i]) return 0;
  return 1;
}

int orc_inverse6(const double A[36], double Ainv[36]) {
  /* Eigen MatrixXd::inverse() (visodo.cpp:1409) -> PartialPivLU; restated as Gauss-Jordan with partial pivoting */
  double M[6][12];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) { M[i][j] = A[i * 6 + j]; M[i][6 + j] = (i == j); }
  for (int c = 0; c < 6; ++c) {
    int p = c;
    for (int r = c + 1; r < 6; ++r) if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
    if (p != c) for (int j = 0; j < 12; ++j) { double t = M[c][j]; M[c][j] = M[p][j]; M[p][j] = t; }
    double piv = M[c][c];
    for (int j = 0; j < 12; ++j) M[c][j] /= piv;
    for (int r = 0; r < 6; ++r) if (r != c) {
      double f = M[r][c];
      if (f != 0.0) for (int j = 0; j < 12; ++j) M[r][j] -= f * M[c][j];
    }
  }
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) Ainv[i * 6 + j] = M[i][6 + j];
  return 1;
}

/* 6x6 helpers for covariance propagation */
static void m6_zero(double A[36]) { memset(A, 0, 36 * sizeof(double)); }
static void m6_set_block(double A[36], int r0, int c0, const double B[9], double scale) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[(r0 + i) * 6 + c0 + j] = scale * B[i * 3 + j];
}
static void m6_ABAt_add(const double J[36], const double C[36], double out[36]) {
  /* out += J*C*J^T */
  double T[36];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) {
    double s = 0; for (int k = 0; k < 6; ++k) s += J[i * 6 + k] * C[k * 6 + j];
    T[i * 6 + j] = s;
  }
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) {
    double s = 0; for (int k = 0; k < 6; ++k) s += T[i * 6 + k] * J[j * 6 + k];
    out[i * 6 + j] += s;
  }
}

/* ------------------------------------------------------------------ tracker state */
#define ORC_MAX_LEVELS 8

typedef struct { int id; double R[9], t[3]; } sink_pose;
typedef struct { int ini, end, type; double R[9], t[3], cov[36]; } sink_constraint;
typedef struct { int id; double R[9], t[3], Rrel[9], trel[3]; uint8_t* mask; uint8_t* colors; float* iD; float* normals; } sink_keyframe;

struct orc_tracker {
  orc_tracker_config c;
  /* streams to the back-end (f-3) */
  sink_pose* sp; int n_sp, cap_sp;
  sink_constraint* sc; int n_sc, cap_sc;
  sink_keyframe* sk; int n_sk, cap_sk;
  int custom_registration; orc_custom_calib cc;   /* prepareImagesCustomCalibration instead of prepareImages */
  int global_time, lost;
  int odoKF_count, integrKF_count, last_odoKF_index, last_integrKF_index;
  double velocity[3], omega[3];
  float delta_t;
  /* pose history */
  int n_poses, cap_poses;
  double* rmats; double* tvecs;           /* 9 / 3 per pose */
  int n_odo, cap_odo;
  double* odo_rmats; double* odo_tvecs; double* odo_cov; /* 9 / 3 / 36 */
  double last_est_R[9], last_est_t[3];
  double delta_R[9], delta_t_[3], delta_cov[36];
  double odoKF_R[9], odoKF_t[3], integrKF_R[9], integrKF_t[3];
  double o2i_last_R[9], o2i_last_t[3], o2i_last_cov[36];
  double o2i_next_R[9], o2i_next_t[3], o2i_next_cov[36];
  /* images */
  float* iD_curr[ORC_MAX_LEVELS]; float* I_curr[ORC_MAX_LEVELS];
  float* iD_kf[ORC_MAX_LEVELS]; float* I_kf[ORC_MAX_LEVELS];
  float* iD_kf_f[ORC_MAX_LEVELS]; float* I_kf_f[ORC_MAX_LEVELS];
  float* gxI[ORC_MAX_LEVELS]; float* gyI[ORC_MAX_LEVELS]; float* gxD[ORC_MAX_LEVELS]; float* gyD[ORC_MAX_LEVELS];
  float* gxI_c[ORC_MAX_LEVELS]; float* gyI_c[ORC_MAX_LEVELS]; float* gxD_c[ORC_MAX_LEVELS]; float* gyD_c[ORC_MAX_LEVELS];
  float* wiD[ORC_MAX_LEVELS]; float* wI[ORC_MAX_LEVELS];
  float* res_I; float* res_D;
  float *r_curr, *g_curr, *b_curr;
  float *iD_integr, *iD_integr_raw, *w_integr, *warped_iD_integr, *warped_w;
  float *vmap, *nmap, *gxD_integr, *gyD_integr;
  uint8_t* colors_integr; uint8_t* overlap_mask;
  orc_frame_info info;
  int force_odo, force_integr;   /* test hook (orc_tracker_force_kf_decisions): -1 = decide naturally, 0 / 1 = imposed for the next frame */
};

void orc_tracker_default_config(orc_tracker_config* c) {
  /* ctor defaults include/visodo.h:54-68 + src/visodo.cpp:65 + shipped ini config_data/visodoRGBDconfig.ini (App. A.11)
   * + factory calibration config_data/calibration_factory.ini */
  memset(c, 0, sizeof(*c));
  c->rows = 480; c->cols = 640; c->levels = 3;
  c->iters[0] = 10; c->iters[1] = 5; c->iters[2] = 3;
  c->mestimator = ORC_STUDENT; c->motion_model = ORC_CONSTANT_VELOCITY; c->sigma_estimator = ORC_SIGMA_PDF;
  c->weighting = ORC_INDEPENDENT; c->warping = ORC_PYR_FIRST;
  c->max_odoKF_count = 9999999; c->finest_level = 0; c->termination = ORC_ALL_ITERS;
  c->visratio_odo = 0.9f; c->image_filtering = ORC_NO_FILTERS; c->visratio_integr = 0.7f;
  c->max_integrKF_count = 9999999; c->nsamples = 10000;
  c->fx = 525.f; c->fy = 525.f; c->cx = 319.5f; c->cy = 239.5f; c->factor_depth = 1.f;
  c->interp_mode = ORC_INTERP_TEX8;
  c->delta_t = 0.03333f;
}

static float* falloc(size_t n) { return (float*)calloc(n, sizeof(float)); }

orc_tracker* orc_tracker_create(const orc_tracker_config* c) {
  orc_tracker* t = (orc_tracker*)calloc(1, sizeof(orc_tracker));
  t->c = *c;
  size_t n0 = (size_t)c->rows * c->cols;
  for (int l = 0; l < c->levels; ++l) {
    size_t n = (size_t)(c->rows >> l) * (c->cols >> l);
    t->iD_curr[l] = falloc(n); t->I_curr[l] = falloc(n);
    t->iD_kf[l] = falloc(n); t->I_kf[l] = falloc(n);
    t->iD_kf_f[l] = falloc(n); t->I_kf_f[l] = falloc(n);
    t->gxI[l] = falloc(n); t->gyI[l] = falloc(n); t->gxD[l] = falloc(n); t->gyD[l] = falloc(n);
    t->gxI_c[l] = falloc(n); t->gyI_c[l] = falloc(n); t->gxD_c[l] = falloc(n); t->gyD_c[l] = falloc(n);
    t->wiD[l] = falloc(n); t->wI[l] = falloc(n);
  }
  t->res_I = falloc(n0); t->res_D = falloc(n0);
  t->r_curr = falloc(n0); t->g_curr = falloc(n0); t->b_curr = falloc(n0);
  t->iD_integr = falloc(n0); t->iD_integr_raw = falloc(n0); t->w_integr = falloc(n0);
  t->warped_iD_integr = falloc(n0); t->warped_w = falloc(n0);
  t->vmap = falloc(3 * n0); t->nmap = falloc(3 * n0); t->gxD_integr = falloc(n0); t->gyD_integr = falloc(n0);
  t->colors_integr = (uint8_t*)calloc(3 * n0, 1); t->overlap_mask = (uint8_t*)calloc(n0, 1);
  t->cap_poses = 1024; t->rmats = (double*)malloc(9 * sizeof(double) * t->cap_poses); t->tvecs = (double*)malloc(3 * sizeof(double) * t->cap_poses);
  t->cap_odo = 1024; t->odo_rmats = (double*)malloc(9 * sizeof(double) * t->cap_odo); t->odo_tvecs = (double*)malloc(3 * sizeof(double) * t->cap_odo);
  t->odo_cov = (double*)malloc(36 * sizeof(double) * t->cap_odo);
  /* reset() visodo.cpp:519-553 */
  t->global_time = 0; t->lost = 0;
  t->force_odo = t->force_integr = -1;
  m3_id(t->rmats); memset(t->tvecs, 0, 3 * sizeof(double)); t->n_poses = 1;
  m3_id(t->last_est_R); memset(t->last_est_t, 0, sizeof(t->last_est_t));
  return t;
}

void orc_tracker_destroy(orc_tracker* t) {
  if (!t) return;
  for (int l = 0; l < t->c.levels; ++l) {
    free(t->iD_curr[l]); free(t->I_curr[l]); free(t->iD_kf[l]); free(t->I_kf[l]); free(t->iD_kf_f[l]); free(t->I_kf_f[l]);
    free(t->gxI[l]); free(t->gyI[l]); free(t->gxD[l]); free(t->gyD[l]);
    free(t->gxI_c[l]); free(t->gyI_c[l]); free(t->gxD_c[l]); free(t->gyD_c[l]); free(t->wiD[l]); free(t->wI[l]);
  }
  free(t->res_I); free(t->res_D); free(t->r_curr); free(t->g_curr); free(t->b_curr);
  free(t->iD_integr); free(t->iD_integr_raw); free(t->w_integr); free(t->warped_iD_integr); free(t->warped_w);
  free(t->vmap); free(t->nmap); free(t->gxD_integr); free(t->gyD_integr); free(t->colors_integr); free(t->overlap_mask);
  free(t->rmats); free(t->tvecs); free(t->odo_rmats); free(t->odo_tvecs); free(t->odo_cov);
  for (int i = 0; i < t->n_sk; ++i) { free(t->sk[i].mask); free(t->sk[i].colors); free(t->sk[i].iD); free(t->sk[i].normals); }
  free(t->sp); free(t->sc); free(t->sk);
  free(t);
}

static void push_pose(orc_tracker* t, const double R[9], const double tv[3]) {
  if (t->n_poses == t->cap_poses) {
    t->cap_poses *= 2;
    t->rmats = (double*)realloc(t->rmats, 9 * sizeof(double) * t->cap_poses);
    t->tvecs = (double*)realloc(t->tvecs, 3 * sizeof(double) * t->cap_poses);
  }
  memcpy(t->rmats + 9 * t->n_poses, R, 9 * sizeof(double));
  memcpy(t->tvecs + 3 * t->n_poses, tv, 3 * sizeof(double));
  t->n_poses++;
}
static void push_odo(orc_tracker* t, const double R[9], const double tv[3], const double cov[36]) {
  if (t->n_odo == t->cap_odo) {
    t->cap_odo *= 2;
    t->odo_rmats = (double*)realloc(t->odo_rmats, 9 * sizeof(double) * t->cap_odo);
    t->odo_tvecs = (double*)realloc(t->odo_tvecs, 3 * sizeof(double) * t->cap_odo);
    t->odo_cov = (double*)realloc(t->odo_cov, 36 * sizeof(double) * t->cap_odo);
  }
  memcpy(t->odo_rmats + 9 * t->n_odo, R, 9 * sizeof(double));
  memcpy(t->odo_tvecs + 3 * t->n_odo, tv, 3 * sizeof(double));
  memcpy(t->odo_cov + 36 * t->n_odo, cov, 36 * sizeof(double));
  t->n_odo++;
}

/* K*R*K^-1 and K*t in float, as the host does with Eigen float matrices (visodo.cpp:1108-1114).
 * Eigen's expression order: (K*Rf)*Kinv. */
static void project_trafo(const orc_tracker_config* c, int level, const double R[9], const double tv[3], float Rp[9], float tp[3]) {
  int div = 1 << level; /* getCalibMatrix visodo.cpp:1885-1900 */
  float fx = c->fx / div, fy = c->fy / div, cx = c->cx / div, cy = c->cy / div;
  float K[9] = { fx, 0.f, cx, 0.f, fy, cy, 0.f, 0.f, 1.f };
  float Ki[9] = { 1.f / fx, 0.f, -cx / fx, 0.f, 1.f / fy, -cy / fy, 0.f, 0.f, 1.f };
  float Rf[9], T[9];
  for (int i = 0; i < 9; ++i) Rf[i] = (float)R[i];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
    T[i * 3 + j] = K[i * 3] * Rf[j] + K[i * 3 + 1] * Rf[3 + j] + K[i * 3 + 2] * Rf[6 + j];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
    Rp[i * 3 + j] = T[i * 3] * Ki[j] + T[i * 3 + 1] * Ki[3 + j] + T[i * 3 + 2] * Ki[6 + j];
  float tf[3] = { (float)tv[0], (float)tv[1], (float)tv[2] };
  for (int i = 0; i < 3; ++i) tp[i] = K[i * 3] * tf[0] + K[i * 3 + 1] * tf[1] + K[i * 3 + 2] * tf[2];
}

static orc_intr cfg_intr(const orc_tracker_config* c) { orc_intr k = { c->fx, c->fy, c->cx, c->cy }; return k; }

static void m3f_mul(const float A[9], const float B[9], float C[9]) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
static void m3f_inv(const float A[9], float I[9]) {
  /* Eigen Matrix3f::inverse(): cofactors / determinant, in float */
  float c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  float det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  float id = 1.f / det;
  I[0] = c00 * id; I[1] = (A[2] * A[7] - A[1] * A[8]) * id; I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  I[3] = c01 * id; I[4] = (A[0] * A[8] - A[2] * A[6]) * id; I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  I[6] = c02 * id; I[7] = (A[1] * A[6] - A[0] * A[7]) * id; I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

static void prepare_images(orc_tracker* t, const uint16_t* depth, const uint8_t* rgb) {
  /* prepareImages visodo.cpp:760-773 */
  const orc_tracker_config* c = &t->c;
  if (t->custom_registration) {
    /* prepareImagesCustomCalibration visodo.cpp:775-824 */
    const orc_custom_calib* cc = &t->cc;
    size_t n = (size_t)c->rows * c->cols;
    float* I_dist = falloc(n); float* iD_dist = falloc(n); float* iD_pre = falloc(n); float* inter = falloc(9 * n);
    orc_intensity(rgb, I_dist, c->rows, c->cols);
    orc_decompose_rgb(rgb, t->r_curr, t->g_curr, t->b_curr, c->rows, c->cols);
    orc_depth2invdepth(depth, iD_dist, c->rows, c->cols, c->factor_depth);
    orc_undistort_intensity(I_dist, c->rows, c->cols, cc->rgb, c->interp_mode, t->I_curr[0]);
    orc_undistort_depthinv(iD_dist, c->rows, c->cols, cc->depth, cc->dist, NULL, iD_pre);
    /* Kd*dRc*Kc^-1, Kd*t_dc, inverse: all Eigen float (:792-801); Kc from getCalibMatrix(0) = tracker fx.. (no distortion) */
    float Kc[9] = { c->fx, 0, c->cx, 0, c->fy, c->cy, 0, 0, 1 }, Kd[9] = { cc->depth.fx, 0, cc->depth.cx, 0, cc->depth.fy, cc->depth.cy, 0, 0, 1 };
    float Kci[9], T[9], dRc_proj[9], cRd_proj[9], t_proj[3];
    m3f_inv(Kc, Kci);
    m3f_mul(Kd, cc->dRc, T); m3f_mul(T, Kci, dRc_proj);
    for (int i = 0; i < 3; ++i) t_proj[i] = Kd[i * 3] * cc->t_dc[0] + Kd[i * 3 + 1] * cc->t_dc[1] + Kd[i * 3 + 2] * cc->t_dc[2];
    m3f_inv(dRc_proj, cRd_proj);
    orc_register_depthinv(iD_pre, c->rows, c->cols, 3 * c->rows, 3 * c->cols, dRc_proj, t_proj, cRd_proj, inter, t->iD_curr[0]);
    free(I_dist); free(iD_dist); free(iD_pre); free(inter);
    for (int i = 1; i < c->levels; ++i) {
      orc_pyr_down(t->I_curr[i - 1], c->rows >> (i - 1), c->cols >> (i - 1), t->I_curr[i]);
      orc_pyr_down(t->iD_curr[i - 1], c->rows >> (i - 1), c->cols >> (i - 1), t->iD_curr[i]);
    }
    return;
  }
  orc_intensity(rgb, t->I_curr[0], c->rows, c->cols);
  orc_decompose_rgb(rgb, t->r_curr, t->g_curr, t->b_curr, c->rows, c->cols);
  orc_depth2invdepth(depth, t->iD_curr[0], c->rows, c->cols, c->factor_depth);
  for (int i = 1; i < c->levels; ++i) {
    orc_pyr_down(t->I_curr[i - 1], c->rows >> (i - 1), c->cols >> (i - 1), t->I_curr[i]);
    orc_pyr_down(t->iD_curr[i - 1], c->rows >> (i - 1), c->cols >> (i - 1), t->iD_curr[i]);
  }
}

static void save_odo_keyframe(orc_tracker* t) {
  /* saveCurrentImagesAsOdoKeyframes visodo.cpp:826-878 */
  const orc_tracker_config* c = &t->c;
  const float sigma_int_ref = 3.f, sigma_depthinv_ref = 0.0025f;
  for (int i = 0; i < c->levels; ++i) {
    size_t n = (size_t)(c->rows >> i) * (c->cols >> i);
    memcpy(t->iD_kf[i], t->iD_curr[i], n * sizeof(float));
    memcpy(t->I_kf[i], t->I_curr[i], n * sizeof(float));
  }
  orc_bilateral(t->iD_kf[0], c->rows, c->cols, 2.f * sigma_depthinv_ref, t->iD_kf_f[0]);
  orc_bilateral(t->I_kf[0], c->rows, c->cols, sigma_int_ref, t->I_kf_f[0]);
  orc_gradient(t->I_kf_f[0], c->rows, c->cols, t->gxI_c[0], t->gyI_c[0]);
  orc_gradient(t->iD_kf_f[0], c->rows, c->cols, t->gxD_c[0], t->gyD_c[0]);
  for (int i = 1; i < c->levels; ++i) {
    int r = c->rows >> i, cc = c->cols >> i;
    orc_pyr_down(t->iD_kf_f[i - 1], c->rows >> (i - 1), c->cols >> (i - 1), t->iD_kf_f[i]);
    orc_pyr_down(t->I_kf_f[i - 1], c->rows >> (i - 1), c->cols >> (i - 1), t->I_kf_f[i]);
    orc_gradient(t->I_kf_f[i], r, cc, t->gxI_c[i], t->gyI_c[i]);
    orc_gradient(t->iD_kf_f[i], r, cc, t->gxD_c[i], t->gyD_c[i]);
  }
  for (int i = 0; i < c->levels; ++i) {
    int r = c->rows >> i, cc = c->cols >> i;
    size_t n = (size_t)r * cc;
    if (c->image_filtering == ORC_FILTER_GRADS) {
      memcpy(t->gxI[i], t->gxI_c[i], n * sizeof(float)); memcpy(t->gyI[i], t->gyI_c[i], n * sizeof(float));
      memcpy(t->gxD[i], t->gxD_c[i], n * sizeof(float)); memcpy(t->gyD[i], t->gyD_c[i], n * sizeof(float));
    } else {
      orc_gradient(t->I_kf[i], r, cc, t->gxI[i], t->gyI[i]);
      orc_gradient(t->iD_kf[i], r, cc, t->gxD[i], t->gyD[i]);
    }
  }
}

static void save_integr_keyframe(orc_tracker* t, const uint8_t* rgb) {
  /* saveCurrentImagesAsIntegrationKeyframes visodo.cpp:880-893 */
  const orc_tracker_config* c = &t->c;
  size_t n0 = (size_t)c->rows * c->cols;
  memcpy(t->iD_integr, t->iD_curr[0], n0 * sizeof(float));
  memcpy(t->iD_integr_raw, t->iD_curr[0], n0 * sizeof(float));
  memcpy(t->colors_integr, rgb, 3 * n0);
  orc_init_weight(t->iD_curr[0], t->w_integr, c->rows, c->cols);
  orc_vmap(t->iD_integr, c->rows, c->cols, cfg_intr(c), t->vmap);
  orc_gradient(t->iD_integr, c->rows, c->cols, t->gxD_integr, t->gyD_integr);
  orc_nmap_gradients(t->iD_integr, t->gxD_integr, t->gyD_integr, c->rows, c->cols, cfg_intr(c), t->nmap);
}

static float compute_covisibility(orc_tracker* t, const double R_AtoB[9], const double t_AtoB[3], const float* iD_A, const float* iD_B) {
  /* computeCovisibility visodo.cpp:1481-1514 */
  const orc_tracker_config* c = &t->c;
  float Rab[9], tab[3], Rba[9], tba[3];
  project_trafo(c, 0, R_AtoB, t_AtoB, Rab, tab);
  double Ri[9], ti[3];
  m3_inv(R_AtoB, Ri);
  /* translation_BtoA_f = -K*Rinv.cast<float>()*t.cast<float>() : float arithmetic */
  {
    float fx = c->fx, fy = c->fy, cx = c->cx, cy = c->cy;
    float K[9] = { fx, 0.f, cx, 0.f, fy, cy, 0.f, 0.f, 1.f };
    float Rf[9], T[9], tf[3] = { (float)t_AtoB[0], (float)t_AtoB[1], (float)t_AtoB[2] };
    for (int i = 0; i < 9; ++i) Rf[i] = (float)Ri[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
      T[i * 3 + j] = -K[i * 3] * Rf[j] + -K[i * 3 + 1] * Rf[3 + j] + -K[i * 3 + 2] * Rf[6 + j];
    for (int i = 0; i < 3; ++i) tba[i] = T[i * 3] * tf[0] + T[i * 3 + 1] * tf[1] + T[i * 3 + 2] * tf[2];
    (void)ti;
  }
  {
    double zero[3] = { 0, 0, 0 };
    float dummy[3];
    project_trafo(c, 0, Ri, zero, Rba, dummy);
  }
  float vis_BtoA = orc_visibility_ratio(iD_B, iD_A, c->rows, c->cols, Rab, tab, NULL, NULL, NULL);
  float vis_AtoB = orc_visibility_ratio(iD_A, iD_B, c->rows, c->cols, Rba, tba, NULL, NULL, NULL);
  return fminf(vis_AtoB, vis_BtoA);
}

static void reset_odometry_keyframe(orc_tracker* t) {
  /* resetOdometryKeyframe visodo.cpp:1541-1575 */
  t->odoKF_count = 0;
  double J[36], tn[3], S[9];
  m6_zero(J);
  m6_set_block(J, 0, 0, t->o2i_next_R, 1.0);
  m6_set_block(J, 3, 3, t->o2i_next_R, 1.0);
  m3_mulv(t->o2i_next_R, t->delta_t_, tn);
  skew3(tn, S);
  m6_set_block(J, 0, 3, S, 1.0);
  m6_ABAt_add(J, t->delta_cov, t->o2i_next_cov);
  for (int i = 0; i < 3; ++i) t->o2i_next_t[i] = tn[i] + t->o2i_next_t[i];
  m3_mul(t->o2i_next_R, t->delta_R, t->o2i_next_R);
  t->last_odoKF_index = t->global_time;
  memcpy(t->odoKF_R, t->last_est_R, sizeof(t->odoKF_R));
  memcpy(t->odoKF_t, t->last_est_t, sizeof(t->odoKF_t));
  m3_id(t->delta_R); memset(t->delta_t_, 0, sizeof(t->delta_t_)); m6_zero(t->delta_cov);
}

static void sink_push_pose(orc_tracker* t, int id, const double R[9], const double tv[3]) {
  if (t->n_sp == t->cap_sp) { t->cap_sp = t->cap_sp ? 2 * t->cap_sp : 64; t->sp = (sink_pose*)realloc(t->sp, sizeof(sink_pose) * t->cap_sp); }
  sink_pose* p = &t->sp[t->n_sp++];
  p->id = id; memcpy(p->R, R, sizeof(p->R)); memcpy(p->t, tv, sizeof(p->t));
}
static void sink_push_constraint(orc_tracker* t, int ini, int end, int type, const double R[9], const double tv[3], const double cov[36]) {
  if (t->n_sc == t->cap_sc) { t->cap_sc = t->cap_sc ? 2 * t->cap_sc : 64; t->sc = (sink_constraint*)realloc(t->sc, sizeof(sink_constraint) * t->cap_sc); }
  sink_constraint* q = &t->sc[t->n_sc++];
  q->ini = ini; q->end = end; q->type = type;
  memcpy(q->R, R, sizeof(q->R)); memcpy(q->t, tv, sizeof(q->t)); memcpy(q->cov, cov, sizeof(q->cov));
}

static void reset_integration_keyframe(orc_tracker* t) {
  /* resetIntegrationKeyframe visodo.cpp:1577-1672 */
  t->integrKF_count = 0;
  double J[36], tn[3], S[9];
  m6_zero(J);
  m6_set_block(J, 0, 0, t->o2i_next_R, 1.0);
  m6_set_block(J, 3, 3, t->o2i_next_R, 1.0);
  m3_mulv(t->o2i_next_R, t->delta_t_, tn);
  skew3(tn, S);
  m6_set_block(J, 0, 3, S, 1.0);
  m6_ABAt_add(J, t->delta_cov, t->o2i_next_cov);
  for (int i = 0; i < 3; ++i) t->o2i_next_t[i] = tn[i] + t->o2i_next_t[i];
  m3_mul(t->o2i_next_R, t->delta_R, t->o2i_next_R);
  {
    /* :1612-1652  T{k-1,k} = inv(T{odo,k-1}) T{odo,k}, its covariance, the Keyframe record (4 downloads) and the SEQ_KF constraint */
    double lastT[9], Rkf[9], d[3], tkf[3], Jn[36], Jl[36], Sk[9], SR[9], ckf[36];
    m3_T(t->o2i_last_R, lastT);
    m3_mul(lastT, t->o2i_next_R, Rkf);
    for (int i = 0; i < 3; ++i) d[i] = t->o2i_next_t[i] - t->o2i_last_t[i];
    m3_mulv(lastT, d, tkf);
    m6_zero(Jn); m6_set_block(Jn, 0, 0, lastT, 1.0); m6_set_block(Jn, 3, 3, lastT, 1.0);
    m6_zero(Jl); m6_set_block(Jl, 0, 0, lastT, -1.0); m6_set_block(Jl, 3, 3, lastT, -1.0);
    skew3(tkf, Sk); m3_mul(Sk, lastT, SR); m6_set_block(Jl, 0, 3, SR, 1.0);
    m6_zero(ckf);
    m6_ABAt_add(Jl, t->o2i_last_cov, ckf);
    m6_ABAt_add(Jn, t->o2i_next_cov, ckf);
    if (t->n_sk == t->cap_sk) { t->cap_sk = t->cap_sk ? 2 * t->cap_sk : 16; t->sk = (sink_keyframe*)realloc(t->sk, sizeof(sink_keyframe) * t->cap_sk); }
    sink_keyframe* k = &t->sk[t->n_sk++];
    size_t n0 = (size_t)t->c.rows * t->c.cols;
    k->id = t->last_integrKF_index;
    memcpy(k->R, t->integrKF_R, sizeof(k->R)); memcpy(k->t, t->integrKF_t, sizeof(k->t));
    memcpy(k->Rrel, Rkf, sizeof(k->Rrel)); memcpy(k->trel, tkf, sizeof(k->trel));
    k->mask = (uint8_t*)malloc(n0); memcpy(k->mask, t->overlap_mask, n0);
    k->colors = (uint8_t*)malloc(3 * n0); memcpy(k->colors, t->colors_integr, 3 * n0);
    k->iD = (float*)malloc(n0 * sizeof(float)); memcpy(k->iD, t->iD_integr, n0 * sizeof(float));
    k->normals = (float*)malloc(3 * n0 * sizeof(float)); memcpy(k->normals, t->nmap, 3 * n0 * sizeof(float));
    sink_push_constraint(t, t->last_integrKF_index, t->global_time, ORC_SEQ_KF, Rkf, tkf, ckf);
  }
  t->last_integrKF_index = t->global_time;
  memcpy(t->integrKF_R, t->last_est_R, sizeof(t->integrKF_R));
  memcpy(t->integrKF_t, t->last_est_t, sizeof(t->integrKF_t));
  memcpy(t->o2i_last_R, t->delta_R, sizeof(t->o2i_last_R));
  memcpy(t->o2i_last_t, t->delta_t_, sizeof(t->o2i_last_t));
  memcpy(t->o2i_last_cov, t->delta_cov, sizeof(t->o2i_last_cov));
  m3_id(t->o2i_next_R); memset(t->o2i_next_t, 0, sizeof(t->o2i_next_t)); m6_zero(t->o2i_next_cov);
}

static void integrate_into_keyframe(orc_tracker* t, const float* iD_src, const double dR[9], const double dt[3]) {
  /* integrateImagesIntoKeyframes visodo.cpp:1674-1764: here K R K^-1 is formed in DOUBLE, inverted, then cast */
  const orc_tracker_config* c = &t->c;
  double K[9] = { (double)c->fx, 0, (double)c->cx, 0, (double)c->fy, (double)c->cy, 0, 0, 1 };
  double Ki[9], T[9], Rp[9], tp[3], Rpi[9], tpi[3];
  m3_inv(K, Ki);
  m3_mul(K, dR, T); m3_mul(T, Ki, Rp);
  m3_mulv(K, dt, tp);
  m3_inv(Rp, Rpi);
  m3_mulv(Rpi, tp, tpi);
  float Rf[9], tf[3];
  for (int i = 0; i < 9; ++i) Rf[i] = (float)Rpi[i];
  for (int i = 0; i < 3; ++i) tf[i] = (float)(-tpi[i]);
  orc_warp_invdepth_weighted(iD_src, t->iD_integr, c->rows, c->cols, Rf, tf, t->warped_iD_integr, t->warped_w);
  orc_integrate_warped(t->warped_iD_integr, t->warped_w, t->iD_integr, t->w_integr, c->rows, c->cols);
  orc_vmap(t->iD_integr, c->rows, c->cols, cfg_intr(c), t->vmap);
  orc_gradient(t->iD_integr, c->rows, c->cols, t->gxD_integr, t->gyD_integr);
  orc_nmap_gradients(t->iD_integr, t->gxD_integr, t->gyD_integr, c->rows, c->cols, cfg_intr(c), t->nmap);
}

/* one warp pair at `level` with the KF-relative pose (R,t): visodo.cpp:1066-1126 (PYR_FIRST) */
static void warp_level(orc_tracker* t, int level, const double R[9], const double tv[3]) {
  const orc_tracker_config* c = &t->c;
  double Ri[9], ti[3];
  m3_inv(R, Ri);
  m3_mulv(Ri, tv, ti);
  ti[0] = -ti[0]; ti[1] = -ti[1]; ti[2] = -ti[2];
  float Rp[9], tp[3];
  int r = c->rows >> level, cc = c->cols >> level;
  project_trafo(c, level, Ri, ti, Rp, tp);
  orc_warp_invdepth(t->iD_curr[level], t->iD_kf[level], r, cc, Rp, tp, t->wiD[level]);
  orc_warp_intensity(t->I_curr[level], t->wiD[level], r, cc, Rp, tp, c->interp_mode, t->wI[level]);
}

static int has_nan3(const double R[9], const double tv[3]) {
  for (int i = 0; i < 9; ++i) if (R[i] != R[i]) return 1;
  for (int i = 0; i < 3; ++i) if (tv[i] != tv[i]) return 1;
  return 0;
}

/* estimateVisualOdometry visodo.cpp:944-1479.  R,t in/out (KF-relative pose of previous / current frame) */
static int estimate_visual_odometry(orc_tracker* t, double R_io[9], double t_io[3], double cov[36]) {
  const orc_tracker_config* c = &t->c;
  const float sigma_int_ref = 5.f, sigma_depthinv_ref = 0.0025f;
  double prevR[9], prevt[3], curR[9], curt[3];
  memcpy(prevR, R_io, sizeof(prevR)); memcpy(prevt, t_io, sizeof(prevt));
  if ((t->global_time > 1) && (c->motion_model == ORC_CONSTANT_VELOCITY) && (!t->lost)) {
    /* :1016-1027 */
    double vt[3], wt[3], dR[9], dt[3], tmp[3];
    for (int i = 0; i < 3; ++i) { vt[i] = t->velocity[i] * t->delta_t; wt[i] = t->omega[i] * t->delta_t; }
    orc_expmap(wt, vt, dR, dt);
    m3_mulv(prevR, dt, tmp);
    for (int i = 0; i < 3; ++i) curt[i] = tmp[i] + prevt[i];
    m3_mul(prevR, dR, curR);
  } else {
    memcpy(curR, prevR, sizeof(curR)); memcpy(curt, prevt, sizeof(curt));
  }
  float sigma_int = 40.f, sigma_depthinv = 5.f, bias_int = 0.f, bias_depthinv = 0.f, nu_int = 5.f, nu_depthinv = 5.f;
  double A[36], b[6];
  float RMSE = 9999.f, RMSE_prev = 9999.f;   /* :1041 (declared once, carried across levels) */
  double last_inc_inv[9], last_tinc[3];
  m3_id(last_inc_inv); memset(last_tinc, 0, sizeof(last_tinc));
  int iters0 = c->iters[0];
  for (int level = c->levels - 1; level >= c->finest_level; --level) {
    int iter_num = level == 0 ? iters0 : c->iters[level];
    int r = c->rows >> level, cc = c->cols >> level;
    for (int iter = 0; iter < iter_num; ++iter) {
      if (c->warping == ORC_WARP_FIRST) {
        /* :1078-1105: warp at level 0, then pyrDown the warped maps down to `level` */
        warp_level(t, 0, curR, curt);
        for (int i = 1; i < level + 1; ++i) {
          orc_pyr_down(t->wI[i - 1], c->rows >> (i - 1), c->cols >> (i - 1), t->wI[i]);
          orc_pyr_down(t->wiD[i - 1], c->rows >> (i - 1), c->cols >> (i - 1), t->wiD[i]);
        }
      } else {
        warp_level(t, level, curR, curt);
      }
      if ((c->termination == ORC_CHI_SQUARED) && (iter != 0)) {
        /* :1134-1164: full-lattice residuals of the LEVEL-0 warped maps (fresh only with WARP_FIRST; with PYR_FIRST the reference
         * reads whatever the last level-0 warp left there), RMSE must not grow, else undo the last increment and end the level */
        float chi_square, chi_test, ndof;
        int nI = orc_error_lattice(t->wI[0], t->I_kf[0], c->rows, c->cols, 9999999, t->res_I, NULL, NULL, NULL);
        orc_error_lattice(t->wiD[0], t->iD_kf[0], c->rows, c->cols, 9999999, t->res_D, NULL, NULL, NULL);
        orc_chi_square(t->res_I, t->res_D, nI, sigma_int_ref, sigma_depthinv_ref, c->mestimator, &chi_square, &chi_test, &ndof);
        RMSE = sqrtf(chi_square) / sqrtf(ndof);
        if (iter != 1) {
          { float m_ = fabsf(RMSE - RMSE_prev) / fmaxf(RMSE_prev, 1e-30f); if (m_ < t->info.chi_stop_margin_frame) t->info.chi_stop_margin_frame = m_; }   /* test diagnostic */
          if (RMSE > RMSE_prev) {
            ++t->info.chi_stops_frame;
            double d[3], tmp2[3];
            for (int i = 0; i < 3; ++i) d[i] = curt[i] - last_tinc[i];
            m3_mulv(last_inc_inv, d, tmp2);
            memcpy(curt, tmp2, sizeof(tmp2));
            m3_mul(last_inc_inv, curR, curR);
            break;
          }
        }
        RMSE_prev = RMSE;
      }
      sigma_int = 5.f; sigma_depthinv = 0.0025f; bias_int = 0.f; bias_depthinv = 0.f; nu_int = 5.f; nu_depthinv = 5.f; /* :1168-1173 */
      if (c->sigma_estimator == ORC_SIGMA_PDF) {
        int nI = orc_error_lattice(t->wI[level], t->I_kf[level], r, cc, c->nsamples, t->res_I, NULL, NULL, NULL);
        int nD = orc_error_lattice(t->wiD[level], t->iD_kf[level], r, cc, c->nsamples, t->res_D, NULL, NULL, NULL);
        orc_sigma_nu_student_margin(t->res_I, nI, &bias_int, &sigma_int, &nu_int, c->mestimator, &t->info.sigma_stop_margin_int);
        orc_sigma_nu_student_margin(t->res_D, nD, &bias_depthinv, &sigma_depthinv, &nu_depthinv, c->mestimator, &t->info.sigma_stop_margin_depthinv);
        if (t->info.sigma_stop_margin_int < t->info.sigma_stop_margin_frame) t->info.sigma_stop_margin_frame = t->info.sigma_stop_margin_int;
        if (t->info.sigma_stop_margin_depthinv < t->info.sigma_stop_margin_frame) t->info.sigma_stop_margin_frame = t->info.sigma_stop_margin_depthinv;
        nu_int = fmaxf(nu_int, nu_depthinv); /* :1186 */
      } else if (c->sigma_estimator == ORC_SIGMA_CONS) {
        sigma_int = (float)exp(log((double)sigma_int_ref)); sigma_depthinv = (float)exp(log((double)sigma_depthinv_ref));
      }
      orc_build_system(t->iD_kf[level], t->I_kf[level], t->gxD[level], t->gyD[level], t->gxI[level], t->gyI[level],
                       t->wiD[level], t->wI[level], r, cc, 1, c->mestimator, c->weighting,
                       sigma_depthinv, sigma_int, bias_depthinv, bias_int, nu_depthinv, nu_int,
                       orc_intr_level(cfg_intr(c), level), A, b);
      double x[6];
      orc_llt_solve6(A, b, x);
      /* :1252-1263 */
      double inc_inv[9], inc[9], tinc[3], tmp[3];
      orc_expmap_rot(x + 3, inc_inv);
      m3_inv(inc_inv, inc);
      m3_mulv(inc, x, tinc);
      tinc[0] = -tinc[0]; tinc[1] = -tinc[1]; tinc[2] = -tinc[2];
      m3_mulv(inc, curt, tmp);
      for (int i = 0; i < 3; ++i) curt[i] = tmp[i] + tinc[i];
      m3_mul(inc, curR, curR);
      memcpy(last_inc_inv, inc_inv, sizeof(last_inc_inv)); memcpy(last_tinc, tinc, sizeof(last_tinc));
      if (has_nan3(curR, curt)) { /* :1265-1274 */
        memcpy(R_io, prevR, sizeof(prevR)); memcpy(t_io, prevt, sizeof(prevt));
        m6_zero(cov); for (int i = 0; i < 6; ++i) cov[i * 7] = 100.0;
        return 0;
      }
    }
  }
  t->info.sigma_int = sigma_int; t->info.sigma_depthinv = sigma_depthinv; t->info.nu_int = nu_int; t->info.nu_depthinv = nu_depthinv;
  t->info.bias_int = bias_int; t->info.bias_depthinv = bias_depthinv;
  {
    /* covariance pass :1283-1417 */
    int fl = c->finest_level;
    int r = c->rows >> fl, cc = c->cols >> fl;
    warp_level(t, fl, curR, curt);
    sigma_int = (float)exp(log((double)sigma_int_ref)); sigma_depthinv = (float)exp(log((double)sigma_depthinv_ref));
    orc_build_system(t->iD_kf[fl], t->I_kf[fl], t->gxD_c[fl], t->gyD_c[fl], t->gxI_c[fl], t->gyI_c[fl],
                     t->wiD[fl], t->wI[fl], r, cc, 0, ORC_STUDENT, c->weighting,
                     sigma_depthinv, sigma_int, 0.f, 0.f, 5.f, 5.f, orc_intr_level(cfg_intr(c), fl), A, b);
    memcpy(R_io, curR, sizeof(curR)); memcpy(t_io, curt, sizeof(curt));
    orc_inverse6(A, cov);
  }
  {
    /* :1459-1468 */
    double pT[9], dR[9], d[3], dt[3], twist[6];
    m3_T(prevR, pT);
    m3_mul(pT, curR, dR);
    for (int i = 0; i < 3; ++i) d[i] = curt[i] - prevt[i];
    m3_mulv(pT, d, dt);
    orc_logmap(dR, dt, twist);
    float inv_dt = 1.f / t->delta_t;
    for (int i = 0; i < 3; ++i) { t->velocity[i] = twist[i] * (double)inv_dt; t->omega[i] = twist[3 + i] * (double)inv_dt; }
  }
  return 1;
}

static int tracker_track(orc_tracker* t, const uint16_t* depth, const uint8_t* rgb);
int orc_tracker_track(orc_tracker* t, const uint16_t* depth, const uint8_t* rgb) {
#if defined(ORC_CUDA_NUMERICS) && defined(__SSE2__)
  /* --ftz=true for every fp32 operation of the frame: flush-to-zero + denormals-are-zero in this thread's MXCSR (the image loops of
   * this build are single-threaded); the host-side double arithmetic never comes near the subnormal range */
  unsigned int csr = __builtin_ia32_stmxcsr();
  __builtin_ia32_ldmxcsr(csr | 0x8040u);
  int r = tracker_track(t, depth, rgb);
  __builtin_ia32_ldmxcsr(csr);
  return r;
#else
  return tracker_track(t, depth, rgb);
#endif
}
static int tracker_track(orc_tracker* t, const uint16_t* depth, const uint8_t* rgb) {
  /* trackNewFrame visodo.cpp:1967-2247 */
  const orc_tracker_config* c = &t->c;
  t->delta_t = c->delta_t; /* computeInterframeTime :1929-1964 with compute_deltat_flag_ off */
  memset(&t->info, 0, sizeof(t->info));
  t->info.sigma_stop_margin_frame = 1e30f;
  t->info.chi_stop_margin_frame = 1e30f; t->info.chi_stops_frame = 0;
  const int force_odo = t->force_odo, force_integr = t->force_integr;   /* one-shot */
  t->force_odo = t->force_integr = -1;
  prepare_images(t, depth, rgb);
  if (t->global_time == 0) {
    ++t->global_time;
    t->odoKF_count = 0; t->last_odoKF_index = 0;
    memcpy(t->odoKF_R, t->rmats, 9 * sizeof(double)); memcpy(t->odoKF_t, t->tvecs, 3 * sizeof(double));
    t->integrKF_count = 0; t->last_integrKF_index = 0;
    m3_id(t->integrKF_R); memset(t->integrKF_t, 0, sizeof(t->integrKF_t));
    m3_id(t->delta_R); memset(t->delta_t_, 0, sizeof(t->delta_t_)); m6_zero(t->delta_cov);
    push_odo(t, t->delta_R, t->delta_t_, t->delta_cov);
    m3_id(t->o2i_last_R); memset(t->o2i_last_t, 0, sizeof(t->o2i_last_t)); m6_zero(t->o2i_last_cov);
    m3_id(t->o2i_next_R); memset(t->o2i_next_t, 0, sizeof(t->o2i_next_t)); m6_zero(t->o2i_next_cov);
    save_odo_keyframe(t);
    save_integr_keyframe(t, rgb);
    memset(t->overlap_mask, 0, (size_t)c->rows * c->cols);
    { double I3[9], z3[3] = { 0, 0, 0 }; m3_id(I3); sink_push_pose(t, 0, I3, z3); }  /* :2034-2041 */
    return 0;
  }
  double dR_prev[9], dt_prev[3], dcov_prev[36];
  memcpy(dR_prev, t->delta_R, sizeof(dR_prev)); memcpy(dt_prev, t->delta_t_, sizeof(dt_prev)); memcpy(dcov_prev, t->delta_cov, sizeof(dcov_prev));

  int ok = estimate_visual_odometry(t, t->delta_R, t->delta_t_, t->delta_cov);
  t->info.odometry_success = ok;
  if (!t->lost) {
    double tmp[3];
    m3_mulv(t->odoKF_R, t->delta_t_, tmp);
    for (int i = 0; i < 3; ++i) t->last_est_t[i] = t->odoKF_t[i] + tmp[i];
    m3_mul(t->odoKF_R, t->delta_R, t->last_est_R);
    push_pose(t, t->last_est_R, t->last_est_t);
    if (!ok) { /* :2066-2097 */
      t->lost = 1;
      {
        /* dummy odometry constraint (zero motion, covariance 100 I) + a pose that repeats the last one in the back-end's list */
        double I3[9], z3[3] = { 0, 0, 0 }, c100[36];
        m3_id(I3); m6_zero(c100);
        for (int i = 0; i < 6; ++i) c100[i * 7] = 100.0;
        sink_push_constraint(t, t->global_time - 1, t->global_time, ORC_SEQ_ODO, I3, z3, c100);
        sink_push_pose(t, t->global_time, t->sp[t->n_sp - 1].R, t->sp[t->n_sp - 1].t);
      }
      reset_odometry_keyframe(t);
      reset_integration_keyframe(t);
      save_odo_keyframe(t);
      save_integr_keyframe(t, rgb);
      ++t->global_time;
      t->info.lost = 1; t->info.odo_kf_switched = 1; t->info.integr_kf_switched = 1;
      return 0;
    }
  } else {
    if (ok) { /* :2103-2109 */
      t->lost = 0;
      double tmp[3];
      m3_mulv(t->odoKF_R, t->delta_t_, tmp);
      for (int i = 0; i < 3; ++i) t->last_est_t[i] = t->odoKF_t[i] + tmp[i];
      m3_mul(t->odoKF_R, t->delta_R, t->last_est_R);
      push_pose(t, t->last_est_R, t->last_est_t);
    } else {
      save_odo_keyframe(t);
      save_integr_keyframe(t, rgb);
      t->info.lost = 1;
      return 0;
    }
  }
  t->odoKF_count++; t->integrKF_count++;
  /* sequential constraint + covariance :2128-2152 */
  {
    double pT[9], Rseq[9], d[3], tseq[3], Jn[36], Jl[36], S[9], SR[9], cseq[36];
    m3_T(dR_prev, pT);
    m3_mul(pT, t->delta_R, Rseq);
    for (int i = 0; i < 3; ++i) d[i] = t->delta_t_[i] - dt_prev[i];
    m3_mulv(pT, d, tseq);
    m6_zero(Jn); m6_set_block(Jn, 0, 0, pT, 1.0); m6_set_block(Jn, 3, 3, pT, 1.0);
    m6_zero(Jl); m6_set_block(Jl, 0, 0, pT, -1.0); m6_set_block(Jl, 3, 3, pT, -1.0);
    skew3(tseq, S); m3_mul(S, pT, SR); m6_set_block(Jl, 0, 3, SR, 1.0);
    m6_zero(cseq);
    m6_ABAt_add(Jl, dcov_prev, cseq);
    m6_ABAt_add(Jn, t->delta_cov, cseq);
    push_odo(t, Rseq, tseq, cseq);
    sink_push_constraint(t, t->global_time - 1, t->global_time, ORC_SEQ_ODO, Rseq, tseq, cseq);   /* :2154-2165 */
    {
      /* the new pose continues the BACK-END's last pose (which a pose-graph optimisation may have moved), not last_estimated_* (:2161-2162) */
      const sink_pose* b = &t->sp[t->n_sp - 1];
      double Rn[9], tn2[3];
      m3_mul(b->R, Rseq, Rn);
      m3_mulv(b->R, tseq, tn2);
      for (int i = 0; i < 3; ++i) tn2[i] += b->t[i];
      sink_push_pose(t, t->global_time, Rn, tn2);
    }
  }
  memcpy(t->info.delta_R, t->delta_R, sizeof(t->delta_R)); memcpy(t->info.delta_t, t->delta_t_, sizeof(t->delta_t_));
  memcpy(t->info.delta_cov, t->delta_cov, sizeof(t->delta_cov));
  /* odometry keyframe switch :2172-2180 */
  float vis_odo = compute_covisibility(t, t->delta_R, t->delta_t_, t->iD_kf[0], t->iD_curr[0]);
  t->info.visratio_odo = vis_odo;
  int sw_odo = (t->odoKF_count >= c->max_odoKF_count) || (vis_odo < c->visratio_odo);
  t->info.odo_kf_natural = sw_odo;
  if (force_odo >= 0) sw_odo = force_odo;
  if (sw_odo) {
    reset_odometry_keyframe(t);
    save_odo_keyframe(t);
    t->info.odo_kf_switched = 1;
  }
  /* integration keyframe :2182-2211 */
  double iRi[9], dIR[9], d[3], dIt[3];
  m3_inv(t->integrKF_R, iRi);
  m3_mul(iRi, t->last_est_R, dIR);
  for (int i = 0; i < 3; ++i) d[i] = t->last_est_t[i] - t->integrKF_t[i];
  m3_mulv(iRi, d, dIt);
  float vis_int = compute_covisibility(t, dIR, dIt, t->iD_integr_raw, t->iD_curr[0]);
  t->info.visratio_integr = vis_int;
  int sw_int = (t->integrKF_count >= c->max_integrKF_count) || (vis_int < c->visratio_integr);
  t->info.integr_kf_natural = sw_int;
  if (force_integr >= 0) sw_int = force_integr;
  if (sw_int) {
    reset_integration_keyframe(t);
    { /* computeOverlapping visodo.cpp:1517-1539 */
      float Rab[9], tab[3];
      project_trafo(c, 0, dIR, dIt, Rab, tab);
      orc_visibility_ratio(t->iD_curr[0], t->iD_integr_raw, c->rows, c->cols, Rab, tab, t->overlap_mask, NULL, NULL);
    }
    save_integr_keyframe(t, rgb);
    t->info.integr_kf_switched = 1;
  } else {
    integrate_into_keyframe(t, t->iD_curr[0], dIR, dIt);
  }
  ++t->global_time;
  return 1;
}

int orc_tracker_num_poses(const orc_tracker* t) { return t->n_poses; }
void orc_tracker_get_pose(const orc_tracker* t, int i, double R[9], double tv[3]) {
  memcpy(R, t->rmats + 9 * i, 9 * sizeof(double)); memcpy(tv, t->tvecs + 3 * i, 3 * sizeof(double));
}
int orc_tracker_num_odo(const orc_tracker* t) { return t->n_odo; }
void orc_tracker_get_odo(const orc_tracker* t, int i, double R[9], double tv[3], double cov[36]) {
  memcpy(R, t->odo_rmats + 9 * i, 9 * sizeof(double)); memcpy(tv, t->odo_tvecs + 3 * i, 3 * sizeof(double));
  memcpy(cov, t->odo_cov + 36 * i, 36 * sizeof(double));
}
void orc_tracker_last_info(const orc_tracker* t, orc_frame_info* info) { *info = t->info; }
void orc_tracker_force_kf_decisions(orc_tracker* t, int odo_switch, int integr_switch) { t->force_odo = odo_switch; t->force_integr = integr_switch; }
int orc_tracker_num_sink_poses(const orc_tracker* t) { return t->n_sp; }
void orc_tracker_get_sink_pose(const orc_tracker* t, int i, int* id, double R[9], double tv[3]) {
  *id = t->sp[i].id; memcpy(R, t->sp[i].R, 72); memcpy(tv, t->sp[i].t, 24);
}
int orc_tracker_num_constraints(const orc_tracker* t) { return t->n_sc; }
void orc_tracker_get_constraint(const orc_tracker* t, int i, int* ini, int* end, int* type, double R[9], double tv[3], double cov[36]) {
  *ini = t->sc[i].ini; *end = t->sc[i].end; *type = t->sc[i].type;
  memcpy(R, t->sc[i].R, 72); memcpy(tv, t->sc[i].t, 24); memcpy(cov, t->sc[i].cov, 288);
}
int orc_tracker_num_keyframes(const orc_tracker* t) { return t->n_sk; }
void orc_tracker_get_keyframe(const orc_tracker* t, int i, int* id, double R[9], double tv[3], double R_rel[9], double t_rel[3],
                              const uint8_t** mask, const uint8_t** colors, const float** iD, const float** normals) {
  const sink_keyframe* k = &t->sk[i];
  *id = k->id; memcpy(R, k->R, 72); memcpy(tv, k->t, 24); memcpy(R_rel, k->Rrel, 72); memcpy(t_rel, k->trel, 24);
  *mask = k->mask; *colors = k->colors; *iD = k->iD; *normals = k->normals;
}
void orc_tracker_set_custom_calibration(orc_tracker* t, const orc_custom_calib* cc) { t->custom_registration = cc != NULL; if (cc) t->cc = *cc; }
const float* orc_tracker_cur_depthinv(const orc_tracker* t) { return t->iD_curr[0]; }
const float* orc_tracker_cur_intensity(const orc_tracker* t) { return t->I_curr[0]; }
const float* orc_tracker_kf_depthinv(const orc_tracker* t) { return t->iD_integr; }
const float* orc_tracker_kf_weight(const orc_tracker* t) { return t->w_integr; }
const float* orc_tracker_kf_normals(const orc_tracker* t) { return t->nmap; }
const float* orc_tracker_kf_vertices(const orc_tracker* t) { return t->vmap; }
const uint8_t* orc_tracker_kf_overlap_mask(const orc_tracker* t) { return t->overlap_mask; }

int orc_align_pair(const orc_tracker_config* c, const uint16_t* depth0, const uint8_t* rgb0,
                   const uint16_t* depth1, const uint8_t* rgb1, double R[9], double tv[3], double cov[36]) {
  orc_tracker* t = orc_tracker_create(c);
  prepare_images(t, depth0, rgb0);
  save_odo_keyframe(t);
  prepare_images(t, depth1, rgb1);
  t->global_time = 1; t->delta_t = c->delta_t;
  int ok = estimate_visual_odometry(t, R, tv, cov);
  orc_tracker_destroy(t);
  return ok;
}

int orc_keyframe_align(int rows, int cols, const float* depthinv_ini, const uint8_t* grey_ini, const float* depthinv_end,
                       const uint8_t* grey_end, orc_intr k, int interp_mode, double R[9], double tv[3], double cov[36]) {
  /* src/keyframe_align.cpp:115-357 */
  enum { L = 4 };
  const int iters[L] = { 5, 5, 3, 0 };
  float *iDa[L], *iDb[L], *Ia[L], *Ib[L], *gxD[L], *gyD[L], *gxI[L], *gyI[L], *wD[L], *wI[L];
  size_t n0 = (size_t)rows * cols;
  for (int l = 0; l < L; ++l) {
    size_t n = (size_t)(rows >> l) * (cols >> l);
    iDa[l] = falloc(n); iDb[l] = falloc(n); Ia[l] = falloc(n); Ib[l] = falloc(n);
    gxD[l] = falloc(n); gyD[l] = falloc(n); gxI[l] = falloc(n); gyI[l] = falloc(n); wD[l] = falloc(n); wI[l] = falloc(n);
  }
  float* resD = falloc(n0); float* resI = falloc(n0);
  memcpy(iDa[0], depthinv_ini, n0 * sizeof(float)); memcpy(iDb[0], depthinv_end, n0 * sizeof(float));
  for (size_t i = 0; i < n0; ++i) { Ia[0][i] = (float)grey_ini[i]; Ib[0][i] = (float)grey_end[i]; }
  for (int l = 1; l < L; ++l) {
    orc_pyr_down(iDa[l - 1], rows >> (l - 1), cols >> (l - 1), iDa[l]); orc_pyr_down(iDb[l - 1], rows >> (l - 1), cols >> (l - 1), iDb[l]);
    orc_pyr_down(Ia[l - 1], rows >> (l - 1), cols >> (l - 1), Ia[l]); orc_pyr_down(Ib[l - 1], rows >> (l - 1), cols >> (l - 1), Ib[l]);
  }
  for (int l = 0; l < L; ++l) {
    orc_gradient(iDa[l], rows >> l, cols >> l, gxD[l], gyD[l]);
    orc_gradient(Ia[l], rows >> l, cols >> l, gxI[l], gyI[l]);
  }
  double curR[9], curt[3], A[36], b[6];
  memcpy(curR, R, sizeof(curR)); memcpy(curt, tv, sizeof(curt));
  orc_tracker_config cfg;
  orc_tracker_default_config(&cfg);
  cfg.fx = k.fx; cfg.fy = k.fy; cfg.cx = k.cx; cfg.cy = k.cy;
  for (int level = L - 1; level >= 0; --level) {
    int r = rows >> level, cc = cols >> level;
    for (int it = 0; it < iters[level]; ++it) {
      double Ri[9], ti[3];
      m3_inv(curR, Ri); m3_mulv(Ri, curt, ti);
      ti[0] = -ti[0]; ti[1] = -ti[1]; ti[2] = -ti[2];
      float Rp[9], tp[3];
      project_trafo(&cfg, level, Ri, ti, Rp, tp);
      orc_warp_invdepth(iDb[level], iDa[level], r, cc, Rp, tp, wD[level]);
      orc_warp_intensity(Ib[level], iDa[level], r, cc, Rp, tp, interp_mode, wI[level]);
      int nD = orc_error_lattice(wD[level], iDa[level], r, cc, 19200, resD, NULL, NULL, NULL);
      int nI = orc_error_lattice(wI[level], Ia[level], r, cc, 19200, resI, NULL, NULL, NULL);
      float nu_d = 5.f, nu_i = 5.f;
      orc_nu_student(resD, nD, 0.f, 0.0025f, &nu_d);
      orc_nu_student(resI, nI, 0.f, 5.f, &nu_i);
      orc_build_system(iDa[level], Ia[level], gxD[level], gyD[level], gxI[level], gyI[level], wD[level], wI[level], r, cc, 1, ORC_STUDENT,
                       ORC_INDEPENDENT, 0.0025f, 5.f, 0.f, 0.f, nu_d, nu_d, orc_intr_level(k, level), A, b);
      double x[6], inc_inv[9], inc[9], tinc[3], tmp[3];
      orc_llt_solve6(A, b, x);
      orc_expmap_rot(x + 3, inc_inv);
      m3_inv(inc_inv, inc);
      m3_mulv(inc, x, tinc);
      m3_mulv(inc, curt, tmp);
      for (int i = 0; i < 3; ++i) curt[i] = tmp[i] - tinc[i];
      m3_mul(inc, curR, curR);
    }
  }
  memcpy(R, curR, sizeof(curR)); memcpy(tv, curt, sizeof(curt));
  orc_inverse6(A, cov);
  for (int l = 0; l < L; ++l) { free(iDa[l]); free(iDb[l]); free(Ia[l]); free(Ib[l]); free(gxD[l]); free(gyD[l]); free(gxI[l]); free(gyI[l]); free(wD[l]); free(wI[l]); }
  free(resD); free(resI);
  return 1;
}
