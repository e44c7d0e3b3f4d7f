// se3.h -- small fixed-size double-precision linear algebra + SE(3) maps, usable from host C++ and from
// device code (the batched engine runs the Gauss-Newton solve and the pose update on the GPU so that no
// iteration needs a host round trip).  Restates, Eigen-free:
//   src/util_funcs.cpp:31-155 (logMap, expMap, expMapRot, forceOrthogonalisation), include/util_funcs.h:50-58 (skew),
//   Eigen LLT solve (visodo.cpp:1249), Eigen 6x6 inverse (visodo.cpp:1409), fixed 3x3 inverse.
// Matrices are row-major (Matrix3ft is Eigen::RowMajor double, include/types.h:490-496).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define RGBID_HD __host__ __device__ __forceinline__
#else
#define RGBID_HD inline
#endif

// Device compilations evaluate these functions WITHOUT fused multiply-add contraction (hipcc's default would fuse a * b + c wherever its optimiser
// sees one, differently in different kernels): every translation unit that inlines them -- the engine's per-lane kernels, the persistent
// Gauss-Newton level kernel -- then computes bit-identical poses, and the device agrees with the host build (g++ does not contract on x86-64).
// The pragma sits INSIDE each function body: including this header changes nothing about the floating-point state of the including file.
// Evaluate a function body WITHOUT fused multiply-add contraction, whatever the including translation unit's -ffp-contract / pragma state is, and
// without changing that state for the code that follows (the pragma is scoped to the compound statement it opens).
#ifndef RGBID_FP_STRICT
#if defined(__clang__)
#define RGBID_FP_STRICT _Pragma("clang fp contract(off)")
#else
#define RGBID_FP_STRICT
#endif
#endif

// Full unrolling of the fixed-size loops below: on the device every index becomes a compile-time constant and the small matrices live in registers (a
// dynamically indexed local array is scratch memory there: 300 dependent scratch round trips were most of the 12 us of a one-thread 6x6 solve).  Same
// operations in the same order: results unchanged.
#if defined(__clang__)
#define RGBID_UNROLL _Pragma("unroll")
#else
#define RGBID_UNROLL
#endif

namespace rgbid {
namespace se3 {

RGBID_HD void m3_copy(const double* A, double* B) { RGBID_FP_STRICT RGBID_UNROLL for (int i = 0; i < 9; ++i) B[i] = A[i]; }
RGBID_HD void m3_id(double* A) { RGBID_FP_STRICT RGBID_UNROLL for (int i = 0; i < 9; ++i) A[i] = 0.0; A[0] = A[4] = A[8] = 1.0; }
RGBID_HD void m3_mul(const double* A, const double* B, double* C) { RGBID_FP_STRICT
  double T[9];
  RGBID_UNROLL for (int i = 0; i < 3; ++i)
    RGBID_UNROLL for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  m3_copy(T, C);
}
RGBID_HD void m3_mulv(const double* A, const double* v, double* r) { RGBID_FP_STRICT
  double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
  double t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
  double t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  r[0] = t0; r[1] = t1; r[2] = t2;
}
RGBID_HD void m3_T(const double* A, double* T) { RGBID_FP_STRICT
  double t[9] = {A[0], A[3], A[6], A[1], A[4], A[7], A[2], A[5], A[8]};
  m3_copy(t, T);
}
// cofactor inverse (what Eigen's fixed-size 3x3 inverse() computes)
RGBID_HD void m3_inv(const double* A, double* I) { RGBID_FP_STRICT
  double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  double id = 1.0 / (A[0] * c00 + A[1] * c01 + A[2] * c02);
  double t[9];
  t[0] = c00 * id; t[1] = (A[2] * A[7] - A[1] * A[8]) * id; t[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  t[3] = c01 * id; t[4] = (A[0] * A[8] - A[2] * A[6]) * id; t[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  t[6] = c02 * id; t[7] = (A[1] * A[6] - A[0] * A[7]) * id; t[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  m3_copy(t, I);
}
RGBID_HD void skew(const double* w, double* S) { RGBID_FP_STRICT
  S[0] = 0; S[1] = -w[2]; S[2] = w[1];
  S[3] = w[2]; S[4] = 0; S[5] = -w[0];
  S[6] = -w[1]; S[7] = w[0]; S[8] = 0;
}

// forceOrthogonalisation (util_funcs.cpp:150-155): U V^T of the SVD = the orthogonal polar factor
// R = M (M^T M)^(-1/2); the symmetric 3x3 inverse square root comes from cyclic Jacobi rotations.
// Fast path for what this library actually feeds it -- a rotation up to rounding (exp-map output, products of rotations): ONE Newton step of the polar iteration
// X <- (X + X^-T) / 2 squares the distance to U V^T (1e-16 -> 1e-32), so when the step moves M by less than 2^-40 its result IS the polar factor to the last bit
// or two (~95 instructions instead of ~700: the per-lane solve kernels run on one thread).  Anything farther from a rotation takes the general path below.
RGBID_HD void force_orthogonal(const double* M, double* R) { RGBID_FP_STRICT
  {
    double Xi[9], N[9], moved = 0.0;
    m3_inv(M, Xi);
    RGBID_UNROLL for (int i = 0; i < 3; ++i)
      RGBID_UNROLL for (int j = 0; j < 3; ++j) {
        const double n = 0.5 * (M[i * 3 + j] + Xi[j * 3 + i]);
        const double d = fabs(n - M[i * 3 + j]);
        N[i * 3 + j] = n;
        moved = d > moved ? d : moved;   // a NaN never raises `moved`: it travels into R and is caught by the caller's has_nan
      }
    if (moved < 0x1p-40) { m3_copy(N, R); return; }
  }
  double S[9], V[9], Mt[9];
  m3_T(M, Mt);
  m3_mul(Mt, M, S);
  m3_id(V);
  for (int sweep = 0; sweep < 12; ++sweep) {
    // converged once the off-diagonal mass is below the rounding level of the diagonal: a further rotation by an angle of that
    // size cannot change a double (the input is a rotation up to rounding, so this is normally reached after one or two sweeps)
    double off = fabs(S[1]) + fabs(S[2]) + fabs(S[5]);
    if (off <= 1e-19 * (fabs(S[0]) + fabs(S[4]) + fabs(S[8]))) break;
    RGBID_UNROLL for (int p = 0; p < 2; ++p)
      RGBID_UNROLL for (int q = p + 1; q < 3; ++q) {
        double apq = S[p * 3 + q];
        if (fabs(apq) < 1e-300) continue;
        double theta = (S[q * 3 + q] - S[p * 3 + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        RGBID_UNROLL for (int k = 0; k < 3; ++k) {  // S <- S J
          double skp = S[k * 3 + p], skq = S[k * 3 + q];
          S[k * 3 + p] = c * skp - s * skq;
          S[k * 3 + q] = s * skp + c * skq;
        }
        RGBID_UNROLL for (int k = 0; k < 3; ++k) {  // S <- J^T S
          double spk = S[p * 3 + k], sqk = S[q * 3 + k];
          S[p * 3 + k] = c * spk - s * sqk;
          S[q * 3 + k] = s * spk + c * sqk;
        }
        RGBID_UNROLL for (int k = 0; k < 3; ++k) {  // V <- V J
          double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq;
          V[k * 3 + q] = s * vkp + c * vkq;
        }
      }
  }
  double D[9], Vt[9], T[9];
  RGBID_UNROLL for (int i = 0; i < 3; ++i)
    RGBID_UNROLL for (int j = 0; j < 3; ++j) D[i * 3 + j] = V[i * 3 + j] / sqrt(S[j * 3 + j]);  // V diag(d^-1/2)
  m3_T(V, Vt);
  m3_mul(D, Vt, T);  // (M^T M)^(-1/2)
  m3_mul(M, T, R);
}

// expMapRot util_funcs.cpp:124-148
RGBID_HD void expmap_rot(const double* w, double* R) { RGBID_FP_STRICT
  double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double O[9], O2[9], Rr[9];
  skew(w, O);
  m3_mul(O, O, O2);
  double a, b;
  if (theta < 0.00001) { a = 1.0; b = 0.5; }
  else { a = sin(theta) / theta; b = (1 - cos(theta)) / (theta * theta); }
  m3_id(Rr);
  RGBID_UNROLL for (int i = 0; i < 9; ++i) Rr[i] += a * O[i] + b * O2[i];
  force_orthogonal(Rr, R);
}

// expMap util_funcs.cpp:86-122
RGBID_HD void expmap(const double* w, const double* v, double* R, double* t) { RGBID_FP_STRICT
  double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double O[9], O2[9], Rr[9], Q[9];
  skew(w, O);
  m3_mul(O, O, O2);
  double a, b, qa, qb;
  if (theta < 0.00001) { a = 1.0; b = 0.5; qa = 0.5; qb = 1.0 / 6.0; }
  else {
    a = sin(theta) / theta; b = (1 - cos(theta)) / (theta * theta);
    qa = b; qb = (1 - (sin(theta) / theta)) / (theta * theta);
  }
  m3_id(Rr); m3_id(Q);
  RGBID_UNROLL for (int i = 0; i < 9; ++i) { Rr[i] += a * O[i] + b * O2[i]; Q[i] += qa * O[i] + qb * O2[i]; }
  force_orthogonal(Rr, R);
  m3_mulv(Q, v, t);
}

// logMap util_funcs.cpp:31-83 -> twist = (v, omega)
RGBID_HD void logmap(const double* M, const double* trans, double* twist) { RGBID_FP_STRICT
  double R[9];
  force_orthogonal(M, R);
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = acos(c), theta2 = theta * theta, th_by_sinth;
  if (s < 1e-5) th_by_sinth = 1.0 + (1.0 / 6.0) * theta2 + (7.0 / 360.0) * theta2 * theta2;
  else th_by_sinth = theta / s;
  double vth = th_by_sinth / 2.0;
  rx *= vth; ry *= vth; rz *= vth;
  double om[3] = {rx, ry, rz}, O[9], O2[9], Q[9], Qi[9];
  skew(om, O);
  m3_mul(O, O, O2);
  double th = sqrt(rx * rx + ry * ry + rz * rz);
  m3_id(Q);
  if (th < 0.00001) { RGBID_UNROLL for (int i = 0; i < 9; ++i) Q[i] += 0.5 * O[i] + (1.0 / 6.0) * O2[i]; }
  else {
    double qa = (1 - cos(theta)) / (theta * theta), qb = (1 - (sin(theta) / theta)) / (theta * theta);
    RGBID_UNROLL for (int i = 0; i < 9; ++i) Q[i] += qa * O[i] + qb * O2[i];
  }
  m3_inv(Q, Qi);
  double v[3];
  m3_mulv(Qi, trans, v);
  twist[0] = v[0]; twist[1] = v[1]; twist[2] = v[2];
  twist[3] = rx; twist[4] = ry; twist[5] = rz;
}

// A.llt().solve(b): Cholesky without pivoting; a non-PD matrix propagates NaN as Eigen's does (sqrt of a negative).  NOT operation for operation Eigen's
// LLT: Eigen divides every column entry and every substituted element by the pivot; here ONE reciprocal per pivot is taken and multiplied in by its column
// and by both substitutions -- 6 double-precision divisions instead of 27 (each ~12 dependent instructions on the one thread the device solve runs on).
// A product with the rounded reciprocal is within 1.5 ulp of the quotient (0.5 reciprocal + 0.5 product, relative), so x agrees with a dividing solve to
// a few ulp times the condition number: tests/test_cpu_host.py compares both on matrices up to cond 1e12 (host and device share this body, so both
// tracker modes and the oracle's double-precision solve stay within the pose bar by > 4 orders of magnitude).
RGBID_HD void llt_solve6(const double* A, const double* b, double* x) { RGBID_FP_STRICT
  double L[36], iL[6];
  RGBID_UNROLL for (int i = 0; i < 36; ++i) L[i] = 0.0;
  RGBID_UNROLL for (int j = 0; j < 6; ++j) {
    double d = A[j * 6 + j];
    RGBID_UNROLL for (int k = 0; k < j; ++k) d -= L[j * 6 + k] * L[j * 6 + k];
    double ljj = sqrt(d);
    L[j * 6 + j] = ljj;
    const double inv = 1.0 / ljj;
    iL[j] = inv;
    RGBID_UNROLL for (int i = j + 1; i < 6; ++i) {
      double s = A[i * 6 + j];
      RGBID_UNROLL for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = s * inv;
    }
  }
  double y[6];
  RGBID_UNROLL for (int i = 0; i < 6; ++i) {
    double s = b[i];
    RGBID_UNROLL for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k];
    y[i] = s * iL[i];
  }
  RGBID_UNROLL for (int i = 5; i >= 0; --i) {
    double s = y[i];
    RGBID_UNROLL for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
    x[i] = s * iL[i];
  }
}

// general 6x6 inverse, Gauss-Jordan with partial pivoting (Eigen: PartialPivLU)
RGBID_HD void inverse6(const double* A, double* Ainv) { RGBID_FP_STRICT
  double M[6][12];
  RGBID_UNROLL for (int i = 0; i < 6; ++i)
    RGBID_UNROLL for (int j = 0; j < 6; ++j) { M[i][j] = A[i * 6 + j]; M[i][6 + j] = (i == j) ? 1.0 : 0.0; }
  RGBID_UNROLL for (int c = 0; c < 6; ++c) {
    // partial pivoting without a dynamic row index: the candidate rows are compared in the reference order (first strict maximum wins), then the pivot
    // row is brought up by selects -- the same values in the same places as "swap rows c and p".  The pivot row is then scaled by ONE reciprocal
    // (<= 1.5 ulp per entry from a dividing elimination; Eigen's PartialPivLU divides): same pivots, last-bit differences in the inverse
    int p = c;
    double best = fabs(M[c][c]);
    RGBID_UNROLL for (int r = c + 1; r < 6; ++r) { const double v = fabs(M[r][c]); const bool g = v > best; p = g ? r : p; best = g ? v : best; }
    RGBID_UNROLL for (int r = c + 1; r < 6; ++r) {
      const bool sw = p == r;
      RGBID_UNROLL for (int j = 0; j < 12; ++j) { const double t = M[c][j]; M[c][j] = sw ? M[r][j] : t; M[r][j] = sw ? t : M[r][j]; }
    }
    const double ipiv = 1.0 / M[c][c];   // one division per pivot row instead of twelve
    RGBID_UNROLL for (int j = 0; j < 12; ++j) M[c][j] *= ipiv;
    RGBID_UNROLL for (int r = 0; r < 6; ++r) if (r != c) {
      double f = M[r][c];
      if (f != 0.0) RGBID_UNROLL for (int j = 0; j < 12; ++j) M[r][j] -= f * M[c][j];
    }
  }
  RGBID_UNROLL for (int i = 0; i < 6; ++i) RGBID_UNROLL for (int j = 0; j < 6; ++j) Ainv[i * 6 + j] = M[i][6 + j];
}

RGBID_HD void m6_zero(double* A) { RGBID_FP_STRICT RGBID_UNROLL for (int i = 0; i < 36; ++i) A[i] = 0.0; }
RGBID_HD void m6_set_block(double* A, int r0, int c0, const double* B, double scale) { RGBID_FP_STRICT
  RGBID_UNROLL for (int i = 0; i < 3; ++i) RGBID_UNROLL for (int j = 0; j < 3; ++j) A[(r0 + i) * 6 + c0 + j] = scale * B[i * 3 + j];
}
// out += J C J^T
RGBID_HD void m6_JCJt_add(const double* J, const double* C, double* out) { RGBID_FP_STRICT
  double T[36];
  RGBID_UNROLL for (int i = 0; i < 6; ++i) RGBID_UNROLL for (int j = 0; j < 6; ++j) {
    double s = 0; RGBID_UNROLL for (int k = 0; k < 6; ++k) s += J[i * 6 + k] * C[k * 6 + j];
    T[i * 6 + j] = s;
  }
  RGBID_UNROLL for (int i = 0; i < 6; ++i) RGBID_UNROLL for (int j = 0; j < 6; ++j) {
    double s = 0; RGBID_UNROLL for (int k = 0; k < 6; ++k) s += T[i * 6 + k] * J[j * 6 + k];
    out[i * 6 + j] += s;
  }
}

// K R K^-1 and K t in float, as the host does with Eigen float matrices (visodo.cpp:1108-1114): (K*Rf)*Kinv
RGBID_HD void project_trafo(float fx, float fy, float cx, float cy, const double* R, const double* tv, float* Rp, float* tp) { RGBID_FP_STRICT
  float K[9] = {fx, 0.f, cx, 0.f, fy, cy, 0.f, 0.f, 1.f};
  float Ki[9] = {1.f / fx, 0.f, -cx / fx, 0.f, 1.f / fy, -cy / fy, 0.f, 0.f, 1.f};
  float Rf[9], T[9];
  RGBID_UNROLL for (int i = 0; i < 9; ++i) Rf[i] = (float)R[i];
  RGBID_UNROLL for (int i = 0; i < 3; ++i) RGBID_UNROLL for (int j = 0; j < 3; ++j)
    T[i * 3 + j] = K[i * 3] * Rf[j] + K[i * 3 + 1] * Rf[3 + j] + K[i * 3 + 2] * Rf[6 + j];
  RGBID_UNROLL for (int i = 0; i < 3; ++i) RGBID_UNROLL for (int j = 0; j < 3; ++j)
    Rp[i * 3 + j] = T[i * 3] * Ki[j] + T[i * 3 + 1] * Ki[3 + j] + T[i * 3 + 2] * Ki[6 + j];
  float tf[3] = {(float)tv[0], (float)tv[1], (float)tv[2]};
  RGBID_UNROLL for (int i = 0; i < 3; ++i) tp[i] = K[i * 3] * tf[0] + K[i * 3 + 1] * tf[1] + K[i * 3 + 2] * tf[2];
}

RGBID_HD bool has_nan(const double* R, const double* t) { RGBID_FP_STRICT
  RGBID_UNROLL for (int i = 0; i < 9; ++i) if (R[i] != R[i]) return true;
  RGBID_UNROLL for (int i = 0; i < 3; ++i) if (t[i] != t[i]) return true;
  return false;
}

}  // namespace se3
}  // namespace rgbid

