// kfalign.hip -- batched, device-resident KeyframeAlign (C-ABI in include/rgbid_kfalign.h).
//
// KeyframeAlign::alignKeyframes (src/keyframe_align.cpp:115-357) for `pairs` keyframe pairs in lock-step: the reference's call sequence -- pyramids of both
// keyframes (:155-162), Sobel gradients of the first (:166-174), per level {5,5,3,0} iterations of [warp of the second keyframe's inverse depth, warp of its
// intensity SAMPLED ON THE FIRST KEYFRAME'S INVERSE DEPTH (:239-242), residual lattice of >= 19 200 samples, nu by bisection at the fixed sigmas (0.0025, 5),
// normal equations with nu_depthinv for BOTH channels (:308), LLT solve, pre-multiplied exp-map update] -- as one launch sequence of the batched kernels of
// kernels.h (block id -> pair), with the double-precision pose algebra of the host loop in small per-pair kernels (se3.h, no contraction), so that nothing
// crosses to the host between the upload of the keyframes and the read-back of the poses.  Per pair: its own intrinsics (SysParams / WarpParams are per lane).
//
// What is NOT computed: the intensity channel's residual lattice and its nu -- the reference estimates nu_intensity and then passes nu_depthinv for both
// channels (:300-308), so that work has no consumer (two launches per iteration).
#include "../../include/rgbid_kfalign.h"
#include "ctx.h"
#include "kernels.h"
#include "../../include/rgbid/se3.h"

#include <cstring>
#include <new>
#include <vector>

// The per-lane scalar kernels hold 6x6 / 3x3 double matrices in registers: one wave per SIMD may take the whole register file (the default budget of 128
// VGPRs spilled 660 - 1 250 bytes per thread to scratch, and dependent scratch round trips were most of these kernels' run time)
#define RGBID_SCALAR_KERNEL __attribute__((amdgpu_waves_per_eu(1, 1)))

#pragma clang fp contract(off)   // the per-pair scalar kernels below: operation by operation, as the host loop of KeyframeAlign (g++, no contraction)

using namespace rgbid;

namespace {

constexpr int KFA_LEVELS = 4;                       // keyframe_align.h:50
constexpr int KFA_ITERS[KFA_LEVELS] = {5, 5, 3, 0}; // keyframe_align.cpp:44
constexpr int KFA_NSAMPLES = 19200;                 // :262-263

struct KfaState { double R[9], t[3], A[36]; float fx, fy, cx, cy; };

// dense u8 [B][rows][cols] -> pitched float maps: grey_image_.convertTo(CV_32F) (:122-129)
__global__ __launch_bounds__(256) void k_kfa_grey(const unsigned char* src, ImgB dst) {
  const int lane = blockIdx.z, y = blockIdx.y;
  const unsigned char* sp = src + ((size_t)lane * dst.rows + y) * dst.cols;
  float* dp = row_ptr<float>(dst, lane, y);
  for (int x = blockIdx.x * 256 + threadIdx.x; x < dst.cols; x += gridDim.x * 256) dp[x] = (float)sp[x];
}

// pose of a pair -> the projected inverse transform its next warps use (:208-231), with the pair's intrinsics at `level`
__device__ void kfa_set_warp(const KfaState& s, int level, WarpParams& wp) {
  double Ri[9], ti[3];
  se3::m3_inv(s.R, Ri);
  se3::m3_mulv(Ri, s.t, ti);
  ti[0] = -ti[0]; ti[1] = -ti[1]; ti[2] = -ti[2];
  const int div = 1 << level;
  se3::project_trafo(s.fx / div, s.fy / div, s.cx / div, s.cy / div, Ri, ti, wp.R, wp.t);
}
// the start values every iteration gives computeNuStudent (:265) and the constants of its normal equations at `level`
__device__ void kfa_set_iteration(const KfaState& s, int level, SysParams& p, SigmaIO& io) {
  const int div = 1 << level;
  p.fx = s.fx / div; p.fy = s.fy / div; p.cx = s.cx / div; p.cy = s.cy / div;   // cam_intrinsics(level_index)
  p.sigma_d = 0.0025f; p.sigma_i = 5.f; p.bias_d = 0.f; p.bias_i = 0.f; p.nu_d = 5.f; p.nu_i = 5.f;
  p.mestimator = RGBID_STUDENT; p.weighting = RGBID_INDEPENDENT; p.student_nu = 1; p.nu_i_max = 0;
  io.bias = 0.f; io.sigma = 0.0025f; io.nu = 5.f;
}

__global__ __launch_bounds__(64) RGBID_SCALAR_KERNEL void k_kfa_begin(KfaState* st, const double* R, const double* t, const float* K, WarpParams* wp, SysParams* sp, SigmaIO* io, int level, int B) {
  const int lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= B) return;
  KfaState& s = st[lane];
  for (int i = 0; i < 9; ++i) s.R[i] = R[lane * 9 + i];
  for (int i = 0; i < 3; ++i) s.t[i] = t[lane * 3 + i];
  for (int i = 0; i < 36; ++i) s.A[i] = 0.0;
  s.fx = K[lane * 4 + 0]; s.fy = K[lane * 4 + 1]; s.cx = K[lane * 4 + 2]; s.cy = K[lane * 4 + 3];
  kfa_set_warp(s, level, wp[lane]);
  kfa_set_iteration(s, level, sp[lane], io[lane]);
}

// nu_depthinv for both channels (:308)
__global__ void k_kfa_set_nu(const SigmaIO* io, SysParams* sp, int B) {
  const int lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane < B) { sp[lane].nu_d = io[lane].nu; sp[lane].nu_i = io[lane].nu; }
}

// one update of one pair: fixed-order reduction of its partial sums (the order of k_reduce_system, kernels_system.hip), LLT solve, exp-map, pre-multiplied
// pose update (:312-335), then what the next iteration needs (warp at next_level, start values)
// NT: 256 threads per pair, or ONE wave per pair once there are more pairs than compute units (the kernel keeps the whole register file per wave: a 256-thread
// workgroup occupies a compute unit alone while three of its waves idle) -- four slices of the reduction per thread then, the same doubles in the same order
template <int NT>
__global__ __launch_bounds__(NT) RGBID_SCALAR_KERNEL void k_kfa_solve(const double* partials, int nblk, KfaState* st, WarpParams* wp, SysParams* sp, SigmaIO* io, int next_level) {
  static_assert(NT == 256 || NT == 64, "8 slices of 32 threads, or 2 x 4");
  const int lane = blockIdx.x, tid = threadIdx.x;
  __shared__ double sm[8][32];
  __shared__ double sums[SYS_TERMS];
  const int k = tid & 31;
  for (int sl = tid >> 5; sl < 8; sl += NT / 32) {
    double acc = 0.0;
    if (k < SYS_TERMS) {
      const double* p = partials + (size_t)lane * nblk * SYS_TERMS + k;
      for (int b = sl; b < nblk; b += 8) acc += p[(size_t)b * SYS_TERMS];
    }
    sm[sl][k] = acc;
  }
  __syncthreads();
  if (tid < SYS_TERMS) {
    double r = 0.0;
    for (int i = 0; i < 8; ++i) r += sm[i][tid];
    sums[tid] = r;
  }
  __syncthreads();
  if (tid != 0) return;
  KfaState& s = st[lane];
  double b[6], x[6];
  int shift = 0;  // estimate_VO.cu:774-786
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) {
      const double v = sums[shift++];
      if (j == 6) b[i] = v; else s.A[j * 6 + i] = s.A[i * 6 + j] = v;
    }
  se3::llt_solve6(s.A, b, x);
  double inc_inv[9], inc[9], tinc[3], tmp[3];
  se3::expmap_rot(x + 3, inc_inv);
  se3::m3_inv(inc_inv, inc);
  se3::m3_mulv(inc, x, tinc);
  se3::m3_mulv(inc, s.t, tmp);
  for (int i = 0; i < 3; ++i) s.t[i] = tmp[i] - tinc[i];
  se3::m3_mul(inc, s.R, s.R);
  if (next_level >= 0) {
    kfa_set_warp(s, next_level, wp[lane]);
    kfa_set_iteration(s, next_level, sp[lane], io[lane]);
  }
}

__global__ __launch_bounds__(64) RGBID_SCALAR_KERNEL void k_kfa_finish(const KfaState* st, double* R, double* t, double* cov, int B) {
  const int lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= B) return;
  const KfaState& s = st[lane];
  for (int i = 0; i < 9; ++i) R[lane * 9 + i] = s.R[i];
  for (int i = 0; i < 3; ++i) t[lane * 3 + i] = s.t[i];
  se3::inverse6(s.A, cov + (size_t)lane * 36);   // covariance_ini2end = A_final.inverse() :343
}

}  // namespace

struct rgbid_kfalign {
  rgbid_ctx* ctx = nullptr;
  int rows = 0, cols = 0, cap = 0;
  std::vector<void*> allocs;
  size_t bytes = 0;
  ImgB iD_ini[KFA_LEVELS], I_ini[KFA_LEVELS], iD_end[KFA_LEVELS], I_end[KFA_LEVELS], W1[KFA_LEVELS], I1[KFA_LEVELS];
  ImgB gxD[KFA_LEVELS], gyD[KFA_LEVELS], gxI[KFA_LEVELS], gyI[KFA_LEVELS];
  unsigned char *grey_a = nullptr, *grey_b = nullptr;   // dense staging of the two grey images
  float *dense_a = nullptr, *dense_b = nullptr;         // dense staging of the two inverse-depth maps (host-input entry point)
  float* res = nullptr; size_t res_cap = 0;             // residual lattice [cap][res_cap]
  double* partials = nullptr; int nblk_cap = 0;
  KfaState* state = nullptr;
  WarpParams* wp = nullptr; SysParams* sp = nullptr; SigmaIO* io = nullptr;
  double *R_dev = nullptr, *t_dev = nullptr, *cov_dev = nullptr; float* K_dev = nullptr;
  int launches = 0;
};

namespace {

int kfa_alloc(rgbid_kfalign* a, void** p, size_t bytes) {
  hipError_t err = hipMalloc(p, bytes);
  if (err != hipSuccess) return err == hipErrorOutOfMemory ? RGBID_E_NOMEM : (int)err;
  a->allocs.push_back(*p);
  a->bytes += bytes;
  return RGBID_OK;
}
int kfa_img(rgbid_kfalign* a, ImgB* im, int rows, int cols) {
  const size_t pitch = ((size_t)cols * 4 + 255) & ~(size_t)255, lane_stride = pitch * rows;
  void* p = nullptr;
  int r = kfa_alloc(a, &p, lane_stride * a->cap);
  if (r) return r;
  *im = ImgB{p, pitch, lane_stride, rows, cols};
  return RGBID_OK;
}
inline ImgB dense_view(const void* p, int rows, int cols, int elem) { return ImgB{const_cast<void*>(p), (size_t)cols * elem, (size_t)rows * cols * elem, rows, cols}; }
const LaneMask ALL{nullptr, 0};

}  // namespace

extern "C" {

int rgbid_kfalign_create(rgbid_kfalign** out, rgbid_ctx* ctx, int rows, int cols, int max_pairs) {
  if (!out || !ctx || max_pairs < 1 || rows < (4 << (KFA_LEVELS - 1)) || cols < (4 << (KFA_LEVELS - 1)) || cols > (1 << 20) ||
      (unsigned long long)rows * (((unsigned long long)cols * 4 + 255) & ~255ull) >= (1ull << 32))   // 24-bit row-offset arithmetic (common.h row_ptr)
    return RGBID_E_INVALID;
  *out = nullptr;
  rgbid_kfalign* a = new (std::nothrow) rgbid_kfalign();
  if (!a) return RGBID_E_NOMEM;
  a->ctx = ctx; a->rows = rows; a->cols = cols; a->cap = max_pairs;
  hipSetDevice(ctx->device);
  int r = RGBID_OK;
  for (int l = 0; l < KFA_LEVELS && !r; ++l) {
    const int pr = rows >> l, pc = cols >> l;
    ImgB* maps[] = {&a->iD_ini[l], &a->I_ini[l], &a->iD_end[l], &a->I_end[l], &a->W1[l], &a->I1[l], &a->gxD[l], &a->gyD[l], &a->gxI[l], &a->gyI[l]};
    for (ImgB* im : maps) if (!r) r = kfa_img(a, im, pr, pc);
    const size_t n = (size_t)lattice_samples(pr, pc, KFA_NSAMPLES);
    if (n > a->res_cap) a->res_cap = n;
    // the partials buffer must hold the launch plan of EVERY pair count up to max_pairs: the plan of few pairs (<= 32) comes from a schedule-length
    // model, so blocks per pair are not monotonic in the pair count
    for (int b = 1; b <= max_pairs; b = (b < 64 || b == max_pairs) ? b + 1 : (2 * b < max_pairs ? 2 * b : max_pairs)) {   // 1 .. 64, powers of two, max_pairs
      const int nb = system_blocks_per_lane(pr, pc, b);
      if (nb > a->nblk_cap) a->nblk_cap = nb;
    }
  }
  const size_t N = (size_t)rows * cols, B = (size_t)max_pairs;
  if (!r) r = kfa_alloc(a, (void**)&a->grey_a, N * B);
  if (!r) r = kfa_alloc(a, (void**)&a->grey_b, N * B);
  if (!r) r = kfa_alloc(a, (void**)&a->res, sizeof(float) * a->res_cap * B);
  if (!r) r = kfa_alloc(a, (void**)&a->partials, sizeof(double) * SYS_TERMS * (size_t)a->nblk_cap * B);
  if (!r) r = kfa_alloc(a, (void**)&a->state, sizeof(KfaState) * B);
  if (!r) r = kfa_alloc(a, (void**)&a->wp, sizeof(WarpParams) * B);
  if (!r) r = kfa_alloc(a, (void**)&a->sp, sizeof(SysParams) * B);
  if (!r) r = kfa_alloc(a, (void**)&a->io, sizeof(SigmaIO) * B);
  if (!r) r = kfa_alloc(a, (void**)&a->R_dev, sizeof(double) * 9 * B);
  if (!r) r = kfa_alloc(a, (void**)&a->t_dev, sizeof(double) * 3 * B);
  if (!r) r = kfa_alloc(a, (void**)&a->cov_dev, sizeof(double) * 36 * B);
  if (!r) r = kfa_alloc(a, (void**)&a->K_dev, sizeof(float) * 4 * B);
  if (r) { rgbid_kfalign_destroy(a); return r; }
  *out = a;
  return RGBID_OK;
}

int rgbid_kfalign_destroy(rgbid_kfalign* a) {
  if (!a) return RGBID_OK;
  hipSetDevice(a->ctx->device);
  hipStreamSynchronize(a->ctx->stream);
  for (void* p : a->allocs) hipFree(p);
  delete a;
  return RGBID_OK;
}

int rgbid_kfalign_batched(rgbid_kfalign* a, int pairs, const float* iD_ini_dev, const unsigned char* grey_ini_dev, const float* iD_end_dev,
                          const unsigned char* grey_end_dev, const float* K, double* R, double* t, double* cov) {
  if (!a || pairs < 1 || pairs > a->cap || !iD_ini_dev || !grey_ini_dev || !iD_end_dev || !grey_end_dev || !K || !R || !t || !cov) return RGBID_E_INVALID;
  hipSetDevice(a->ctx->device);
  hipStream_t s = a->ctx->stream;
  const int B = pairs, rows = a->rows, cols = a->cols;
  const int tb = 64, gb = div_up(B, tb);
  a->launches = 0;
  for (int l = 0; l < KFA_LEVELS; ++l)   // the normal equations' launch plan of THIS pair count must fit the partials buffer (sized at creation)
    if (system_blocks_per_lane(rows >> l, cols >> l, B) > a->nblk_cap) return RGBID_E_INVALID;
#define KFA_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
  KFA_HIP(hipMemcpyAsync(a->R_dev, R, sizeof(double) * 9 * B, hipMemcpyHostToDevice, s));
  KFA_HIP(hipMemcpyAsync(a->t_dev, t, sizeof(double) * 3 * B, hipMemcpyHostToDevice, s));
  KFA_HIP(hipMemcpyAsync(a->K_dev, K, sizeof(float) * 4 * B, hipMemcpyHostToDevice, s));
  // level 0: the inverse-depth maps into the aligner's pitched buffers, the grey images converted to float (:120-129)
  launch_copy_bytes(s, B, dense_view(iD_ini_dev, rows, cols, 4), a->iD_ini[0], 4, ALL);
  launch_copy_bytes(s, B, dense_view(iD_end_dev, rows, cols, 4), a->iD_end[0], 4, ALL);
  hipLaunchKernelGGL(k_kfa_grey, dim3(div_up(cols, 256), rows, B), dim3(256), 0, s, grey_ini_dev, a->I_ini[0]);
  hipLaunchKernelGGL(k_kfa_grey, dim3(div_up(cols, 256), rows, B), dim3(256), 0, s, grey_end_dev, a->I_end[0]);
  a->launches += 4;
  for (int l = 1; l < KFA_LEVELS; ++l) {   // :155-162
    launch_pyr_down2(s, B, a->iD_ini[l - 1], a->iD_ini[l], a->iD_end[l - 1], a->iD_end[l], ALL);
    launch_pyr_down2(s, B, a->I_ini[l - 1], a->I_ini[l], a->I_end[l - 1], a->I_end[l], ALL);
    a->launches += 2;
  }
  for (int l = 0; l < KFA_LEVELS; ++l) {   // :166-174
    launch_gradient2(s, B, a->iD_ini[l], a->gxD[l], a->gyD[l], a->I_ini[l], a->gxI[l], a->gyI[l], ALL);
    a->launches += 1;
  }
  // the levels that iterate, coarse to fine
  int stage_level[KFA_LEVELS], n_stages = 0;
  for (int l = KFA_LEVELS - 1; l >= 0; --l) if (KFA_ITERS[l] > 0) stage_level[n_stages++] = l;
  hipLaunchKernelGGL(k_kfa_begin, dim3(gb), dim3(tb), 0, s, a->state, a->R_dev, a->t_dev, a->K_dev, a->wp, a->sp, a->io, n_stages ? stage_level[0] : 0, B);
  a->launches += 1;
  for (int st = 0; st < n_stages; ++st) {
    const int l = stage_level[st];
    int n, lr, lc, stride;
    lattice_geometry(a->iD_ini[l].rows, a->iD_ini[l].cols, KFA_NSAMPLES, &n, &lr, &lc, &stride);
    for (int it = 0; it < KFA_ITERS[l]; ++it) {
      const bool last_of_level = it == KFA_ITERS[l] - 1;
      const int next_level = !last_of_level ? l : (st + 1 < n_stages ? stage_level[st + 1] : -1);
      launch_warp_invdepth(s, B, a->iD_end[l], a->iD_ini[l], a->W1[l], nullptr, a->wp, ALL);
      launch_warp_intensity(s, B, a->I_end[l], a->iD_ini[l], a->I1[l], nullptr, a->wp, a->ctx->interp_mode, ALL);   // sampled on the KEYFRAME inverse depth (:239-242)
      launch_error_lattice(s, B, a->W1[l], a->iD_ini[l], a->res, a->res_cap, lr, lc, stride, ALL);
      launch_sigma(s, B, 1, a->res, a->res_cap, n, a->io, RGBID_STUDENT, ALL);                                        // computeNuStudent (:300)
      hipLaunchKernelGGL(k_kfa_set_nu, dim3(gb), dim3(tb), 0, s, a->io, a->sp, B);
      const int nblk = launch_build_system(s, B, a->iD_ini[l], a->I_ini[l], a->gxD[l], a->gyD[l], a->gxI[l], a->gyI[l], a->W1[l], a->I1[l], nullptr, a->sp, a->partials, ALL,
                                           l < 2 ? l : 2);
      if (B > 256) hipLaunchKernelGGL(k_kfa_solve<64>, dim3(B), dim3(64), 0, s, a->partials, nblk, a->state, a->wp, a->sp, a->io, next_level);
      else hipLaunchKernelGGL(k_kfa_solve<256>, dim3(B), dim3(256), 0, s, a->partials, nblk, a->state, a->wp, a->sp, a->io, next_level);
      a->launches += 7;
    }
  }
  hipLaunchKernelGGL(k_kfa_finish, dim3(gb), dim3(tb), 0, s, a->state, a->R_dev, a->t_dev, a->cov_dev, B);
  a->launches += 1;
  KFA_HIP(hipGetLastError());
  KFA_HIP(hipMemcpyAsync(R, a->R_dev, sizeof(double) * 9 * B, hipMemcpyDeviceToHost, s));
  KFA_HIP(hipMemcpyAsync(t, a->t_dev, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, s));
  KFA_HIP(hipMemcpyAsync(cov, a->cov_dev, sizeof(double) * 36 * B, hipMemcpyDeviceToHost, s));
  KFA_HIP(hipStreamSynchronize(s));
  return RGBID_OK;
}

int rgbid_kfalign_batched_host(rgbid_kfalign* a, int pairs, const float* iD_ini, const unsigned char* grey_ini, const float* iD_end, const unsigned char* grey_end,
                               const float* K, double* R, double* t, double* cov) {
  if (!a || pairs < 1 || pairs > a->cap || !iD_ini || !grey_ini || !iD_end || !grey_end) return RGBID_E_INVALID;
  hipSetDevice(a->ctx->device);
  hipStream_t s = a->ctx->stream;
  const size_t N = (size_t)a->rows * a->cols;
  if (!a->dense_a) {   // only this entry point needs dense device staging of the inverse-depth maps
    int r = kfa_alloc(a, (void**)&a->dense_a, sizeof(float) * N * a->cap);
    if (!r) r = kfa_alloc(a, (void**)&a->dense_b, sizeof(float) * N * a->cap);
    if (r) return r;
  }
  KFA_HIP(hipMemcpyAsync(a->dense_a, iD_ini, sizeof(float) * N * pairs, hipMemcpyHostToDevice, s));
  KFA_HIP(hipMemcpyAsync(a->dense_b, iD_end, sizeof(float) * N * pairs, hipMemcpyHostToDevice, s));
  KFA_HIP(hipMemcpyAsync(a->grey_a, grey_ini, N * pairs, hipMemcpyHostToDevice, s));
  KFA_HIP(hipMemcpyAsync(a->grey_b, grey_end, N * pairs, hipMemcpyHostToDevice, s));
  return rgbid_kfalign_batched(a, pairs, a->dense_a, a->grey_a, a->dense_b, a->grey_b, K, R, t, cov);
#undef KFA_HIP
}

int rgbid_kfalign_launches(const rgbid_kfalign* a) { return a ? a->launches : 0; }
int rgbid_kfalign_bytes(const rgbid_kfalign* a, size_t* bytes) { if (!a || !bytes) return RGBID_E_INVALID; *bytes = a->bytes; return RGBID_OK; }

}  // extern "C"
