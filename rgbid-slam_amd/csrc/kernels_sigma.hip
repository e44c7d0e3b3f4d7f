// kernels_sigma.hip -- residual lattice + robust scale estimation (bias, sigma, Student-t nu) and the
// chi-square statistic for gfx950.  Replaces src/cuda/sigmaFuncs.cu of the reference.
//
// The reference runs every IRLS / bisection step as two kernel launches + two stream syncs + a 4-8 byte
// D2H copy (16-32 launches and 7 cudaMalloc/cudaFree pairs per call, sigmaFuncs.cu:858-1066), with
// boost::math::digamma evaluated on the host.  Here one 512-thread workgroup per lane keeps its
// <=40 residual samples per thread in VGPRs (19 200 samples at every pyramid level of a 640x480
// frame), runs ALL iterations in-kernel (moments -> wave64 shuffle reduction -> LDS across 8 waves ->
// broadcast), evaluates digamma on the device, and writes (bias, sigma, nu): one launch, no host trips.
#include "kernels.h"
#include <cstdlib>
#include "sigma_device.h"

#pragma clang fp contract(off)

namespace rgbid {

// ---- lattice geometry: computeErrorGridStride sigmaFuncs.cu:701-765 --------------------------------
void lattice_geometry(int rows, int cols, int min_nsamples, int* n, int* lrows, int* lcols, int* stride) {
  int error_size = cols * rows;
  int cols_prev = cols, rows_prev = rows;
  if (min_nsamples < error_size) {
    for (;;) {
      int cols_curr = cols_prev / 2, rows_curr = rows_prev / 2;
      if (((2 * cols_curr - cols_prev) != 0) || ((2 * rows_curr - rows_prev) != 0) || (min_nsamples > cols_curr * rows_curr)) {
        error_size = cols_prev * rows_prev;
        break;
      }
      cols_prev = cols_curr;
      rows_prev = rows_curr;
    }
  }
  int s = 1;
  while ((long long)(s + 1) * (s + 1) <= (long long)((rows * cols) / error_size)) ++s;  // == (int)sqrt(.)
  *n = error_size; *lrows = rows_prev; *lcols = cols_prev; *stride = s;
}

// errorHandler::computeErrorGridStride sigmaFuncs.cu:90-135
__global__ __launch_bounds__(256) void k_error_lattice(ImgB im1, ImgB im0, float* err, size_t err_lane_stride, int lrows, int lcols, int stride, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= lcols || y >= lrows) return;
  float v = px<float>(im1, lane, stride * y, stride * x) - px<float>(im0, lane, stride * y, stride * x);
  err[(size_t)lane * err_lane_stride + (size_t)y * lcols + x] = v;
}
void launch_error_lattice(hipStream_t s, int B, ImgB im1, ImgB im0, float* err, size_t err_lane_stride, int lrows, int lcols, int stride, LaneMask m) {
  hipLaunchKernelGGL(k_error_lattice, dim3(div_up(lcols, 64), div_up(lrows, 4), B), dim3(64, 4), 0, s, im1, im0, err, err_lane_stride, lrows, lcols, stride, m);
}

// ---- digamma (device.hpp:76-80 -> boost::math::digamma<float>): recurrence + asymptotic series in double.
// The bisection of computeSigmaAndNuStudent only ever visits nu in {2, 2.5, ..., 10}, so the four
// transcendental terms of C(nu) are tabulated once on the host (the reference also evaluates them on the
// host) and passed to the kernel by value; the kernel keeps the reference's left-to-right fp32 evaluation.
static float digamma_host(float x) {
  double xd = x, r = 0.0;
  while (xd < 10.0) { r -= 1.0 / xd; xd += 1.0; }
  double f = 1.0 / (xd * xd);
  double s = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return (float)(r + log(xd) - 0.5 / xd + s);
}
static const NuTable& nu_table() {
  static const NuTable T = [] {
    NuTable t;
    for (int k = 0; k < 17; ++k) {
      float nu = 2.f + 0.5f * (float)k;
      t.t[k][0] = -digamma_host(nu / 2.f);
      t.t[k][1] = logf(nu / 2.f);
      t.t[k][2] = digamma_host((nu + 1.f) / 2.f);
      t.t[k][3] = logf((nu + 1.f) / 2.f);
    }
    return t;
  }();
  return T;
}
const NuTable& sigma_nu_table() { return nu_table(); }   // sigma_device.h: the same table for the per-lane persistent kernel (tools/experiments/persistent_gn_level/kernels_gnlevel.hip: a measured experiment, profiles/r04_experiments/persistent_gn_level.md)


template <bool REG>
__global__ __launch_bounds__(SIG_T) void k_sigma(NuTable T, int mode, const float* err, size_t err_lane_stride, int n, SigmaIO* io, int mestimator, LaneMask m) {
  int lane = blockIdx.x;
  if (!m.on(lane)) return;
  __shared__ double sm_[SIG_SM];
  BlockSum sm(sm_);
  Samples<REG, ArrayGetter> S(ArrayGetter{err + (size_t)lane * err_lane_stride, 0}, n, threadIdx.x);
  SigmaIO v = io[lane];
  float bias = v.bias, sigma = v.sigma, nu = v.nu;
  sigma_core(S, T, mode, mestimator, bias, sigma, nu, sm);
  if (threadIdx.x == 0) { v.bias = bias; v.sigma = sigma; v.nu = nu; io[lane] = v; }
}

void launch_sigma(hipStream_t s, int B, int mode, const float* err, size_t err_lane_stride, int n, SigmaIO* io, int mestimator, LaneMask m) {
  if (n <= SIG_T * SIG_MAXPT)
    hipLaunchKernelGGL(k_sigma<true>, dim3(B), dim3(SIG_T), 0, s, nu_table(), mode, err, err_lane_stride, n, io, mestimator, m);
  else
    hipLaunchKernelGGL(k_sigma<false>, dim3(B), dim3(SIG_T), 0, s, nu_table(), mode, err, err_lane_stride, n, io, mestimator, m);
}

// ---- batched engine: both channels of one GN iteration in ONE launch (grid = lanes x 2) -----------------
// blockIdx.y == 0: inverse depth (W1 - W0), == 1: intensity (I1 - I0).  The residual lattice of
// computeErrorGridStride (sigmaFuncs.cu:90-135) is sampled straight from the maps, the start values are the
// ones visodo.cpp:1168-1173 sets each iteration, and (bias, sigma, nu) land in the lane's SysParams.
struct LatticeGetter {
  ImgB a, b;  // a - b
  int lane, lcols, stride;
  int y, x, sy, sx;  // cursor: lattice row / column of the current sample and the (row, column) advance of SIG_T samples
  __device__ __forceinline__ float at(int ly, int lx) const {
    return px<float>(a, lane, stride * ly, stride * lx) - px<float>(b, lane, stride * ly, stride * lx);
  }
  __device__ __forceinline__ float operator()(int i) const { int ly = i / lcols; return at(ly, i - ly * lcols); }
  __device__ __forceinline__ void seek(int i) { y = i / lcols; x = i - y * lcols; sy = SIG_T / lcols; sx = SIG_T - sy * lcols; }
  __device__ __forceinline__ float load() const { return at(y, x); }
  __device__ __forceinline__ void step() { y += sy; x += sx; if (x >= lcols) { x -= lcols; ++y; } }
};

template <bool REG>
__global__ __launch_bounds__(SIG_T) void k_sigma_pair(NuTable T, ImgB W1, ImgB W0, ImgB I1, ImgB I0, int lrows, int lcols, int stride,
                                                      SysParams* sp, int mestimator, LaneMask m) {
  int lane = blockIdx.x, ch = blockIdx.y;
  if (!m.on(lane)) return;
  __shared__ double sm_[SIG_SM];
  BlockSum sm(sm_);
  LatticeGetter g{ch == 0 ? W1 : I1, ch == 0 ? W0 : I0, lane, lcols, stride, 0, 0, 0, 0};
  Samples<REG, LatticeGetter> S(g, lrows * lcols, threadIdx.x);
  float bias = 0.f, sigma = ch == 0 ? 0.0025f : 5.f, nu = 5.f;
  sigma_core(S, T, 0, mestimator, bias, sigma, nu, sm);
  if (threadIdx.x == 0) {
    if (ch == 0) { sp[lane].bias_d = bias; sp[lane].sigma_d = sigma; sp[lane].nu_d = nu; }
    else { sp[lane].bias_i = bias; sp[lane].sigma_i = sigma; sp[lane].nu_i = nu; }
  }
}
void launch_sigma_pair(hipStream_t s, int B, ImgB W1, ImgB W0, ImgB I1, ImgB I0, int min_nsamples, SysParams* sp, int mestimator, LaneMask m) {
  int n, lr, lc, st;
  lattice_geometry(W0.rows, W0.cols, min_nsamples, &n, &lr, &lc, &st);
  if (n <= SIG_T * SIG_MAXPT)
    hipLaunchKernelGGL(k_sigma_pair<true>, dim3(B, 2), dim3(SIG_T), 0, s, nu_table(), W1, W0, I1, I0, lr, lc, st, sp, mestimator, m);
  else
    hipLaunchKernelGGL(k_sigma_pair<false>, dim3(B, 2), dim3(SIG_T), 0, s, nu_table(), W1, W0, I1, I0, lr, lc, st, sp, mestimator, m);
}


// Two kernels.  With the warps inside the sigma / nu kernel a thread walks its <= 19 samples one after the other, each through three dependent
// memory round trips (keyframe iD -> point sample -> bilinear taps): +85 us of exposed latency per launch (and batching the round trips over 3
// samples, the most 64 VGPRs allow: 190 us, no better than the pair below).  Here one
// thread per lattice sample warps its pixel (9.8 M independent threads at 512 lanes: the latency hides behind occupancy) and parks both
// residuals in res[lane][channel][n]; the sigma / nu kernel then reads them as plain coalesced arrays.
// kf_lat (nullable): the keyframe side of the lattice, packed once per keyframe by k_lattice_pack -- [lane][2][n] = W0 | I0 at the lattice
// points.  The lattice takes every stride-th pixel of every stride-th row, i.e. a quarter of the cache lines of each map it samples at
// level 0; the two keyframe maps do not change between keyframes, so their samples are read here as two coalesced arrays instead.
__global__ __launch_bounds__(256) void k_lattice_residuals_fused(ImgB Wcur, ImgB W0, ImgB Icur, ImgB I0, const WarpParams* wp, int interp_mode, int n, int lcols,
                                                                 int stride, float* res, size_t res_lane_stride, const float* kf_lat, size_t kf_lat_lane_stride,
                                                                 LaneMask m, int fast) {
  const int lane = blockIdx.y;
  if (!m.on(lane)) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  FusedLatticeGetter g{Wcur, Icur, W0, I0, wp[lane], lane, stride, interp_mode, fast};
  const int ly = i / lcols, lx = i - ly * lcols;
  float rd, ri;
  if (kf_lat) {
    const float* k = kf_lat + (size_t)lane * kf_lat_lane_stride;
    g.both_given(ly, lx, k[i], k[n + i], rd, ri);
  } else {
    g.both(ly, lx, rd, ri);
  }
  float* r = res + (size_t)lane * res_lane_stride;
  r[i] = rd; r[n + i] = ri;
}
struct LatPackLevel { ImgB W0, I0; int n, lcols, stride; float* out; };
struct LatPackLevels { LatPackLevel l[4]; };
// blockIdx.z: pyramid level (the keyframe side of every level's lattice in ONE launch at a keyframe switch)
__global__ __launch_bounds__(256) void k_lattice_pack(LatPackLevels L, size_t out_lane_stride, LaneMask m) {
  const int lane = blockIdx.y;
  if (!m.on(lane)) return;
  const LatPackLevel& P = L.l[blockIdx.z];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P.n) return;
  const int ly = i / P.lcols, lx = i - ly * P.lcols;
  float* o = P.out + (size_t)lane * out_lane_stride;
  o[i] = px<float>(P.W0, lane, ly * P.stride, lx * P.stride);
  o[P.n + i] = px<float>(P.I0, lane, ly * P.stride, lx * P.stride);
}
void launch_lattice_pack_levels(hipStream_t s, int B, int n_levels, const ImgB* W0, const ImgB* I0, int min_nsamples, float* const* out, size_t out_lane_stride, LaneMask m) {
  for (int first = 0; first < n_levels; first += 4) {
    LatPackLevels L{};
    const int cnt = n_levels - first < 4 ? n_levels - first : 4;
    int nmax = 0;
    for (int k = 0; k < cnt; ++k) {
      int n, lr, lc, st;
      lattice_geometry(W0[first + k].rows, W0[first + k].cols, min_nsamples, &n, &lr, &lc, &st);
      L.l[k] = LatPackLevel{W0[first + k], I0[first + k], n, lc, st, out[first + k]};
      nmax = n > nmax ? n : nmax;
    }
    hipLaunchKernelGGL(k_lattice_pack, dim3(div_up(nmax, 256), B, cnt), dim3(256), 0, s, L, out_lane_stride, m);
  }
}
void launch_lattice_pack(hipStream_t s, int B, ImgB W0, ImgB I0, int min_nsamples, float* out, size_t out_lane_stride, LaneMask m) {
  launch_lattice_pack_levels(s, B, 1, &W0, &I0, min_nsamples, &out, out_lane_stride, m);
}
template <bool REG>
__global__ __launch_bounds__(SIG_T) void k_sigma_pair_arrays(NuTable T, const float* res, size_t res_lane_stride, int n, SysParams* sp, int mestimator, LaneMask m) {
  int lane = blockIdx.x, ch = blockIdx.y;
  if (!m.on(lane)) return;
  __shared__ double sm_[SIG_SM];
  BlockSum sm(sm_);
  Samples<REG, ArrayGetter> S(ArrayGetter{res + (size_t)lane * res_lane_stride + (size_t)ch * n, 0}, n, threadIdx.x);
  float bias = 0.f, sigma = ch == 0 ? 0.0025f : 5.f, nu = 5.f;
  sigma_core(S, T, 0, mestimator, bias, sigma, nu, sm);
  if (threadIdx.x == 0) {
    if (ch == 0) { sp[lane].bias_d = bias; sp[lane].sigma_d = sigma; sp[lane].nu_d = nu; }
    else { sp[lane].bias_i = bias; sp[lane].sigma_i = sigma; sp[lane].nu_i = nu; }
  }
}
int lattice_samples(int rows, int cols, int min_nsamples) {
  int n, lr, lc, st;
  lattice_geometry(rows, cols, min_nsamples, &n, &lr, &lc, &st);
  return n;
}

// the lattice pre-pass of the fused path alone: both channels' residuals of every lattice sample into res[lane][2][n].  `fast` is the RESOLVED
// numerics class of the level (kernels.h gn_fast_supported): the caller decides it once for the lattice and the normal equations.
void launch_lattice_residuals_fused(hipStream_t s, int B, ImgB Wcur, ImgB W0, ImgB Icur, ImgB I0, const WarpParams* lane_wp, int interp_mode, int min_nsamples,
                                    LaneMask m, bool fast, float* res, size_t res_lane_stride, const float* kf_lat, size_t kf_lat_lane_stride) {
  int n, lr, lc, st;
  lattice_geometry(W0.rows, W0.cols, min_nsamples, &n, &lr, &lc, &st);
  hipLaunchKernelGGL(k_lattice_residuals_fused, dim3(div_up(n, 256), B), dim3(256), 0, s, Wcur, W0, Icur, I0, lane_wp, interp_mode, n, lc, st, res, res_lane_stride,
                     kf_lat_lane_stride >= 2 * (size_t)n ? kf_lat : nullptr, kf_lat_lane_stride, m, fast ? 1 : 0);
}
// computeSigmaAndNuStudent of both channels on res[lane][2][n] (start values of visodo.cpp:1168-1173), results into sp[lane]
void launch_sigma_pair_arrays(hipStream_t s, int B, const float* res, size_t res_lane_stride, int n, SysParams* sp, int mestimator, LaneMask m) {
  if (n <= SIG_T * SIG_MAXPT)
    hipLaunchKernelGGL(k_sigma_pair_arrays<true>, dim3(B, 2), dim3(SIG_T), 0, s, nu_table(), res, res_lane_stride, n, sp, mestimator, m);
  else
    hipLaunchKernelGGL(k_sigma_pair_arrays<false>, dim3(B, 2), dim3(SIG_T), 0, s, nu_table(), res, res_lane_stride, n, sp, mestimator, m);
}
void launch_sigma_pair_fused(hipStream_t s, int B, ImgB Wcur, ImgB W0, ImgB Icur, ImgB I0, const WarpParams* lane_wp, int interp_mode,
                             int min_nsamples, SysParams* sp, int mestimator, LaneMask m, bool fast, float* res, size_t res_lane_stride,
                             const float* kf_lat, size_t kf_lat_lane_stride) {
  // res: [lane][2][n] scratch of at least 2 * lattice_samples() floats per lane (the engine sizes it at creation)
  launch_lattice_residuals_fused(s, B, Wcur, W0, Icur, I0, lane_wp, interp_mode, min_nsamples, m, fast, res, res_lane_stride, kf_lat, kf_lat_lane_stride);
  launch_sigma_pair_arrays(s, B, res, res_lane_stride, lattice_samples(W0.rows, W0.cols, min_nsamples), sp, mestimator, m);
}

// ---- computeChiSquare sigmaFuncs.cu:1225-1297 (+ :137-150, :541-646) --------------------------------
__global__ __launch_bounds__(SIG_T) void k_chi_square(const float* err_int, const float* err_depth, size_t err_lane_stride, int n,
                                                      float sigma_int, float sigma_depth, int mest, float* out, LaneMask m) {
  int lane = blockIdx.x;
  if (!m.on(lane)) return;
  __shared__ double sm_[SIG_SM];
  BlockSum sm(sm_);
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  for (int half = 0; half < 2; ++half) {
    const float* e = (half == 0 ? err_int : err_depth) + (size_t)lane * err_lane_stride;
    float sg = half == 0 ? sigma_int : sigma_depth;
    for (int i = threadIdx.x; i < n; i += SIG_T) {
      float en = e[i] / sg;
      float rho = 0.f;
      if (!isinf(en) && !isnan(en)) {
        a[0] += 1.f;
        rho = (en * en) / 2.f;
        if ((mest == 1) && (fabsf(en) > TH_HUBER)) rho = TH_HUBER * (fabsf(en) - TH_HUBER / 2.f);
        else if (mest == 2) {
          if (fabsf(en) < TH_TUKEY) {
            float aux1 = (en / TH_TUKEY) * (en / TH_TUKEY);
            float aux2 = (1.f - aux1) * (1.f - aux1) * (1.f - aux1);
            rho = ((TH_TUKEY * TH_TUKEY) / 6.f) * (1.f - aux2);
          } else rho = ((TH_TUKEY * TH_TUKEY) / 6.f);
        } else if (mest == 3) rho = ((STUDENT_DOF + 1.f) / 2.f) * logf(1.f + (en * en) / STUDENT_DOF);
      }
      a[1] += rho;
    }
  }
  double t[4];
  block_sum4(a, t, sm);
  if (threadIdx.x == 0) {
    float fN = (float)t[0], frho = (float)t[1];
    float chi = frho / fN;
    float z_gauss = (chi - fN) / (sqrtf(2.f * fN));
    out[lane * 3 + 0] = chi;
    out[lane * 3 + 1] = 0.5f * (1.f + erff(z_gauss / sqrtf(2.f)));
    out[lane * 3 + 2] = fN;
  }
}
void launch_chi_square(hipStream_t s, int B, const float* err_int, const float* err_depth, size_t err_lane_stride, int n,
                       float sigma_int, float sigma_depth, int mestimator, float* out, LaneMask m) {
  hipLaunchKernelGGL(k_chi_square, dim3(B), dim3(SIG_T), 0, s, err_int, err_depth, err_lane_stride, n, sigma_int, sigma_depth, mestimator, out, m);
}

}  // namespace rgbid
