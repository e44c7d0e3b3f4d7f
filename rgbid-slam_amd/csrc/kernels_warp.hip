// kernels_warp.hip -- inverse warps, keyframe iD fusion, visibility ratio, vertex/normal maps and the
// preview image for gfx950.  Replaces src/cuda/warping_registration.cu (:297-669, 825-1095),
// src/cuda/maps.cu (:63-90,134-179) and src/cuda/image_generator.cu (:66-185) of the reference.
//
// The reference samples through CUDA texture objects created and destroyed on every call; CDNA has
// no texture path for fp32 linear filtering of pitched memory that would be worth it, so the point
// sample is a plain gather and the bilinear sample is four gathers combined in fp32, with the texture
// unit's 1.8 fixed-point weight quantisation available as RGBID_INTERP_TEX8 (what tex2D computes).
// The gathers hit L2 / Infinity Cache: a warped tile's footprint in the source frame is compact.
#include "kernels.h"
#include <cstdlib>
#include "warp_device.h"

// Whole file: no FMA contraction, so every fp32 expression is evaluated operation by operation exactly
// like the scalar oracle (divisions/sqrt are IEEE by hipcc default).  These kernels are bandwidth-bound.
#pragma clang fp contract(off)

namespace rgbid {

static constexpr int TX = 64, TY = 4, RPB = 4;  // a workgroup sweeps RPB stacked 64x4 tiles (see kernels_prep.hip)
static inline dim3 grid2d(int cols, int rows, int B) { return dim3(div_up(cols, TX), div_up(rows, TY * RPB), B); }
static inline dim3 grid2d_full(int cols, int rows, int B) { return dim3(div_up(cols, TX), div_up(rows, TY), B); }
static inline bool vec4_ok(const ImgB& a) { return ((a.pitch & 15) == 0) && ((a.lane_stride & 15) == 0) && ((((uintptr_t)a.base) & 15) == 0) && (a.cols % 4 == 0); }
#define RGBID_FOR_ROWS(yv) for (int it_ = 0, yv = blockIdx.y * (TY * RPB) + threadIdx.y; it_ < RPB; ++it_, yv += TY)

// ---- trafo3DKernelInvDepthGridStride (:505-546) ---------------------------------------------------
template <class PS>
__global__ __launch_bounds__(256) void k_warp_invdepth(ImgB src, ImgB grid, ImgB dst, PS ps, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x;
  const WarpParams P = ps.get(lane);
  const FMap S(src, lane);
  if (x >= dst.cols) return;
  // RPB independent pixels per thread: all grid loads first, then the RPB projection / gather chains (interleaved
  // by the scheduler), then the stores -- the dependent chain load -> divide -> project -> gather is long
  const int yb = blockIdx.y * (TY * RPB) + threadIdx.y;
  float wv[RPB], out[RPB];
#pragma unroll
  for (int i = 0; i < RPB; ++i) { int y = yb + i * TY; wv[i] = (y < dst.rows) ? px<float>(grid, lane, y, x) : qnan(); }
  // one straight-line pass with the fast exact reciprocal for all RPB pixels; a single (never taken on real data) fallback
  RcpFast fast;
#pragma unroll
  for (int i = 0; i < RPB; ++i) out[i] = warp_invdepth_px_t(S, x, yb + i * TY, wv[i], P, fast);
  if (__builtin_expect(fast.failed(), 0)) {
    RcpIeee ieee;
#pragma unroll
    for (int i = 0; i < RPB; ++i) out[i] = warp_invdepth_px_t(S, x, yb + i * TY, wv[i], P, ieee);
  }
#pragma unroll
  for (int i = 0; i < RPB; ++i) { int y = yb + i * TY; if (y < dst.rows) px<float>(dst, lane, y, x) = out[i]; }
}
void launch_warp_invdepth(hipStream_t s, int B, ImgB src, ImgB grid, ImgB dst, const WarpParams* hp, const WarpParams* lp, LaneMask m) {
  dim3 g = grid2d(dst.cols, dst.rows, B), b(TX, TY);
  if (lp) hipLaunchKernelGGL(k_warp_invdepth<ByLane<WarpParams>>, g, b, 0, s, src, grid, dst, ByLane<WarpParams>{lp}, m);
  else hipLaunchKernelGGL(k_warp_invdepth<ByValue<WarpParams>>, g, b, 0, s, src, grid, dst, ByValue<WarpParams>{*hp}, m);
}

// ---- trafo3DKernelIntensityWithInvDepthGridStride (:465-501) --------------------------------------
template <class PS>
__global__ __launch_bounds__(256) void k_warp_intensity(ImgB src, ImgB grid, ImgB dst, PS ps, int interp_mode, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x;
  const WarpParams P = ps.get(lane);
  const FMap S(src, lane);
  if (x >= dst.cols) return;
  const int yb = blockIdx.y * (TY * RPB) + threadIdx.y;
  float wv[RPB], out[RPB];
#pragma unroll
  for (int i = 0; i < RPB; ++i) { int y = yb + i * TY; wv[i] = (y < dst.rows) ? px<float>(grid, lane, y, x) : qnan(); }
  RcpFast fast;
#pragma unroll
  for (int i = 0; i < RPB; ++i) out[i] = warp_intensity_px_t(S, x, yb + i * TY, wv[i], P, interp_mode, fast);
  if (__builtin_expect(fast.failed(), 0)) {
    RcpIeee ieee;
#pragma unroll
    for (int i = 0; i < RPB; ++i) out[i] = warp_intensity_px_t(S, x, yb + i * TY, wv[i], P, interp_mode, ieee);
  }
#pragma unroll
  for (int i = 0; i < RPB; ++i) { int y = yb + i * TY; if (y < dst.rows) px<float>(dst, lane, y, x) = out[i]; }
}
void launch_warp_intensity(hipStream_t s, int B, ImgB src, ImgB grid, ImgB dst, const WarpParams* hp, const WarpParams* lp, int interp_mode, LaneMask m) {
  dim3 g = grid2d(dst.cols, dst.rows, B), b(TX, TY);
  if (lp) hipLaunchKernelGGL(k_warp_intensity<ByLane<WarpParams>>, g, b, 0, s, src, grid, dst, ByLane<WarpParams>{lp}, interp_mode, m);
  else hipLaunchKernelGGL(k_warp_intensity<ByValue<WarpParams>>, g, b, 0, s, src, grid, dst, ByValue<WarpParams>{*hp}, interp_mode, m);
}

// ---- engine: both warps of a Gauss-Newton iteration in one pass ---------------------------------------------------------------
// W1 = warp of the current inverse depth onto the keyframe grid, I1 = warp of the current intensity sampled with W1 (the tracker
// passes the WARPED inverse depth as the sampling grid, visodo.cpp:1098-1100).  Same device functions as the two kernels above, so
// the maps are bit-identical; W1 is consumed from registers (one load + one launch less per iteration).
template <class PS>
__global__ __launch_bounds__(256) void k_warp_pair(ImgB src_iD, ImgB src_I, ImgB grid, ImgB dst_iD, ImgB dst_I, PS ps, int interp_mode, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x;
  const WarpParams P = ps.get(lane);
  const FMap SD(src_iD, lane), SI(src_I, lane), G(grid, lane);
  const FMapW DW(dst_iD, lane), DI(dst_I, lane);
  if (x >= dst_iD.cols) return;
  const int yb = blockIdx.y * (TY * RPB) + threadIdx.y;
  float wv[RPB], w1[RPB], i1[RPB];
#pragma unroll
  for (int i = 0; i < RPB; ++i) { int y = yb + i * TY; wv[i] = (y < dst_iD.rows) ? G.at(y, x) : qnan(); }
  RcpFast fast;
#pragma unroll
  for (int i = 0; i < RPB; ++i) w1[i] = warp_invdepth_px_t(SD, x, yb + i * TY, wv[i], P, fast);
#pragma unroll
  for (int i = 0; i < RPB; ++i) i1[i] = warp_intensity_px_t(SI, x, yb + i * TY, w1[i], P, interp_mode, fast);
  if (__builtin_expect(fast.failed(), 0)) {
    RcpIeee ieee;
#pragma unroll
    for (int i = 0; i < RPB; ++i) w1[i] = warp_invdepth_px_t(SD, x, yb + i * TY, wv[i], P, ieee);
#pragma unroll
    for (int i = 0; i < RPB; ++i) i1[i] = warp_intensity_px_t(SI, x, yb + i * TY, w1[i], P, interp_mode, ieee);
  }
#pragma unroll
  for (int i = 0; i < RPB; ++i) {
    int y = yb + i * TY;
    if (y < dst_iD.rows) { DW.store(y, x, w1[i]); DI.store(y, x, i1[i]); }
  }
}
void launch_warp_pair(hipStream_t s, int B, ImgB src_iD, ImgB src_I, ImgB grid, ImgB dst_iD, ImgB dst_I, const WarpParams* lp, int interp_mode, LaneMask m, const WarpParams* hp) {
  dim3 g = grid2d(dst_iD.cols, dst_iD.rows, B), b(TX, TY);
  if (lp) hipLaunchKernelGGL(k_warp_pair<ByLane<WarpParams>>, g, b, 0, s, src_iD, src_I, grid, dst_iD, dst_I, ByLane<WarpParams>{lp}, interp_mode, m);
  else hipLaunchKernelGGL(k_warp_pair<ByValue<WarpParams>>, g, b, 0, s, src_iD, src_I, grid, dst_iD, dst_I, ByValue<WarpParams>{*hp}, interp_mode, m);
}

// ---- engine, fast numerics (warp_device.h fastnum): the same pair in the reference build's class of arithmetic ------------------------
// 1-D grid, XCD-contiguous tile order (TileMap): a lane's tiles run on one XCD, so neighbouring tiles share their gather footprints in L2.
// Per thread: x fixed, RPB rows TY apart -- the ray q = R (x, y, 1) is formed once (6 FMAs) and stepped down the rows with 3 adds, and it
// serves BOTH projections of every pixel (iD warp with the keyframe inverse depth, intensity warp with the warped inverse depth).
// Geometry as the exact kernel (lane <-> pixel: a wave covers 64 consecutive pixels of a row, RPB rows TY apart per thread): measured against
// a 4-consecutive-pixels-per-thread layout with 16-byte grid loads / stores (15 instead of 32 vector-memory instructions per 4 pixels) this is
// 15 % FASTER -- a gather instruction whose 64 lanes sit on consecutive pixels touches 2-3 cache lines, strided lanes touch 8-12.
template <class PS>
__global__ __launch_bounds__(256) void k_warp_pair_fast(ImgB src_iD, ImgB src_I, ImgB grid, ImgB dst_iD, ImgB dst_I, PS ps, int interp_mode, LaneMask m, TileMap tm) {
  const TileId t = tm.tile(blockIdx.x);
  const int lane = t.lane;
  if (!m.on(lane)) return;
  const int x = t.bx * TX + threadIdx.x;
  const WarpParams P = ps.get(lane);
  const FMap SD(src_iD, lane), SI(src_I, lane), G(grid, lane);
  const FMapW DW(dst_iD, lane), DI(dst_I, lane);
  if (x >= dst_iD.cols) return;
  const int yb = t.by * (TY * RPB) + threadIdx.y;
  float wv[RPB], w1[RPB], i1[RPB];
#pragma unroll
  for (int i = 0; i < RPB; ++i) { int y = yb + i * TY; wv[i] = (y < dst_iD.rows) ? G.at(y, x) : qnan(); }
  // the ray of a pixel is a function of the pixel alone (warp_device.h ray_at), so this kernel, the fused normal-equation kernel and the
  // sigma / nu lattice agree bit for bit
  const fastnum::Guard GB = fastnum::lane_guard(P, dst_iD.cols, dst_iD.rows);
  // phases over the thread's RPB pixels (a branch between a gather and its use would serialise their chains): projections (+ the oracle's
  // coordinates inside the guard band) -> point-sample gathers -> warped inverse depths -> tap loads (+ the oracle's in-image predicate) -> blends
  fastnum::Ray q[RPB];
  fastnum::IdProj pr[RPB];
  float s2[RPB];
  bool fr[RPB];
#pragma unroll
  for (int i = 0; i < RPB; ++i) {
    const int y = yb + i * TY;
    q[i] = fastnum::ray(P, (float)x, (float)y);
    bool fc;
    pr[i] = fastnum::id_project(q[i], wv[i], P, GB, SD.cols, SD.rows, fc);
    if (__builtin_expect(fc, 0)) fastnum::id_fix_coords(pr[i], x, y, P, SD.cols, SD.rows);
  }
#pragma unroll
  for (int i = 0; i < RPB; ++i) s2[i] = SD.at(pr[i].iy, pr[i].ix);   // unclamped (warp_device.h)
#pragma unroll
  for (int i = 0; i < RPB; ++i) w1[i] = fastnum::id_finish(pr[i], s2[i], P, GB, fr[i]);
  if (__builtin_expect(fr[0] | fr[1] | fr[2] | fr[3], 0)) {
#pragma unroll
    for (int i = 0; i < RPB; ++i) if (fr[i]) w1[i] = warp_invdepth_px(SD, x, yb + i * TY, wv[i], P);
  }
  fastnum::IntensityTaps tp[RPB];
#pragma unroll
  for (int i = 0; i < RPB; ++i) {
    bool bd;
    tp[i] = fastnum::intensity_taps(SI, q[i], w1[i], P, GB, interp_mode, bd);
    if (__builtin_expect(bd, 0)) tp[i].ok = fastnum::intensity_fix_border(SI, q[i], x, yb + i * TY, w1[i], P, GB);
  }
#pragma unroll
  for (int i = 0; i < RPB; ++i) {
    bool nan_tap;
    i1[i] = fastnum::intensity_finish(tp[i], nan_tap);
    if (__builtin_expect(nan_tap, 0)) i1[i] = warp_intensity_px(SI, x, yb + i * TY, w1[i], P, interp_mode);   // a NaN tap: the oracle's texel pair decides
  }
#pragma unroll
  for (int i = 0; i < RPB; ++i) {
    int y = yb + i * TY;
    if (y < dst_iD.rows) { DW.store(y, x, w1[i]); DI.store(y, x, i1[i]); }
  }
}
bool launch_warp_pair_fast(hipStream_t s, int B, ImgB src_iD, ImgB src_I, ImgB grid, ImgB dst_iD, ImgB dst_I, const WarpParams* hp, const WarpParams* lp, int interp_mode, LaneMask m) {
  if (src_I.cols < 2 || src_I.rows < 2) return false;   // the bilinear taps need a 2 x 2 neighbourhood; the caller runs the exact kernel
  dim3 g = grid2d(dst_iD.cols, dst_iD.rows, B);
  TileMap tm;
  if (!make_tile_map((int)g.x, (int)g.y, B, &tm)) return false;
  if (lp) hipLaunchKernelGGL(k_warp_pair_fast<ByLane<WarpParams>>, dim3(tm.n), dim3(TX, TY), 0, s, src_iD, src_I, grid, dst_iD, dst_I, ByLane<WarpParams>{lp}, interp_mode, m, tm);
  else hipLaunchKernelGGL(k_warp_pair_fast<ByValue<WarpParams>>, dim3(tm.n), dim3(TX, TY), 0, s, src_iD, src_I, grid, dst_iD, dst_I, ByValue<WarpParams>{*hp}, interp_mode, m, tm);
  return true;
}

// device self-test of fastnum::cvt_flr (v_cvt_flr_i32_f32) against floor + saturating convert over a stride of the 2^32 float patterns (NaN excluded)
__global__ void k_selftest_cvt_flr(unsigned long long* mismatches, unsigned stride) {
  unsigned long long bad = 0;
  for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 32); u += (unsigned long long)gridDim.x * blockDim.x * stride) {
    const float x = __uint_as_float((unsigned)u);
    if (x == x && fastnum::cvt_flr(x) != cvt_rd(x)) ++bad;   // NaN: v_cvt_i32_f32 gives 0, v_cvt_flr_i32_f32 does not -- every index is clamped before it addresses memory
  }
  if (bad) atomicAdd(mismatches, bad);
}
void launch_selftest_cvt_flr(hipStream_t s, unsigned long long* mismatches_dev, unsigned stride) {
  hipLaunchKernelGGL(k_selftest_cvt_flr, dim3(4096), dim3(256), 0, s, mismatches_dev, stride);
}

// ---- trafo3DKernelInvDepthWeightedGridStride (:549-594) -------------------------------------------
template <class PS>
__global__ __launch_bounds__(256) void k_warp_invdepth_weighted(ImgB src, ImgB grid, ImgB dst, ImgB weight, PS ps, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x;
  const WarpParams P = ps.get(lane);
  const FMap S(src, lane);
  if (x >= dst.cols) return;
  // same structure as k_warp_invdepth: RPB independent pixels, straight-line exact reciprocals, one IEEE fallback
  const int yb = blockIdx.y * (TY * RPB) + threadIdx.y;
  float wv[RPB], out[RPB], wt[RPB];
  bool st[RPB];
#pragma unroll
  for (int i = 0; i < RPB; ++i) { int y = yb + i * TY; wv[i] = (y < dst.rows) ? px<float>(grid, lane, y, x) : qnan(); }
  RcpFast fast;
#pragma unroll
  for (int i = 0; i < RPB; ++i) out[i] = warp_invdepth_weighted_px_t(S, x, yb + i * TY, wv[i], P, fast, wt[i], st[i]);
  if (__builtin_expect(fast.failed(), 0)) {
    RcpIeee ieee;
#pragma unroll
    for (int i = 0; i < RPB; ++i) out[i] = warp_invdepth_weighted_px_t(S, x, yb + i * TY, wv[i], P, ieee, wt[i], st[i]);
  }
#pragma unroll
  for (int i = 0; i < RPB; ++i) {
    int y = yb + i * TY;
    if (y < dst.rows) {
      px<float>(dst, lane, y, x) = out[i];
      if (st[i]) px<float>(weight, lane, y, x) = wt[i];  // untouched otherwise, as the reference
    }
  }
}
void launch_warp_invdepth_weighted(hipStream_t s, int B, ImgB src, ImgB grid, ImgB dst, ImgB weight, const WarpParams* hp, const WarpParams* lp, LaneMask m) {
  dim3 g = grid2d(dst.cols, dst.rows, B), b(TX, TY);
  if (lp) hipLaunchKernelGGL(k_warp_invdepth_weighted<ByLane<WarpParams>>, g, b, 0, s, src, grid, dst, weight, ByLane<WarpParams>{lp}, m);
  else hipLaunchKernelGGL(k_warp_invdepth_weighted<ByValue<WarpParams>>, g, b, 0, s, src, grid, dst, weight, ByValue<WarpParams>{*hp}, m);
}

// ---- integrateWarpedFrameKernel (:637-669) --------------------------------------------------------
__global__ __launch_bounds__(256) void k_integrate(ImgB warped, ImgB wweight, ImgB kf, ImgB kfw, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x;
  RGBID_FOR_ROWS(y) {
    if (x >= kf.cols || y >= kf.rows) continue;
    float w_sum = px<float>(warped, lane, y, x);
    if (!isnan(w_sum)) {
      float w_KF = px<float>(kf, lane, y, x);
      float dw = fabsf(w_sum - w_KF);
      if (isnan(w_KF)) {
        px<float>(kf, lane, y, x) = w_sum;
        px<float>(kfw, lane, y, x) = px<float>(wweight, lane, y, x);
      } else if (dw < 3 * 0.0075f) {  // 3*DEPTHINV_INTEGR_TH (:80)
        float q = px<float>(kfw, lane, y, x), qs = px<float>(wweight, lane, y, x);
        float new_weight = q + qs;
        px<float>(kf, lane, y, x) = (w_KF * q + w_sum * qs) / new_weight;
        px<float>(kfw, lane, y, x) = new_weight;
      }
    }
  }
}
// 4 pixels per thread (16-byte accesses); pixels that keep their value are simply written back unchanged
__device__ __forceinline__ void integrate_px(float w_sum, float qs, float& w_KF, float& q) {
  if (!isnan(w_sum)) {
    float dw = fabsf(w_sum - w_KF);
    if (isnan(w_KF)) { w_KF = w_sum; q = qs; }
    else if (dw < 3 * 0.0075f) {
      float new_weight = q + qs;
      w_KF = (w_KF * q + w_sum * qs) / new_weight;
      q = new_weight;
    }
  }
}
__global__ __launch_bounds__(256) void k_integrate4(ImgB warped, ImgB wweight, ImgB kf, ImgB kfw, int cols4, int units, LaneMask m) {
  int lane = blockIdx.y;
  if (!m.on(lane)) return;
  for (int u = blockIdx.x * 256 + threadIdx.x; u < units; u += gridDim.x * 256) {
    int y = u / cols4, x = (u - y * cols4) * 4;
    float4 ws = *reinterpret_cast<const float4*>(row_ptr<float>(warped, lane, y) + x);
    float4 qs = *reinterpret_cast<const float4*>(row_ptr<float>(wweight, lane, y) + x);
    float4* kp = reinterpret_cast<float4*>(row_ptr<float>(kf, lane, y) + x);
    float4* qp = reinterpret_cast<float4*>(row_ptr<float>(kfw, lane, y) + x);
    float4 k = *kp, q = *qp;
    integrate_px(ws.x, qs.x, k.x, q.x); integrate_px(ws.y, qs.y, k.y, q.y);
    integrate_px(ws.z, qs.z, k.z, q.z); integrate_px(ws.w, qs.w, k.w, q.w);
    *kp = k; *qp = q;
  }
}
void launch_integrate_warped(hipStream_t s, int B, ImgB warped, ImgB wweight, ImgB kf, ImgB kfw, LaneMask m) {
  if (vec4_ok(warped) && vec4_ok(wweight) && vec4_ok(kf) && vec4_ok(kfw)) {
    int cols4 = kf.cols / 4, units = cols4 * kf.rows;
    hipLaunchKernelGGL(k_integrate4, dim3(div_up(units, 256 * 2), B), dim3(256), 0, s, warped, wweight, kf, kfw, cols4, units, m);
    return;
  }
  hipLaunchKernelGGL(k_integrate, grid2d(kf.cols, kf.rows, B), dim3(TX, TY), 0, s, warped, wweight, kf, kfw, m);
}

// ---- engine: integrateImagesIntoKeyframes (visodo.cpp:1674-1764) as ONE kernel -----------------------------------------------------
// warpInvDepthWithTrafo3DWeighted writes the warped inverse depth + its weight (8 B/px) which integrateWarpedFrame reads straight back
// together with the keyframe map and weight it updates in place.  Fused, a thread owns 4 consecutive keyframe pixels: the keyframe inverse
// depth is both the warp's sampling grid and the fusion's w_KF (one 16-byte load), the warped value and weight stay in registers, and the
// keyframe map / weight are written back once (40 -> ~25 B/px).  The reference's warped-weight buffer is kept EXACTLY as the two kernels
// leave it -- written only where the weight is positive, stale elsewhere -- because the fusion reads it wherever the warped value is
// valid, which can (only with infinite intermediates) include pixels whose weight was not stored: full groups store 16 bytes, mixed groups
// store their pixels one by one, and the stale value is fetched only in that exotic case.  Same device functions, bit-identical maps.
// FUSE_UNITS 4-pixel groups per thread, all loads of the thread's groups issued first, then all projections and gathers, then the blends and
// stores.  Measured at 1 024 lanes (tools/kernel_bench.py): 1 group 1.75-1.79 us per lane, 2 groups 1.87-1.94, 4 groups 1.84-1.92 (the
// round-2 grid-stride loop of 2 groups per thread: 1.83) -- the kernel is not bound by the latency of its chain but by its traffic: half of its
// 24 B/px are writes, read-modify-write of two maps in place.  One group per thread; and the FAST class does not carry the reference's
// warped-weight buffer at all (20 B/px): that buffer only exists between the reference's two kernels, and the one case in which the fused
// kernel reads it back -- a valid warped value whose weight was not positive, i.e. an infinite intermediate -- takes weight 0 there.
static constexpr int FUSE_UNITS = 1;
template <class PS, bool FAST>
__global__ __launch_bounds__(256) void k_fuse_frame4(ImgB src, ImgB kf, ImgB kfw, ImgB wweight, PS ps, int cols4, int units, LaneMask m) {
  // (round 4: the XCD-contiguous grid order that pays for the gather-only kernels was measured here too -- 1.09 instead of 1.01 us per lane: this
  // kernel rewrites two maps in place and the natural order spreads those writes over the XCDs; kept as it was)
  int lane = blockIdx.y;
  if (!m.on(lane)) return;
  const WarpParams P = ps.get(lane);
  const FMap S(src, lane);
  const int u0 = blockIdx.x * (256 * FUSE_UNITS) + threadIdx.x;   // the thread's groups: u0, u0 + 256, ... (a wave's loads stay contiguous)
  float4 k4[FUSE_UNITS], q4[FUSE_UNITS];
  int xs_[FUSE_UNITS], ys_[FUSE_UNITS];
  bool on[FUSE_UNITS];
#pragma unroll
  for (int g = 0; g < FUSE_UNITS; ++g) {
    const int u = u0 + g * 256;
    on[g] = u < units;
    const int y = on[g] ? u / cols4 : 0, x = on[g] ? (u - y * cols4) * 4 : 0;
    xs_[g] = x; ys_[g] = y;
    {  // the keyframe's inverse depth and weight are read once and rewritten below: non-temporal (-4 % FAST, -8 % EXACT per launch)
      typedef float f4v __attribute__((ext_vector_type(4)));
      const f4v a_ = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(row_ptr<float>(kf, lane, y) + x)),
                b_ = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(row_ptr<float>(kfw, lane, y) + x));
      k4[g] = make_float4(a_.x, a_.y, a_.z, a_.w); q4[g] = make_float4(b_.x, b_.y, b_.z, b_.w);
    }
  }
  float ws[FUSE_UNITS][4], wt[FUSE_UNITS][4], eps[FUSE_UNITS][4];
  bool st[FUSE_UNITS][4];
  if (FAST) {   // fast values, the oracle's selection (warp_device.h fastnum)
    const fastnum::Guard G = fastnum::lane_guard(P, kf.cols, kf.rows);
#pragma unroll
    for (int g = 0; g < FUSE_UNITS; ++g) {
      const float k[4] = {k4[g].x, k4[g].y, k4[g].z, k4[g].w};
      const fastnum::RowRay rr = fastnum::row_ray(P, (float)ys_[g]);
      // the four projections (+ the oracle's coordinates inside the guard band), then the four gathers, then the four values
      fastnum::IdProj pr[4];
      float w2[4];
      bool fr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bool fc;
        pr[i] = fastnum::id_project(fastnum::ray_at(P, rr, (float)(xs_[g] + i)), k[i], P, G, S.cols, S.rows, fc);
        if (__builtin_expect(fc, 0)) fastnum::id_fix_coords(pr[i], xs_[g] + i, ys_[g], P, S.cols, S.rows);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) w2[i] = S.at(pr[i].iy, pr[i].ix);   // unclamped (warp_device.h)
#pragma unroll
      for (int i = 0; i < 4; ++i) ws[g][i] = fastnum::id_finish_weighted(pr[i], w2[i], P, G, wt[g][i], st[g][i], eps[g][i], fr[i]);
      if (__builtin_expect(fr[0] | fr[1] | fr[2] | fr[3], 0)) {   // the sign of the oracle's value is not implied (guard_band.h (4)): the oracle's pixel
        RcpIeee ieee;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (fr[i]) { ws[g][i] = warp_invdepth_weighted_px_t(S, xs_[g] + i, ys_[g], k[i], P, ieee, wt[g][i], st[g][i]); eps[g][i] = 0.f; }
      }
    }
  } else {
    RcpFast fast;
#pragma unroll
    for (int g = 0; g < FUSE_UNITS; ++g) {
      const float k[4] = {k4[g].x, k4[g].y, k4[g].z, k4[g].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) ws[g][i] = warp_invdepth_weighted_px_t(S, xs_[g] + i, ys_[g], k[i], P, fast, wt[g][i], st[g][i]);
    }
    if (__builtin_expect(fast.failed(), 0)) {
      RcpIeee ieee;
#pragma unroll
      for (int g = 0; g < FUSE_UNITS; ++g) {
        const float k[4] = {k4[g].x, k4[g].y, k4[g].z, k4[g].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) ws[g][i] = warp_invdepth_weighted_px_t(S, xs_[g] + i, ys_[g], k[i], P, ieee, wt[g][i], st[g][i]);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < FUSE_UNITS; ++g) {
    if (!on[g]) continue;
    float k[4] = {k4[g].x, k4[g].y, k4[g].z, k4[g].w}, q[4] = {q4[g].x, q4[g].y, q4[g].z, q4[g].w};
    if (FAST) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float wsi = ws[g][i], qsi = st[g][i] ? wt[g][i] : 0.f;
        // the fusion gate |w_s - w_KF| < 3 * 0.0075 (:653) is open when the fast value lies within its error bound of the threshold
        // (guard_band.h (5)): that pixel is warped again with the oracle's instruction sequence and fuses ITS value and weight
        const float gap = fabsf(fabsf(wsi - k[i]) - 3 * 0.0075f);
        if (__builtin_expect(gap <= __builtin_fmaf(fabsf(wsi), eps[g][i], 4.f * 0x1p-24f * 0.0225f), 0)) {
          RcpIeee ieee;
          float wte; bool ste;
          wsi = warp_invdepth_weighted_px_t(S, xs_[g] + i, ys_[g], k[i], P, ieee, wte, ste);
          qsi = ste ? wte : 0.f;
        }
        integrate_px(wsi, qsi, k[i], q[i]);
      }
    } else {
      float* wp = row_ptr<float>(wweight, lane, ys_[g]) + xs_[g];
      const bool all_st = st[g][0] & st[g][1] & st[g][2] & st[g][3];
      if (all_st) *reinterpret_cast<float4*>(wp) = make_float4(wt[g][0], wt[g][1], wt[g][2], wt[g][3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float qs = wt[g][i];
        if (!st[g][i]) {
          if (!isnan(ws[g][i])) qs = wp[i];   // valid warped value whose weight was not stored: the reference fuses with the stale weight
        } else if (!all_st) wp[i] = wt[g][i];
        integrate_px(ws[g][i], qs, k[i], q[i]);
      }
    }
    st16_stream(row_ptr<float>(kf, lane, ys_[g]) + xs_[g], k[0], k[1], k[2], k[3]);
    st16_stream(row_ptr<float>(kfw, lane, ys_[g]) + xs_[g], q[0], q[1], q[2], q[3]);
  }
}
bool launch_fuse_frame(hipStream_t s, int B, ImgB src, ImgB kf, ImgB kfw, ImgB wweight, const WarpParams* lp, LaneMask m, bool fast) {
  if (!(vec4_ok(kf) && vec4_ok(kfw) && vec4_ok(wweight))) return false;   // caller falls back to the two kernels
  int cols4 = kf.cols / 4, units = cols4 * kf.rows;
  const dim3 g(div_up(units, 256 * FUSE_UNITS), B);
  if (fast) hipLaunchKernelGGL((k_fuse_frame4<ByLane<WarpParams>, true>), g, dim3(256), 0, s, src, kf, kfw, wweight, ByLane<WarpParams>{lp}, cols4, units, m);
  else hipLaunchKernelGGL((k_fuse_frame4<ByLane<WarpParams>, false>), g, dim3(256), 0, s, src, kf, kfw, wweight, ByLane<WarpParams>{lp}, cols4, units, m);
  return true;
}

// ---- integrateWarpedRGBKernel (:673-708): colour + inverse-depth fusion (bridge function; the reference's only caller,
// visodo.cpp:1826, sits in a routine trackNewFrame no longer invokes) ---------------------------------------------
__global__ __launch_bounds__(256) void k_integrate_rgb(ImgB warped, ImgB r_w, ImgB g_w, ImgB b_w, ImgB wweight, ImgB kf, ImgB colors, ImgB kfw, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x;
  const float TH = 0.0075f;  // DEPTHINV_INTEGR_TH (:80)
  RGBID_FOR_ROWS(y) {
    if (x >= kf.cols || y >= kf.rows) continue;
    float ws = px<float>(warped, lane, y, x), r = px<float>(r_w, lane, y, x), g = px<float>(g_w, lane, y, x), b = px<float>(b_w, lane, y, x);
    if (isnan(ws) || isnan(r) || isnan(g) || isnan(b)) continue;
    float wk = px<float>(kf, lane, y, x);
    uint8_t* c = row_ptr<uint8_t>(colors, lane, y) + 3 * x;
    float qs = px<float>(wweight, lane, y, x);
    if (isnan(wk)) {
      px<float>(kf, lane, y, x) = ws;
      c[0] = (uint8_t)f2i_rn(r); c[1] = (uint8_t)f2i_rn(g); c[2] = (uint8_t)f2i_rn(b);
      px<float>(kfw, lane, y, x) = qs;
    } else if (((wk - ws) < TH) && ((ws - wk) < TH)) {
      float q = px<float>(kfw, lane, y, x);
      float new_weight = q + qs;
      px<float>(kf, lane, y, x) = (wk * q + ws * qs) / new_weight;
      c[0] = (uint8_t)f2i_rn(((float)c[0] * q + r * qs) / new_weight);
      c[1] = (uint8_t)f2i_rn(((float)c[1] * q + g * qs) / new_weight);
      c[2] = (uint8_t)f2i_rn(((float)c[2] * q + b * qs) / new_weight);
      px<float>(kfw, lane, y, x) = new_weight;
    }
  }
}
void launch_integrate_warped_rgb(hipStream_t s, int B, ImgB warped, ImgB r, ImgB g, ImgB b, ImgB wweight, ImgB kf, ImgB colors, ImgB kfw, LaneMask m) {
  hipLaunchKernelGGL(k_integrate_rgb, grid2d(kf.cols, kf.rows, B), dim3(TX, TY), 0, s, warped, r, g, b, wweight, kf, colors, kfw, m);
}

// ---- partialVisibility(WithOverlapMask)Kernel (:297-437) -----------------------------------------
// The reference reduces float counters through shared memory + a second kernel + a per-call malloc.
// Here a workgroup sweeps a 64 x 32 pixel strip; every wave ballots its predicates, popcounts into scalar
// registers, the four waves meet in LDS and the workgroup issues ONE integer atomic per counter
// (150 atomics per counter for a 640x480 lane: no contention, exact integer counts).
static constexpr int VIS_ROWS = 96;   // rows per workgroup: the single-direction kernel mostly runs with few lanes on (overlap masks), so few workgroups
template <class PS>
__global__ __launch_bounds__(256) void k_visibility(ImgB src, ImgB dst, ImgB mask, PS ps, unsigned int* counts, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  __shared__ unsigned int sm[2][4];
  const WarpParams P = ps.get(lane);
  const FMap D(dst, lane);
  const int x = blockIdx.x * TX + threadIdx.x;
  const bool xin = x < src.cols;
  unsigned int nvis = 0, nval = 0;
  for (int g = 0; g < VIS_ROWS / (TY * 4); ++g) {
    // 4 independent pixels per thread and pass: loads, then 4 projection/gather chains, then the ballots
    const int yb = blockIdx.y * VIS_ROWS + g * (TY * 4) + threadIdx.y;
    float w[4];
    bool valid[4], visible[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { int y = yb + i * TY; w[i] = (xin && y < src.rows) ? px<float>(src, lane, y, x) : qnan(); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      valid[i] = !isnan(w[i]);
      float xd, yd;
      float w_dst = register_pixel(xd, yd, x, yb + i * TY, valid[i] ? w[i] : 1.f, P);
      bool inside_img = (xd > 0) && (xd < (float)(src.cols - 1)) && (yd > 0) && (yd < (float)(src.rows - 1));
      int xi = clampi(__float2int_rn(xd), src.cols - 1), yi = clampi(__float2int_rn(yd), src.rows - 1);
      visible[i] = valid[i] && inside_img && (fabsf(w_dst - D.at(yi, xi)) < 0.020f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (mask.base && valid[i]) px<uint8_t>(mask, lane, yb + i * TY, x) = visible[i] ? 1 : 0;
      nvis += (unsigned int)__popcll(__ballot(visible[i]));  // wave-uniform
      nval += (unsigned int)__popcll(__ballot(valid[i]));
    }
  }
  if (threadIdx.x == 0) { sm[0][threadIdx.y] = nvis; sm[1][threadIdx.y] = nval; }  // one wave per threadIdx.y (TX == 64)
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    unsigned int a = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3], b = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
    if (a) atomicAdd(&counts[2 * lane + 0], a);
    if (b) atomicAdd(&counts[2 * lane + 1], b);
  }
}
// engine: computeCovisibility (visodo.cpp:1481-1514) evaluates the ratio in BOTH directions between the same two maps.  One kernel does
// the pair: each thread loads its pixel of A and of B (coalesced) and runs the two projection -> gather -> gate chains side by side --
// the single-direction kernel is latency-bound (SQ counters: waves parked 75 % of their cycles), so the second independent chain is
// nearly free, and two of the four launches per frame disappear.  Counters: counts_ab / counts_ba as in k_visibility.
struct VisOther { ImgB Bm; const WarpParams *p_ab, *p_ba; unsigned int *counts_ab, *counts_ba; };
// o1: a second "other" map the same frame A is compared with in the same launch (trackNewFrame's two covisibility checks, against the odometry and the
// integration keyframe: one launch instead of two); its workgroups are the second half of the grid (wgs_per_other each)
template <bool FAST>
__global__ __launch_bounds__(256) void k_visibility_pair(ImgB A, VisOther o0, VisOther o1, int nbx, int nby, unsigned wgs_per_other, LaneMask m) {
  // 1-D grid in XCD-contiguous order (common.h xcd_slab_index): a lane's 64-pixel-wide strips run on one XCD, so the gather footprints neighbouring
  // strips share in the other map are fetched once (round 3: 1.15 x the algorithmic traffic with the natural order)
  unsigned V = xcd_slab_index(blockIdx.x, gridDim.x);
  const bool second = V >= wgs_per_other;
  if (second) V -= wgs_per_other;
  const VisOther& O = second ? o1 : o0;
  const ImgB& Bm = O.Bm;
  const WarpParams* const p_ab = O.p_ab; const WarpParams* const p_ba = O.p_ba;
  unsigned int* const counts_ab = O.counts_ab; unsigned int* const counts_ba = O.counts_ba;
  const unsigned per_lane = (unsigned)nbx * (unsigned)nby;
  const int lane = (int)(V / per_lane);
  const unsigned tl = V - (unsigned)lane * per_lane;
  const int by_ = (int)(tl / (unsigned)nbx), bx_ = (int)(tl - (unsigned)by_ * (unsigned)nbx);
  if (!m.on(lane)) return;
  __shared__ unsigned int sm[4][4];
  const WarpParams Pab = p_ab[lane], Pba = p_ba[lane];
  const FMap FA(A, lane), FB(Bm, lane);
  const int cols = A.cols, rows = A.rows;
  const fastnum::Guard Gab = FAST ? fastnum::lane_guard(Pab, cols, rows) : fastnum::Guard{}, Gba = FAST ? fastnum::lane_guard(Pba, cols, rows) : fastnum::Guard{};
  const int x = bx_ * TX + threadIdx.x;
  const bool xin = x < cols;
  unsigned int n[4] = {0, 0, 0, 0};  // visible a->b, valid a, visible b->a, valid b
  for (int g = 0; g < VIS_ROWS / (TY * 2); ++g) {
    const int yb = by_ * VIS_ROWS + g * (TY * 2) + threadIdx.y;
    float wa[2], wb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int y = yb + i * TY;
      bool in = xin && y < rows;
      wa[i] = in ? FA.at(y, x) : qnan();
      wb[i] = in ? FB.at(y, x) : qnan();
    }
    if (FAST) {
      // all four projections (+ the oracle's coordinates where the guard band asks for them), then the four gathers, then the four gates
      fastnum::VisProj pa[2], pb[2];
      float da[2], db[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int y = yb + i * TY;
        pa[i] = fastnum::vis_project(cols, rows, fastnum::ray(Pab, (float)x, (float)y), x, y, wa[i], Pab, Gab);
        pb[i] = fastnum::vis_project(cols, rows, fastnum::ray(Pba, (float)x, (float)y), x, y, wb[i], Pba, Gba);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) { da[i] = FB.at(pa[i].yi, pa[i].xi); db[i] = FA.at(pb[i].yi, pb[i].xi); }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int y = yb + i * TY;
        const bool va = !isnan(wa[i]), vb = !isnan(wb[i]);
        const bool sa = va & fastnum::vis_gate(pa[i], da[i], x, y, Pab, Gab), sb = vb & fastnum::vis_gate(pb[i], db[i], x, y, Pba, Gba);
        n[0] += (unsigned int)__popcll(__ballot(sa)); n[1] += (unsigned int)__popcll(__ballot(va));
        n[2] += (unsigned int)__popcll(__ballot(sb)); n[3] += (unsigned int)__popcll(__ballot(vb));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int y = yb + i * TY;
        const bool va = !isnan(wa[i]), vb = !isnan(wb[i]);
        const bool sa = visible_px_exact(FB, cols, rows, x, y, wa[i], va, Pab), sb = visible_px_exact(FA, cols, rows, x, y, wb[i], vb, Pba);
        n[0] += (unsigned int)__popcll(__ballot(sa)); n[1] += (unsigned int)__popcll(__ballot(va));
        n[2] += (unsigned int)__popcll(__ballot(sb)); n[3] += (unsigned int)__popcll(__ballot(vb));
      }
    }
  }
  if (threadIdx.x == 0) { for (int k = 0; k < 4; ++k) sm[k][threadIdx.y] = n[k]; }  // one wave per threadIdx.y (TX == 64)
  __syncthreads();
  if (threadIdx.x < 4 && threadIdx.y == 0) {
    const int k = threadIdx.x;
    unsigned int t = sm[k][0] + sm[k][1] + sm[k][2] + sm[k][3];
    unsigned int* c = (k < 2 ? counts_ab : counts_ba) + 2 * lane + (k & 1);
    if (t) atomicAdd(c, t);
  }
}
void launch_visibility_pair(hipStream_t s, int B, ImgB a, ImgB b, const WarpParams* p_ab, const WarpParams* p_ba, unsigned int* counts_ab,
                            unsigned int* counts_ba, LaneMask m, bool fast) {
  const int nbx = div_up(a.cols, TX), nby = div_up(a.rows, VIS_ROWS);
  const unsigned wgs = (unsigned)nbx * (unsigned)nby * (unsigned)B;
  dim3 g(wgs), blk(TX, TY);
  const VisOther o{b, p_ab, p_ba, counts_ab, counts_ba};
  if (fast) hipLaunchKernelGGL(k_visibility_pair<true>, g, blk, 0, s, a, o, o, nbx, nby, wgs, m);
  else hipLaunchKernelGGL(k_visibility_pair<false>, g, blk, 0, s, a, o, o, nbx, nby, wgs, m);
}
// frame `a` against two other maps of the same geometry in one launch
void launch_visibility_pair2(hipStream_t s, int B, ImgB a, ImgB b0, const WarpParams* p_ab0, const WarpParams* p_ba0, unsigned int* counts_ab0, unsigned int* counts_ba0,
                             ImgB b1, const WarpParams* p_ab1, const WarpParams* p_ba1, unsigned int* counts_ab1, unsigned int* counts_ba1, LaneMask m, bool fast) {
  const int nbx = div_up(a.cols, TX), nby = div_up(a.rows, VIS_ROWS);
  const unsigned wgs = (unsigned)nbx * (unsigned)nby * (unsigned)B;
  dim3 g(2u * wgs), blk(TX, TY);
  const VisOther o0{b0, p_ab0, p_ba0, counts_ab0, counts_ba0}, o1{b1, p_ab1, p_ba1, counts_ab1, counts_ba1};
  if (fast) hipLaunchKernelGGL(k_visibility_pair<true>, g, blk, 0, s, a, o0, o1, nbx, nby, wgs, m);
  else hipLaunchKernelGGL(k_visibility_pair<false>, g, blk, 0, s, a, o0, o1, nbx, nby, wgs, m);
}

void launch_visibility(hipStream_t s, int B, ImgB src, ImgB dst, ImgB mask, const WarpParams* hp, const WarpParams* lp, unsigned int* counts, LaneMask m) {
  dim3 g(div_up(src.cols, TX), div_up(src.rows, VIS_ROWS), B), b(TX, TY);
  if (lp) hipLaunchKernelGGL(k_visibility<ByLane<WarpParams>>, g, b, 0, s, src, dst, mask, ByLane<WarpParams>{lp}, counts, m);
  else hipLaunchKernelGGL(k_visibility<ByValue<WarpParams>>, g, b, 0, s, src, dst, mask, ByValue<WarpParams>{*hp}, counts, m);
}

// ---- computeVmapKernel (maps.cu:63-90) -----------------------------------------------------------
__global__ __launch_bounds__(256) void k_vmap(ImgB depthinv, ImgB vmap, IntrP k, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int u = blockIdx.x * TX + threadIdx.x;
  int rows = depthinv.rows;
  float fx_inv = 1.f / k.fx, fy_inv = 1.f / k.fy;
  RGBID_FOR_ROWS(v) {
    if (u >= depthinv.cols || v >= rows) continue;
    float z = rcp_exact(px<float>(depthinv, lane, v, u));
    if (!isnan(z)) {
      px<float>(vmap, lane, v, u) = z * ((float)u - k.cx) * fx_inv;
      px<float>(vmap, lane, v + rows, u) = z * ((float)v - k.cy) * fy_inv;
      px<float>(vmap, lane, v + 2 * rows, u) = z;
    } else {
      px<float>(vmap, lane, v, u) = qnan();
    }
  }
}
// 4 pixels per thread, 16-byte stores.  The reference leaves planes 1/2 untouched at invalid pixels: a group with an invalid pixel
// reads the old vector back and blends (read-modify-write), every group then stores full vectors.
__device__ __forceinline__ float4 blend4(float4 oldv, const float n[4], const bool keep[4]) {
  return make_float4(keep[0] ? n[0] : oldv.x, keep[1] ? n[1] : oldv.y, keep[2] ? n[2] : oldv.z, keep[3] ? n[3] : oldv.w);
}
__global__ __launch_bounds__(256) void k_vmap4(ImgB depthinv, ImgB vmap, IntrP k, int cols4, int units, LaneMask m) {
  int lane = blockIdx.y;
  if (!m.on(lane)) return;
  const int rows = depthinv.rows;
  const float fx_inv = 1.f / k.fx, fy_inv = 1.f / k.fy;
  for (int un = blockIdx.x * 256 + threadIdx.x; un < units; un += gridDim.x * 256) {
    int v = un / cols4, u = (un - v * cols4) * 4;
    float4 w = *reinterpret_cast<const float4*>(row_ptr<float>(depthinv, lane, v) + u);
    float wi[4] = {w.x, w.y, w.z, w.w}, X[4], Y[4], Z[4];
    bool ok[4], all = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float z = rcp_exact(wi[i]);
      ok[i] = !isnan(z);
      all = all && ok[i];
      X[i] = ok[i] ? z * ((float)(u + i) - k.cx) * fx_inv : qnan();
      Y[i] = z * ((float)v - k.cy) * fy_inv;
      Z[i] = z;
    }
    float4* p0 = reinterpret_cast<float4*>(row_ptr<float>(vmap, lane, v) + u);
    float4* p1 = reinterpret_cast<float4*>(row_ptr<float>(vmap, lane, v + rows) + u);
    float4* p2 = reinterpret_cast<float4*>(row_ptr<float>(vmap, lane, v + 2 * rows) + u);
    float4 y4 = make_float4(Y[0], Y[1], Y[2], Y[3]), z4 = make_float4(Z[0], Z[1], Z[2], Z[3]);
    if (!all) { y4 = blend4(*p1, Y, ok); z4 = blend4(*p2, Z, ok); }
    *p0 = make_float4(X[0], X[1], X[2], X[3]);
    *p1 = y4;
    *p2 = z4;
  }
}
void launch_vmap(hipStream_t s, int B, ImgB depthinv, ImgB vmap, IntrP k, LaneMask m) {
  if (vec4_ok(depthinv) && vec4_ok(vmap)) {
    int cols4 = depthinv.cols / 4, units = cols4 * depthinv.rows;
    hipLaunchKernelGGL(k_vmap4, dim3(div_up(units, 256 * 2), B), dim3(256), 0, s, depthinv, vmap, k, cols4, units, m);
    return;
  }
  hipLaunchKernelGGL(k_vmap, grid2d(depthinv.cols, depthinv.rows, B), dim3(TX, TY), 0, s, depthinv, vmap, k, m);
}

// ---- computeNmapKernel (maps.cu:92-133): normals from the cross product of forward differences of the vertex map ----
__global__ __launch_bounds__(256) void k_nmap_cross(ImgB vmap, ImgB nmap, int rows, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int u = blockIdx.x * TX + threadIdx.x;
  int cols = vmap.cols;
  RGBID_FOR_ROWS(v) {
    if (u >= cols || v >= rows) continue;
    float n0 = qnan();
    if (!(u == cols - 1 || v == rows - 1)) {
      float ax = px<float>(vmap, lane, v, u), bx = px<float>(vmap, lane, v, u + 1), cx = px<float>(vmap, lane, v + 1, u);
      if (!isnan(ax) && !isnan(bx) && !isnan(cx)) {
        float ay = px<float>(vmap, lane, v + rows, u), by = px<float>(vmap, lane, v + rows, u + 1), cy = px<float>(vmap, lane, v + 1 + rows, u);
        float az = px<float>(vmap, lane, v + 2 * rows, u), bz = px<float>(vmap, lane, v + 2 * rows, u + 1), cz = px<float>(vmap, lane, v + 1 + 2 * rows, u);
        float d1x = bx - ax, d1y = by - ay, d1z = bz - az, d2x = cx - ax, d2y = cy - ay, d2z = cz - az;
        float rx = d1y * d2z - d1z * d2y, ry = d1z * d2x - d1x * d2z, rz = d1x * d2y - d1y * d2x;
        float inv = rcp_exact(sqrtf(rx * rx + ry * ry + rz * rz));
        n0 = rx * inv;
        px<float>(nmap, lane, v + rows, u) = ry * inv;
        px<float>(nmap, lane, v + 2 * rows, u) = rz * inv;
      }
    }
    px<float>(nmap, lane, v, u) = n0;
  }
}
void launch_nmap_cross(hipStream_t s, int B, ImgB vmap, ImgB nmap, LaneMask m) {
  int rows = vmap.rows / 3;
  hipLaunchKernelGGL(k_nmap_cross, grid2d(vmap.cols, rows, B), dim3(TX, TY), 0, s, vmap, nmap, rows, m);
}

// ---- computeNmapGradientsKernel (maps.cu:134-179) -------------------------------------------------
__global__ __launch_bounds__(256) void k_nmap_grad(ImgB depthinv, ImgB gx_, ImgB gy_, ImgB nmap, IntrP k, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int u = blockIdx.x * TX + threadIdx.x;
  int rows = depthinv.rows;
  RGBID_FOR_ROWS(v) {
    if (u >= depthinv.cols || v >= rows) continue;
    float w = px<float>(depthinv, lane, v, u), gx = px<float>(gx_, lane, v, u), gy = px<float>(gy_, lane, v, u);
    float n0 = qnan();
    if (!(isnan(w) || isnan(gx) || isnan(gy))) {
      float nx = gx * k.fx, ny = gy * k.fy, nz = gx * (k.cx - (float)u) + gy * (k.cy - (float)v) + w;
      float rn = rcp_exact(sqrtf(nx * nx + ny * ny + nz * nz));
      nx *= rn; ny *= rn; nz *= rn;
      float z = rcp_exact(w);
      float vx = z * ((float)u - k.cx) * (1.f / k.fx), vy = z * ((float)v - k.cy) * (1.f / k.fy), vz = z;
      float rv = rcp_exact(sqrtf(vx * vx + vy * vy + vz * vz));
      vx *= rv; vy *= rv; vz *= rv;
      float acos_vn = vx * nx + vy * ny + vz * nz;
      if ((double)acos_vn > 0.1) {
        n0 = nx;
        px<float>(nmap, lane, v + rows, u) = ny;
        px<float>(nmap, lane, v + 2 * rows, u) = nz;
      }
    }
    px<float>(nmap, lane, v, u) = n0;
  }
}
__global__ __launch_bounds__(256) void k_nmap_grad4(ImgB depthinv, ImgB gx_, ImgB gy_, ImgB nmap, IntrP k, int cols4, int units, LaneMask m) {
  int lane = blockIdx.y;
  if (!m.on(lane)) return;
  const int rows = depthinv.rows;
  for (int un = blockIdx.x * 256 + threadIdx.x; un < units; un += gridDim.x * 256) {
    int v = un / cols4, u = (un - v * cols4) * 4;
    float4 w4 = *reinterpret_cast<const float4*>(row_ptr<float>(depthinv, lane, v) + u);
    float4 gx4 = *reinterpret_cast<const float4*>(row_ptr<float>(gx_, lane, v) + u);
    float4 gy4 = *reinterpret_cast<const float4*>(row_ptr<float>(gy_, lane, v) + u);
    float wi[4] = {w4.x, w4.y, w4.z, w4.w}, gxi[4] = {gx4.x, gx4.y, gx4.z, gx4.w}, gyi[4] = {gy4.x, gy4.y, gy4.z, gy4.w};
    float N0[4], N1[4], N2[4];
    bool keep[4], all = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float w = wi[i], gx = gxi[i], gy = gyi[i];
      float uu = (float)(u + i);
      float nx = gx * k.fx, ny = gy * k.fy, nz = gx * (k.cx - uu) + gy * (k.cy - (float)v) + w;
      float rn = rcp_exact(sqrtf(nx * nx + ny * ny + nz * nz));
      nx *= rn; ny *= rn; nz *= rn;
      float z = rcp_exact(w);
      float vx = z * (uu - k.cx) * (1.f / k.fx), vy = z * ((float)v - k.cy) * (1.f / k.fy), vz = z;
      float rv = rcp_exact(sqrtf(vx * vx + vy * vy + vz * vz));
      vx *= rv; vy *= rv; vz *= rv;
      float acos_vn = vx * nx + vy * ny + vz * nz;
      keep[i] = !(isnan(w) || isnan(gx) || isnan(gy)) && ((double)acos_vn > 0.1);
      all = all && keep[i];
      N0[i] = keep[i] ? nx : qnan(); N1[i] = ny; N2[i] = nz;
    }
    float4* p0 = reinterpret_cast<float4*>(row_ptr<float>(nmap, lane, v) + u);
    float4* p1 = reinterpret_cast<float4*>(row_ptr<float>(nmap, lane, v + rows) + u);
    float4* p2 = reinterpret_cast<float4*>(row_ptr<float>(nmap, lane, v + 2 * rows) + u);
    float4 a = make_float4(N1[0], N1[1], N1[2], N1[3]), b = make_float4(N2[0], N2[1], N2[2], N2[3]);
    if (!all) { a = blend4(*p1, N1, keep); b = blend4(*p2, N2, keep); }
    *p0 = make_float4(N0[0], N0[1], N0[2], N0[3]);
    *p1 = a;
    *p2 = b;
  }
}
void launch_nmap_gradients(hipStream_t s, int B, ImgB depthinv, ImgB gx, ImgB gy, ImgB nmap, IntrP k, LaneMask m) {
  if (vec4_ok(depthinv) && vec4_ok(gx) && vec4_ok(gy) && vec4_ok(nmap)) {
    int cols4 = depthinv.cols / 4, units = cols4 * depthinv.rows;
    hipLaunchKernelGGL(k_nmap_grad4, dim3(div_up(units, 256 * 2), B), dim3(256), 0, s, depthinv, gx, gy, nmap, k, cols4, units, m);
    return;
  }
  hipLaunchKernelGGL(k_nmap_grad, grid2d(depthinv.cols, depthinv.rows, B), dim3(TX, TY), 0, s, depthinv, gx, gy, nmap, k, m);
}

// ---- ImageGenerator(RGB) (image_generator.cu:66-185) ----------------------------------------------
template <class PS>
__global__ __launch_bounds__(256) void k_generate_image(ImgB vmap, ImgB nmap, ImgB rgb, ImgB dst, PS ps, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x, y = blockIdx.y * TY + threadIdx.y;
  if (x >= dst.cols || y >= dst.rows) return;
  int rows = dst.rows;
  uint8_t c0 = 0, c1 = 0, c2 = 0;
  float vx = px<float>(vmap, lane, y, x), nx = px<float>(nmap, lane, y, x);
  if (!isnan(vx) && !isnan(nx)) {
    const LightP& L = ps.get(lane);
    float vy = px<float>(vmap, lane, y + rows, x), vz = px<float>(vmap, lane, y + 2 * rows, x);
    float ny = px<float>(nmap, lane, y + rows, x), nz = px<float>(nmap, lane, y + 2 * rows, x);
    float dx = L.x - vx, dy = L.y - vy, dz = L.z - vz;
    float rd = rcp_exact(sqrtf(dx * dx + dy * dy + dz * dz));
    dx *= rd; dy *= rd; dz *= rd;
    float weight = 1.f;
    weight *= fabsf(dx * nx + dy * ny + dz * nz);
    int br = (int)(205 * weight) + 50;
    br = max(0, min(255, br));
    if (rgb.base) {
      float br_f = (float)br / 255.f;
      const uint8_t* p = row_ptr<uint8_t>(rgb, lane, y) + 3 * x;
      c0 = (uint8_t)f2i_rn((float)p[0] * br_f);
      c1 = (uint8_t)f2i_rn((float)p[1] * br_f);
      c2 = (uint8_t)f2i_rn((float)p[2] * br_f);
    } else {
      c0 = c1 = c2 = (uint8_t)br;
    }
  }
  uint8_t* d = row_ptr<uint8_t>(dst, lane, y) + 3 * x;
  d[0] = c0; d[1] = c1; d[2] = c2;
}
void launch_generate_image(hipStream_t s, int B, ImgB vmap, ImgB nmap, ImgB rgb, ImgB dst, const LightP* hl, const LightP* ll, LaneMask m) {
  dim3 g = grid2d_full(dst.cols, dst.rows, B), b(TX, TY);
  if (ll) hipLaunchKernelGGL(k_generate_image<ByLane<LightP>>, g, b, 0, s, vmap, nmap, rgb, dst, ByLane<LightP>{ll}, m);
  else hipLaunchKernelGGL(k_generate_image<ByValue<LightP>>, g, b, 0, s, vmap, nmap, rgb, dst, ByValue<LightP>{*hl}, m);
}

}  // namespace rgbid
