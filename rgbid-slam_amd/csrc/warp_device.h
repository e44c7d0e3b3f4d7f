// warp_device.h -- per-pixel inverse warps shared by the stand-alone warp kernels (kernels_warp.hip), the fused
// Gauss-Newton kernel (kernels_system.hip) and the lattice pre-pass of the sigma/nu kernel (kernels_sigma.hip).
// All three therefore produce bit-identical W1 / I1 values.
//
// Numerics: fp contraction is off inside these functions and the divisions / reciprocals are IEEE-exact (common.h rcp_exact), so every value and every
// floor()/rint() pixel selection is bit-identical to the scalar oracle.  Control flow: branch-free.  The CUDA
// kernels nest three `if`s per pixel (valid iD, in bounds, res > 0); a wave64 would serialise all of them, so the
// arithmetic runs unconditionally on sanitised inputs, every gather uses a clamped (always legal) address, and
// the predicates only select the final value.  v_cvt_i32_f32 saturates and maps NaN to 0 -- exactly CUDA's
// __float2int_rd/_rn semantics that registerPixel's callers rely on -- so the conversions need no range checks.
#pragma once
#include "common.h"

namespace rgbid {

__device__ __forceinline__ int cvt_rd(float x) { return __float2int_rd(x); }  // v_floor_f32 + v_cvt_i32_f32 (saturating)
__device__ __forceinline__ bool inside(int ix, int iy, int cols, int rows) {
  return ((unsigned)ix < (unsigned)cols) & ((unsigned)iy < (unsigned)rows);  // == !(ix<0 || iy<0 || ix>=cols || iy>=rows)
}
// clamp(v, lo, hi) as ONE v_med3_i32 (lo <= hi always holds here: hi = rows - 1 or cols - 1 of a non-empty image); the compiler keeps
// min(max()) as two instructions because it cannot prove the ordering, and the gather kernels are VALU-bound
__device__ __forceinline__ int clampi(int v, int hi) { int r; asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(hi)); return r; }
__device__ __forceinline__ int clamp_from_m1(int v, int hi) { int r; asm("v_med3_i32 %0, %1, -1, %2" : "=v"(r) : "v"(v), "v"(hi)); return r; }

// lane-local view of a float map for gathers: a raw buffer descriptor over the lane's image (wave-uniform: the lane is a block index) and
// 32-bit BYTE offsets formed with 24-bit multiplies.  v_mul_lo_u32 issues at quarter rate and a 64-bit flat address costs two more VALU
// instructions per access, which matters in the VALU-bound warps (rows, pitch < 2^24 and the image < 4 GB: checked at the C-ABI).  The EXACT
// functions clamp every coordinate to the image before it gets here; the fastnum functions do NOT clamp the coordinates of pixels they
// reject anyway and rely on the descriptor's range check (an offset outside the lane's image reads 0, a wrapped one some other texel of the
// same image: either value is discarded by the pixel's predicate).
struct FMap {
  const float* base;
  int pitch, rows, cols;
  __amdgpu_buffer_rsrc_t rsrc;   // raw buffer descriptor over the lane's image: loads take a 32-bit BYTE offset, no 64-bit address arithmetic
  unsigned pitch_b;
  __device__ __forceinline__ FMap(const ImgB& im, int lane)
      : base(row_ptr<float>(im, lane, 0)), pitch((int)(im.pitch >> 2)), rows(im.rows), cols(im.cols),
        rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row_ptr<float>(im, lane, 0)), 0, (int)((unsigned)im.rows * (unsigned)im.pitch), 0x00020000)),
        pitch_b((unsigned)im.pitch) {}
  __device__ __forceinline__ unsigned row(int y) const { return __umul24((unsigned)y, pitch_b); }              // byte offset of a row
  __device__ __forceinline__ float at_off(unsigned row_b, int x) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, row_b + ((unsigned)x << 2), 0, 0));
  }
  __device__ __forceinline__ float at(int y, int x) const { return at_off(row(y), x); }
  // two horizontally adjacent texels (x, x + 1) in ONE 8-byte load: the texture addresser handles 4 lanes per clock whatever the access
  // width, so a gather kernel's time is its number of vector-memory instructions, not its bytes (x + 1 must be a valid column)
  __device__ __forceinline__ float2 at2_off(unsigned row_b, int x) const {
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, row_b + ((unsigned)x << 2), 0, 0));
  }
  // the same with the byte offset split into a per-pixel part and a wave-uniform part (the instruction's scalar offset operand)
  __device__ __forceinline__ float2 at2_raw(unsigned voff, unsigned soff) const {
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0));
  }
  __device__ __forceinline__ float4 at4(int y, int x) const {   // x % 4 == 0, 16-byte aligned rows
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, row(y) + ((unsigned)x << 2), 0, 0));
  }
};

// writable counterpart (kernel outputs): raw buffer stores with a 32-bit byte offset (a 24-bit multiply-add) on a wave-uniform descriptor
struct FMapW {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned pitch_b;
  __device__ __forceinline__ FMapW(const ImgB& im, int lane)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(row_ptr<float>(im, lane, 0), 0, (int)((unsigned)im.rows * (unsigned)im.pitch), 0x00020000)),
        pitch_b((unsigned)im.pitch) {}
  __device__ __forceinline__ void store(int y, int x, float v) const {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rsrc, __umul24((unsigned)y, pitch_b) + ((unsigned)x << 2), 0, 0);
  }
  typedef int v4i __attribute__((ext_vector_type(4)));
  __device__ __forceinline__ void store4(int y, int x, float4 v) const {   // x % 4 == 0, 16-byte aligned rows
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), rsrc, __umul24((unsigned)y, pitch_b) + ((unsigned)x << 2), 0, 0);
  }
};

// CUDA linear filtering at unnormalised coordinates with clamp addressing (what tex2D<float> computes for the
// reference's cudaFilterModeLinear texture, warping_registration.cu:938-944); mode 1 = 1.8 fixed-point weights
__device__ __forceinline__ float tex2d_linear(const FMap& src, float xs, float ys, int mode) {
#pragma clang fp contract(off)
  float xB = xs - 0.5f, yB = ys - 0.5f;
  float fx0 = floorf(xB), fy0 = floorf(yB);
  float a = xB - fx0, b = yB - fy0;
  if (mode == 1) {
    a = rintf(a * 256.f) * 0.00390625f;
    b = rintf(b * 256.f) * 0.00390625f;
  }
  // clamp addressing of both taps, i0 -> clamp(i0, 0, n-1), i1 -> clamp(i0 + 1, 0, n-1), evaluated so that the saturated
  // conversions (huge / infinite coordinates give INT_MAX) can never overflow the `+ 1` (signed overflow is undefined behaviour the
  // optimiser may exploit): c = clamp(i0, -1, n-1); i1 = min(c + 1, n-1); i0 = max(c, 0)  -- the same values in 5 instead of 6 ops
  const int ic = clamp_from_m1(__float2int_rd(fx0), src.cols - 1), jc = clamp_from_m1(__float2int_rd(fy0), src.rows - 1);
  const int i1 = min(ic + 1, src.cols - 1), j1 = min(jc + 1, src.rows - 1);
  const int i0 = max(ic, 0), j0 = max(jc, 0);
  const unsigned r0 = src.row(j0), r1 = src.row(j1);
  float T00 = src.at_off(r0, i0), T10 = src.at_off(r0, i1), T01 = src.at_off(r1, i0), T11 = src.at_off(r1, i1);
  float oa = 1.f - a, ob = 1.f - b;
  return (oa * ob) * T00 + (a * ob) * T10 + (oa * b) * T01 + (a * b) * T11;
}

// trafo3DKernelInvDepthGridStride, warping_registration.cu:505-546 (one pixel; w = keyframe inverse depth)
template <class RCP>
__device__ __forceinline__ float warp_invdepth_px_t(const FMap& src, int x, int y, float w, const WarpParams& P, RCP& rcp) {
#pragma clang fp contract(off)
  const bool valid = !isnan(w);
  const float ws = valid ? w : 1.f;
  float xs, ys;
  float w3 = register_pixel_t(xs, ys, x, y, ws, P, rcp);
  xs += 0.5f; ys += 0.5f;
  int ix = cvt_rd(xs), iy = cvt_rd(ys);
  const bool inb = inside(ix, iy, src.cols, src.rows);
  float w2 = src.at(clampi(iy, src.rows - 1), clampi(ix, src.cols - 1));
  float tz = P.t[2];
  float v1_z = (rcp(w3) - tz) * ws;
  float res = (v1_z / (1.f - w2 * tz)) * w2;
  return (valid & inb & (res > 0.f)) ? res : qnan();
}
__device__ __forceinline__ float warp_invdepth_px(const FMap& src, int x, int y, float w, const WarpParams& P) {
  RcpFast f;
  float res = warp_invdepth_px_t(src, x, y, w, P, f);
  if (__builtin_expect(f.failed(), 0)) { RcpIeee s; res = warp_invdepth_px_t(src, x, y, w, P, s); }
  return res;
}

// trafo3DKernelInvDepthWeightedGridStride, warping_registration.cu:549-594 (one pixel): the warped inverse depth plus the weight
// (1 - w2 tz)^4 / v^2; `store_weight` tells the caller whether the reference would have written the weight (only when > 0)
template <class RCP>
__device__ __forceinline__ float warp_invdepth_weighted_px_t(const FMap& src, int x, int y, float w, const WarpParams& P, RCP& rcp,
                                                              float& weight_res, bool& store_weight) {
#pragma clang fp contract(off)
  const bool valid = !isnan(w);
  const float ws = valid ? w : 1.f;
  float xs, ys;
  float w3 = register_pixel_t(xs, ys, x, y, ws, P, rcp);
  xs += 0.5f; ys += 0.5f;
  int ix = cvt_rd(xs), iy = cvt_rd(ys);
  const bool inb = inside(ix, iy, src.cols, src.rows);
  float w2 = src.at(clampi(iy, src.rows - 1), clampi(ix, src.cols - 1));
  float tz = P.t[2];
  float v1_z = (rcp(w3) - tz) * ws;
  float w_factor = 1.f - w2 * tz;
  float w_factor2 = w_factor * w_factor;
  weight_res = (w_factor2 * w_factor2) / (v1_z * v1_z);
  float res = (v1_z / w_factor) * w2;
  store_weight = valid & inb & (weight_res > 0.f);
  return (valid & inb & (res > 0.f)) ? res : qnan();
}

// trafo3DKernelIntensityWithInvDepthGridStride, warping_registration.cu:465-501 (one pixel; w = sampling-grid iD)
template <class RCP>
__device__ __forceinline__ float warp_intensity_px_t(const FMap& src, int x, int y, float w, const WarpParams& P, int interp_mode, RCP& rcp) {
#pragma clang fp contract(off)
  const bool valid = !isnan(w);
  const float ws = valid ? w : 1.f;
  float xs, ys;
  register_pixel_t(xs, ys, x, y, ws, P, rcp);
  xs += 0.5f; ys += 0.5f;
  const bool inb = inside(cvt_rd(xs), cvt_rd(ys), src.cols, src.rows);
  float res = tex2d_linear(src, xs, ys, interp_mode);
  res = fmaxf(0.f, fminf(res, 255.f));  // NaN -> 255, as CUDA's min/max
  return (valid & inb) ? res : qnan();
}
__device__ __forceinline__ float warp_intensity_px(const FMap& src, int x, int y, float w, const WarpParams& P, int interp_mode) {
  RcpFast f;
  float res = warp_intensity_px_t(src, x, y, w, P, interp_mode, f);
  if (__builtin_expect(f.failed(), 0)) { RcpIeee s; res = warp_intensity_px_t(src, x, y, w, P, interp_mode, s); }
  return res;
}

// =====================================================================================================================================
// "Reference-build-class" numerics for the ENGINE's gather kernels (rgbid_engine_config.fast_numerics, default on).
//
// The reference compiles its kernels with --prec-div=false --prec-sqrt=false --ftz=true (CMakeLists.txt:105) and nvcc's default FMA
// contraction: its own pixel selection is that of approximate reciprocals and fused multiply-adds, not of IEEE arithmetic.  The functions
// above reproduce the IEEE evaluation of the scalar oracle bit for bit (the compat bridge always uses them); the functions below evaluate
// the same formulas in the reference build's class of arithmetic -- v_rcp_f32 (1 ulp) for every division, explicit FMAs, the ray
// q = R (x, y, 1) formed once per pixel and shared by every projection from that pixel (registerPixel, warping_registration.cu:129-146,
// re-associated: R (x z, y z, z) + t = z q + t, and projected from the scaled point w X = q + w t so that no reciprocal of the inverse depth
// is taken), 1 / (1 / X.z) taken as X.z -- which removes ~45 % of the instructions of kernels that
// are VALU-bound.  What changes: a coordinate that lands within an ulp of a pixel boundary may select the neighbouring pixel.  Measured
// against the exact kernels (tests/test_gpu_engine.py::test_engine_fast_numerics_vs_exact): a few boundary pixels per map, poses
// within 1e-6; the oracle's model of the reference's nvcc flags moves poses by the same order (tests/test_oracle_cuda_numerics.py).
namespace fastnum {

// Rejected pixels (coordinates outside the image, invalid inverse depth) are not clamped to a legal address first, as the exact functions do:
// the gathers go through the map's raw buffer descriptor, whose range check returns 0 for any offset outside the lane's image, an offset
// that wraps lands on some other texel of the same image, and either value is discarded by the pixel's predicate.
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// floor + saturating convert in one instruction (NaN -> 0); verified against v_floor_f32 + v_cvt_i32_f32 by rgbid_selftest_cvt_flr
__device__ __forceinline__ int cvt_flr(float x) { int r; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }

struct Ray { float q0, q1, q2; };   // K R^-1 K^-1 (x, y, 1)
__device__ __forceinline__ Ray ray(const WarpParams& P, float xf, float yf) {
  Ray r;
  r.q0 = __builtin_fmaf(P.R[0], xf, __builtin_fmaf(P.R[1], yf, P.R[2]));
  r.q1 = __builtin_fmaf(P.R[3], xf, __builtin_fmaf(P.R[4], yf, P.R[5]));
  r.q2 = __builtin_fmaf(P.R[6], xf, __builtin_fmaf(P.R[7], yf, P.R[8]));
  return r;
}
__device__ __forceinline__ Ray ray_step(const Ray& r, float d0, float d1, float d2) { return Ray{r.q0 + d0, r.q1 + d1, r.q2 + d2}; }
// The back-projected, transformed point X = q / w + t of a pixel with inverse depth w, scaled by w: Y = w X = q + w t.  Y projects to the same
// pixel as X (Y.x / Y.z = X.x / X.z) and needs no reciprocal of w; the depth along the keyframe ray comes out as (X.z - t_z) w = q_z.
struct Scaled { float y0, y1, y2; };
__device__ __forceinline__ Scaled scaled_point(const Ray& q, float w, const WarpParams& P) {
  return Scaled{__builtin_fmaf(P.t[0], w, q.q0), __builtin_fmaf(P.t[1], w, q.q1), __builtin_fmaf(P.t[2], w, q.q2)};
}

// trafo3DKernelInvDepthGridStride (:505-546), one pixel
__device__ __forceinline__ float warp_invdepth_px(const FMap& src, const Ray& q, float w, const WarpParams& P) {
  const bool valid = w > 0.f;   // an inverse depth of 0 (a point at infinity) is rejected like NaN, as the exact path ends up doing
  const float ws = valid ? w : 1.f;
  const Scaled Y = scaled_point(q, ws, P);
  const float wc = rcp(Y.y2);
  const float xs = __builtin_fmaf(Y.y0, wc, 0.5f), ys = __builtin_fmaf(Y.y1, wc, 0.5f);
  const int ix = cvt_flr(xs), iy = cvt_flr(ys);
  const bool inb = inside(ix, iy, src.cols, src.rows);
  const float w2 = src.at(iy, ix);   // unclamped: see the note on rejected pixels above
  const float res = (q.q2 * rcp(__builtin_fmaf(-w2, P.t[2], 1.f))) * w2;   // v1_z = (X.z - t_z) w = q_z
  return (valid & inb & (res > 0.f)) ? res : qnan();
}

// the weighted variant (:549-594): warped inverse depth + weight (1 - w2 tz)^4 / v^2
__device__ __forceinline__ float warp_invdepth_weighted_px(const FMap& src, const Ray& q, float w, const WarpParams& P, float& weight_res, bool& store_weight) {
  const bool valid = w > 0.f;   // an inverse depth of 0 (a point at infinity) is rejected like NaN, as the exact path ends up doing
  const float ws = valid ? w : 1.f;
  const Scaled Y = scaled_point(q, ws, P);
  const float wc = rcp(Y.y2);
  const float xs = __builtin_fmaf(Y.y0, wc, 0.5f), ys = __builtin_fmaf(Y.y1, wc, 0.5f);
  const int ix = cvt_flr(xs), iy = cvt_flr(ys);
  const bool inb = inside(ix, iy, src.cols, src.rows);
  const float w2 = src.at(iy, ix);   // unclamped: see the note on rejected pixels above
  const float v1_z = q.q2;
  const float w_factor = __builtin_fmaf(-w2, P.t[2], 1.f);
  const float rv = rcp(v1_z), wf2 = w_factor * w_factor;
  weight_res = (wf2 * wf2) * (rv * rv);
  const float res = (v1_z * rcp(w_factor)) * w2;
  store_weight = valid & inb & (weight_res > 0.f);
  return (valid & inb & (res > 0.f)) ? res : qnan();
}

// the same intensity warp in two halves, so that a caller can put other memory traffic between the tap loads and their use:
// intensity_taps() projects, forms the 1.8 fixed-point weights and ISSUES the two 8-byte tap loads; intensity_finish() blends.
struct IntensityTaps { float2 p0, p1; float a, b; bool ok; };
// Clamp addressing without integer clamps or per-tap selects: the sample coordinate itself is clamped to [0, n - 1) (one v_med3_f32 per
// axis; the upper end is the float just below n - 1), so that floor() is a legal column / row with a legal right / lower neighbour and the
// two 8-byte loads (row j and row j + 1: the same offset, the pitch in the instruction's scalar offset) ARE the four texels.  Left / top
// border: the weight becomes 0 on texel 0, as clamp addressing gives; right / bottom border: the coordinate is the float just below n - 1, so
// the weight of the last texel is 1 - ulp(n - 1) (1 - 2^-14 at 640 columns) -- exactly 1 after the 1.8 fixed-point rounding of
// RGBID_INTERP_TEX8, the engine's mode; with RGBID_INTERP_EXACT the last texel is blended with its neighbour by that 2^-14 (at most 0.016 grey
// levels), a bound of the FAST class only (the exact functions above clamp the texel indices instead).
__device__ __forceinline__ IntensityTaps intensity_taps(const FMap& src, const Ray& q, float w, const WarpParams& P, int interp_mode) {
  IntensityTaps t;
  const bool valid = w > 0.f;   // an inverse depth of 0 (a point at infinity) is rejected like NaN, as the exact path ends up doing
  const float ws = valid ? w : 1.f;
  const Scaled Y = scaled_point(q, ws, P);
  const float wc = rcp(Y.y2);
  const float xB = Y.y0 * wc, yB = Y.y1 * wc;
  t.ok = valid & (xB >= -0.5f) & (xB < (float)src.cols - 0.5f) & (yB >= -0.5f) & (yB < (float)src.rows - 0.5f);
  const float hx = __builtin_bit_cast(float, __builtin_bit_cast(int, (float)(src.cols - 1)) - 1);   // wave-uniform
  const float hy = __builtin_bit_cast(float, __builtin_bit_cast(int, (float)(src.rows - 1)) - 1);
  const float xc = __builtin_amdgcn_fmed3f(xB, 0.f, hx), yc = __builtin_amdgcn_fmed3f(yB, 0.f, hy);
  t.a = __builtin_amdgcn_fractf(xc); t.b = __builtin_amdgcn_fractf(yc);   // v_fract_f32: x - floor(x), kept below 1
  if (interp_mode == 1) {
    t.a = rintf(t.a * 256.f) * 0.00390625f;
    t.b = rintf(t.b * 256.f) * 0.00390625f;
  }
  const unsigned off = src.row(cvt_flr(yc)) + ((unsigned)cvt_flr(xc) << 2);
  t.p0 = src.at2_raw(off, 0u); t.p1 = src.at2_raw(off, src.pitch_b);
  return t;
}
__device__ __forceinline__ float intensity_finish(const IntensityTaps& t) {
  const float top = __builtin_fmaf(t.a, t.p0.y - t.p0.x, t.p0.x), bot = __builtin_fmaf(t.a, t.p1.y - t.p1.x, t.p1.x);
  float res = __builtin_fmaf(t.b, bot - top, top);
  res = fmaxf(0.f, fminf(res, 255.f));
  return t.ok ? res : qnan();
}

// trafo3DKernelIntensityWithInvDepthGridStride (:465-501), one pixel; bilinear tap in the lerp form
__device__ __forceinline__ float warp_intensity_px(const FMap& src, const Ray& q, float w, const WarpParams& P, int interp_mode) {
  return intensity_finish(intensity_taps(src, q, w, P, interp_mode));
}

// one direction of computeCovisibility's gate (partialVisibilityKernel :297-360)
__device__ __forceinline__ bool visible_px(const FMap& D, int cols, int rows, const Ray& q, float w, bool valid, const WarpParams& P) {
  const float ws = valid ? w : 1.f;
  const Scaled Y = scaled_point(q, ws, P);
  const float ry = rcp(Y.y2), wc = ws * ry;                 // inverse depth of the point in the other frame: 1 / X.z = w / Y.z
  const float xd = Y.y0 * ry, yd = Y.y1 * ry;
  const bool inside_img = (xd > 0) & (xd < (float)(cols - 1)) & (yd > 0) & (yd < (float)(rows - 1));
  const int xi = __float2int_rn(xd), yi = __float2int_rn(yd);   // inside_img: both already in the image
  return valid & inside_img & (fabsf(wc - D.at(yi, xi)) < 0.020f);
}

}  // namespace fastnum

// Workgroup -> tile mapping that keeps a lane's tiles on ONE XCD.  The hardware deals consecutive workgroup ids round-robin onto the 8
// XCDs (id % 8), each with its own 4 MiB L2, so with the natural (x, y, lane) order the neighbours of a tile -- whose gather footprints
// share cache lines with it (a 64-px row segment shifted by the warp touches 3 lines instead of 2, plus a halo row) -- always run on
// other XCDs and every shared line is fetched once per XCD.  With a 1-D grid and V = (id % 8) * (n / 8) + id / 8, XCD k walks a
// contiguous slab of (lane, tile) space: spatial neighbours meet in the same L2.  The two divisions by launch constants are
// multiply-high by host-computed magic numbers (wave-uniform: they stay on the scalar unit; the generic division the compiler emits for
// `V / nx` costs ~40 VALU instructions per workgroup, more than the sharing gains in a VALU-bound kernel).
struct TileMap {
  unsigned nx, tiles, per, n, magic_tiles, magic_nx;
  __device__ __forceinline__ TileId tile(unsigned id) const {
    const unsigned V = (id < (per << 3)) ? (id & 7u) * per + (id >> 3) : id;   // tail ids keep their place
    TileId t;
    const unsigned lane = tiles == 1 ? V : __umulhi(V, magic_tiles);   // 2^32 / 1 + 1 does not fit the magic: divisor 1 is taken apart
    const unsigned tl = V - lane * tiles;
    const unsigned by = nx == 1 ? tl : __umulhi(tl, magic_nx);
    t.lane = (int)lane; t.by = (int)by; t.bx = (int)(tl - by * nx);
    return t;
  }
};
inline bool make_tile_map(int nx, int ny, int B, TileMap* tm) {
  const unsigned long long tiles = (unsigned long long)nx * ny, n = tiles * B;
  if (n >= (1ull << 31) || tiles * n >= (1ull << 32) || (unsigned long long)nx * tiles >= (1ull << 32)) return false;   // mulhi-by-magic exactness bound
  tm->nx = (unsigned)nx; tm->tiles = (unsigned)tiles; tm->n = (unsigned)n; tm->per = (unsigned)(n >> 3);
  tm->magic_tiles = (unsigned)((1ull << 32) / tiles + 1);
  tm->magic_nx = (unsigned)((1ull << 32) / (unsigned)nx + 1);
  return true;
}

}  // namespace rgbid
