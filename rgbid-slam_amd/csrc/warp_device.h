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
#include "guard_band.h"

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

// one direction of computeCovisibility's gate (partialVisibilityKernel :297-360), the oracle's evaluation
__device__ __forceinline__ bool visible_px_exact(const FMap& D, int cols, int rows, int x, int y, float w, bool valid, const WarpParams& P) {
#pragma clang fp contract(off)
  float xd, yd;
  float w_dst = register_pixel(xd, yd, x, y, valid ? w : 1.f, P);
  bool inside_img = (xd > 0) && (xd < (float)(cols - 1)) && (yd > 0) && (yd < (float)(rows - 1));
  int xi = clampi(__float2int_rn(xd), cols - 1), yi = clampi(__float2int_rn(yd), rows - 1);
  return valid && inside_img && (fabsf(w_dst - D.at(yi, xi)) < 0.020f);
}

// =====================================================================================================================================
// FAST numerics class of the ENGINE's gather kernels (rgbid_engine_config.fast_numerics, default on): cheap VALUES, the oracle's SELECTION.
//
// The reference compiles its kernels with --prec-div=false --prec-sqrt=false --ftz=true (CMakeLists.txt:105) and nvcc's default FMA
// contraction: its own arithmetic is approximate reciprocals and fused multiply-adds, not IEEE.  The functions above reproduce the IEEE
// evaluation of the scalar oracle bit for bit (the compat bridge always uses them); the functions below evaluate the same formulas in
// the reference build's class of arithmetic -- v_rcp_f32 (1 ulp) for every division, explicit FMAs, the ray q = R (x, y, 1) formed once per
// pixel and shared by every projection from that pixel (registerPixel, warping_registration.cu:129-146, re-associated:
// R (x z, y z, z) + t = z q + t, and projected from the scaled point w X = q + w t so that no reciprocal of the inverse depth is taken),
// 1 / (1 / X.z) taken as X.z -- which removes ~45 % of the instructions of kernels that are VALU-bound.
//
// Round 4: every DISCRETE decision -- which source pixel is point-sampled, whether a projection lies inside the image, the sign test of the
// warped value, the covisibility lattice point, the covisibility / fusion gates, the texel pair of a bilinear sample that meets a NaN texel -- is
// the ORACLE's, bit for bit.  A cheap coordinate can only flip such a decision when it lies within a proven error bound (guard_band.h) of the
// decision's threshold; those pixels (a few per thousand) are recomputed with the exact instruction sequence under a wave-level branch,
// everything else keeps the fast arithmetic.  What still differs from the oracle are float values in their last bits, and the 1.8 fixed-point
// bilinear weight where a coordinate sits within the bound of a 1/256 step (a value step of 1/256 of the local contrast, no selection).
// Structure rule for the kernels: a branch between a gather and its use makes the wave wait for that gather there, which serialises the
// independent chains of a thread's pixels -- so kernels run the projections (+ coordinate fixes) of ALL their pixels, then the gathers, then the
// finishing steps (measured: the covisibility pair 2.0 x slower, the fusion 13 % slower with per-pixel branch -> gather -> branch chains).
namespace fastnum {

// Rejected pixels (coordinates outside the image, invalid inverse depth) are not clamped to a legal address first, as the exact functions do:
// the gathers go through the map's raw buffer descriptor, whose range check returns 0 for any offset outside the lane's image, an offset
// that wraps lands on some other texel of the same image, and either value is discarded by the pixel's predicate.
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// floor + saturating convert in one instruction (NaN -> 0); verified against v_floor_f32 + v_cvt_i32_f32 by rgbid_selftest_cvt_flr
__device__ __forceinline__ int cvt_flr(float x) { int r; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }

// the guard constants of a lane as wave-uniform scalars (guard_band.h make_guard on uniform inputs; readfirstlane pins them to SGPRs)
__device__ __forceinline__ float uniform_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ Guard lane_guard(const WarpParams& P, int cols, int rows) {
  Guard g = make_guard(P.R, P.t, cols, rows);
  g.d1 = uniform_f(g.d1); g.c2 = uniform_f(g.c2); g.d2 = uniform_f(g.d2); g.q0 = uniform_f(g.q0); g.q1 = uniform_f(g.q1); g.db = uniform_f(g.db);
  g.g0 = uniform_f(g.g0); g.g1 = uniform_f(g.g1); g.e0 = uniform_f(g.e0); g.e1 = uniform_f(g.e1);
  g.bL = uniform_f(g.bL); g.cL = uniform_f(g.cL); g.kL = uniform_f(g.kL); g.wcore = uniform_f(g.wcore);
  g.zsafe = __builtin_amdgcn_readfirstlane(g.zsafe);
#ifdef RGBID_EXPERIMENT_GUARD_NEVER_FIRES   // timing experiment only (tools/build_variant.sh): the checks run, no pixel is ever recomputed
  g.d1 = uniform_f(0.f); g.c2 = uniform_f(10.f); g.db = uniform_f(-1e9f); g.d2 = uniform_f(0.f);
  g.bL = uniform_f(0.5f); g.cL = uniform_f(10.f); g.kL = uniform_f(0.f);
#endif
  return g;
}

// the part of the ray q = K R^-1 K^-1 (x, y, 1) that a pixel row shares, and the ray of one pixel of the row: q_r = fl(R[3r] x + fl(R[3r+1] y + R[3r+2]))
// (two FMAs: the evaluation the bound (F) of guard_band.h is proven for; a function of the pixel alone, whatever kernel asks)
struct RowRay { float c0, c1, c2; };
struct Ray { float q0, q1, q2; };
__device__ __forceinline__ RowRay row_ray(const WarpParams& P, float yf) {
  return RowRay{__builtin_fmaf(P.R[1], yf, P.R[2]), __builtin_fmaf(P.R[4], yf, P.R[5]), __builtin_fmaf(P.R[7], yf, P.R[8])};
}
__device__ __forceinline__ Ray ray_at(const WarpParams& P, const RowRay& c, float xf) {
  return Ray{__builtin_fmaf(P.R[0], xf, c.c0), __builtin_fmaf(P.R[3], xf, c.c1), __builtin_fmaf(P.R[6], xf, c.c2)};
}
__device__ __forceinline__ Ray ray(const WarpParams& P, float xf, float yf) { return ray_at(P, row_ray(P, yf), xf); }
// The back-projected, transformed point X = q / w + t of a pixel with inverse depth w, scaled by w: Y = w X = q + w t.  Y projects to the same
// pixel as X (Y.x / Y.z = X.x / X.z) and needs no reciprocal of w; the depth along the keyframe ray comes out as (X.z - t_z) w = q_z.
struct Scaled { float y0, y1, y2; };
__device__ __forceinline__ Scaled scaled_point(const Ray& q, float w, const WarpParams& P) {
  return Scaled{__builtin_fmaf(P.t[0], w, q.q0), __builtin_fmaf(P.t[1], w, q.q1), __builtin_fmaf(P.t[2], w, q.q2)};
}
// a grid inverse depth of the FAST domain [W_LO, W_HI], or the stand-in for an invalid one (NaN, <= 0, out of the domain): one v_med3_f32 (NaN
// operands make it a min3, which skips them) + one compare -- the cost of the `w > 0` + select it replaces
__device__ __forceinline__ float sanitised(float w, bool& valid) {
  const float ws = __builtin_amdgcn_fmed3f(w, W_LO, W_HI);
  valid = ws == w;
  return ws;
}

// v_max3_f32 (1.35 FMA-equivalents on this chip; v_max_f32 costs 1.5): NaN operands are skipped -- callers make sure a NaN cannot hide an open decision
__device__ __forceinline__ float max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// ---- trafo3DKernelInvDepthGridStride (:505-546), one pixel, in steps so that a kernel can run the steps of several pixels side by side:
// id_project (coordinates + guard verdict) -> [id_fix_coords for flagged pixels] -> gather -> id_finish -> [exact pixel where the sign is open]
struct IdProj {
  float ws, q2;       // sanitised inverse depth; depth along the keyframe ray times w (= the oracle's v1_z)
  int ix, iy;         // source pixel of the point sample
  bool ok;            // inverse depth valid and source pixel inside the image
};
// `fixc`: the coordinates lie inside the guard band -- the caller replaces them by the oracle's (id_fix_coords) before it gathers
__device__ __forceinline__ IdProj id_project(const Ray& q, float w, const WarpParams& P, const Guard& G, int cols, int rows, bool& fixc) {
  IdProj r;
  bool valid;
  r.ws = sanitised(w, valid);
  r.q2 = q.q2;
  const Scaled Y = scaled_point(q, r.ws, P);
  const float wc = rcp(Y.y2);
  // guard_band.h (3'): the coordinate biased DOWN by the lane's band dLa (bL = 0.5 - dLa); the oracle's coordinate then lies in [xs, xs + 2 dLa] and
  // floor() of the two agrees when fract(xs) < 1 - 2 dLa -- two v_fract, one multiply, one v_max3, one compare for both axes and the bound on |wc|
  const float xs = __builtin_fmaf(Y.y0, wc, G.bL), ys = __builtin_fmaf(Y.y1, wc, G.bL);
  r.ix = cvt_flr(xs); r.iy = cvt_flr(ys);
  r.ok = valid & inside(r.ix, r.iy, cols, rows);
  // NaN / inf: wc is finite or +-inf for a sane lane (Y_2 finite), |wc| = inf fails the test; a lane that is not sane has cL = -1 (never safe)
  const float m = max3(__builtin_amdgcn_fractf(xs), __builtin_amdgcn_fractf(ys), fabsf(wc) * G.kL);
  fixc = valid & !(m < G.cL);
  return r;
}
// cold path: the oracle's coordinates (register_pixel: no contraction, IEEE reciprocals) of a valid pixel
__device__ __forceinline__ void id_fix_coords(IdProj& r, int x, int y, const WarpParams& P, int cols, int rows) {
#pragma clang fp contract(off)
  float xe, ye;
  register_pixel(xe, ye, x, y, r.ws, P);
  xe += 0.5f; ye += 0.5f;
  r.ix = cvt_rd(xe); r.iy = cvt_rd(ye);
  r.ok = inside(r.ix, r.iy, cols, rows);
}
// after the gather of w2 = src(iy, ix): the warped value; `fixr`: the sign of the oracle's result is not implied (recompute the pixel exactly)
__device__ __forceinline__ float id_finish(const IdProj& r, float w2, const WarpParams& P, const Guard& G, bool& fixr) {
  const float rwf = rcp(__builtin_fmaf(-w2, P.t[2], 1.f));
  const float res = (r.q2 * rwf) * w2;   // v1_z = (X.z - t_z) w = q_z
  fixr = r.ok & ((G.zsafe == 0) | (fabsf(rwf) > 0x1p19f));   // guard_band.h (4): per lane (wave-uniform), and per pixel
  return (r.ok & (res > 0.f)) ? res : qnan();
}
// the same for callers that carry validity as a mask (the fused normal-equation kernel): `okd` = the warped value is valid; the returned value is finite
// either way (the sanitised keyframe inverse depth stands in for an invalid one, so that a residual formed from it stays finite under a zero weight)
__device__ __forceinline__ float id_finish_m(const IdProj& r, float w2, const WarpParams& P, const Guard& G, bool& okd, bool& fixr) {
  const float rwf = rcp(__builtin_fmaf(-w2, P.t[2], 1.f));
  const float res = (r.q2 * rwf) * w2;
  fixr = r.ok & ((G.zsafe == 0) | (fabsf(rwf) > 0x1p19f));
  okd = r.ok & (res > 0.f);
  return okd ? res : r.ws;
}
// the whole pixel for kernels with one pixel per thread (lattice pre-pass, scalar path of the normal equations)
__device__ __forceinline__ float warp_invdepth_px(const FMap& src, const Ray& q, int x, int y, float w, const WarpParams& P, const Guard& G) {
  bool fixc, fixr;
  IdProj r = id_project(q, w, P, G, src.cols, src.rows, fixc);
  if (__builtin_expect(fixc, 0)) id_fix_coords(r, x, y, P, src.cols, src.rows);
  float res = id_finish(r, src.at(r.iy, r.ix), P, G, fixr);   // unclamped: see the note on rejected pixels above
  if (__builtin_expect(fixr, 0)) res = rgbid::warp_invdepth_px(src, x, y, w, P);
  return res;
}

// the weighted variant (:549-594) after the gather: warped inverse depth + weight (1 - w2 tz)^4 / v^2.  eps_res: bound of the relative distance between
// this value and the oracle's (guard_band.h (5)), for the gate of the fusion that follows.  `fixr` as id_finish.
__device__ __forceinline__ float id_finish_weighted(const IdProj& r, float w2, const WarpParams& P, const Guard& G, float& weight_res, bool& store_weight,
                                                    float& eps_res, bool& fixr) {
  const float v1_z = r.q2;
  const float w_factor = __builtin_fmaf(-w2, P.t[2], 1.f);
  const float rwf = rcp(w_factor);
  const float rv = rcp(v1_z), wf2 = w_factor * w_factor;
  weight_res = (wf2 * wf2) * (rv * rv);
  const float res = (v1_z * rwf) * w2;
  store_weight = r.ok & (weight_res > 0.f);
  eps_res = __builtin_fmaf(fabsf(rwf), G.e1, G.e0);
  fixr = r.ok & ((G.zsafe == 0) | (fabsf(rwf) > 0x1p19f));
  return (r.ok & (res > 0.f)) ? res : qnan();
}

// the intensity warp in two halves, so that a caller can put other memory traffic between the tap loads and their use:
// intensity_taps() projects, forms the 1.8 fixed-point weights and ISSUES the two 8-byte tap loads; intensity_finish() blends.
struct IntensityTaps { float2 p0, p1; float a, b; bool ok; };
// Clamp addressing without integer clamps or per-tap selects: the sample coordinate itself is clamped to [0, n - 1) (one v_med3_f32 per
// axis; the upper end is the float just below n - 1), so that floor() is a legal column / row with a legal right / lower neighbour and the
// two 8-byte loads (row j and row j + 1: the same offset, the pitch in the instruction's scalar offset) ARE the four texels.  Left / top
// border: the weight becomes 0 on texel 0, as clamp addressing gives; right / bottom border: the coordinate is the float just below n - 1, so
// the weight of the last texel is 1 - ulp(n - 1) (1 - 2^-14 at 640 columns) -- exactly 1 after the 1.8 fixed-point rounding of
// RGBID_INTERP_TEX8, the engine's mode; with RGBID_INTERP_EXACT the last texel is blended with its neighbour by that 2^-14 (at most 0.016 grey
// levels), a bound of the FAST class only (the exact functions above clamp the texel indices instead).
// The bilinear VALUE is continuous across texel boundaries (a tap pair one texel off carries weight 0 / 1 there), so the discrete decisions of
// this warp are the in-image predicate (:486-487) -- taken as "inside" / "outside" when the coordinate is farther than the guard band from the
// image border, and from the oracle's coordinates otherwise (`border`: the caller runs intensity_fix_border for those pixels) -- and the texel
// pair of a sample that meets a NaN texel (intensity_finish).
// Round 5 (guard_band.h (3'')): the hot path takes "inside" where it is certain at no extra cost -- the coordinate equals its own clamp to [0, n - 1) (the
// v_med3_f32 the tap addresses need anyway) and |wc| <= wcore -- and leaves every other pixel of the domain (the half-pixel ring around the image,
// projections outside it) to the cold path intensity_fix_border, which sorts them into surely inside / surely outside / the oracle's coordinates.
__device__ __forceinline__ float tex8_weight(float a) {
  // rint(a * 256) / 256 for a in [0, 1] in two full-rate adds: 1.5 * 2^15 has ulp 2^-8, round-to-nearest-even on the same grid as rintf (bit-identical)
  return (a + 49152.f) - 49152.f;
}
// `valid_in`: what the caller already knows about w (true: nothing; the fused kernel passes the mask of its warped inverse depth)
__device__ __forceinline__ IntensityTaps intensity_taps(const FMap& src, const Ray& q, float w, const WarpParams& P, const Guard& G, int interp_mode, bool& border, bool valid_in = true) {
  IntensityTaps t;
  bool valid;
  const float ws = sanitised(w, valid);
  valid &= valid_in;
  const Scaled Y = scaled_point(q, ws, P);
  const float wc = rcp(Y.y2);
  const float xB = Y.y0 * wc, yB = Y.y1 * wc;
  const float hx = __builtin_bit_cast(float, __builtin_bit_cast(int, (float)(src.cols - 1)) - 1);   // wave-uniform
  const float hy = __builtin_bit_cast(float, __builtin_bit_cast(int, (float)(src.rows - 1)) - 1);
  const float xc = __builtin_amdgcn_fmed3f(xB, 0.f, hx), yc = __builtin_amdgcn_fmed3f(yB, 0.f, hy);
  const bool core = (xc == xB) & (yc == yB) & (fabsf(wc) <= G.wcore);   // NaN anywhere: not core
  t.ok = valid & core;
  border = valid & !core;
  t.a = __builtin_amdgcn_fractf(xc); t.b = __builtin_amdgcn_fractf(yc);   // v_fract_f32: x - floor(x), kept below 1
  if (interp_mode == 1) { t.a = tex8_weight(t.a); t.b = tex8_weight(t.b); }
  const unsigned off = src.row(cvt_flr(yc)) + ((unsigned)cvt_flr(xc) << 2);
  t.p0 = src.at2_raw(off, 0u); t.p1 = src.at2_raw(off, src.pitch_b);
  return t;
}
// cold path of a pixel whose projection is not in the core: safely inside (farther than the band from the border), surely outside, or the oracle's predicate
__device__ __forceinline__ bool intensity_fix_border(const FMap& src, const Ray& q, int x, int y, float w, const WarpParams& P, const Guard& G) {
#pragma clang fp contract(off)
  const Scaled Y = scaled_point(q, w, P);
  const float wc = rcp(Y.y2);
  const float xB = Y.y0 * wc, yB = Y.y1 * wc;
  const bool priced = fabsf(wc) <= RHO_BORDER;   // db = RHO_BORDER d1 + d2
  const bool surely_in = priced & (xB >= -0.5f + G.db) & (xB < (float)src.cols - 0.5f - G.db) & (yB >= -0.5f + G.db) & (yB < (float)src.rows - 0.5f - G.db);
  const bool surely_out = priced & ((xB < -0.5f - G.db) | (xB >= (float)src.cols - 0.5f + G.db) | (yB < -0.5f - G.db) | (yB >= (float)src.rows - 0.5f + G.db));
  if (surely_in | surely_out) return surely_in;
  float xe, ye;
  register_pixel(xe, ye, x, y, w, P);
  xe += 0.5f; ye += 0.5f;
  return inside(cvt_rd(xe), cvt_rd(ye), src.cols, src.rows);
}
// `nan_tap`: a tap is NaN (or the blend is: inf - inf) -- the one case in which the texel PAIR matters: a pair one texel off carries weight 0 / 1 at
// the boundary, but 0 * NaN is NaN -- so the caller takes the oracle's pixel (rgbid::warp_intensity_px) for it.  Intensity maps hold NaN only at
// the three corner pixels of pyramid levels >= 1.
__device__ __forceinline__ float intensity_finish(const IntensityTaps& t, bool& nan_tap) {
  const float top = __builtin_fmaf(t.a, t.p0.y - t.p0.x, t.p0.x), bot = __builtin_fmaf(t.a, t.p1.y - t.p1.x, t.p1.x);
  float res = __builtin_fmaf(t.b, bot - top, top);
  nan_tap = t.ok & (res != res);
  res = __builtin_amdgcn_fmed3f(res, 0.f, 255.f);   // the clamp to [0, 255] as one instruction; a NaN blend is replaced by the caller (nan_tap)
  return t.ok ? res : qnan();
}
// for callers that carry validity as a mask (t.ok): no NaN is written into the value
__device__ __forceinline__ float intensity_finish_m(const IntensityTaps& t, bool& nan_tap) {
  const float top = __builtin_fmaf(t.a, t.p0.y - t.p0.x, t.p0.x), bot = __builtin_fmaf(t.a, t.p1.y - t.p1.x, t.p1.x);
  const float res = __builtin_fmaf(t.b, bot - top, top);
  nan_tap = t.ok & (res != res);
  return __builtin_amdgcn_fmed3f(res, 0.f, 255.f);
}

// trafo3DKernelIntensityWithInvDepthGridStride (:465-501), one pixel; bilinear tap in the lerp form
__device__ __forceinline__ float warp_intensity_px(const FMap& src, const Ray& q, int x, int y, float w, const WarpParams& P, const Guard& G, int interp_mode) {
  bool border, nan_tap;
  IntensityTaps t = intensity_taps(src, q, w, P, G, interp_mode, border);
  if (__builtin_expect(border, 0)) t.ok = intensity_fix_border(src, q, x, y, w, P, G);
  float res = intensity_finish(t, nan_tap);
  if (__builtin_expect(nan_tap, 0)) res = rgbid::warp_intensity_px(src, x, y, w, P, interp_mode);
  return res;
}

// one direction of computeCovisibility's gate (partialVisibilityKernel :297-360), in two steps around the gather of the other frame's inverse depth.
// Discrete decisions: the lattice point rint(xd), rint(yd) (open within delta of a half-integer), the comparisons with the image border (open within
// delta of 0 / cols - 1 / rows - 1), the gate |w' - D| < 0.020 (open within eps_w |w'|).  An open coordinate is replaced by the oracle's BEFORE the
// gather; an open gate re-evaluates w' the oracle's way on the same texel.
struct VisProj { float ws, wc, ry; int xi, yi; bool ok, exact; };   // ok: inverse depth of the domain and projection strictly inside the image
__device__ __forceinline__ VisProj vis_project(int cols, int rows, const Ray& q, int x, int y, float w, const WarpParams& P, const Guard& G) {
#pragma clang fp contract(off)
  VisProj v;
  bool in_domain;
  v.ws = sanitised(w, in_domain);
  const Scaled Y = scaled_point(q, v.ws, P);
  v.ry = rcp(Y.y2);
  v.wc = v.ws * v.ry;                                       // inverse depth of the point in the other frame: 1 / X.z = w / Y.z
  float xd = Y.y0 * v.ry, yd = Y.y1 * v.ry;
  const float xm = (float)(cols - 1), ym = (float)(rows - 1);
  const float hx = 0.5f * xm, hy = 0.5f * ym;               // exact: the image centre and half extent
  // signed distance to the image border along each axis (negative inside); m < 0 <=> 0 < xd < cols - 1 and 0 < yd < rows - 1 whenever |m| > delta
  const float m = fmaxf(fabsf(xd - hx) - hx, fabsf(yd - hy) - hy);
  float rx = rintf(xd), ryy = rintf(yd);
  const float delta = __builtin_fmaf(fabsf(v.ry), G.d1, G.d2);
  // open: within delta of a half-integer (the lattice point) or of the image border.  NaN / inf anywhere: open
  v.exact = in_domain & !((fmaxf(fabsf(xd - rx), fabsf(yd - ryy)) + delta < 0.5f) & (fabsf(m) > delta));
  bool inside_img = m < 0.f;
  if (__builtin_expect(v.exact, 0)) {
    v.wc = register_pixel(xd, yd, x, y, v.ws, P);
    inside_img = (xd > 0) & (xd < xm) & (yd > 0) & (yd < ym);
    rx = rintf(xd); ryy = rintf(yd);
  }
  v.xi = __float2int_rz(rx); v.yi = __float2int_rz(ryy);   // integral values; inside_img: both already in the image (NaN -> 0, saturating)
  v.ok = in_domain & inside_img;
  return v;
}
__device__ __forceinline__ bool vis_gate(const VisProj& v, float d, int x, int y, const WarpParams& P, const Guard& G) {
#pragma clang fp contract(off)
  float dgap = fabsf(v.wc - d);
  // the gate is open within eps_w |w'| of the threshold (guard_band.h (5)): the band itself is the test (two FMAs; round 4 screened with a fixed 2^-10
  // first, which an in-domain pixel with |w'| |1 / Y_2| of the order of 10^3 could pass although its gate was open: ADVICE r4)
  const float band = __builtin_fmaf(fabsf(v.wc), __builtin_fmaf(fabsf(v.ry), G.g1, G.g0), 4.f * 0x1p-24f * 0.020f);
  if (__builtin_expect(v.ok & !v.exact & (fabsf(dgap - 0.020f) <= band), 0)) {   // a NaN gap (NaN texel: common) is not an open gate; a non-finite band only occurs where `exact` is set
    float xe, ye;
    dgap = fabsf(register_pixel(xe, ye, x, y, v.ws, P) - d);
  }
  return v.ok & (dgap < 0.020f);
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
