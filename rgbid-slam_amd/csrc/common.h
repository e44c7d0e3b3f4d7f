// common.h -- shared device helpers / launch descriptors for the gfx950 kernels.
//
// Every kernel in csrc/ is natively BATCHED: an image argument is an ImgB = one allocation holding
// `B` lanes (independent frame pairs / tracker lanes) of identical geometry, lane l at
// base + l*lane_stride.  The single-image C-ABI calls are the B == 1 case (lane_stride == 0).
// Per-lane scalars come either by value in the kernarg segment (compat wrappers) or from a device
// array indexed by lane (batched engine) -- see ByValue / ByLane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Evaluate a function body WITHOUT fused multiply-add contraction, whatever the including translation unit's -ffp-contract / pragma state is, and
// without changing that state for the code that follows (the pragma is scoped to the compound statement it opens).
#ifndef RGBID_FP_STRICT
#if defined(__clang__)
#define RGBID_FP_STRICT _Pragma("clang fp contract(off)")
#else
#define RGBID_FP_STRICT
#endif
#endif

namespace rgbid {

struct ImgB {
  void* base;
  size_t pitch;        // bytes between rows
  size_t lane_stride;  // bytes between lanes (0 for a single image)
  int rows, cols;
};

// The row offset is a 24-bit multiply (rows and the pitch in bytes are < 2^24, a lane's image < 4 GB): the 64-bit y * pitch it replaces
// compiles to quarter-rate v_mad_u64_u32 / v_mul_lo_u32 pairs, several per pixel in kernels that are VALU-bound.  The lane term is
// wave-uniform in every kernel (the lane is a block index) and stays on the scalar unit.
template <typename T>
__device__ __forceinline__ T* row_ptr(const ImgB& im, int lane, int y) {
#ifdef RGBID_ROW_PTR_MUL64
  return reinterpret_cast<T*>(static_cast<char*>(im.base) + (size_t)lane * im.lane_stride + (size_t)y * im.pitch);
#else
  return reinterpret_cast<T*>(static_cast<char*>(im.base) + (size_t)lane * im.lane_stride + (size_t)__umul24((unsigned)y, (unsigned)im.pitch));
#endif
}
template <typename T>
__device__ __forceinline__ T& px(const ImgB& im, int lane, int y, int x) { return row_ptr<T>(im, lane, y)[x]; }

template <class T> struct ByValue {
  T v;
  __device__ __forceinline__ const T& get(int) const { return v; }
};
template <class T> struct ByLane {
  const T* p;
  __device__ __forceinline__ const T& get(int lane) const { return p[lane]; }
};

// lane predicate: kernels skip lanes whose mask byte != want (mask == nullptr -> all lanes run)
struct LaneMask {
  const int* flags;
  int want;
  __device__ __forceinline__ bool on(int lane) const { return flags == nullptr || flags[lane] == want; }
};

// 16-byte store of four consecutive outputs of a plane that this kernel does not read again: non-temporal, the written lines do not displace the
// streams the kernel is reading from the XCD's L2 (Sobel pair -3.5 %, keyframe maps -3.9 %, frame preparation -1.5 % per launch at 1 024 lanes)
__device__ __forceinline__ void st16_stream(float* p, float a, float b, float c, float d) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(f4v{a, b, c, d}, reinterpret_cast<f4v*>(p));
}

__device__ __forceinline__ float qnan() { return __int_as_float(0x7fffffff); }  // utils.hpp:76-77

// CUDA __float2int_rd / __float2int_rn: saturating, NaN -> 0
__device__ __forceinline__ int f2i_rd(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.f) return 2147483647;
  if (x <= -2147483648.f) return (-2147483647 - 1);
  return (int)floorf(x);
}
__device__ __forceinline__ int f2i_rn(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.f) return 2147483647;
  if (x <= -2147483648.f) return (-2147483647 - 1);
  return (int)rintf(x);
}

struct WarpParams {  // K R^-1 K^-1 (row-major) and K t^-1, float (visodo.cpp:1108-1114)
  float R[9];
  float t[3];
};

// Correctly rounded reciprocal in 4 VALU instructions instead of the 11 of the generic IEEE division sequence hipcc emits for
// 1.f / x: v_rcp_f32 (1 ulp) followed by ONE Newton step in FMA form returns exactly the IEEE quotient whenever the result is a
// normal number (x itself not denormal) -- verified exhaustively over all 2^32 inputs (rgbid_selftest_rcp,
// tests/test_gpu_kernels.py).  RcpFast evaluates that straight line and remembers whether any result left the verified range
// (zero / denormal / inf / NaN result: v_cmp_class); the caller then recomputes that pixel with RcpIeee.  On real data no wave
// ever takes the fallback; the warps are VALU-bound with four reciprocals per pixel, so this removes ~25 % of their instructions
// without changing a single bit.
struct RcpIeee {
  __device__ __forceinline__ float operator()(float x) { return 1.f / x; }
  __device__ __forceinline__ bool failed() const { return false; }
};
struct RcpFast {
  bool bad = false;
  __device__ __forceinline__ float operator()(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    bad |= !__builtin_amdgcn_classf(r, 0x108);  // 0x108 = negative normal | positive normal
    return r;
  }
  __device__ __forceinline__ bool failed() const { return bad; }
};
__device__ __forceinline__ float rcp_exact(float x) {
  RcpFast f;
  float r = f(x);
  if (__builtin_expect(f.failed(), 0)) r = 1.f / x;
  return r;
}

// x / c for a wave-uniform constant c, with rc = RN(1 / c) computed once on the host: q = RN(x rc), r = x - q c (exact in one FMA),
// q' = RN(q + r rc) is the correctly rounded quotient (Markstein) -- 3 VALU instructions instead of the 11 of the IEEE division sequence --
// whenever the result is a normal number and |x| >= 2^-100 (below that the remainder itself underflows).  `ok` reports that (or x == 0, whose quotient is a zero; its SIGN may differ from IEEE's for
// x = -0, which no caller can observe: the bilateral filter squares it).  rgbid_selftest_div_const verifies the claim for a given c over
// all 2^32 x on the device; callers use it only for constants that test has passed and recompute with `/` when !ok.
struct DivConst { float c, rc; };
__device__ __forceinline__ float div_const_fast(float x, const DivConst& d, bool& ok) {
  float q = x * d.rc;
  const float r = __builtin_fmaf(-d.c, q, x);
  q = __builtin_fmaf(r, d.rc, q);
  // the remainder x - q c carries bits down to 2^(exponent(x) - 46): it is exact only while that stays above the denormal floor 2^-149
  ok = (__builtin_amdgcn_classf(q, 0x108) & (fabsf(x) >= 0x1p-100f)) | (x == 0.f);  // 0x108 = negative normal | positive normal
  return q;
}

// registerPixel (warping_registration.cu:129-146).  fp contraction is OFF here so the projected
// coordinates -- and therefore every floor()/rint() pixel selection -- are bit-identical to the
// scalar oracle; the reciprocals are IEEE-exact (see above).
template <class RCP>
__device__ __forceinline__ float register_pixel_t(float& xc, float& yc, int xd, int yd, float wd, const WarpParams& P, RCP& rcp) {
#pragma clang fp contract(off)
  float zd = rcp(wd);
  float X = (float)xd * zd, Y = (float)yd * zd;
  float X0 = (P.R[0] * X + P.R[1] * Y + P.R[2] * zd) + P.t[0];
  float X1 = (P.R[3] * X + P.R[4] * Y + P.R[5] * zd) + P.t[1];
  float X2 = (P.R[6] * X + P.R[7] * Y + P.R[8] * zd) + P.t[2];
  float wc = rcp(X2);
  xc = X0 * wc;
  yc = X1 * wc;
  return wc;
}
__device__ __forceinline__ float register_pixel(float& xc, float& yc, int xd, int yd, float wd, const WarpParams& P) {
  RcpFast f;
  float wc = register_pixel_t(xc, yc, xd, yd, wd, P, f);
  if (__builtin_expect(f.failed(), 0)) { RcpIeee s; wc = register_pixel_t(xc, yc, xd, yd, wd, P, s); }
  return wc;
}

// XCD-aware tile order for the gather kernels.  The hardware places consecutive workgroups (x fastest, then y, then z) on consecutive
// XCDs, each with its own L2, so spatially adjacent tiles -- whose bilinear / point-sample footprints share cache lines (a 64-px row
// segment shifted by the warp touches 3 lines instead of 2, plus a halo row) -- would be fetched from HBM once PER XCD.  The tiles are
// renumbered so that XCD k walks a contiguous slab of (lane, tile row, tile column) space: neighbours then meet in the same L2.
struct TileId { int bx, by, lane; };
__device__ __forceinline__ TileId xcd_slab_tile() {
  const unsigned nx = gridDim.x, ny = gridDim.y, n = nx * ny * gridDim.z;
  const unsigned L = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
  const unsigned per = n >> 3;
  const unsigned V = (L < (per << 3)) ? (L & 7u) * per + (L >> 3) : L;   // tail tiles keep their id
  const unsigned r = V / nx;
  TileId t;
  t.bx = (int)(V - r * nx);
  t.lane = (int)(r / ny);
  t.by = (int)(r - (unsigned)t.lane * ny);
  return t;
}

// For PREDICATED launches (only some lanes work: keyframe switches) a slab of whole lanes per XCD would leave the XCDs unevenly loaded.  This variant keeps the
// lane of a workgroup (blockIdx.z) and renumbers only the tiles INSIDE the lane: XCD k owns the contiguous eighth [k nt / 8, (k + 1) nt / 8) of every lane's
// tiles -- every active lane loads all eight XCDs equally, and spatial neighbours still meet in one L2.  Needs nt % 8 == 0 (640x480, 1280x960 tilings);
// otherwise the natural order is kept.
__device__ __forceinline__ TileId xcd_lane_local_tile() {
  const unsigned nx = gridDim.x, nt = nx * gridDim.y;
  const unsigned r = blockIdx.x + nx * blockIdx.y;
  const unsigned t = (nt & 7u) == 0 ? (r & 7u) * (nt >> 3) + (r >> 3) : r;
  TileId o;
  o.lane = (int)blockIdx.z;
  o.by = (int)(t / nx);
  o.bx = (int)(t - (unsigned)o.by * nx);
  return o;
}

// the same renumbering for a 1-D grid: hardware id -> logical id such that XCD k owns the contiguous range [k n / 8, (k + 1) n / 8)
__device__ __forceinline__ unsigned xcd_slab_index(unsigned id, unsigned n) {
  const unsigned per = n >> 3;
  return (id < (per << 3)) ? (id & 7u) * per + (id >> 3) : id;   // tail ids keep their place
}

__device__ __forceinline__ bool in_bounds_rd(float xs, float ys, int cols, int rows) {
  int ix = f2i_rd(xs), iy = f2i_rd(ys);
  return !(ix < 0 || iy < 0 || ix >= cols || iy >= rows);
}

// wave64 reductions on the VALU with DPP (no LDS traffic, no ds_bpermute round trips):
// xor-1 / xor-2 quad permutes, half-row mirror, row mirror leave every lane of a 16-lane row holding the
// row sum; row_bcast:15 / row_bcast:31 fold the four rows so lane 63 holds the wave total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  int s = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
  return v + __int_as_float(s);
}
// total valid in lane 63 only
__device__ __forceinline__ float wave_sum_l63(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
  return v;
}
// wave total broadcast to every lane (uniform, lives in an SGPR)
__device__ __forceinline__ float wave_sum(float v) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_l63(v)), 63));
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

inline int div_up(int a, int b) { return (a + b - 1) / b; }

}  // namespace rgbid
