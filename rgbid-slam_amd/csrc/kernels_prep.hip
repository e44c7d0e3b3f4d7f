// kernels_prep.hip -- frame preparation for gfx950: converters, copies/fills, Sobel, NaN-aware 5x5
// pyramid reduction, NaN-aware 5x5 bilateral.  Replaces src/cuda/misc.cu, pyrdown.cu, filters.cu of
// the reference (citations per kernel).  Geometry is CDNA4-native: 256-thread workgroups shaped
// 64x4 so each wave64 covers one 256-byte row segment (fully coalesced), blockIdx.z = lane.
#include "kernels.h"
#include "warp_device.h"

// Whole file: no FMA contraction, so every fp32 expression is evaluated operation by operation exactly
// like the scalar oracle (divisions/sqrt are IEEE by hipcc default).  These kernels are bandwidth-bound.
#pragma clang fp contract(off)

namespace rgbid {

static constexpr int TX = 64, TY = 4;  // one wave per tile row
// every workgroup sweeps RPB vertically stacked 64x4 tiles: 4x fewer workgroups to dispatch (a 64-lane 640x480
// launch is 19 200 instead of 76 800 WGs), which matters most for launches whose lanes are predicated off
static constexpr int RPB = 16;   // an all-lanes-off 640x480 x 512-lane launch: 54 us of pure workgroup dispatch at 4, 14 us at 16

static inline dim3 grid2d(int cols, int rows, int B) { return dim3(div_up(cols, TX), div_up(rows, TY * RPB), B); }
#define RGBID_FOR_TILES(y0v) for (int it_ = 0, y0v = blockIdx.y * (TY * RPB); it_ < RPB; ++it_, y0v += TY)

// ---- convertDepth2InvDepth (misc.cu:105-124) ---------------------------------------------------
__global__ __launch_bounds__(256) void k_depth_to_invdepth(ImgB src, ImgB dst, float factor_depth, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x;
  RGBID_FOR_TILES(y0) {
    int y = y0 + threadIdx.y;
    if (x >= dst.cols || y >= dst.rows) continue;
    int value = px<uint16_t>(src, lane, y, x);
    float r = qnan();
    if (value > 0) r = (1.f / factor_depth) * 1000.f / (float)max(0, min(value, 10000));
    px<float>(dst, lane, y, x) = r;
  }
}
void launch_depth_to_invdepth(hipStream_t s, int B, ImgB src, ImgB dst, float factor_depth, LaneMask m) {
  hipLaunchKernelGGL(k_depth_to_invdepth, grid2d(dst.cols, dst.rows, B), dim3(TX, TY), 0, s, src, dst, factor_depth, m);
}

// ---- depth2floatKernel (misc.cu:86-102): u16 millimetres -> metres (bridge function convertDepth2Float, unused by the tracker)
__global__ __launch_bounds__(256) void k_depth_to_float(ImgB src, ImgB dst, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x;
  RGBID_FOR_TILES(y0) {
    int y = y0 + threadIdx.y;
    if (x >= dst.cols || y >= dst.rows) continue;
    int value = px<uint16_t>(src, lane, y, x);
    px<float>(dst, lane, y, x) = value > 0 ? (float)max(0, min(value, 10000)) / 1000.f : qnan();
  }
}
void launch_depth_to_float(hipStream_t s, int B, ImgB src, ImgB dst, LaneMask m) {
  hipLaunchKernelGGL(k_depth_to_float, grid2d(dst.cols, dst.rows, B), dim3(TX, TY), 0, s, src, dst, m);
}
// ---- float2ucharKernel (misc.cu:289-324): grey visualisation of a float map, NaN / inf colour coded (convertFloat2RGB)
__global__ __launch_bounds__(256) void k_float_to_rgb(ImgB src, ImgB dst, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x;
  RGBID_FOR_TILES(y0) {
    int y = y0 + threadIdx.y;
    if (x >= src.cols || y >= src.rows) continue;
    float v = px<float>(src, lane, y, x);
    uint8_t* c = row_ptr<uint8_t>(dst, lane, y) + 3 * x;
    const float min_val = 0.f, max_val = 255.f;
    if (isnan(v)) { c[0] = 200; c[1] = 150; c[2] = 150; }
    else if (isinf(v)) { c[0] = 150; c[1] = 150; c[2] = 250; }
    else {
      uint8_t grey = (uint8_t)max(0, min(f2i_rn(255 * (v - min_val) / (max_val - min_val)), 255));
      c[0] = grey; c[1] = grey; c[2] = grey;
    }
  }
}
void launch_float_to_rgb(hipStream_t s, int B, ImgB src, ImgB dst, LaneMask m) {
  hipLaunchKernelGGL(k_float_to_rgb, grid2d(src.cols, src.rows, B), dim3(TX, TY), 0, s, src, dst, m);
}

// ---- computeIntensity (misc.cu:128-147) / decomposeRGBInChannels (misc.cu:151-172) --------------
__global__ __launch_bounds__(256) void k_intensity(ImgB rgb, ImgB dst, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x;
  RGBID_FOR_TILES(y0) {
    int y = y0 + threadIdx.y;
    if (x >= dst.cols || y >= dst.rows) continue;
    const uint8_t* p = row_ptr<uint8_t>(rgb, lane, y) + 3 * x;
    float v = 0.2126f * (float)p[0] + 0.7152f * (float)p[1] + 0.0722f * (float)p[2];
    px<float>(dst, lane, y, x) = fmaxf(0.f, fminf(v, 255.f));
  }
}
void launch_intensity(hipStream_t s, int B, ImgB rgb, ImgB dst, LaneMask m) {
  hipLaunchKernelGGL(k_intensity, grid2d(dst.cols, dst.rows, B), dim3(TX, TY), 0, s, rgb, dst, m);
}

__global__ __launch_bounds__(256) void k_decompose(ImgB rgb, ImgB r, ImgB g, ImgB b, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x;
  RGBID_FOR_TILES(y0) {
    int y = y0 + threadIdx.y;
    if (x >= r.cols || y >= r.rows) continue;
    const uint8_t* p = row_ptr<uint8_t>(rgb, lane, y) + 3 * x;
    px<float>(r, lane, y, x) = (float)p[0];
    px<float>(g, lane, y, x) = (float)p[1];
    px<float>(b, lane, y, x) = (float)p[2];
  }
}
void launch_decompose_rgb(hipStream_t s, int B, ImgB rgb, ImgB r, ImgB g, ImgB b, LaneMask m) {
  hipLaunchKernelGGL(k_decompose, grid2d(r.cols, r.rows, B), dim3(TX, TY), 0, s, rgb, r, g, b, m);
}

// ---- engine frame preparation: the three converters above in ONE pass, 4 pixels per thread -------------------
// reads the u16 depth (8 B) and the packed rgb (12 B) of a 4-pixel group once, writes five 16-byte vectors
// (iD, luma, r, g, b planes).  Same per-pixel arithmetic as k_depth_to_invdepth / k_intensity / k_decompose.
static inline bool same_geometry(const ImgB& a, const ImgB& b) { return a.rows == b.rows && a.cols == b.cols; }
static inline bool vec4_ok(const ImgB& a, int elem) {
  return ((a.pitch & 15) == 0) && ((a.lane_stride & 15) == 0) && ((((uintptr_t)a.base) & 15) == 0) && (((size_t)a.cols * elem) % (4 * elem) == 0);
}
__global__ __launch_bounds__(256) void k_prep_frame4(ImgB depth, ImgB rgb, ImgB iD, ImgB I, ImgB r, ImgB g, ImgB b, float factor_depth, int cols4, int units, LaneMask m) {
  int lane = blockIdx.y;
  if (!m.on(lane)) return;
  const float scale = (1.f / factor_depth) * 1000.f;
  for (int u = blockIdx.x * 256 + threadIdx.x; u < units; u += gridDim.x * 256) {
    int y = u / cols4, x = (u - y * cols4) * 4;
    uint2 dq = *reinterpret_cast<const uint2*>(row_ptr<uint16_t>(depth, lane, y) + x);
    const uint32_t* cp = reinterpret_cast<const uint32_t*>(row_ptr<uint8_t>(rgb, lane, y) + 3 * x);
    uint32_t c0 = cp[0], c1 = cp[1], c2 = cp[2];
    int dv[4] = {(int)(dq.x & 0xffffu), (int)(dq.x >> 16), (int)(dq.y & 0xffffu), (int)(dq.y >> 16)};
    float R[4] = {(float)(c0 & 255u), (float)(c0 >> 24), (float)((c1 >> 16) & 255u), (float)((c2 >> 8) & 255u)};
    float G[4] = {(float)((c0 >> 8) & 255u), (float)(c1 & 255u), (float)(c1 >> 24), (float)((c2 >> 16) & 255u)};
    float Bl[4] = {(float)((c0 >> 16) & 255u), (float)((c1 >> 8) & 255u), (float)(c2 & 255u), (float)(c2 >> 24)};
    float w[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = dv[i] > 0 ? scale / (float)min(dv[i], 10000) : qnan();
      float v = 0.2126f * R[i] + 0.7152f * G[i] + 0.0722f * Bl[i];
      l[i] = fmaxf(0.f, fminf(v, 255.f));
    }
    st16_stream(row_ptr<float>(iD, lane, y) + x, w[0], w[1], w[2], w[3]);
    st16_stream(row_ptr<float>(I, lane, y) + x, l[0], l[1], l[2], l[3]);
    st16_stream(row_ptr<float>(r, lane, y) + x, R[0], R[1], R[2], R[3]);
    st16_stream(row_ptr<float>(g, lane, y) + x, G[0], G[1], G[2], G[3]);
    st16_stream(row_ptr<float>(b, lane, y) + x, Bl[0], Bl[1], Bl[2], Bl[3]);
  }
}
void launch_prep_frame(hipStream_t s, int B, ImgB depth, ImgB rgb, ImgB iD, ImgB I, ImgB r, ImgB g, ImgB b, float factor_depth, LaneMask m) {
  bool vec = (iD.cols % 4 == 0) && vec4_ok(iD, 4) && vec4_ok(I, 4) && vec4_ok(r, 4) && vec4_ok(g, 4) && vec4_ok(b, 4) &&
             ((depth.pitch & 7) == 0) && ((depth.lane_stride & 7) == 0) && ((((uintptr_t)depth.base) & 7) == 0) &&
             ((rgb.pitch & 3) == 0) && ((rgb.lane_stride & 3) == 0) && ((((uintptr_t)rgb.base) & 3) == 0);
  if (!vec) {
    launch_intensity(s, B, rgb, I, m);
    launch_decompose_rgb(s, B, rgb, r, g, b, m);
    launch_depth_to_invdepth(s, B, depth, iD, factor_depth, m);
    return;
  }
  int cols4 = iD.cols / 4, units = cols4 * iD.rows;
  hipLaunchKernelGGL(k_prep_frame4, dim3(div_up(units, 256 * 2), B), dim3(256), 0, s, depth, rgb, iD, I, r, g, b, factor_depth, cols4, units, m);
}

// ---- gradientKernel (misc.cu:176-220): 3x3 Sobel/8, replicate border -----------------------------
// LDS tile (TY+2)x(TX+2): one coalesced load of the halo'd tile, 9 taps from LDS.  Accumulation order
// is the reference's (dx outer, dy inner) so the oracle comparison is exact.
__global__ __launch_bounds__(256) void k_gradient(ImgB src, ImgB gx, ImgB gy, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  __shared__ float tile[TY + 2][TX + 2 + 1];
  int x0 = blockIdx.x * TX;
  int tid = threadIdx.y * TX + threadIdx.x;
  RGBID_FOR_TILES(y0) {
    if (y0 >= src.rows) break;
    __syncthreads();
    for (int i = tid; i < (TY + 2) * (TX + 2); i += TX * TY) {
      int ty = i / (TX + 2), tx = i - ty * (TX + 2);
      int cx = min(max(0, x0 + tx - 1), src.cols - 1);
      int cy = min(max(0, y0 + ty - 1), src.rows - 1);
      tile[ty][tx] = px<float>(src, lane, cy, cx);
    }
    __syncthreads();
    int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= src.cols || y >= src.rows) continue;
    float res_hor = 0.f, res_vert = 0.f;
#pragma unroll
    for (int dx = -1; dx < 2; dx++)
#pragma unroll
      for (int dy = -1; dy < 2; dy++) {
        float t = tile[threadIdx.y + 1 + dy][threadIdx.x + 1 + dx];
        res_hor += t * (float)(dx * (2 - dy * dy));
        res_vert += t * (float)(dy * (2 - dx * dx));
      }
    px<float>(gx, lane, y, x) = res_hor / 8.f;
    px<float>(gy, lane, y, x) = res_vert / 8.f;
  }
}
// 4 pixels per thread with a rolling 3-row window in registers: per output row one 16-byte load plus the two
// neighbour columns (cache hits), two 16-byte stores.  No LDS, no barriers; tap order and arithmetic as above.
static constexpr int GR_ROWS = 8;  // output rows per thread
__device__ __forceinline__ void sobel_load_row(const ImgB& src, int lane, int y, int x, int xl, int xr, float r[6]) {
  const float* rp = row_ptr<float>(src, lane, y);
  float4 v = *reinterpret_cast<const float4*>(rp + x);
  r[0] = rp[xl]; r[1] = v.x; r[2] = v.y; r[3] = v.z; r[4] = v.w; r[5] = rp[xr];
}
// COPY: the source row is also written to `keep` (the keyframe's own copy of a current-frame map: the copy kernel's 4 B/px read and one launch saved)
struct GradSet { ImgB src, gx, gy, keep; };
// blockIdx.z: which of (up to) two maps of the same geometry (intensity and inverse depth of a pyramid level: one launch instead of two)
template <bool COPY>
__global__ __launch_bounds__(256) void k_gradient4(GradSet s0, GradSet s1, int cols4, int strips, LaneMask m) {
  int lane = blockIdx.y;
  if (!m.on(lane)) return;
  const GradSet& S = blockIdx.z ? s1 : s0;
  const ImgB& src = S.src; const ImgB& gx = S.gx; const ImgB& gy = S.gy; const ImgB& keep = S.keep;
  int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= cols4 * strips) return;
  int strip = u / cols4, x = (u - strip * cols4) * 4;
  int y_begin = strip * GR_ROWS, y_end = min(y_begin + GR_ROWS, src.rows);
  int xl = max(x - 1, 0), xr = min(x + 4, src.cols - 1);
  float a[6], b[6], c[6];  // rows y-1, y, y+1 (replicated at the border)
  sobel_load_row(src, lane, max(y_begin - 1, 0), x, xl, xr, a);
  sobel_load_row(src, lane, y_begin, x, xl, xr, b);
  for (int y = y_begin; y < y_end; ++y) {
    sobel_load_row(src, lane, min(y + 1, src.rows - 1), x, xl, xr, c);
    float h[4], v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float res_hor = 0.f, res_vert = 0.f;
#pragma unroll
      for (int dx = -1; dx < 2; dx++)
#pragma unroll
        for (int dy = -1; dy < 2; dy++) {
          float t = dy < 0 ? a[i + 1 + dx] : (dy == 0 ? b[i + 1 + dx] : c[i + 1 + dx]);
          res_hor += t * (float)(dx * (2 - dy * dy));
          res_vert += t * (float)(dy * (2 - dx * dx));
        }
      h[i] = res_hor / 8.f; v[i] = res_vert / 8.f;
    }
    st16_stream(row_ptr<float>(gx, lane, y) + x, h[0], h[1], h[2], h[3]);
    st16_stream(row_ptr<float>(gy, lane, y) + x, v[0], v[1], v[2], v[3]);
    if (COPY) st16_stream(row_ptr<float>(keep, lane, y) + x, b[1], b[2], b[3], b[4]);
#pragma unroll
    for (int i = 0; i < 6; ++i) { a[i] = b[i]; b[i] = c[i]; }
  }
}
void launch_gradient(hipStream_t s, int B, ImgB src, ImgB gx, ImgB gy, LaneMask m) {
  if ((src.cols % 4 == 0) && vec4_ok(src, 4) && vec4_ok(gx, 4) && vec4_ok(gy, 4)) {
    int cols4 = src.cols / 4, strips = div_up(src.rows, GR_ROWS);
    const GradSet g{src, gx, gy, src};
    hipLaunchKernelGGL(k_gradient4<false>, dim3(div_up(cols4 * strips, 256), B), dim3(256), 0, s, g, g, cols4, strips, m);
    return;
  }
  hipLaunchKernelGGL(k_gradient, grid2d(src.cols, src.rows, B), dim3(TX, TY), 0, s, src, gx, gy, m);
}
void launch_gradient2(hipStream_t s, int B, ImgB src0, ImgB gx0, ImgB gy0, ImgB src1, ImgB gx1, ImgB gy1, LaneMask m) {
  const bool v = (src0.cols % 4 == 0) && same_geometry(src0, src1) && vec4_ok(src0, 4) && vec4_ok(gx0, 4) && vec4_ok(gy0, 4) && vec4_ok(src1, 4) && vec4_ok(gx1, 4) && vec4_ok(gy1, 4);
  if (!v) { launch_gradient(s, B, src0, gx0, gy0, m); launch_gradient(s, B, src1, gx1, gy1, m); return; }
  int cols4 = src0.cols / 4, strips = div_up(src0.rows, GR_ROWS);
  hipLaunchKernelGGL(k_gradient4<false>, dim3(div_up(cols4 * strips, 256), B, 2), dim3(256), 0, s, GradSet{src0, gx0, gy0, src0}, GradSet{src1, gx1, gy1, src1}, cols4, strips, m);
}
// Sobel pair of `src` + a copy of `src` into `keep` in one pass (keyframe switch: the current-frame map becomes the keyframe's); false: the
// geometry is not the 16-byte path's, nothing launched (the caller copies, then takes the gradient)
bool launch_gradient_keep(hipStream_t s, int B, ImgB src, ImgB gx, ImgB gy, ImgB keep, LaneMask m) {
  if (!((src.cols % 4 == 0) && vec4_ok(src, 4) && vec4_ok(gx, 4) && vec4_ok(gy, 4) && vec4_ok(keep, 4) && keep.rows == src.rows && keep.cols == src.cols)) return false;
  int cols4 = src.cols / 4, strips = div_up(src.rows, GR_ROWS);
  const GradSet g{src, gx, gy, keep};
  hipLaunchKernelGGL(k_gradient4<true>, dim3(div_up(cols4 * strips, 256), B), dim3(256), 0, s, g, g, cols4, strips, m);
  return true;
}
static bool gradient_keep_ok(const ImgB& src, const ImgB& gx, const ImgB& gy, const ImgB& keep) {
  return (src.cols % 4 == 0) && vec4_ok(src, 4) && vec4_ok(gx, 4) && vec4_ok(gy, 4) && vec4_ok(keep, 4) && keep.rows == src.rows && keep.cols == src.cols;
}
// both maps of a level in one launch; false: nothing launched (the caller takes the one-map path for each)
bool launch_gradient_keep2(hipStream_t s, int B, ImgB src0, ImgB gx0, ImgB gy0, ImgB keep0, ImgB src1, ImgB gx1, ImgB gy1, ImgB keep1, LaneMask m) {
  if (!(gradient_keep_ok(src0, gx0, gy0, keep0) && gradient_keep_ok(src1, gx1, gy1, keep1) && same_geometry(src0, src1))) return false;
  int cols4 = src0.cols / 4, strips = div_up(src0.rows, GR_ROWS);
  hipLaunchKernelGGL(k_gradient4<true>, dim3(div_up(cols4 * strips, 256), B, 2), dim3(256), 0, s, GradSet{src0, gx0, gy0, keep0}, GradSet{src1, gx1, gy1, keep1}, cols4, strips, m);
  return true;
}

// ---- engine: vertex map + Sobel + normal map of the fused keyframe in ONE pass --------------------------------------------------
// createVMap (maps.cu:63-90), computeGradientDepth (misc.cu:176-220) and createNMapGradients (maps.cu:134-179) run back to back on the
// same inverse-depth map after every fusion step (visodo.cpp:1758-1762, 886-892).  As three kernels they move 4+12, 4+8 and 12+12 B/px
// plus the read-modify-write blends of the two 4-px kernels; fused, the map is read once through k_gradient4's rolling 3-row window, the
// gradients stay in registers, and the six output planes are written with full 16-byte stores (24 B/px).  Per-pixel arithmetic is that
// of the three kernels, so every value the reference defines is bit-identical.  One deliberate difference: planes 1 and 2 of an INVALID
// pixel (plane 0 = NaN), which the reference leaves untouched -- i.e. stale or uninitialised -- are written as NaN here.
// (The reciprocal / normalisations as v_rcp_f32 / v_rsq_f32 instead of the exact sequences change nothing: 1.76 vs 1.69 us per lane at 1 024
// lanes -- the kernel is bound by its 24 B/px of stores -- so there is only the exact class.)
__global__ __launch_bounds__(256) void k_kf_maps4(ImgB src, ImgB vmap, ImgB nmap, IntrP k, int cols4, int strips, LaneMask m) {
  int lane = blockIdx.y;
  if (!m.on(lane)) return;
  int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= cols4 * strips) return;
  const int rows = src.rows;
  int strip = u / cols4, x = (u - strip * cols4) * 4;
  int y_begin = strip * GR_ROWS, y_end = min(y_begin + GR_ROWS, rows);
  int xl = max(x - 1, 0), xr = min(x + 4, src.cols - 1);
  const float fx_inv = 1.f / k.fx, fy_inv = 1.f / k.fy;
  float a[6], b[6], c[6];  // rows y-1, y, y+1 (replicated at the border)
  sobel_load_row(src, lane, max(y_begin - 1, 0), x, xl, xr, a);
  sobel_load_row(src, lane, y_begin, x, xl, xr, b);
  for (int y = y_begin; y < y_end; ++y) {
    sobel_load_row(src, lane, min(y + 1, rows - 1), x, xl, xr, c);
    float X[4], Y[4], Z[4], N0[4], N1[4], N2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // Sobel / 8 (k_gradient4)
      float res_hor = 0.f, res_vert = 0.f;
#pragma unroll
      for (int dx = -1; dx < 2; dx++)
#pragma unroll
        for (int dy = -1; dy < 2; dy++) {
          float t = dy < 0 ? a[i + 1 + dx] : (dy == 0 ? b[i + 1 + dx] : c[i + 1 + dx]);
          res_hor += t * (float)(dx * (2 - dy * dy));
          res_vert += t * (float)(dy * (2 - dx * dx));
        }
      const float gx = res_hor / 8.f, gy = res_vert / 8.f;
      const float w = b[i + 1];
      const float uu = (float)(x + i), vv = (float)y;
      // vertex (k_vmap4)
      const float z = rcp_exact(w);
      const bool okv = !isnan(z);
      X[i] = okv ? z * (uu - k.cx) * fx_inv : qnan();
      Y[i] = z * (vv - k.cy) * fy_inv;     // NaN when invalid
      Z[i] = z;
      // normal (k_nmap_grad4)
      float nx = gx * k.fx, ny = gy * k.fy, nz = gx * (k.cx - uu) + gy * (k.cy - vv) + w;
      float rn = rcp_exact(sqrtf(nx * nx + ny * ny + nz * nz));
      nx *= rn; ny *= rn; nz *= rn;
      float vx = z * (uu - k.cx) * (1.f / k.fx), vy = z * (vv - k.cy) * (1.f / k.fy), vz = z;
      float rv = rcp_exact(sqrtf(vx * vx + vy * vy + vz * vz));
      vx *= rv; vy *= rv; vz *= rv;
      float acos_vn = vx * nx + vy * ny + vz * nz;
      const bool keep = !(isnan(w) || isnan(gx) || isnan(gy)) && ((double)acos_vn > 0.1);
      N0[i] = keep ? nx : qnan(); N1[i] = keep ? ny : qnan(); N2[i] = keep ? nz : qnan();
    }
    st16_stream(row_ptr<float>(vmap, lane, y) + x, X[0], X[1], X[2], X[3]);
    st16_stream(row_ptr<float>(vmap, lane, y + rows) + x, Y[0], Y[1], Y[2], Y[3]);
    st16_stream(row_ptr<float>(vmap, lane, y + 2 * rows) + x, Z[0], Z[1], Z[2], Z[3]);
    st16_stream(row_ptr<float>(nmap, lane, y) + x, N0[0], N0[1], N0[2], N0[3]);
    st16_stream(row_ptr<float>(nmap, lane, y + rows) + x, N1[0], N1[1], N1[2], N1[3]);
    st16_stream(row_ptr<float>(nmap, lane, y + 2 * rows) + x, N2[0], N2[1], N2[2], N2[3]);
#pragma unroll
    for (int i = 0; i < 6; ++i) { a[i] = b[i]; b[i] = c[i]; }
  }
}
bool launch_kf_maps(hipStream_t s, int B, ImgB depthinv, ImgB vmap, ImgB nmap, IntrP k, LaneMask m) {
  if (!((depthinv.cols % 4 == 0) && vec4_ok(depthinv, 4) && vec4_ok(vmap, 4) && vec4_ok(nmap, 4))) return false;  // caller falls back to the three kernels
  int cols4 = depthinv.cols / 4, strips = div_up(depthinv.rows, GR_ROWS);
  hipLaunchKernelGGL(k_kf_maps4, dim3(div_up(cols4 * strips, 256), B), dim3(256), 0, s, depthinv, vmap, nmap, k, cols4, strips, m);
  return true;
}

// ---- copies / fills (misc.cu:225-287,327-341) --------------------------------------------------
// row-wise byte copy: 16 B per thread where the row allows it
__global__ __launch_bounds__(256) void k_copy_rows(ImgB src, ImgB dst, int row_bytes, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  for (int y = blockIdx.y; y < src.rows; y += gridDim.y) {   // a workgroup walks rows: few workgroups to dispatch when the lane is off
    const char* sp = row_ptr<char>(src, lane, y);
    char* dp = row_ptr<char>(dst, lane, y);
    bool vec = ((row_bytes & 15) == 0) && ((((uintptr_t)sp | (uintptr_t)dp) & 15) == 0);
    if (vec) {
      int n16 = row_bytes >> 4;
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x)
        { typedef float f4v __attribute__((ext_vector_type(4))); __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const f4v*>(sp) + i), reinterpret_cast<f4v*>(dp) + i); }   // one pass over both images
    } else {
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < row_bytes; i += gridDim.x * blockDim.x) dp[i] = sp[i];
    }
  }
}
void launch_copy_bytes(hipStream_t s, int B, ImgB src, ImgB dst, int elem_size, LaneMask m) {
  int row_bytes = src.cols * elem_size;
  int gx = max(1, min(8, div_up(row_bytes / 16, 256)));
  hipLaunchKernelGGL(k_copy_rows, dim3(gx, min(src.rows, 32), B), dim3(256), 0, s, src, dst, row_bytes, m);
}

__global__ __launch_bounds__(256) void k_fill(ImgB dst, int elem_size, uint32_t bits, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x;
  RGBID_FOR_TILES(y0) {
    int y = y0 + threadIdx.y;
    if (x >= dst.cols || y >= dst.rows) continue;
    if (elem_size == 4) px<uint32_t>(dst, lane, y, x) = bits;
    else px<uint8_t>(dst, lane, y, x) = (uint8_t)bits;
  }
}
void launch_fill(hipStream_t s, int B, ImgB dst, int elem_size, uint32_t bits, LaneMask m) {
  hipLaunchKernelGGL(k_fill, grid2d(dst.cols, dst.rows, B), dim3(TX, TY), 0, s, dst, elem_size, bits, m);
}

// ---- pyrDownKernelGridStridef (pyrdown.cu:84-132) -------------------------------------------------
// 5x5 window [2x-2, 2x+2] x [2y-2, 2y+2] clipped to the image, NaN taps skipped, `count > 12` validity rule, tap order
// cy outer / cx inner -- all as in the reference.  The Gaussian weights exp(-d2/2), d2 = dx^2+dy^2 in {0,1,2,4,5,8}, are
// evaluated ONCE on the host with expf (the reference evaluates __expf per tap: 25 transcendentals per output pixel)
// and reach the kernel as scalar arguments.
struct PyrWeights { float w[9]; float sum_all; };
static const PyrWeights& pyr_weights() {
  static const PyrWeights W = [] {
    PyrWeights t;
    for (int d2 = 0; d2 < 9; ++d2) t.w[d2] = expf(-((float)d2 * 0.5f));
    // the weight sum of a window without an invalid tap, added in the kernel's tap order: fmaf(1, w, s) = fl(s + w), so this IS the value the mask-weighted
    // chain reaches when every mask is 1 (round 6: k_pyr_down_dpp's all-valid fast path divides by it)
    volatile float s = 0.f;
    for (int dy = -2; dy <= 2; ++dy) for (int dx = -2; dx <= 2; ++dx) s = s + t.w[dx * dx + dy * dy];
    t.sum_all = getenv("RGBID_PYRDOWN_NO_FASTPATH") ? 0.f : (float)s;   // 0: the fast path is off (bit-identity test, A/B timing)
    return t;
  }();
  return W;
}
static constexpr int PD_ROWS = 8;    // output rows a thread walks down (16 rows: < 4 % either way)
static constexpr int PD_AHEAD = 1;   // outputs whose two new source rows are in flight ahead of the one being computed (deeper: < 4 %)
// A thread owns one output column and walks PD_ROWS output rows downwards with the 5-row source window in registers (no LDS, no barriers); the
// row loop is fully unrolled (the window lives in renamed registers) and software-pipelined: the two source rows of output j + 1 are ISSUED
// before output j is computed and only unpacked after it.  The window COLUMNS are shared between neighbouring lanes: a thread that loaded and
// sanitised all five columns of each row itself (round 2: three 8-byte loads and 15 compare / select instructions per source row, 0.47 us per
// lane) repeats what its neighbours hold.  Here lane l of a wave owns the source column pair
// (2x, 2x + 1) of output column x = 62 wave + l - 1: ONE coalesced 8-byte load per lane and row (a wave reads 512 contiguous bytes), two values
// sanitised, and the other three window columns arrive by DPP wave shifts from lanes l - 1 (columns 2x - 2, 2x - 1) and l + 1 (column 2x + 2).
// Lanes 0 and 63 of a wave are halo providers only (62 outputs per wave), so no lane ever needs data of another wave.  Same tap order, unfused
// sum1, exact-FMA mask sum (adding +0 for an invalid tap leaves the sums bit-identical to skipping it: they start at +0 and can never become
// -0; m * w is exact) and an integer-valued tap count: bit-identical to the oracle's validity pattern.  0.34 us per lane (0.56 of the HBM peak
// on 5 B per source pixel), VALU-bound at 170 instructions per output.
template <int CTRL>
__device__ __forceinline__ float dpp_shift(float v) {   // 0x138 = wave_shr:1 (lane l reads lane l - 1), 0x130 = wave_shl:1 (lane l reads lane l + 1); edge lanes get 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
struct PyrRow { float v[5], m[5]; };
__device__ __forceinline__ float2 pyr_issue_pair(const FMap& S, int rows, int cy, unsigned colb) {
  const unsigned rb = S.row((cy >= 0 && cy < rows) ? cy : 0);
  return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(S.rsrc, rb + colb, 0, 0));
}
// returns: both columns of EVERY lane of the wave are valid in this row (wave-uniform)
__device__ __forceinline__ bool pyr_finish_shared(float2 p, bool row_in, bool c0, bool c1, PyrRow& r) {
  const bool ok0 = row_in && c0 && !isnan(p.x), ok1 = row_in && c1 && !isnan(p.y);
  const float v0 = ok0 ? p.x : 0.f, v1 = ok1 ? p.y : 0.f, m0 = ok0 ? 1.f : 0.f, m1 = ok1 ? 1.f : 0.f;
  r.v[2] = v0; r.v[3] = v1; r.m[2] = m0; r.m[3] = m1;
  r.v[0] = dpp_shift<0x138>(v0); r.v[1] = dpp_shift<0x138>(v1); r.m[0] = dpp_shift<0x138>(m0); r.m[1] = dpp_shift<0x138>(m1);
  r.v[4] = dpp_shift<0x130>(v0); r.m[4] = dpp_shift<0x130>(m0);
  return __ballot(ok0 && ok1) == ~0ull;
}
static constexpr int PD_WAVE_OUT = 62;   // outputs per wave (lanes 1 .. 62)
// src1 / dst1: a second map of the SAME geometry reduced by the same launch (the intensity and inverse-depth pyramids of a frame: one launch per
// level instead of two -- below ~100 lanes a step is bound by its number of dependent launches); its workgroups are the second half of the grid
__global__ __launch_bounds__(256) void k_pyr_down_dpp(ImgB src0, ImgB dst0, ImgB src1, ImgB dst1, PyrWeights W, int strips, int wpr, int wgs_per_lane, int wgs_per_map, LaneMask m) {
  // 1-D grid in XCD-contiguous order (common.h): the workgroups of vertically adjacent strips -- which share 3 of their 19 source rows -- run on ONE
  // XCD one after the other and find the shared rows in its L2 (with the natural order they sit on different XCDs and each fetches them from HBM:
  // round 3 counted 1.48 x the algorithmic reads)
  unsigned V = xcd_slab_index(blockIdx.x, gridDim.x);
  const bool second = V >= (unsigned)wgs_per_map;   // wave-uniform
  if (second) V -= (unsigned)wgs_per_map;
  const ImgB& src = second ? src1 : src0;
  const ImgB& dst = second ? dst1 : dst0;
  const int lane = (int)(V / (unsigned)wgs_per_lane), wg = (int)(V - (unsigned)lane * (unsigned)wgs_per_lane);
  if (!m.on(lane)) return;
  const int wave = (wg * 256 + threadIdx.x) >> 6, lid = threadIdx.x & 63;
  if (wave >= strips * wpr) return;                       // whole waves only: every lane of a live wave takes part in the shifts
  const int strip = wave / wpr, wx = wave - strip * wpr;
  const int x = wx * PD_WAVE_OUT + lid - 1;                // -1 and >= dst.cols: halo / idle lanes (they still provide their column pair)
  const int y_begin = strip * PD_ROWS;
  const FMap S(src, lane);
  const bool c0 = x >= 0 && 2 * x < src.cols, c1 = x >= 0 && 2 * x + 1 < src.cols;
  const unsigned colb = (unsigned)max(2 * x, 0) << 3 >> 1;   // byte offset of column 2x (0 for the halo lane left of the image, whose values are masked off)
  const bool writer = lid >= 1 && lid <= PD_WAVE_OUT && x < dst.cols;
  PyrRow win[2 * PD_ROWS + 3];
  bool full[2 * PD_ROWS + 3];   // round 6: rows without an invalid tap anywhere in the wave (scalars) -- five of them in a row make an all-valid window for every output
  float2 raw[2 * PD_ROWS + 3];
#pragma unroll
  for (int r = 0; r < 3 + 2 * PD_AHEAD; ++r) raw[r] = pyr_issue_pair(S, src.rows, 2 * y_begin - 2 + r, colb);
#pragma unroll
  for (int r = 0; r < 3; ++r) { const int cy = 2 * y_begin - 2 + r; full[r] = pyr_finish_shared(raw[r], cy >= 0 && cy < src.rows, c0, c1, win[r]); }
#pragma unroll
  for (int j = 0; j < PD_ROWS; ++j) {
    const int y = y_begin + j;
    if (y < dst.rows) {   // wave-uniform
      if (j + PD_AHEAD < PD_ROWS && y + PD_AHEAD < dst.rows) {
        raw[2 * (j + PD_AHEAD) + 3] = pyr_issue_pair(S, src.rows, 2 * (y + PD_AHEAD) + 1, colb);
        raw[2 * (j + PD_AHEAD) + 4] = pyr_issue_pair(S, src.rows, 2 * (y + PD_AHEAD) + 2, colb);
      }
      full[2 * j + 3] = pyr_finish_shared(raw[2 * j + 3], 2 * y + 1 < src.rows, c0, c1, win[2 * j + 3]);
      full[2 * j + 4] = pyr_finish_shared(raw[2 * j + 4], 2 * y + 2 < src.rows, c0, c1, win[2 * j + 4]);
      if (W.sum_all > 0.f && full[2 * j] && full[2 * j + 1] && full[2 * j + 2] && full[2 * j + 3] && full[2 * j + 4]) {
        // all 25 taps of every output of the wave are valid (an intensity map away from the image border, an inverse-depth map without holes here): the
        // tap count is 25 and the mask-weighted sum is the constant the same chain reaches with every mask at 1 (PyrWeights::sum_all) -- the 50 instructions
        // that form them are skipped; sum1 is the same chain, the quotient the same IEEE division: bit-identical
        float sum1 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) sum1 = sum1 + win[2 * j + dy].v[dx] * W.w[(dx - 2) * (dx - 2) + (dy - 2) * (dy - 2)];
        if (writer) px<float>(dst, lane, y, x) = sum1 / W.sum_all;
        continue;
      }
      // (a per-row tap count shared by the outputs that hold the row saves 12 adds per output but costs 6 VGPRs: 7 -> 6 waves / SIMD, 0.34 -> 0.43 us
      // per lane; forcing 8 waves / SIMD spills: 0.49)
      float sum1 = 0.f, sum2 = 0.f, count = 0.f;
#pragma unroll
      for (int dy = 0; dy < 5; ++dy)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
          const float weight = W.w[(dx - 2) * (dx - 2) + (dy - 2) * (dy - 2)];
          sum1 = sum1 + win[2 * j + dy].v[dx] * weight;
          sum2 = __builtin_fmaf(win[2 * j + dy].m[dx], weight, sum2);
          count += win[2 * j + dy].m[dx];                  // small integers: exact in fp32
        }
      if (writer) px<float>(dst, lane, y, x) = count > 12.f ? sum1 / sum2 : qnan();
    }
  }
}
void launch_pyr_down(hipStream_t s, int B, ImgB src, ImgB dst, LaneMask m) {
  int strips = div_up(dst.rows, PD_ROWS);
  const int wpr = div_up(dst.cols, PD_WAVE_OUT);
  const int wgs = div_up(strips * wpr, 4);
  hipLaunchKernelGGL(k_pyr_down_dpp, dim3((unsigned)wgs * (unsigned)B), dim3(256), 0, s, src, dst, src, dst, pyr_weights(), strips, wpr, wgs, wgs * B, m);
}
void launch_pyr_down2(hipStream_t s, int B, ImgB src0, ImgB dst0, ImgB src1, ImgB dst1, LaneMask m) {
  if (!(same_geometry(src0, src1) && same_geometry(dst0, dst1))) { launch_pyr_down(s, B, src0, dst0, m); launch_pyr_down(s, B, src1, dst1, m); return; }
  int strips = div_up(dst0.rows, PD_ROWS);
  const int wpr = div_up(dst0.cols, PD_WAVE_OUT);
  const int wgs = div_up(strips * wpr, 4);
  hipLaunchKernelGGL(k_pyr_down_dpp, dim3(2u * (unsigned)wgs * (unsigned)B), dim3(256), 0, s, src0, dst0, src1, dst1, pyr_weights(), strips, wpr, wgs, wgs * B, m);
}

// (the bilateral filter lives in kernels_bilateral.hip)
// exhaustive check of div_const_fast for one constant: every x whose fast result is flagged ok must equal x / c bit for bit (a zero result
// only up to its sign); returns the number of violations over all 2^32 bit patterns of x
__global__ __launch_bounds__(256) void k_selftest_div_const(DivConst d, unsigned long long* mismatches) {
  const uint32_t hi = blockIdx.x;
  unsigned int bad = 0;
  for (uint32_t lo = threadIdx.x; lo < 65536u; lo += blockDim.x) {
    const float x = __uint_as_float((hi << 16) | lo);
    bool ok;
    const float a = div_const_fast(x, d, ok), b = x / d.c;
    const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a == 0.f && b == 0.f);
    bad += (ok && !same) ? 1u : 0u;
  }
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}
void launch_selftest_div_const(hipStream_t s, float c, unsigned long long* mismatches_dev) {
  hipLaunchKernelGGL(k_selftest_div_const, dim3(65536), dim3(256), 0, s, DivConst{c, 1.0f / c}, mismatches_dev);
}

}  // namespace rgbid
