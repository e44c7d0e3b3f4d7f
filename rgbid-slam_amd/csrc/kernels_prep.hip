// kernels_prep.hip -- frame preparation for gfx950: converters, copies/fills, Sobel, NaN-aware 5x5
// pyramid reduction, NaN-aware 5x5 bilateral.  Replaces src/cuda/misc.cu, pyrdown.cu, filters.cu of
// the reference (citations per kernel).  Geometry is CDNA4-native: 256-thread workgroups shaped
// 64x4 so each wave64 covers one 256-byte row segment (fully coalesced), blockIdx.z = lane.
#include "kernels.h"

// Whole file: no FMA contraction, so every fp32 expression is evaluated operation by operation exactly
// like the scalar oracle (divisions/sqrt are IEEE by hipcc default).  These kernels are bandwidth-bound.
#pragma clang fp contract(off)

namespace rgbid {

static constexpr int TX = 64, TY = 4;  // one wave per tile row
// every workgroup sweeps RPB vertically stacked 64x4 tiles: 4x fewer workgroups to dispatch (a 64-lane 640x480
// launch is 19 200 instead of 76 800 WGs), which matters most for launches whose lanes are predicated off
static constexpr int RPB = 4;

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
void launch_gradient(hipStream_t s, int B, ImgB src, ImgB gx, ImgB gy, LaneMask m) {
  hipLaunchKernelGGL(k_gradient, grid2d(src.cols, src.rows, B), dim3(TX, TY), 0, s, src, gx, gy, m);
}

// ---- copies / fills (misc.cu:225-287,327-341) --------------------------------------------------
// row-wise byte copy: 16 B per thread where the row allows it
__global__ __launch_bounds__(256) void k_copy_rows(ImgB src, ImgB dst, int row_bytes, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int y = blockIdx.y;
  const char* sp = row_ptr<char>(src, lane, y);
  char* dp = row_ptr<char>(dst, lane, y);
  bool vec = ((row_bytes & 15) == 0) && ((((uintptr_t)sp | (uintptr_t)dp) & 15) == 0);
  if (vec) {
    int n16 = row_bytes >> 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x)
      reinterpret_cast<float4*>(dp)[i] = reinterpret_cast<const float4*>(sp)[i];
  } else {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < row_bytes; i += gridDim.x * blockDim.x) dp[i] = sp[i];
  }
}
void launch_copy_bytes(hipStream_t s, int B, ImgB src, ImgB dst, int elem_size, LaneMask m) {
  int row_bytes = src.cols * elem_size;
  int gx = max(1, min(8, div_up(row_bytes / 16, 256)));
  hipLaunchKernelGGL(k_copy_rows, dim3(gx, src.rows, B), dim3(256), 0, s, src, dst, row_bytes, m);
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
// dst tile 64x4 <- src tile (2*64+3) x (2*4+3) staged in LDS (window [2x-2, 2x+2] clipped).  The 5x5
// Gaussian weights exp(-d2/2), d2 in {0,1,2,4,5,8}, are evaluated with expf like the reference; the
// validity rule `count > 12` and the tap order (cy outer, cx inner) are the reference's.
static constexpr int PSX = 2 * TX + 3, PSY = 2 * TY + 3;
// exp(-d2/2) for the nine possible squared tap distances d2 = dx^2 + dy^2 in {0,1,2,4,5,8}: evaluated ONCE on the host
// with expf (the reference evaluates __expf per tap: 25 transcendentals per output pixel)
struct PyrWeights { float w[9]; };
static const PyrWeights& pyr_weights() {
  static const PyrWeights W = [] { PyrWeights t; for (int d2 = 0; d2 < 9; ++d2) t.w[d2] = expf(-((float)d2 * 0.5f)); return t; }();
  return W;
}
__global__ __launch_bounds__(256) void k_pyr_down(ImgB src, ImgB dst, PyrWeights W, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  __shared__ float tile[PSY][PSX + 1];
  __shared__ float wt[9];
  int x0 = blockIdx.x * TX;
  int tid = threadIdx.y * TX + threadIdx.x;
  if (tid < 9) wt[tid] = W.w[tid];
  RGBID_FOR_TILES(y0) {
    if (y0 >= dst.rows) break;
    int sx0 = 2 * x0 - 2, sy0 = 2 * y0 - 2;
    __syncthreads();
    for (int i = tid; i < PSY * PSX; i += TX * TY) {
      int ty = i / PSX, tx = i - ty * PSX;
      int cx = sx0 + tx, cy = sy0 + ty;
      float v = qnan();
      if (cx >= 0 && cy >= 0 && cx < src.cols && cy < src.rows) v = px<float>(src, lane, cy, cx);
      tile[ty][tx] = v;
    }
    __syncthreads();
    int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= dst.cols || y >= dst.rows) continue;
    const int br = 2;
    int tx_end = min(2 * x + br + 1, src.cols), ty_end = min(2 * y + br + 1, src.rows);
    float sum1 = 0.f, sum2 = 0.f;
    int count = 0;
    for (int cy = max(0, 2 * y - br); cy < ty_end; ++cy)
      for (int cx = max(0, 2 * x - br); cx < tx_end; ++cx) {
        float val = tile[cy - sy0][cx - sx0];
        if (!isnan(val)) {
          float weight = wt[(2 * x - cx) * (2 * x - cx) + (2 * y - cy) * (2 * y - cy)];
          sum1 += val * weight;
          sum2 += weight;
          ++count;
        }
      }
    float res = qnan();
    if (count > 12) res = sum1 / sum2;
    px<float>(dst, lane, y, x) = res;
  }
}
void launch_pyr_down(hipStream_t s, int B, ImgB src, ImgB dst, LaneMask m) {
  hipLaunchKernelGGL(k_pyr_down, grid2d(dst.cols, dst.rows, B), dim3(TX, TY), 0, s, src, dst, pyr_weights(), m);
}

// ---- bilateralKernel (filters.cu:86-135), clipped 5x5 window -------------------------------------
static constexpr int BR = 2;
__global__ __launch_bounds__(256) void k_bilateral(ImgB src, ImgB dst, float sigma_floatmap, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  __shared__ float tile[TY + 2 * BR][TX + 2 * BR + 1];
  int x0 = blockIdx.x * TX;
  int tid = threadIdx.y * TX + threadIdx.x;
  RGBID_FOR_TILES(y0) {
    if (y0 >= src.rows) break;
    __syncthreads();
    for (int i = tid; i < (TY + 2 * BR) * (TX + 2 * BR); i += TX * TY) {
      int ty = i / (TX + 2 * BR), tx = i - ty * (TX + 2 * BR);
      int cx = x0 + tx - BR, cy = y0 + ty - BR;
      float v = qnan();
      if (cx >= 0 && cy >= 0 && cx < src.cols && cy < src.rows) v = px<float>(src, lane, cy, cx);
      tile[ty][tx] = v;
    }
    __syncthreads();
    int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= src.cols || y >= src.rows) continue;
    float value = tile[threadIdx.y + BR][threadIdx.x + BR];
    if (isnan(value)) { px<float>(dst, lane, y, x) = qnan(); continue; }
    int tx_end = min(x + BR + 1, src.cols), ty_end = min(y + BR + 1, src.rows);
    float sum1 = 0.f, sum2 = 0.f;
    const float sigma_space = 5.f;
    const float s2ih = (float)(0.5 / (double)(sigma_space * sigma_space));
    for (int cy = max(y - BR, 0); cy < ty_end; ++cy)
      for (int cx = max(x - BR, 0); cx < tx_end; ++cx) {
        float tmp = tile[cy - y0 + BR][cx - x0 + BR];
        if (!isnan(tmp)) {
          float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
          float fn = (value - tmp) / sigma_floatmap;
          // the source mixes float and double here (`0.5*fn*fn`): keep the double evaluation
          double arg = (double)(s2ih * space2) + (0.5 * (double)fn) * (double)fn;
          float weight = expf((float)(-arg));
          sum1 += tmp * weight;
          sum2 += weight;
        }
      }
    px<float>(dst, lane, y, x) = sum1 / sum2;
  }
}
void launch_bilateral(hipStream_t s, int B, ImgB src, ImgB dst, float sigma_floatmap, LaneMask m) {
  hipLaunchKernelGGL(k_bilateral, grid2d(src.cols, src.rows, B), dim3(TX, TY), 0, s, src, dst, sigma_floatmap, m);
}

}  // namespace rgbid
