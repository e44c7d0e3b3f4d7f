/*
 * rgbid_oracle.c -- per-kernel CPU restatement of the reference's dense RGB-iD front-end.
 * TEST INFRASTRUCTURE ONLY (see rgbid_oracle.h).  PARITY UNPINNED by reference tests (there
 * are none); each function cites the reference file:line it restates.
 *
 * fp32 everywhere the CUDA kernels are fp32; reductions are carried in double (the reference's
 * summation order depends on its launch geometry, so there is no canonical fp32 order).
 * Compile with -ffp-contract=off (see oracle/Makefile).
 */
#include "rgbid_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_NAN (nanf(""))

/* ---- numerics of the reference's CUDA build, as a sensitivity model ---------------------------------------------------------
 * The reference compiles src/cuda with  --ftz=true --prec-div=false --prec-sqrt=false  (CMakeLists.txt:105) on top of nvcc's default
 * --fmad=true, and calls __expf in its Gaussian taps (pyrdown.cu:116, filters.cu:124): its own results are NOT those of IEEE
 * arithmetic.  Built with -DORC_CUDA_NUMERICS (oracle/Makefile: librgbid_oracle_cudanum.so, also -ffp-contract=fast -mfma so that
 * a*b+c contracts into FMAs as nvcc does) every fp32 division, sqrt, rsqrt and exp of the device code goes through the functions
 * below: the correctly rounded result moved by a deterministic, input-keyed pseudo-random offset within the error bound the CUDA
 * programming guide states for the approximate instruction (div.full / rcp.approx: 2 ulp, sqrt.approx / rsqrt.approx: 2 ulp,
 * ex2.approx behind __expf: 2 + floor(|1.16 x|) ulp), subnormal inputs and results flushed to zero.  The GPU hardware this models
 * is absent, so the offsets are a MODEL of "any result inside the documented bound", not the bits an NVIDIA GPU would return.
 * tests/test_oracle_cuda_numerics.py runs both libraries over the same sequences and bounds how far poses and keyframe decisions
 * move: that is what "within tolerance of the reference's CUDA path" can mean in an image without CUDA.  The default build
 * (ORC_CUDA_NUMERICS undefined) expands the macros to the plain IEEE operators: its bits are unchanged. */
#ifdef ORC_CUDA_NUMERICS
#include <float.h>
static inline uint32_t cn_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float cn_from_bits(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
static inline float cn_ftz(float x) { return (fabsf(x) < FLT_MIN) ? copysignf(0.f, x) : x; }
static inline uint32_t cn_mix(uint32_t a, uint32_t b) {
  uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u + (a << 6) + (a >> 2));
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
/* r moved by k units in the last place (magnitude-wise), kept finite and normal */
static inline float cn_nudge(float r, int k) {
  if (!(fabsf(r) >= FLT_MIN) || isinf(r) || k == 0) return r;
  uint32_t u = cn_bits(r), sign = u & 0x80000000u, mag = u & 0x7fffffffu;
  int64_t m = (int64_t)mag + k;
  if (m < 0x00800000) m = 0x00800000;
  if (m > 0x7f7fffff) m = 0x7f7fffff;
  return cn_from_bits(sign | (uint32_t)m);
}
static inline int cn_off(uint32_t h, int bound) { return (int)(h % (uint32_t)(2 * bound + 1)) - bound; }
static inline float cn_div(float a, float b) {
  a = cn_ftz(a); b = cn_ftz(b);
  return cn_ftz(cn_nudge(a / b, cn_off(cn_mix(cn_bits(a), cn_bits(b)), 2)));
}
static inline float cn_sqrt(float x) { x = cn_ftz(x); return cn_ftz(cn_nudge(sqrtf(x), cn_off(cn_mix(cn_bits(x), 0x51u), 2))); }
static inline float cn_rsqrt(float x) { x = cn_ftz(x); return cn_ftz(cn_nudge((float)(1.0 / sqrt((double)x)), cn_off(cn_mix(cn_bits(x), 0x52u), 2))); }
static inline float cn_exp(float x) {
  x = cn_ftz(x);
  float t = x * 1.44269504088896341f;                       /* __expf(x) = ex2.approx(x * log2 e), the product rounded to fp32 */
  int bound = 2 + (int)floorf(fabsf(1.16f * x));            /* CUDA C Programming Guide, intrinsic __expf: max ulp error 2 + floor(abs(1.16 x)) */
  if (bound > 4096) bound = 4096;                            /* results that far out are flushed to 0 / saturate anyway */
  return cn_ftz(cn_nudge((float)exp2((double)t), cn_off(cn_mix(cn_bits(x), 0x53u), bound)));
}
#define FDIV(a, b) cn_div((a), (b))
#define FSQRT(x) cn_sqrt(x)
#define FRSQRT(x) cn_rsqrt(x)
#define FEXP(x) cn_exp(x)
int orc_cuda_numerics(void) { return 1; }
#else
#define FDIV(a, b) ((a) / (b))
#define FSQRT(x) sqrtf(x)
#define FRSQRT(x) (1.0f / sqrtf(x))
#define FEXP(x) expf(x)
int orc_cuda_numerics(void) { return 0; }
#endif

static int g_threads = 0;
int orc_num_threads(void) {
#ifdef _OPENMP
  return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_num_threads(int n) {
  g_threads = n;
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#endif
}

/* CUDA __float2int_rd / __float2int_rn: saturating, NaN -> 0 */
static inline int f2i_rd(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.f) return INT_MAX;
  if (x <= -2147483648.f) return INT_MIN;
  return (int)floorf(x);
}
static inline int f2i_rn(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.f) return INT_MAX;
  if (x <= -2147483648.f) return INT_MIN;
  return (int)rintf(x); /* round-half-even under the default rounding mode */
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
/* utils.hpp:153-157 normalized(): v * rsqrt(dot(v,v)) */
static inline float rsqrt_f(float x) { return FRSQRT(x); }
static inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

orc_intr orc_intr_level(orc_intr k, int level) {
  /* src/internal.h:128-132 */
  int div = 1 << level;
  orc_intr r = { k.fx / div, k.fy / div, k.cx / div, k.cy / div };
  return r;
}

/* ------------------------------------------------------------------ misc.cu */
void orc_depth2invdepth(const uint16_t* src, float* dst, int rows, int cols, float factor_depth) {
  /* misc.cu:105-124 */
#pragma omp parallel for
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      int value = src[(size_t)y * cols + x];
      float r = ORC_NAN;
      if (value > 0) r = FDIV(FDIV(1.f, factor_depth) * 1000.f, (float)imax(0, imin(value, 10000)));
      dst[(size_t)y * cols + x] = r;
    }
}

void orc_intensity(const uint8_t* rgb, float* dst, int rows, int cols) {
  /* misc.cu:128-147 */
#pragma omp parallel for
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const uint8_t* p = rgb + 3 * ((size_t)y * cols + x);
      float v = 0.2126f * (float)p[0] + 0.7152f * (float)p[1] + 0.0722f * (float)p[2];
      dst[(size_t)y * cols + x] = fmaxf(0.f, fminf(v, 255.f));
    }
}

void orc_decompose_rgb(const uint8_t* rgb, float* r, float* g, float* b, int rows, int cols) {
  /* misc.cu:151-172 */
  for (size_t i = 0; i < (size_t)rows * cols; ++i) {
    r[i] = (float)rgb[3 * i];
    g[i] = (float)rgb[3 * i + 1];
    b[i] = (float)rgb[3 * i + 2];
  }
}

void orc_gradient(const float* src, int rows, int cols, float* gx, float* gy) {
  /* misc.cu:176-220: 3x3 Sobel / 8, replicate border, dx outer / dy inner accumulation order */
#pragma omp parallel for
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float res_hor = 0, res_vert = 0;
      for (int dx = -1; dx < 2; dx++)
        for (int dy = -1; dy < 2; dy++) {
          int cx = imin(imax(0, x + dx), cols - 1);
          int cy = imin(imax(0, y + dy), rows - 1);
          int weight_hor = dx * (2 - dy * dy);
          int weight_vert = dy * (2 - dx * dx);
          float t = src[(size_t)cy * cols + cx];
          res_hor += t * weight_hor;
          res_vert += t * weight_vert;
        }
      gx[(size_t)y * cols + x] = res_hor / 8.f;
      gy[(size_t)y * cols + x] = res_vert / 8.f;
    }
}

void orc_init_weight(const float* src_depth, float* dst_weight, int rows, int cols) {
  /* misc.cu:272-287: weight := 1 everywhere (both branches assign 1) */
  (void)src_depth;
  for (size_t i = 0; i < (size_t)rows * cols; ++i) dst_weight[i] = 1.f;
}

/* ------------------------------------------------------------------ pyrdown.cu */
void orc_pyr_down(const float* src, int rows, int cols, float* dst) {
  /* pyrdown.cu:84-132, blur_radius = 2 (RADIUS_DEPTH == RADIUS_INT == 2), sigma = 1 */
  const int br = 2;
  int drows = rows / 2, dcols = cols / 2;
#pragma omp parallel for
  for (int y = 0; y < drows; ++y)
    for (int x = 0; x < dcols; ++x) {
      float res = ORC_NAN;
      int tx = imin(2 * x + br + 1, cols);
      int ty = imin(2 * y + br + 1, rows);
      float sum1 = 0.f, sum2 = 0.f;
      int count = 0;
      float sigma_space2_inv_half = 0.5f / (1.f * 1.f);
      for (int cy = imax(0, 2 * y - br); cy < ty; ++cy)
        for (int cx = imax(0, 2 * x - br); cx < tx; ++cx) {
          float val = src[(size_t)cy * cols + cx];
          if (!isnan(val)) {
            float space2 = (float)((2 * x - cx) * (2 * x - cx) + (2 * y - cy) * (2 * y - cy));
            float weight = FEXP(-(space2 * sigma_space2_inv_half));
            sum1 += val * weight;
            sum2 += weight;
            ++count;
          }
        }
      int d = 2 * br + 1;
      int area = d * d;
      if (count > (area / 2)) res = FDIV(sum1, sum2);
      dst[(size_t)y * dcols + x] = res;
    }
}

/* ------------------------------------------------------------------ filters.cu */
void orc_bilateral(const float* src, int rows, int cols, float sigma_floatmap, float* dst) {
  /* filters.cu:86-135.  The reference indexes with unsigned x,y so `max(y - RADIUS, 0)` wraps for
   * y < 2 and the window start becomes -2 (out-of-bounds read, undefined behaviour).  The oracle
   * implements the evidently intended clipped 5x5 window; see DESIGN.md. */
  const int R = 2;
  const float sigma_space = 5.f;
#pragma omp parallel for
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float value = src[(size_t)y * cols + x];
      if (isnan(value)) { dst[(size_t)y * cols + x] = ORC_NAN; continue; }
      int tx = imin(x + R + 1, cols);
      int ty = imin(y + R + 1, rows);
      float sum1 = 0, sum2 = 0;
      for (int cy = imax(y - R, 0); cy < ty; ++cy)
        for (int cx = imax(x - R, 0); cx < tx; ++cx) {
          float tmp = src[(size_t)cy * cols + cx];
          if (!isnan(tmp)) {
            float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
            float fn = FDIV(value - tmp, sigma_floatmap);
            /* `0.5 / (sigma_space*sigma_space)` and `0.5*fn*fn` are double expressions in the source */
            float s2ih = (float)(0.5 / (double)(sigma_space * sigma_space));
            double arg = (double)(s2ih * space2) + (0.5 * (double)fn) * (double)fn;
            float weight = FEXP((float)(-arg));
            sum1 += tmp * weight;
            sum2 += weight;
          }
        }
      dst[(size_t)y * cols + x] = FDIV(sum1, sum2);
    }
}

/* ------------------------------------------------------------------ warping_registration.cu */
/* registerPixel, warping_registration.cu:129-146; Mat33*float3 device.hpp:70-74; dot utils.hpp:105-109 */
static inline float register_pixel(float* xc, float* yc, int xd, int yd, float wd, const float R[9], const float t[3]) {
  float zd = FDIV(1.f, wd);
  float Xd[3] = { (float)xd * zd, (float)yd * zd, zd };
  float X0 = dot3(R + 0, Xd) + t[0];
  float X1 = dot3(R + 3, Xd) + t[1];
  float X2 = dot3(R + 6, Xd) + t[2];
  float wc = FDIV(1.f, X2);
  *xc = X0 * wc;
  *yc = X1 * wc;
  return wc;
}

static inline int in_bounds_rd(float xs, float ys, int cols, int rows) {
  /* :486-487 / :526-527 */
  return !(f2i_rd(xs) < 0 || f2i_rd(ys) < 0 || f2i_rd(xs) >= cols || f2i_rd(ys) >= rows);
}

void orc_warp_invdepth(const float* src, const float* grid, int rows, int cols,
                       const float R[9], const float t[3], float* dst) {
  /* trafo3DKernelInvDepthGridStride :505-546; texture = point filter, clamp, unnormalised (:994-998) */
#pragma omp parallel for
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float out = ORC_NAN;
      float w = grid[(size_t)y * cols + x];
      if (!isnan(w)) {
        float xs, ys;
        float w3 = register_pixel(&xs, &ys, x, y, w, R, t);
        xs += 0.5f; ys += 0.5f;
        if (in_bounds_rd(xs, ys, cols, rows)) {
          float w2 = src[(size_t)f2i_rd(ys) * cols + f2i_rd(xs)];
          float tz = t[2];
          float v1_z = (FDIV(1.f, w3) - tz) * w;
          float res = FDIV(v1_z, 1.f - w2 * tz) * w2;
          if (res > 0.f) out = res;
        }
      }
      dst[(size_t)y * cols + x] = out;
    }
}

/* CUDA linear filtering at unnormalised (xs,ys), clamp addressing (programming guide, texture fetching):
 * xB = xs-0.5, i=floor(xB), alpha=frac(xB) [TEX8: stored in 1.8 fixed point] */
static inline float tex2d_linear(const float* src, int rows, int cols, float xs, float ys, int mode) {
  float xB = xs - 0.5f, yB = ys - 0.5f;
  float fx0 = floorf(xB), fy0 = floorf(yB);
  float a = xB - fx0, b = yB - fy0;
  if (mode == ORC_INTERP_TEX8) {
    a = rintf(a * 256.f) * 0.00390625f;
    b = rintf(b * 256.f) * 0.00390625f;
  }
  /* clamp addressing: i0 -> clamp(i0, 0, n-1), i1 -> clamp(i0 + 1, 0, n-1); the conversions saturate (INT_MAX for huge
   * coordinates), so clamp to [-1, n-1] BEFORE the + 1 (no signed overflow); same values */
  int ic = imin(imax(f2i_rd(fx0), -1), cols - 1), jc = imin(imax(f2i_rd(fy0), -1), rows - 1);
  int i1 = imin(ic + 1, cols - 1), j1 = imin(jc + 1, rows - 1);
  int i0 = imax(ic, 0), j0 = imax(jc, 0);
  float T00 = src[(size_t)j0 * cols + i0], T10 = src[(size_t)j0 * cols + i1];
  float T01 = src[(size_t)j1 * cols + i0], T11 = src[(size_t)j1 * cols + i1];
  float oa = 1.f - a, ob = 1.f - b;
  return (oa * ob) * T00 + (a * ob) * T10 + (oa * b) * T01 + (a * b) * T11;
}

void orc_warp_intensity(const float* src, const float* grid, int rows, int cols,
                        const float R[9], const float t[3], int interp_mode, float* dst) {
  /* trafo3DKernelIntensityWithInvDepthGridStride :465-501; texture = linear filter (:942) */
#pragma omp parallel for
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float res = ORC_NAN;
      float w = grid[(size_t)y * cols + x];
      if (!isnan(w)) {
        float xs, ys;
        register_pixel(&xs, &ys, x, y, w, R, t);
        xs += 0.5f; ys += 0.5f;
        if (in_bounds_rd(xs, ys, cols, rows)) {
          res = tex2d_linear(src, rows, cols, xs, ys, interp_mode);
          res = fmaxf(0.f, fminf(res, 255.f)); /* CUDA min/max drop NaN: NaN -> 255 */
        }
      }
      dst[(size_t)y * cols + x] = res;
    }
}

void orc_warp_invdepth_weighted(const float* src, const float* grid, int rows, int cols,
                                const float R[9], const float t[3], float* dst, float* weight) {
  /* trafo3DKernelInvDepthWeightedGridStride :549-594; weight is written only where weight_res > 0 */
#pragma omp parallel for
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      size_t idx = (size_t)y * cols + x;
      dst[idx] = ORC_NAN;
      float w = grid[idx];
      if (!isnan(w)) {
        float xs, ys;
        float w3 = register_pixel(&xs, &ys, x, y, w, R, t);
        xs += 0.5f; ys += 0.5f;
        if (in_bounds_rd(xs, ys, cols, rows)) {
          float w2 = src[(size_t)f2i_rd(ys) * cols + f2i_rd(xs)];
          float tz = t[2];
          float v1_z = (FDIV(1.f, w3) - tz) * w;
          float w_factor = 1.f - w2 * tz;
          float w_factor2 = w_factor * w_factor;
          float weight_res = FDIV(w_factor2 * w_factor2, v1_z * v1_z);
          float res = FDIV(v1_z, w_factor) * w2;
          if (res > 0.f) dst[idx] = res;
          if (weight_res > 0.f) weight[idx] = weight_res;
        }
      }
    }
}

void orc_integrate_warped(const float* warped, const float* warped_weight,
                          float* kf, float* kf_weight, int rows, int cols) {
  /* integrateWarpedFrameKernel :637-669; DEPTHINV_INTEGR_TH = 0.0075f (:80) */
  const float TH = 0.0075f;
  for (size_t i = 0; i < (size_t)rows * cols; ++i) {
    if (!isnan(warped[i])) {
      float w_sum = warped[i];
      float w_KF = kf[i];
      float dw = fabsf(w_sum - w_KF);
      if (isnan(w_KF)) {
        kf[i] = w_sum;
        kf_weight[i] = warped_weight[i];
      } else if (dw < 3 * TH) {
        float new_weight = kf_weight[i] + warped_weight[i];
        kf[i] = FDIV(w_KF * kf_weight[i] + w_sum * warped_weight[i], new_weight);
        kf_weight[i] = new_weight;
      }
    }
  }
}

float orc_visibility_ratio(const float* src, const float* dst, int rows, int cols,
                           const float R[9], const float t[3], uint8_t* mask,
                           float* n_visible, float* n_valid) {
  /* partialVisibility(WithOverlapMask)Kernel :297-437 + host :825-913 */
  double vis = 0, val = 0;
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float w = src[(size_t)y * cols + x];
      if (!isnan(w)) {
        float xd, yd;
        float w_dst = register_pixel(&xd, &yd, x, y, w, R, t);
        uint8_t flag = 0;
        val += 1.0;
        if ((xd > 0) && (xd < (cols - 1)) && (yd > 0) && (yd < (rows - 1))) {
          int xi = f2i_rn(xd), yi = f2i_rn(yd);
          if (fabsf(w_dst - dst[(size_t)yi * cols + xi]) < 0.020f) { vis += 1.0; flag = 1; }
        }
        if (mask) mask[(size_t)y * cols + x] = flag;
      }
    }
  if (n_visible) *n_visible = (float)vis;
  if (n_valid) *n_valid = (float)val;
  float fvis = (float)vis, fval = (float)val;
  return (fval < 1.f) ? 0.f : fvis / fval;
}

/* ------------------------------------------------------------------ maps.cu */
void orc_vmap(const float* depthinv, int rows, int cols, orc_intr k, float* vmap) {
  /* computeVmapKernel maps.cu:63-90 (host passes 1.f/fx, 1.f/fy :340); only plane 0 is NaN-marked */
  float fx_inv = 1.f / k.fx, fy_inv = 1.f / k.fy;
  for (int v = 0; v < rows; ++v)
    for (int u = 0; u < cols; ++u) {
      float z = FDIV(1.f, depthinv[(size_t)v * cols + u]);
      if (!isnan(z)) {
        vmap[(size_t)v * cols + u] = z * (u - k.cx) * fx_inv;
        vmap[(size_t)(v + rows) * cols + u] = z * (v - k.cy) * fy_inv;
        vmap[(size_t)(v + 2 * rows) * cols + u] = z;
      } else {
        vmap[(size_t)v * cols + u] = ORC_NAN;
      }
    }
}

void orc_nmap_gradients(const float* depthinv, const float* gx_, const float* gy_,
                        int rows, int cols, orc_intr k, float* nmap) {
  /* computeNmapGradientsKernel maps.cu:134-179 */
  for (int v = 0; v < rows; ++v)
    for (int u = 0; u < cols; ++u) {
      size_t i = (size_t)v * cols + u;
      nmap[i] = ORC_NAN;
      float w = depthinv[i], gx = gx_[i], gy = gy_[i];
      if (!(isnan(w) || isnan(gx) || isnan(gy))) {
        float n[3] = { gx * k.fx, gy * k.fy, gx * (k.cx - u) + gy * (k.cy - v) + w };
        float rn = rsqrt_f(dot3(n, n));
        n[0] *= rn; n[1] *= rn; n[2] *= rn;
        float z = FDIV(1.f, w);
        float vt[3] = { z * (u - k.cx) * (1.f / k.fx), z * (v - k.cy) * (1.f / k.fy), z };
        float rv = rsqrt_f(dot3(vt, vt));
        vt[0] *= rv; vt[1] *= rv; vt[2] *= rv;
        float acos_vn = dot3(vt, n);
        if ((double)acos_vn > 0.1) {
          nmap[i] = n[0];
          nmap[(size_t)(v + rows) * cols + u] = n[1];
          nmap[(size_t)(v + 2 * rows) * cols + u] = n[2];
        }
      }
    }
}

void orc_nmap_cross(const float* vmap, int rows, int cols, float* nmap) {
  /* computeNmapKernel maps.cu:92-133 (normalized = v * rsqrt(dot), cross utils.hpp:153-162) */
  for (int v = 0; v < rows; ++v)
    for (int u = 0; u < cols; ++u) {
      nmap[(size_t)v * cols + u] = ORC_NAN;
      if (u == cols - 1 || v == rows - 1) continue;
      float ax = vmap[(size_t)v * cols + u], bx = vmap[(size_t)v * cols + u + 1], cx = vmap[(size_t)(v + 1) * cols + u];
      if (!isnan(ax) && !isnan(bx) && !isnan(cx)) {
        float ay = vmap[(size_t)(v + rows) * cols + u], by = vmap[(size_t)(v + rows) * cols + u + 1], cy = vmap[(size_t)(v + 1 + rows) * cols + u];
        float az = vmap[(size_t)(v + 2 * rows) * cols + u], bz = vmap[(size_t)(v + 2 * rows) * cols + u + 1], cz = vmap[(size_t)(v + 1 + 2 * rows) * cols + u];
        float d1x = bx - ax, d1y = by - ay, d1z = bz - az, d2x = cx - ax, d2y = cy - ay, d2z = cz - az;
        float rx = d1y * d2z - d1z * d2y, ry = d1z * d2x - d1x * d2z, rz = d1x * d2y - d1y * d2x;
        float inv = rsqrt_f(rx * rx + ry * ry + rz * rz);
        nmap[(size_t)v * cols + u] = rx * inv;
        nmap[(size_t)(v + rows) * cols + u] = ry * inv;
        nmap[(size_t)(v + 2 * rows) * cols + u] = rz * inv;
      }
    }
}

void orc_integrate_warped_rgb(const float* warped, const float* r, const float* g, const float* b, const float* wweight,
                              float* kf, uint8_t* colors, float* kfw, int rows, int cols) {
  /* integrateWarpedRGBKernel warping_registration.cu:673-708 */
  const float TH = 0.0075f;
  for (size_t i = 0; i < (size_t)rows * cols; ++i) {
    if (isnan(warped[i]) || isnan(r[i]) || isnan(g[i]) || isnan(b[i])) continue;
    uint8_t* c = colors + 3 * i;
    if (isnan(kf[i])) {
      kf[i] = warped[i];
      c[0] = (uint8_t)f2i_rn(r[i]); c[1] = (uint8_t)f2i_rn(g[i]); c[2] = (uint8_t)f2i_rn(b[i]);
      kfw[i] = wweight[i];
    } else if (((kf[i] - warped[i]) < TH) && ((warped[i] - kf[i]) < TH)) {
      float new_weight = kfw[i] + wweight[i];
      float q = kfw[i];
      kf[i] = FDIV(kf[i] * q + warped[i] * wweight[i], new_weight);
      c[0] = (uint8_t)f2i_rn(((float)c[0] * q + r[i] * wweight[i]) / new_weight);
      c[1] = (uint8_t)f2i_rn(((float)c[1] * q + g[i] * wweight[i]) / new_weight);
      c[2] = (uint8_t)f2i_rn(((float)c[2] * q + b[i] * wweight[i]) / new_weight);
      kfw[i] = new_weight;
    }
  }
}

void orc_depth2float(const uint16_t* src, float* dst, int rows, int cols) {
  /* depth2floatKernel misc.cu:86-102 */
  for (size_t i = 0; i < (size_t)rows * cols; ++i) {
    int value = src[i];
    dst[i] = value > 0 ? (float)imax(0, imin(value, 10000)) / 1000.f : ORC_NAN;
  }
}

void orc_float2rgb(const float* src, uint8_t* dst, int rows, int cols) {
  /* float2ucharKernel misc.cu:289-324 */
  const float min_val = 0.f, max_val = 255.f;
  for (size_t i = 0; i < (size_t)rows * cols; ++i) {
    float v = src[i];
    uint8_t* c = dst + 3 * i;
    if (isnan(v)) { c[0] = 200; c[1] = 150; c[2] = 150; }
    else if (isinf(v)) { c[0] = 150; c[1] = 150; c[2] = 250; }
    else {
      uint8_t grey = (uint8_t)imax(0, imin(f2i_rn(255 * (v - min_val) / (max_val - min_val)), 255));
      c[0] = c[1] = c[2] = grey;
    }
  }
}

void orc_generate_image_rgb(const float* vmap, const float* nmap, const uint8_t* rgb,
                            const float light[3], int rows, int cols, uint8_t* dst) {
  /* ImageGeneratorRGB image_generator.cu:122-180, light.number == 1 (visodo.cpp:563-565) */
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      size_t i = (size_t)y * cols + x;
      uint8_t c[3] = { 0, 0, 0 };
      float vx = vmap[i], nx = nmap[i];
      if (!isnan(vx) && !isnan(nx)) {
        float v[3] = { vx, vmap[(size_t)(y + rows) * cols + x], vmap[(size_t)(y + 2 * rows) * cols + x] };
        float n[3] = { nx, nmap[(size_t)(y + rows) * cols + x], nmap[(size_t)(y + 2 * rows) * cols + x] };
        float weight = 1.f;
        float d[3] = { light[0] - v[0], light[1] - v[1], light[2] - v[2] };
        float rd = rsqrt_f(dot3(d, d));
        d[0] *= rd; d[1] *= rd; d[2] *= rd;
        weight *= fabsf(dot3(d, n));
        int br = (int)(205 * weight) + 50;
        br = imax(0, imin(255, br));
        float br_f = (float)br / 255.f;
        c[0] = (uint8_t)f2i_rn((float)rgb[3 * i] * br_f);
        c[1] = (uint8_t)f2i_rn((float)rgb[3 * i + 1] * br_f);
        c[2] = (uint8_t)f2i_rn((float)rgb[3 * i + 2] * br_f);
      }
      dst[3 * i] = c[0]; dst[3 * i + 1] = c[1]; dst[3 * i + 2] = c[2];
    }
}

/* ------------------------------------------------------------------ sigmaFuncs.cu */
/* ================================================================== custom-calibration front-end (f-5) */
static inline void distort_pixel(float uu, float vu, float* ud, float* vd, orc_intr_k k) {
  /* distortPixel undistortion.cu:96-112 */
  float r2 = uu * uu + vu * vu;
  float r4 = r2 * r2;
  float r6 = r2 * r4;
  float factor_r = 1.f + k.k1 * r2 + k.k2 * r4 + k.k5 * r6;
  float a = factor_r * uu;
  a += 2.f * k.k3 * uu * vu + k.k4 * (r2 + 2.f * uu * uu);
  float b = factor_r * vu;
  b += 2.f * k.k4 * uu * vu + k.k3 * (r2 + 2.f * vu * vu);
  *ud = a; *vd = b;
}

static void undistort_image(const float* src, int rows, int cols, orc_intr_k k, int linear, int interp_mode, float* dst) {
  /* undistortKernel undistortion.cu:145-176 */
#pragma omp parallel for
  for (int yu = 0; yu < rows; ++yu)
    for (int xu = 0; xu < cols; ++xu) {
      float res = ORC_NAN;
      float uu = ((float)xu - k.cx) * (1.f / k.fx);
      float vu = ((float)yu - k.cy) * (1.f / k.fy);
      float ud, vd;
      distort_pixel(uu, vu, &ud, &vd, k);
      float xd = k.fx * ud + k.cx + 0.5f;
      float yd = k.fy * vd + k.cy + 0.5f;
      if (!((xd <= 0) || (yd <= 0) || (xd >= (float)cols) || (yd >= (float)rows))) {
        if (linear) res = tex2d_linear(src, rows, cols, xd, yd, interp_mode);
        else res = src[(size_t)imin(imax(f2i_rd(yd), 0), rows - 1) * cols + imin(imax(f2i_rd(xd), 0), cols - 1)];
      }
      dst[(size_t)yu * cols + xu] = res;
    }
}

void orc_undistort_intensity(const float* src, int rows, int cols, orc_intr_k k, int interp_mode, float* dst) {
  undistort_image(src, rows, cols, k, 1, interp_mode, dst);
}

void orc_undistort_depthinv(const float* src, int rows, int cols, orc_intr_k k, orc_depth_dist dp, float* src_corr, float* dst) {
  float* corr = src_corr ? src_corr : (float*)malloc((size_t)rows * cols * sizeof(float));
  /* depthinvCorrectionKernel undistortion.cu:179-211, correctDepthinv :131-142, undistortDepthinv :114-129 */
#pragma omp parallel for
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float res = ORC_NAN;
      int xs = x - dp.xshift, ys = y - dp.yshift;
      if ((xs > 0) && (ys > 0)) {
        float u = ((float)x - k.cx) * (1.f / k.fx);
        float v = ((float)y - k.cy) * (1.f / k.fy);
        float val = src[(size_t)ys * cols + xs];
        float wd = dp.c1 * val + dp.c0;
        float r2 = u * u + v * v;
        float r4 = r2 * r2;
        float r6 = r2 * r4;
        float uv = u * v;
        float u2v = u * u * v;
        float uv2 = u * v * v;
        float D0 = dp.q0[0] + dp.q0[1] * r2 + dp.q0[2] * r4 + dp.q0[3] * r6 + dp.q0[4] * u + dp.q0[5] * v + dp.q0[6] * uv + dp.q0[7] * u2v + dp.q0[8] * uv2;
        float D1 = dp.q1[0] + dp.q1[1] * r2 + dp.q1[2] * r4 + dp.q1[3] * r6 + dp.q1[4] * u + dp.q1[5] * v + dp.q1[6] * uv + dp.q1[7] * u2v + dp.q1[8] * uv2;
        res = (1.f + D1) * wd + D0;
      }
      corr[(size_t)y * cols + x] = res;
    }
  undistort_image(corr, rows, cols, k, 0, 0, dst);
  if (!src_corr) free(corr);
}

void orc_register_depthinv(const float* src, int rows, int cols, int irows, int icols, const float dRc_proj[9], const float t[3],
                           const float cRd_proj[9], float* intermediate, float* dst) {
  int offset_x = (icols - cols) / 2, offset_y = (irows - rows) / 2;
  size_t ni = (size_t)irows * icols;
  int32_t* zbuf = (int32_t*)calloc(ni, sizeof(int32_t));        /* initialiseRegistrationKernel :168-183 */
  for (size_t i = 0; i < ni; ++i) intermediate[i] = ORC_NAN;
  /* depthinvRegistrationTranslationWithDilationKernel :241-288 (serial: the max is order independent) */
  for (int yd = 0; yd < rows; ++yd)
    for (int xd = 0; xd < cols; ++xd) {
      float wd = src[(size_t)yd * cols + xd];
      if (wd != wd) continue;
      float zd = FDIV(1.f, wd);                                        /* registerPixelTranslationOnly :148-165 */
      float X0 = (float)xd * zd - t[0], X1 = (float)yd * zd - t[1], X2 = zd - t[2];
      float wc = FDIV(1.f, X2);
      float xc = X0 * wc, yc = X1 * wc;
      if (wc > 0.01f) {
        float dilation = FDIV(wc, wd);
        int32_t bits; memcpy(&bits, &wc, 4);
        /* the conversions saturate (CUDA semantics); adding the offset / the +1 of the loop bound is done on indices first clamped
         * to [-1, size] so nothing overflows: the clipped loop ranges are the same */
        int xmin = imin(imax(f2i_rn(xc - 0.5f * dilation), -icols), icols) + offset_x, xmax = imin(imax(f2i_rn(xc + 0.5f * dilation), -icols), icols) + offset_x;
        int ymin = imin(imax(f2i_rn(yc - 0.5f * dilation), -irows), irows) + offset_y, ymax = imin(imax(f2i_rn(yc + 0.5f * dilation), -irows), irows) + offset_y;
        for (int x = imax(0, xmin); x < imin(xmax + 1, icols); x++)
          for (int y = imax(0, ymin); y < imin(ymax + 1, irows); y++)
            if (zbuf[(size_t)y * icols + x] < bits) zbuf[(size_t)y * icols + x] = bits;   /* dst is all-NaN during this kernel */
      }
    }
  for (size_t i = 0; i < ni; ++i)                                  /* conversionRegistrationKernel :186-204 */
    if (zbuf[i] != 0) memcpy(&intermediate[i], &zbuf[i], 4);
  free(zbuf);
  /* homographyKernelInvDepthGridStride :597-635 with srcHdst = dRc_proj, dstHsrc = cRd_proj */
#pragma omp parallel for
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float out = ORC_NAN;
      float pd[3] = { (float)x, (float)y, 1.f }, ps[3];
      for (int r = 0; r < 3; ++r) ps[r] = dot3(dRc_proj + 3 * r, pd);
      float iz = FDIV(1.f, ps[2]);
      ps[0] *= iz; ps[1] *= iz; ps[2] *= iz;
      float x_src = ps[0] + 0.5f + (float)offset_x;
      float y_src = ps[1] + 0.5f + (float)offset_y;
      if (!(f2i_rd(x_src) < 0 || f2i_rd(y_src) < 0 || f2i_rd(x_src) >= icols || f2i_rd(y_src) >= irows)) {
        float w_src = intermediate[(size_t)imin(imax(f2i_rd(y_src), 0), irows - 1) * icols + imin(imax(f2i_rd(x_src), 0), icols - 1)];
        float pz = dot3(cRd_proj + 6, ps);
        float res = FDIV(w_src, pz);
        if (res > 0.f) out = res;
      }
      dst[(size_t)y * cols + x] = out;
    }
}

int orc_error_lattice(const float* im1, const float* im0, int rows, int cols, int min_nsamples,
                      float* err, int* out_rows, int* out_cols, int* out_stride) {
  /* computeErrorGridStride sigmaFuncs.cu:701-765 + errorHandler :90-135 */
  int error_size = cols * rows;
  int cols_prev = cols, rows_prev = rows;
  if (min_nsamples < error_size) {
    for (;;) {
      int cols_curr = cols_prev / 2, rows_curr = rows_prev / 2;
      if (((2 * cols_curr - cols_prev) != 0) || ((2 * rows_curr - rows_prev) != 0) ||
          (min_nsamples > cols_curr * rows_curr)) {
        error_size = cols_prev * rows_prev;
        break;
      }
      cols_prev = cols_curr;
      rows_prev = rows_curr;
    }
  }
  int stride = (int)sqrt((double)((rows * cols) / error_size));
  for (int y = 0; y < rows_prev; ++y)
    for (int x = 0; x < cols_prev; ++x)
      err[(size_t)y * cols_prev + x] = im1[(size_t)(stride * y) * cols + stride * x] - im0[(size_t)(stride * y) * cols + stride * x];
  if (out_rows) *out_rows = rows_prev;
  if (out_cols) *out_cols = cols_prev;
  if (out_stride) *out_stride = stride;
  return error_size;
}

float orc_digamma(float x) {
  /* device.hpp:76-80 -> boost::math::digamma<float> (Boost un-vendored, unpinned).  Restated with the
   * published recurrence psi(x) = psi(x+1) - 1/x and the asymptotic series, in double, rounded to float. */
  double xd = x, r = 0.0;
  while (xd < 10.0) { r -= 1.0 / xd; xd += 1.0; }
  double f = 1.0 / (xd * xd);
  double s = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return (float)(r + log(xd) - 0.5 / xd + s);
}

static const float TH_HUBER = 1.345f, TH_TUKEY = 4.685f, STUDENT_DOF = 5.f;

/* sums of one partial+final pass; the four accumulators of sigmaHandler (:179-241 / :258-332) */
typedef struct { float swsr, swr, sw, nel; } moments4;

static moments4 pass_bias_sigma(const float* err, int n, float bias, float sigma, float nu, int mest, int student_variant) {
  double swsr = 0, swr = 0, sw = 0, nel = 0;
  for (int i = 0; i < n; ++i) {
    float is_valid = 0.f, weighted_sq_res = 0.f, weighted_res = 0.f, weight = 0.f;
    float e = err[i];
    if (!isinf(e) && !isnan(e)) {
      if (student_variant) { /* partialBiasAndSigmaStudent :258-332 */
        is_valid = 1.f;
        if (mest == ORC_LSQ) weight = 1.f;
        else {
          float en = FDIV(e - bias, sigma);
          weight = FDIV(nu + 1.f, nu + en * en);
        }
      } else { /* partialBiasAndSigma :179-255 */
        weight = 1.f; is_valid = 1.f;
        float en = FDIV(e - bias, sigma);
        if ((mest == ORC_HUBER) && (fabsf(en) > TH_HUBER)) weight = FDIV(TH_HUBER, fabsf(en));
        else if (mest == ORC_TUKEY) {
          if (fabsf(en) < TH_TUKEY) {
            float aux1 = FDIV(en, TH_TUKEY) * FDIV(en, TH_TUKEY);
            weight = (1.f - aux1) * (1.f - aux1);
          } else { weight = 0.f; is_valid = 0.f; }
        } else if (mest == ORC_STUDENT) weight = FDIV(STUDENT_DOF + 1.f, STUDENT_DOF + en * en);
      }
      weighted_res = e * weight;
      weighted_sq_res = weighted_res * e;
    }
    swsr += weighted_sq_res; swr += weighted_res; sw += weight; nel += is_valid;
  }
  moments4 m = { (float)swsr, (float)swr, (float)sw, (float)nel };
  return m;
}

static void final_bias_sigma(moments4 m, float* bias, float* sigma) {
  /* finalReductionBiasAndSigma :361-407 (fp32) */
  float b = FDIV(m.swr, m.sw);
  *bias = b;
  *sigma = FSQRT(FDIV(m.swsr - 2.f * b * m.swr + b * b * m.sw, m.nel));
}

static float func_weights_nu(const float* err, int n, float bias, float sigma, float nu) {
  /* partialFuncWeightsNu :410-468 + finalReductionFuncWeightsNu :471-512 */
  double sln = 0, sw = 0, nel = 0;
  for (int i = 0; i < n; ++i) {
    float e = err[i];
    if (!isinf(e) && !isnan(e)) {
      float en = FDIV(e - bias, sigma);
      float weight = FDIV(nu + 1.f, nu + en * en);
      sln += logf(weight); sw += weight; nel += 1.0;
    }
  }
  return ((float)sln - (float)sw) / (float)nel;
}

static float C_nu(float nu, float fw) {
  /* sigmaFuncs.cu:951 etc.: float expression, left to right */
  return -orc_digamma(nu / 2.f) + logf(nu / 2.f) + fw + 1.f + orc_digamma((nu + 1.f) / 2.f) - logf((nu + 1.f) / 2.f);
}

static float estimate_nu(const float* err, int n, float bias, float sigma) {
  /* sigmaFuncs.cu:934-1039 == :1100-1205 */
  float nu_up = 10.f, nu_down = 2.f, nu_new = 0.f, nu;
  float C_nu_down = C_nu(nu_down, func_weights_nu(err, n, bias, sigma, nu_down));
  float C_nu_up = C_nu(nu_up, func_weights_nu(err, n, bias, sigma, nu_up));
  if (C_nu_up * C_nu_down > 0) {
    nu = (C_nu_down <= 0.f) ? nu_down : nu_up;
  } else {
    for (int j = 0; j < 5; j++) {
      nu_new = (nu_up + nu_down) / 2;
      if ((nu_up - nu_down) < 1.f) break;
      float C_nu_new = C_nu(nu_new, func_weights_nu(err, n, bias, sigma, nu_new));
      if (C_nu_new * C_nu_up > 0) { C_nu_up = C_nu_new; nu_up = nu_new; }
      else { C_nu_down = C_nu_new; nu_down = nu_new; }
    }
    nu = nu_new;
  }
  return nu;
}

void orc_sigma_nu_student_margin(const float* err, int n, float* bias, float* sigma, float* nu, int mestimator, float* stop_margin) {
  /* computeSigmaAndNuStudent sigmaFuncs.cu:858-1066.  stop_margin (nullable; TEST DIAGNOSTIC, no counterpart in the reference): the smallest
   * distance of the stopping test's ratio |sigma - sigma_prev| / sigma_prev from its threshold 0.1 over the iterations that evaluated it.  The
   * iteration count -- and with it sigma, to a few 1e-3 -- is a discontinuous function of the residuals wherever that ratio sits on the threshold;
   * two implementations whose residuals differ by 1e-6 may then stop one iteration apart. */
  float sh_sigma = *sigma, sh_bias = *bias, sh_nu = 5.f;
  int sh_mest = ORC_LSQ;
  float sigma_prev;
  const int max_iters = 10;
  const float rel_tol = 0.1f;
  float margin = 1e30f;
  for (int i = 0; i < max_iters; i++) {
    moments4 m = pass_bias_sigma(err, n, sh_bias, sh_sigma, sh_nu, sh_mest, 1);
    final_bias_sigma(m, bias, sigma);
    sigma_prev = sh_sigma;
    sh_bias = *bias; sh_sigma = *sigma; sh_mest = mestimator;
    if (i > 0) {
      const float ratio = fabsf(*sigma - sigma_prev) / sigma_prev;
      if (fabsf(ratio - rel_tol) < margin) margin = fabsf(ratio - rel_tol);
      if (ratio < rel_tol) break;
    }
  }
  if (stop_margin) *stop_margin = margin;
  *nu = estimate_nu(err, n, sh_bias, sh_sigma);
}
void orc_sigma_nu_student(const float* err, int n, float* bias, float* sigma, float* nu, int mestimator) {
  orc_sigma_nu_student_margin(err, n, bias, sigma, nu, mestimator, NULL);
}

void orc_nu_student(const float* err, int n, float bias, float sigma, float* nu) {
  /* computeNuStudent sigmaFuncs.cu:1068-1222 */
  *nu = estimate_nu(err, n, bias, sigma);
}

void orc_sigma_pdf(const float* err, int n, float* bias, float* sigma, int mestimator) {
  /* computeSigmaPdf sigmaFuncs.cu:773-854 */
  float sh_sigma = *sigma, sh_bias = *bias;
  int sh_mest = ORC_LSQ;
  for (int i = 0; i < 10; i++) {
    moments4 m = pass_bias_sigma(err, n, sh_bias, sh_sigma, 5.f, sh_mest, 0);
    final_bias_sigma(m, bias, sigma);
    if ((i > 0) && ((fabsf(*sigma - sh_sigma) / sh_sigma) < 0.1f)) break;
    sh_bias = *bias; sh_sigma = *sigma; sh_mest = mestimator;
  }
}

void orc_chi_square(const float* err_int, const float* err_depth, int n, float sigma_int,
                    float sigma_depth, int mest, float* chi_square, float* chi_test, float* ndof) {
  /* computeChiSquare sigmaFuncs.cu:1225-1297 + normalizeAndAppendErrorsKernel :137-150 + chiSquaredHandler :541-646 */
  double sN = 0, srho = 0;
  for (int half = 0; half < 2; ++half)
    for (int i = 0; i < n; ++i) {
      float en = half == 0 ? FDIV(err_int[i], sigma_int) : FDIV(err_depth[i], sigma_depth);
      float rho = 0.f;
      if (!isinf(en) && !isnan(en)) {
        sN += 1.0;
        rho = (en * en) / 2.f;
        if ((mest == ORC_HUBER) && (fabsf(en) > TH_HUBER)) rho = TH_HUBER * (fabsf(en) - TH_HUBER / 2.f);
        else if (mest == ORC_TUKEY) {
          if (fabsf(en) < TH_TUKEY) {
            float aux1 = FDIV(en, TH_TUKEY) * FDIV(en, TH_TUKEY);
            float aux2 = (1.f - aux1) * (1.f - aux1) * (1.f - aux1);
            rho = ((TH_TUKEY * TH_TUKEY) / 6.f) * (1.f - aux2);
          } else rho = ((TH_TUKEY * TH_TUKEY) / 6.f);
        } else if (mest == ORC_STUDENT) rho = ((STUDENT_DOF + 1.f) / 2.f) * logf(1.f + (en * en) / STUDENT_DOF);
      }
      srho += rho;
    }
  float fN = (float)sN, frho = (float)srho;
  *chi_square = frho / fN;
  *ndof = fN;
  float z_gauss = (*chi_square - *ndof) / (sqrtf(2.f * (*ndof)));
  *chi_test = 0.5f * (1.f + erff(z_gauss / sqrtf(2.f)));
}

/* ------------------------------------------------------------------ estimate_VO.cu */
static inline float compute_weight(float error, int mest) {
  /* computeWeight estimate_VO.cu:141-167 */
  float weight = 1.f;
  if (mest == ORC_HUBER) { if (fabsf(error) > TH_HUBER) weight = FDIV(TH_HUBER, fabsf(error)); }
  else if (mest == ORC_TUKEY) {
    if (fabsf(error) < TH_TUKEY) { float aux1 = FDIV(error, TH_TUKEY) * FDIV(error, TH_TUKEY); weight = (1.f - aux1) * (1.f - aux1); }
    else weight = 0.f;
  } else if (mest == ORC_STUDENT) weight = FDIV(STUDENT_DOF + 1.f, STUDENT_DOF + error * error);
  return weight;
}

void orc_build_system(const float* W0, const float* I0, const float* gW0x, const float* gW0y,
                      const float* gI0x, const float* gI0y, const float* W1, const float* I1,
                      int rows, int cols, int student_nu, int mestimator, int weighting,
                      float sigma_depthinv, float sigma_int, float bias_depthinv, float bias_int,
                      float nu_depthinv, float nu_int, orc_intr k, double A[36], double b[6]) {
  /* constraintsHandler estimate_VO.cu:176-262 (rows), :265-350 / :354-439 (accumulate), :627-642 (unpack) */
  double* rowsum = (double*)calloc((size_t)rows * 27, sizeof(double));
#pragma omp parallel for
  for (int y = 0; y < rows; ++y) {
    double* acc = rowsum + (size_t)y * 27;
    for (int x = 0; x < cols; ++x) {
      size_t i = (size_t)y * cols + x;
      float row_int[6] = { 0 }, row_d[6] = { 0 };
      float error_int = 0.f, error_d = 0.f, weight_int = 0.f, weight_d = 0.f, n_factor = 1.f;
      float w0 = W0[i], w1 = W1[i];
      {
        /* invDepthConstraint :214-262 */
        float gradx = gW0x[i], grady = gW0y[i];
        if (!(isnan(w0) || isnan(w1) || isnan(gradx) || isnan(grady))) {
          float p[3] = { FDIV((float)x - k.cx, k.fx), FDIV((float)y - k.cy, k.fy), 1.f };
          float g[3];
          g[0] = gradx * k.fx; g[1] = grady * k.fy; g[2] = -(g[0] * p[0] + g[1] * p[1]);
          float inv_w0 = FDIV(1.f, w0);
          float n[3] = { g[0] * inv_w0, g[1] * inv_w0, g[2] * inv_w0 };
          n[2] += 1.f;
          float rn = rsqrt_f(dot3(n, n));
          n[0] *= rn; n[1] *= rn; n[2] *= rn;
          float rp = rsqrt_f(dot3(p, p));
          float pu[3] = { p[0] * rp, p[1] * rp, p[2] * rp };
          n_factor = fabsf(dot3(n, pu));
          float weight = FDIV(1.f, sigma_depthinv);
          float rt[3] = { g[0] * w0, g[1] * w0, g[2] * w0 };
          rt[2] = rt[2] + w0 * w1;
          g[2] = g[2] + w1;
          /* row_rot = -cross(g, p), utils.hpp:159-163 */
          float rr[3] = { -(g[1] * p[2] - g[2] * p[1]), -(g[2] * p[0] - g[0] * p[2]), -(g[0] * p[1] - g[1] * p[0]) };
          float bb = (w1 - w0);
          for (int c = 0; c < 3; ++c) { row_d[c] = rt[c] * weight; row_d[3 + c] = rr[c] * weight; }
          error_d = -bb * weight;
          float e_unb = error_d - FDIV(bias_depthinv, sigma_depthinv);
          float wgt = student_nu ? FDIV(nu_depthinv + 1.f, nu_depthinv + e_unb * e_unb) : compute_weight(e_unb, mestimator);
          weight_d = wgt * (float)(1 - (weighting == ORC_PHOT_ONLY));
        }
      }
      {
        /* intensityConstraint :176-212 */
        float i0 = I0[i], i1 = I1[i], gradx = gI0x[i], grady = gI0y[i];
        if (!(isnan(w0) || isnan(i0) || isnan(i1) || isnan(gradx) || isnan(grady))) {
          float p[3] = { FDIV((float)x - k.cx, k.fx), FDIV((float)y - k.cy, k.fy), 1.f };
          float g[3];
          g[0] = gradx * k.fx; g[1] = grady * k.fy; g[2] = -(g[0] * p[0] + g[1] * p[1]);
          float weight = FDIV(1.f, sigma_int);
          float rr[3] = { -(g[1] * p[2] - g[2] * p[1]), -(g[2] * p[0] - g[0] * p[2]), -(g[0] * p[1] - g[1] * p[0]) };
          float rt[3] = { g[0] * w0, g[1] * w0, g[2] * w0 };
          float bb = (i1 - i0);
          for (int c = 0; c < 3; ++c) { row_int[c] = rt[c] * weight; row_int[3 + c] = rr[c] * weight; }
          error_int = -bb * weight;
          float e_unb = error_int - FDIV(bias_int, sigma_int);
          float wgt = student_nu ? FDIV(nu_int + 1.f, nu_int + e_unb * e_unb) : compute_weight(e_unb, mestimator);
          weight_int = wgt * (float)(1 - (weighting == ORC_GEOM_ONLY));
        }
      }
      if (weighting == ORC_MIN_WEIGHT) weight_int = fminf(weight_d, weight_int);
      int shift = 0;
      for (int r = 0; r < 6; ++r) {
        for (int c = r; c < 6; ++c)
          acc[shift++] += (double)(weight_int * (row_int[r] * row_int[c]) + n_factor * weight_d * (row_d[r] * row_d[c]));
        acc[shift++] += (double)(weight_int * (row_int[r] * error_int) + n_factor * weight_d * (row_d[r] * error_d));
      }
    }
  }
  double host[27] = { 0 };
  for (int y = 0; y < rows; ++y)
    for (int s = 0; s < 27; ++s) host[s] += rowsum[(size_t)y * 27 + s];
  free(rowsum);
  int shift = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 7; ++c) {
      double value = host[shift++];
      if (c == 6) b[r] = value;
      else A[c * 6 + r] = A[r * 6 + c] = value;
    }
}
