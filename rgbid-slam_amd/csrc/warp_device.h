// warp_device.h -- per-pixel inverse warps shared by the stand-alone warp kernels (kernels_warp.hip), the fused
// Gauss-Newton kernel (kernels_system.hip) and the lattice pre-pass of the sigma/nu kernel (kernels_sigma.hip).
// All three therefore produce bit-identical W1 / I1 values.  fp contraction is off inside these functions so the
// arithmetic is evaluated operation by operation exactly like the scalar oracle (see common.h register_pixel).
#pragma once
#include "common.h"

namespace rgbid {

// CUDA linear filtering at unnormalised coordinates with clamp addressing (what tex2D<float> computes for the
// reference's cudaFilterModeLinear texture, warping_registration.cu:938-944); mode 1 = 1.8 fixed-point weights
__device__ __forceinline__ float tex2d_linear(const ImgB& src, int lane, float xs, float ys, int mode) {
#pragma clang fp contract(off)
  float xB = xs - 0.5f, yB = ys - 0.5f;
  float fx0 = floorf(xB), fy0 = floorf(yB);
  float a = xB - fx0, b = yB - fy0;
  if (mode == 1) {
    a = rintf(a * 256.f) * 0.00390625f;
    b = rintf(b * 256.f) * 0.00390625f;
  }
  int i0 = f2i_rd(fx0), j0 = f2i_rd(fy0);
  int i1 = min(max(i0 + 1, 0), src.cols - 1), j1 = min(max(j0 + 1, 0), src.rows - 1);
  i0 = min(max(i0, 0), src.cols - 1);
  j0 = min(max(j0, 0), src.rows - 1);
  const float* r0 = row_ptr<float>(src, lane, j0);
  const float* r1 = row_ptr<float>(src, lane, j1);
  float T00 = r0[i0], T10 = r0[i1], T01 = r1[i0], T11 = r1[i1];
  float oa = 1.f - a, ob = 1.f - b;
  return (oa * ob) * T00 + (a * ob) * T10 + (oa * b) * T01 + (a * b) * T11;
}

// trafo3DKernelInvDepthGridStride, warping_registration.cu:505-546 (one pixel; w = keyframe inverse depth)
__device__ __forceinline__ float warp_invdepth_px(const ImgB& src, int lane, int x, int y, float w, const WarpParams& P) {
#pragma clang fp contract(off)
  float out = qnan();
  if (!isnan(w)) {
    float xs, ys;
    float w3 = register_pixel(xs, ys, x, y, w, P);
    xs += 0.5f; ys += 0.5f;
    if (in_bounds_rd(xs, ys, src.cols, src.rows)) {
      float w2 = px<float>(src, lane, f2i_rd(ys), f2i_rd(xs));
      float tz = P.t[2];
      float v1_z = (1.f / w3 - tz) * w;
      float res = (v1_z / (1.f - w2 * tz)) * w2;
      if (res > 0.f) out = res;
    }
  }
  return out;
}

// trafo3DKernelIntensityWithInvDepthGridStride, warping_registration.cu:465-501 (one pixel; w = sampling-grid iD)
__device__ __forceinline__ float warp_intensity_px(const ImgB& src, int lane, int x, int y, float w, const WarpParams& P, int interp_mode) {
#pragma clang fp contract(off)
  float res = qnan();
  if (!isnan(w)) {
    float xs, ys;
    register_pixel(xs, ys, x, y, w, P);
    xs += 0.5f; ys += 0.5f;
    if (in_bounds_rd(xs, ys, src.cols, src.rows)) {
      res = tex2d_linear(src, lane, xs, ys, interp_mode);
      res = fmaxf(0.f, fminf(res, 255.f));  // NaN -> 255, as CUDA's min/max
    }
  }
  return res;
}

}  // namespace rgbid
