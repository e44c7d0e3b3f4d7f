// kernels_calib.hip -- custom-calibration front-end for gfx950: lens undistortion of the intensity and
// inverse-depth maps, the depth-sensor distortion model, and depth -> RGB registration.  Replaces src/cuda/undistortion.cu
// and warping_registration.cu:148-288,597-635,720-822 of the reference.  Only used when the calibration file sets
// custom_registration=1 (prepareImagesCustomCalibration, visodo.cpp:775-824).
//
// The reference binds a texture per call (linear filtering for intensity, point for inverse depth); here the sampling is
// the same warp_device.h code the warps use.  The registration splat is a z-buffer: atomicMax on the bits of the positive
// inverse depth (largest inverse depth = nearest surface wins), so the result is independent of thread order and the
// kernel is bit-comparable to the serial oracle.
#include "kernels.h"
#include "warp_device.h"

#pragma clang fp contract(off)

namespace rgbid {

static constexpr int TX = 64, TY = 4;
static inline dim3 grid_full(int cols, int rows, int B) { return dim3(div_up(cols, TX), div_up(rows, TY), B); }

// distortPixel undistortion.cu:96-112
__device__ __forceinline__ void distort_pixel(float uu, float vu, float& ud, float& vd, const IntrK& k) {
  float r2 = uu * uu + vu * vu;
  float r4 = r2 * r2;
  float r6 = r2 * r4;
  float factor_r = 1.f + k.k1 * r2 + k.k2 * r4 + k.k5 * r6;
  ud = factor_r * uu;
  ud += 2.f * k.k3 * uu * vu + k.k4 * (r2 + 2.f * uu * uu);
  vd = factor_r * vu;
  vd += 2.f * k.k4 * uu * vu + k.k3 * (r2 + 2.f * vu * vu);
}

// undistortKernel undistortion.cu:145-176; LINEAR = cudaFilterModeLinear (intensity), else point (inverse depth)
template <bool LINEAR>
__global__ __launch_bounds__(256) void k_undistort(ImgB src, ImgB dst, IntrK k, int interp_mode, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int xu = blockIdx.x * TX + threadIdx.x, yu = blockIdx.y * TY + threadIdx.y;
  if (xu >= dst.cols || yu >= dst.rows) return;
  const FMap S(src, lane);
  float uu = ((float)xu - k.cx) * (1.f / k.fx);
  float vu = ((float)yu - k.cy) * (1.f / k.fy);
  float ud, vd;
  distort_pixel(uu, vu, ud, vd, k);
  float xd = k.fx * ud + k.cx + 0.5f;
  float yd = k.fy * vd + k.cy + 0.5f;
  const bool in = !((xd <= 0) || (yd <= 0) || (xd >= (float)dst.cols) || (yd >= (float)dst.rows));
  float res;
  if (LINEAR) res = tex2d_linear(S, xd, yd, interp_mode);
  else res = S.at(clampi(cvt_rd(yd), S.rows - 1), clampi(cvt_rd(xd), S.cols - 1));
  px<float>(dst, lane, yu, xu) = in ? res : qnan();
}
void launch_undistort(hipStream_t s, int B, ImgB src, ImgB dst, IntrK k, bool linear, int interp_mode, LaneMask m) {
  if (linear) hipLaunchKernelGGL(k_undistort<true>, grid_full(dst.cols, dst.rows, B), dim3(TX, TY), 0, s, src, dst, k, interp_mode, m);
  else hipLaunchKernelGGL(k_undistort<false>, grid_full(dst.cols, dst.rows, B), dim3(TX, TY), 0, s, src, dst, k, interp_mode, m);
}

// depthinvCorrectionKernel undistortion.cu:179-211 (correctDepthinv :131-142, undistortDepthinv :114-129)
__global__ __launch_bounds__(256) void k_depthinv_correction(ImgB src, ImgB dst, IntrK k, DepthDistP dp, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x, y = blockIdx.y * TY + threadIdx.y;
  if (x >= dst.cols || y >= dst.rows) return;
  float res = qnan();
  int xs = x - dp.xshift, ys = y - dp.yshift;
  if ((xs > 0) && (ys > 0)) {
    float u = ((float)x - k.cx) * (1.f / k.fx);
    float v = ((float)y - k.cy) * (1.f / k.fy);
    float val = px<float>(src, lane, ys, xs);
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
  px<float>(dst, lane, y, x) = res;
}
void launch_depthinv_correction(hipStream_t s, int B, ImgB src, ImgB dst, IntrK k, DepthDistP dp, LaneMask m) {
  hipLaunchKernelGGL(k_depthinv_correction, grid_full(dst.cols, dst.rows, B), dim3(TX, TY), 0, s, src, dst, k, dp, m);
}

// ---- registerDepthinv (warping_registration.cu:720-822) ----------------------------------------------------------
// initialiseRegistrationKernel :168-183
__global__ __launch_bounds__(256) void k_reg_init(ImgB inter_i, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x, y = blockIdx.y * TY + threadIdx.y;
  if (x >= inter_i.cols || y >= inter_i.rows) return;
  px<int>(inter_i, lane, y, x) = 0;
}
// depthinvRegistrationTranslationWithDilationKernel :241-288 (registerPixelTranslationOnly :148-165).  The reference tests
// isnan(dst) before every atomicMax, but dst is all-NaN until the conversion kernel runs, so the test is always true.
__global__ __launch_bounds__(256) void k_reg_splat(ImgB src, ImgB inter_i, float tx, float ty, float tz, int offset_x, int offset_y, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int xd = blockIdx.x * TX + threadIdx.x, yd = blockIdx.y * TY + threadIdx.y;
  if (xd >= src.cols || yd >= src.rows) return;
  float wd = px<float>(src, lane, yd, xd);
  if (isnan(wd)) return;
  float zd = 1.f / wd;
  float X0 = (float)xd * zd - tx, X1 = (float)yd * zd - ty, X2 = zd - tz;
  float wc = 1.f / X2;
  float xc = X0 * wc, yc = X1 * wc;
  if (wc > 0.01f) {
    float dilation = wc / wd;
    int bits = __float_as_int(wc);
    // saturating conversions; indices are brought into [-size, size] before the offset / the loop's +1 so nothing can overflow
    const int ic = inter_i.cols, ir = inter_i.rows;
    int xmin = min(max(f2i_rn(xc - 0.5f * dilation), -ic), ic) + offset_x, xmax = min(max(f2i_rn(xc + 0.5f * dilation), -ic), ic) + offset_x;
    int ymin = min(max(f2i_rn(yc - 0.5f * dilation), -ir), ir) + offset_y, ymax = min(max(f2i_rn(yc + 0.5f * dilation), -ir), ir) + offset_y;
    for (int y = max(0, ymin); y < min(ymax + 1, inter_i.rows); y++)
      for (int x = max(0, xmin); x < min(xmax + 1, inter_i.cols); x++) atomicMax(&px<int>(inter_i, lane, y, x), bits);
  }
}
// conversionRegistrationKernel :186-204 fused with the NaN initialisation of the float view
__global__ __launch_bounds__(256) void k_reg_convert(ImgB inter_i, ImgB inter_f, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x, y = blockIdx.y * TY + threadIdx.y;
  if (x >= inter_f.cols || y >= inter_f.rows) return;
  int b = px<int>(inter_i, lane, y, x);
  px<float>(inter_f, lane, y, x) = b != 0 ? __int_as_float(b) : qnan();
}
// homographyKernelInvDepthGridStride :597-635 (srcHdst = dRc_proj, dstHsrc = cRd_proj, point sampling)
__global__ __launch_bounds__(256) void k_reg_homography(ImgB inter_f, ImgB dst, WarpParams srcHdst, float d20, float d21, float d22, float offset_x, float offset_y, LaneMask m) {
  int lane = blockIdx.z;
  if (!m.on(lane)) return;
  int x = blockIdx.x * TX + threadIdx.x, y = blockIdx.y * TY + threadIdx.y;
  if (x >= dst.cols || y >= dst.rows) return;
  const FMap S(inter_f, lane);
  const float* H = srcHdst.R;
  float fx = (float)x, fy = (float)y;
  float p0 = H[0] * fx + H[1] * fy + H[2] * 1.f;
  float p1 = H[3] * fx + H[4] * fy + H[5] * 1.f;
  float p2 = H[6] * fx + H[7] * fy + H[8] * 1.f;
  float iz = 1.f / p2;
  p0 *= iz; p1 *= iz; p2 *= iz;
  float x_src = p0 + 0.5f + offset_x;
  float y_src = p1 + 0.5f + offset_y;
  int ix = cvt_rd(x_src), iy = cvt_rd(y_src);
  const bool in = inside(ix, iy, S.cols, S.rows);
  float w_src = S.at(clampi(iy, S.rows - 1), clampi(ix, S.cols - 1));
  float pz = d20 * p0 + d21 * p1 + d22 * p2;
  float res = w_src / pz;
  px<float>(dst, lane, y, x) = (in && res > 0.f) ? res : qnan();
}
void launch_register_depthinv(hipStream_t s, int B, ImgB src, ImgB inter_f, ImgB inter_i, ImgB dst, const float dRc_proj[9], const float t_dc_proj[3],
                              const float cRd_proj[9], LaneMask m) {
  int offset_x = (inter_f.cols - src.cols) / 2, offset_y = (inter_f.rows - src.rows) / 2;
  dim3 b(TX, TY);
  hipLaunchKernelGGL(k_reg_init, grid_full(inter_i.cols, inter_i.rows, B), b, 0, s, inter_i, m);
  hipLaunchKernelGGL(k_reg_splat, grid_full(src.cols, src.rows, B), b, 0, s, src, inter_i, t_dc_proj[0], t_dc_proj[1], t_dc_proj[2], offset_x, offset_y, m);
  hipLaunchKernelGGL(k_reg_convert, grid_full(inter_f.cols, inter_f.rows, B), b, 0, s, inter_i, inter_f, m);
  WarpParams H;
  for (int i = 0; i < 9; ++i) H.R[i] = dRc_proj[i];
  H.t[0] = H.t[1] = H.t[2] = 0.f;
  hipLaunchKernelGGL(k_reg_homography, grid_full(dst.cols, dst.rows, B), b, 0, s, inter_f, dst, H, cRd_proj[6], cRd_proj[7], cRd_proj[8],
                     (float)offset_x, (float)offset_y, m);
}

}  // namespace rgbid
