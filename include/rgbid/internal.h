// internal.h -- the reference's bridge API (src/internal.h) re-expressed over the HIP library's C-ABI.
//
// Same namespace (RGBID_SLAM::device), same types / enums / defaults (src/internal.h:40-177) and the same
// function prototypes (src/internal.h:187-453) for every function that has a definition in the reference
// and lies on the tracking path, so src/visodo.cpp and src/keyframe_align.cpp compile against this header in
// place of the original.  Each function is a thin inline wrapper over one rgbid_* entry point (rgbid.h cites the
// .cu wrapper it replaces); functions that return `float` return elapsed milliseconds like the reference's
// cudaTimer.  Any error prints "Error: ...\t<file>:<line>" and exit(0)s, exactly as cudaSafeCall did.
//
// The ONE source change a caller needs: `extern cudaDeviceProp dev_prop` (src/internal.h:36) becomes
// `extern DeviceProp dev_prop` (an rgbid_device_prop filled by rgbid_get_device_prop).
#pragma once
#include <cstring>
#include <iostream>

#include "../rgbid.h"
#include "containers.hpp"

// CUDA vector PODs used in the bridge signatures (float3 / uchar3); layout-identical stand-ins unless a HIP/CUDA
// header already provided them
#if !defined(HIP_INCLUDE_HIP_AMD_DETAIL_HIP_VECTOR_TYPES_H) && !defined(__VECTOR_TYPES_H__) && !defined(RGBID_NO_VECTOR_PODS)
struct float3 { float x, y, z; };
struct uchar3 { unsigned char x, y, z; };
struct float4 { float x, y, z, w; };
#endif

using namespace pcl::gpu;

namespace RGBID_SLAM {
namespace device {

typedef rgbid_device_prop DeviceProp;
extern DeviceProp dev_prop;  // defined by the application, like the reference (tools/RGBID_SLAMapp.cpp:68-69)
extern int dev_id;

typedef unsigned short ushort;
typedef unsigned char uchar;
typedef DeviceArray2D<float> MapArr;
typedef DeviceArray2D<ushort> DepthMap;
typedef DeviceArray2D<uchar> IntensityMap;
typedef DeviceArray2D<float> DepthMapf;
typedef DeviceArray2D<float> IntensityMapf;
typedef DeviceArray2D<float> GradientMap;
typedef DeviceArray2D<uchar> BinaryMap;
typedef float4 PointType;
typedef double float_type;

enum { B_SIZE = 6, A_SIZE = (B_SIZE * B_SIZE - B_SIZE) / 2 + B_SIZE, TOTAL_SIZE = A_SIZE + B_SIZE };
enum { LSQ, HUBER, TUKEY, STUDENT };
enum { NO_MM, CONSTANT_VELOCITY };
enum { SIGMA_MAD, SIGMA_PDF, SIGMA_CONS };
enum { INDEPENDENT, MIN_WEIGHT, GEOM_ONLY, PHOT_ONLY };
enum { WARP_FIRST, PYR_FIRST };
enum { CHI_SQUARED, ALL_ITERS };
enum { NO_FILTERS, FILTER_GRADS };

const float THRESHOLD_HUBER = 1.345f;
const float THRESHOLD_TUKEY = 4.685f;
const float STUDENT_DOF = 5.f;
const float FOCAL_LENGTH = 543.78f;
const float CENTER_X = 313.45f;
const float CENTER_Y = 235.00f;
const float FOCAL_LENGTH_DEPTH = 580.f;

const int DEFAULT_MOTION_MODEL = CONSTANT_VELOCITY;
const int DEFAULT_MESTIMATOR = STUDENT;
const int DEFAULT_FINEST_LEVEL = 0;
const int DEFAULT_SIGMA = SIGMA_PDF;
const int DEFAULT_WEIGHTING = INDEPENDENT;
const int DEFAULT_WARPING = WARP_FIRST;
const int DEFAULT_ODO_KF_COUNT = 9999999;
const int DEFAULT_INTEGR_KF_COUNT = 9999999;
const float DEFAULT_VISRATIO_ODO = 0.9f;
const float DEFAULT_VISRATIO_INTEGR = 0.7f;
const int DEFAULT_TERMINATION = ALL_ITERS;
const int DEFAULT_IMAGE_FILTERING = NO_FILTERS;
const int DEFAULT_NSAMPLES = 10000;

struct Intr {
  float fx, fy, cx, cy, k1, k2, k3, k4, k5;
  Intr() {}
  Intr(float fx_, float fy_, float cx_, float cy_, float k1_ = 0.f, float k2_ = 0.f, float k3_ = 0.f, float k4_ = 0.f, float k5_ = 0.f)
      : fx(fx_), fy(fy_), cx(cx_), cy(cy_), k1(k1_), k2(k2_), k3(k3_), k4(k4_), k5(k5_) {}
  Intr operator()(int level_index) const {
    int div = 1 << level_index;
    return (Intr(fx / div, fy / div, cx / div, cy / div, k1, k2, k3, k4, k5));
  }
  friend inline std::ostream& operator<<(std::ostream& os, const Intr& intr) {
    os << "([f = " << intr.fx << ", " << intr.fy << "] [cp = " << intr.cx << ", " << intr.cy << "])";
    return (os);
  }
};

// depth-sensor distortion model (src/internal.h:142-161)
struct DepthDist {
  float c1, c0;
  float q00, q01, q02, q03, q04, q05, q06, q07, q08;
  float q10, q11, q12, q13, q14, q15, q16, q17, q18;
  int xshift, yshift;
  DepthDist() {}
  DepthDist(float c1_, float c0_, float q00_ = 0.f, float q01_ = 0.f, float q02_ = 0.f, float q03_ = 0.f, float q04_ = 0.f, float q05_ = 0.f,
            float q06_ = 0.f, float q07_ = 0.f, float q08_ = 0.f, float q10_ = 1.f, float q11_ = 0.f, float q12_ = 0.f, float q13_ = 0.f,
            float q14_ = 0.f, float q15_ = 0.f, float q16_ = 0.f, float q17_ = 0.f, float q18_ = 0.f, int xshift_ = 4, int yshift_ = 4)
      : c1(c1_), c0(c0_), q00(q00_), q01(q01_), q02(q02_), q03(q03_), q04(q04_), q05(q05_), q06(q06_), q07(q07_), q08(q08_),
        q10(q10_), q11(q11_), q12(q12_), q13(q13_), q14(q14_), q15(q15_), q16(q16_), q17(q17_), q18(q18_), xshift(xshift_), yshift(yshift_) {}
};
struct Mat33 { float3 data[3]; };      // three rows (src/internal.h:166-169)
struct LightSource { float3 pos[1]; int number; };

inline rgbid_intr c_intr(const Intr& k) { rgbid_intr r = {k.fx, k.fy, k.cx, k.cy}; return r; }
inline rgbid_intr_k c_intr_k(const Intr& k) { rgbid_intr_k r = {k.fx, k.fy, k.cx, k.cy, k.k1, k.k2, k.k3, k.k4, k.k5}; return r; }
inline rgbid_depth_dist c_depth_dist(const DepthDist& d) {
  rgbid_depth_dist r = {d.c1, d.c0, {d.q00, d.q01, d.q02, d.q03, d.q04, d.q05, d.q06, d.q07, d.q08},
                        {d.q10, d.q11, d.q12, d.q13, d.q14, d.q15, d.q16, d.q17, d.q18}, d.xshift, d.yshift};
  return r;
}
template <class T> inline rgbid_img c_img(const DeviceArray2D<T>& a) { return a.img(); }
template <class T> inline rgbid_img c_img(const PtrStepSz<T>& a) { rgbid_img i; i.data = (void*)a.data; i.step = a.step; i.rows = a.rows; i.cols = a.cols; return i; }

// ---- debug
inline void showGPUMemoryUsage() {  // misc.cu:526-540
  size_t free_byte, total_byte;
  rgbidSafeCall(rgbid_mem_info(&free_byte, &total_byte));
  double free_db = (double)free_byte, total_db = (double)total_byte, used_db = total_db - free_db;
  std::cout << "GPU memory usage: used =  " << used_db / 1024.0 / 1024.0 << " MB, free = " << free_db / 1024.0 / 1024.0
            << " MB, total = " << total_db / 1024.0 / 1024.0 << std::endl;
}

// ---- pyramid
inline float pyrDownDepth(DepthMapf& src, DepthMapf& dst, int numSMs = -1) {
  (void)numSMs;
  dst.create(src.rows() / 2, src.cols() / 2);  // pyrdown.cu:224
  float ms; rgbid_img a = c_img(src), b = c_img(dst);
  rgbidSafeCall(rgbid_pyr_down(default_ctx(), &a, &b, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float pyrDownIntensity(IntensityMapf& src, IntensityMapf& dst, int numSMs = -1) { return pyrDownDepth(src, dst, numSMs); }

// ---- converters
inline void convertDepth2InvDepth(const DepthMap& src, DepthMapf& dst, float factor_depth) {
  rgbid_img a = c_img(src), b = c_img(dst);
  rgbidSafeCall(rgbid_depth_to_invdepth(default_ctx(), &a, &b, factor_depth));
}
inline void computeIntensity(const PtrStepSz<uchar3>& src, IntensityMapf& dst) {
  rgbid_img a = c_img(src), b = c_img(dst);
  rgbidSafeCall(rgbid_compute_intensity(default_ctx(), &a, &b));
}
inline void decomposeRGBInChannels(const PtrStepSz<uchar3>& src, IntensityMapf& r_dst, IntensityMapf& g_dst, IntensityMapf& b_dst) {
  rgbid_img a = c_img(src), r = c_img(r_dst), g = c_img(g_dst), b = c_img(b_dst);
  rgbidSafeCall(rgbid_decompose_rgb(default_ctx(), &a, &r, &g, &b));
}
inline float computeGradientIntensity(const IntensityMapf& src, GradientMap& dst_hor, GradientMap& dst_vert, int numSMs = -1) {
  (void)numSMs;
  float ms; rgbid_img a = c_img(src), h = c_img(dst_hor), v = c_img(dst_vert);
  rgbidSafeCall(rgbid_compute_gradient(default_ctx(), &a, &h, &v, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float computeGradientDepth(const DepthMapf& src, GradientMap& dst_hor, GradientMap& dst_vert, int numSMs = -1) {
  return computeGradientIntensity(src, dst_hor, dst_vert, numSMs);
}
inline void copyImages(const DepthMapf& src_depth, const IntensityMapf& src_int, DepthMapf& dst_depth, IntensityMapf& dst_int) {
  rgbid_img a = c_img(src_depth), b = c_img(src_int), c = c_img(dst_depth), d = c_img(dst_int);
  rgbidSafeCall(rgbid_copy_images(default_ctx(), &a, &b, &c, &d));
}
inline void copyImage(const DeviceArray2D<float>& src, DeviceArray2D<float>& dst) {
  rgbid_img a = c_img(src), b = c_img(dst);
  rgbidSafeCall(rgbid_copy_image(default_ctx(), &a, &b));
}
inline void copyImageRGB(const PtrStepSz<uchar3>& src, PtrStepSz<uchar3> dst) {
  rgbid_img a = c_img(src), b = c_img(dst);
  rgbidSafeCall(rgbid_copy_image_rgb(default_ctx(), &a, &b));
}
inline void initialiseWeightKeyframe(const DepthMapf& src_depth, DeviceArray2D<float>& dst_weight) {
  rgbid_img a = c_img(src_depth), b = c_img(dst_weight);
  rgbidSafeCall(rgbid_init_weight_keyframe(default_ctx(), &a, &b));
}
template <typename T> inline void initialiseDeviceMemory2D(DeviceArray2D<T>& src, T val, int numSMs = -1) {
  (void)numSMs;
  static_assert(sizeof(T) == 1 || sizeof(T) == 4, "u8, i8, u32, i32, f32 (misc.cu:506-510)");
  uint32_t bits = 0;
  memcpy(&bits, &val, sizeof(T));
  rgbid_img a = c_img(src);
  rgbidSafeCall(rgbid_fill_2d(default_ctx(), &a, (int)sizeof(T), bits));
}

// ---- residuals + scale
inline float computeErrorGridStride(const DeviceArray2D<float>& im1, const DeviceArray2D<float>& im0, DeviceArray<float>& error,
                                    int Nsamples = 9999999, int numSMs = -1) {
  (void)numSMs;
  int n = 0;
  rgbidSafeCall(rgbid_error_lattice_size(im0.rows(), im0.cols(), Nsamples, &n, 0, 0, 0));
  error.create(n);  // sigmaFuncs.cu:738
  float ms; rgbid_img a = c_img(im1), b = c_img(im0);
  rgbidSafeCall(rgbid_compute_error(default_ctx(), &a, &b, error.ptr(), Nsamples, &n, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float computeChiSquare(DeviceArray<float>& error_int, DeviceArray<float>& error_depth, float sigma_int, float sigma_depth,
                              int Mestimator, float& chi_square, float& chi_test, float& Ndof, int numSMs = -1) {
  (void)numSMs;
  float ms;
  rgbidSafeCall(rgbid_chi_square(default_ctx(), error_int.ptr(), error_depth.ptr(), (int)error_depth.size(), sigma_int, sigma_depth, Mestimator,
                                 &chi_square, &chi_test, &Ndof, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float computeSigmaPdf(DeviceArray<float>& error, float& bias, float& sigma, int Mestimator, int numSMs = -1) {
  (void)numSMs;
  float ms;
  rgbidSafeCall(rgbid_sigma_pdf(default_ctx(), error.ptr(), (int)error.size(), &bias, &sigma, Mestimator, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float computeSigmaAndNuStudent(DeviceArray<float>& error, float& bias, float& sigma, float& nu, int Mestimator, int numSMs = -1) {
  (void)numSMs;
  float ms;
  rgbidSafeCall(rgbid_sigma_nu_student(default_ctx(), error.ptr(), (int)error.size(), &bias, &sigma, &nu, Mestimator, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float computeNuStudent(DeviceArray<float>& error, float& bias, float& sigma, float& nu, int numSMs = -1) {
  (void)numSMs;
  float ms;
  rgbidSafeCall(rgbid_nu_student(default_ctx(), error.ptr(), (int)error.size(), bias, sigma, &nu, pcl::gpu::ms_arg(ms)));
  return ms;
}

// ---- normal equations.  delta_trans / delta_rot / size_A / gbuf / mbuf are unused by the kernels of the reference as
// well; they stay in the signature so call sites compile unchanged.
inline float buildSystemGridStride(const float3 delta_trans, const float3 delta_rot, const DepthMapf& W0, const IntensityMapf& I0,
                                   const GradientMap& gradW0_x, const GradientMap& gradW0_y, const GradientMap& gradI0_x,
                                   const GradientMap& gradI0_y, const DepthMapf& W1, const IntensityMapf& I1, int Mestimator, int weighting,
                                   float sigma_depth, float sigma_int, float bias_depth, float bias_int, const Intr& intr, const int size_A,
                                   DeviceArray2D<float_type>& gbuf, DeviceArray<float_type>& mbuf, float_type* matrixA_host,
                                   float_type* vectorB_host, int numSMs = -1) {
  (void)delta_trans; (void)delta_rot; (void)size_A; (void)gbuf; (void)mbuf; (void)numSMs;
  float ms;
  rgbid_img m[8] = {c_img(W0), c_img(I0), c_img(gradW0_x), c_img(gradW0_y), c_img(gradI0_x), c_img(gradI0_y), c_img(W1), c_img(I1)};
  rgbidSafeCall(rgbid_build_system(default_ctx(), &m[0], &m[1], &m[2], &m[3], &m[4], &m[5], &m[6], &m[7], Mestimator, weighting, sigma_depth,
                                   sigma_int, bias_depth, bias_int, c_intr(intr), matrixA_host, vectorB_host, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float buildSystemStudentNuGridStride(const float3 delta_trans, const float3 delta_rot, const DepthMapf& W0, const IntensityMapf& I0,
                                            const GradientMap& gradW0_x, const GradientMap& gradW0_y, const GradientMap& gradI0_x,
                                            const GradientMap& gradI0_y, const DepthMapf& W1, const IntensityMapf& I1, int Mestimator,
                                            int weighting, float sigma_depth, float sigma_int, float bias_depth, float bias_int,
                                            float nu_depth, float nu_int, const Intr& intr, const int size_A,
                                            DeviceArray2D<float_type>& gbuf, DeviceArray<float_type>& mbuf, float_type* matrixA_host,
                                            float_type* vectorB_host, int numSMs = -1) {
  (void)delta_trans; (void)delta_rot; (void)size_A; (void)gbuf; (void)mbuf; (void)numSMs;
  float ms;
  rgbid_img m[8] = {c_img(W0), c_img(I0), c_img(gradW0_x), c_img(gradW0_y), c_img(gradI0_x), c_img(gradI0_y), c_img(W1), c_img(I1)};
  rgbidSafeCall(rgbid_build_system_student_nu(default_ctx(), &m[0], &m[1], &m[2], &m[3], &m[4], &m[5], &m[6], &m[7], Mestimator, weighting,
                                              sigma_depth, sigma_int, bias_depth, bias_int, nu_depth, nu_int, c_intr(intr), matrixA_host,
                                              vectorB_host, pcl::gpu::ms_arg(ms)));
  return ms;
}

// ---- warps / fusion / visibility
inline float warpIntensityWithTrafo3DInvDepth(IntensityMapf& src, IntensityMapf& dst, const DepthMapf& depthinv_prev, Mat33 inv_rotation,
                                              float3 inv_translation, const Intr& intr, int numSMs = -1) {
  (void)intr; (void)numSMs;
  float ms; rgbid_img a = c_img(src), b = c_img(dst), c = c_img(depthinv_prev);
  rgbidSafeCall(rgbid_warp_intensity(default_ctx(), &a, &b, &c, &inv_rotation.data[0].x, &inv_translation.x, pcl::gpu::ms_arg(ms)));
  return ms;
}
// ---- custom-calibration front-end (src/internal.h:354-356,437-440)
inline float undistortIntensity(IntensityMapf& src, IntensityMapf& dst, const Intr& intr_int, int numSMs = -1) {
  (void)numSMs;
  float ms; rgbid_img a = c_img(src), b = c_img(dst); rgbid_intr_k k = c_intr_k(intr_int);
  rgbidSafeCall(rgbid_undistort_intensity(default_ctx(), &a, &b, &k, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float undistortDepthInv(const DepthMapf& src, DepthMapf& src_corr, DepthMapf& dst, const Intr& intr_depth, const DepthDist& dp, int numSMs = -1) {
  (void)numSMs;
  float ms; rgbid_img a = c_img(src), b = c_img(src_corr), c = c_img(dst); rgbid_intr_k k = c_intr_k(intr_depth); rgbid_depth_dist d = c_depth_dist(dp);
  rgbidSafeCall(rgbid_undistort_depthinv(default_ctx(), &a, &b, &c, &k, &d, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float registerDepthinv(const DepthMapf& src, DepthMapf& intermediate, DeviceArray2D<int>& intermediate_as_int, DepthMapf& dst, const Mat33 dRc_proj,
                              float3 t_dc_proj, const Mat33 cRd_proj, int numSMs = -1) {
  (void)numSMs;
  float ms; rgbid_img a = c_img(src), b = c_img(intermediate), c = c_img(intermediate_as_int), d = c_img(dst);
  rgbidSafeCall(rgbid_register_depthinv(default_ctx(), &a, &b, &c, &d, &dRc_proj.data[0].x, &t_dc_proj.x, &cRd_proj.data[0].x, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float warpInvDepthWithTrafo3D(DepthMapf& src, DepthMapf& dst, const DepthMapf& depth_prev, Mat33 inv_rotation, float3 inv_translation,
                                     const Intr& intr, int numSMs = -1) {
  (void)intr; (void)numSMs;
  float ms; rgbid_img a = c_img(src), b = c_img(dst), c = c_img(depth_prev);
  rgbidSafeCall(rgbid_warp_invdepth(default_ctx(), &a, &b, &c, &inv_rotation.data[0].x, &inv_translation.x, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float warpInvDepthWithTrafo3DWeighted(DepthMapf& src, DepthMapf& dst, const DepthMapf& depth_prev, DeviceArray2D<float>& weight_warped,
                                             Mat33 inv_rotation_proj, float3 inv_translation_proj, const Intr& intr, int numSMs = -1) {
  (void)intr; (void)numSMs;
  float ms; rgbid_img a = c_img(src), b = c_img(dst), c = c_img(depth_prev), w = c_img(weight_warped);
  rgbidSafeCall(rgbid_warp_invdepth_weighted(default_ctx(), &a, &b, &c, &w, &inv_rotation_proj.data[0].x, &inv_translation_proj.x, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float integrateWarpedFrame(const DepthMapf& warped_depth_src, const DeviceArray2D<float>& warped_weight_src, DepthMapf& depth_dst,
                                  DeviceArray2D<float>& weight_dst, int numSMs = -1) {
  (void)numSMs;
  float ms; rgbid_img a = c_img(warped_depth_src), b = c_img(warped_weight_src), c = c_img(depth_dst), d = c_img(weight_dst);
  rgbidSafeCall(rgbid_integrate_warped_frame(default_ctx(), &a, &b, &c, &d, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float getVisibilityRatioWithOverlapMask(const DepthMapf& depth_src, const DepthMapf& depth_dst, Mat33 rotation, float3 translation,
                                               const Intr& intr, float& visibility_ratio, float geom_tol, BinaryMap& overlap_mask,
                                               int numSMs = -1) {
  (void)intr; (void)geom_tol; (void)numSMs;
  float ms; rgbid_img a = c_img(depth_src), b = c_img(depth_dst), m = c_img(overlap_mask);
  rgbidSafeCall(rgbid_visibility_ratio(default_ctx(), &a, &b, &rotation.data[0].x, &translation.x, &m, &visibility_ratio, pcl::gpu::ms_arg(ms)));
  return ms;
}
inline float getVisibilityRatio(const DepthMapf& depth_src, const DepthMapf& depth_dst, Mat33 rotation, float3 translation, const Intr& intr,
                                float& visibility_ratio, float geom_tol, int numSMs = -1) {
  (void)intr; (void)geom_tol; (void)numSMs;
  float ms; rgbid_img a = c_img(depth_src), b = c_img(depth_dst);
  rgbidSafeCall(rgbid_visibility_ratio(default_ctx(), &a, &b, &rotation.data[0].x, &translation.x, 0, &visibility_ratio, pcl::gpu::ms_arg(ms)));
  return ms;
}

// ---- maps / preview
inline void createVMap(const Intr& intr, const DepthMapf& depth, MapArr& vmap, int numSMs = -1) {
  (void)numSMs;
  vmap.create(depth.rows() * 3, depth.cols());  // maps.cu:303
  rgbid_img a = c_img(depth), b = c_img(vmap);
  rgbidSafeCall(rgbid_create_vmap(default_ctx(), c_intr(intr), &a, &b));
}
// bridge functions the reference defines but its tracker does not call any more (src/internal.h:217-221,362-365,384-385)
inline void convertDepth2Float(const DepthMap& src, DepthMapf& dst) {
  rgbid_img a = c_img(src), b = c_img(dst);
  rgbidSafeCall(rgbid_depth_to_float(default_ctx(), &a, &b));
}
inline void convertFloat2RGB(const IntensityMapf& src, PtrStepSz<uchar3> dst) {
  rgbid_img a = c_img(src), b = c_img(dst);
  rgbidSafeCall(rgbid_float_to_rgb(default_ctx(), &a, &b));
}
inline void createNMap(const MapArr& vmap, MapArr& nmap, int numSMs = -1) {
  (void)numSMs;
  nmap.create(vmap.rows(), vmap.cols());  // maps.cu:349
  rgbid_img a = c_img(vmap), b = c_img(nmap);
  rgbidSafeCall(rgbid_create_nmap(default_ctx(), &a, &b));
}
inline float integrateWarpedRGB(const DepthMapf& depth_warped_src, const IntensityMapf& r_warped_src, const IntensityMapf& g_warped_src,
                                const IntensityMapf& b_warped_src, const DeviceArray2D<float>& weight_warped_src, DepthMapf& depth_dst,
                                PtrStepSz<uchar3> colors_dst, DeviceArray2D<float>& weight_dst, int numSMs = -1) {
  (void)numSMs;
  float ms; rgbid_img a = c_img(depth_warped_src), r = c_img(r_warped_src), g = c_img(g_warped_src), b = c_img(b_warped_src), w = c_img(weight_warped_src),
                      d = c_img(depth_dst), c = c_img(colors_dst), q = c_img(weight_dst);
  rgbidSafeCall(rgbid_integrate_warped_rgb(default_ctx(), &a, &r, &g, &b, &w, &d, &c, &q, pcl::gpu::ms_arg(ms)));
  return ms;
}
// device_cast (src/internal.h:459-463): bit-copy between layout-compatible host and device PODs (e.g. a row-major float[9] -> Mat33)
template <class D, class S> inline D device_cast(const S& source) {
  static_assert(sizeof(D) <= sizeof(S), "device_cast: destination larger than source");
  D d; std::memcpy(&d, &source, sizeof(D)); return d;
}
inline void createNMapGradients(const Intr& intr, const DepthMapf& depth_inv, const GradientMap& grad_x, const GradientMap& grad_y, MapArr& nmap,
                                int numSMs = -1) {
  (void)numSMs;
  nmap.create(depth_inv.rows() * 3, depth_inv.cols());  // maps.cu:399
  rgbid_img a = c_img(depth_inv), gx = c_img(grad_x), gy = c_img(grad_y), n = c_img(nmap);
  rgbidSafeCall(rgbid_create_nmap_gradients(default_ctx(), c_intr(intr), &a, &gx, &gy, &n));
}
inline void generateImage(const MapArr& vmap, const MapArr& nmap, const LightSource& light, PtrStepSz<uchar3> dst) {
  rgbid_img v = c_img(vmap), n = c_img(nmap), d = c_img(dst);
  rgbidSafeCall(rgbid_generate_image(default_ctx(), &v, &n, 0, &light.pos[0].x, &d));
}
inline void generateImageRGB(const MapArr& vmap, const MapArr& nmap, const PtrStepSz<uchar3>& rgb, const LightSource& light, PtrStepSz<uchar3> dst) {
  rgbid_img v = c_img(vmap), n = c_img(nmap), c = c_img(rgb), d = c_img(dst);
  rgbidSafeCall(rgbid_generate_image(default_ctx(), &v, &n, &c, &light.pos[0].x, &d));
}
inline float bilateralFilter(const DeviceArray2D<float>& src, DeviceArray2D<float>& dst, const float sigma_floatmap, int numSMs = -1) {
  (void)numSMs;
  float ms; rgbid_img a = c_img(src), b = c_img(dst);
  rgbidSafeCall(rgbid_bilateral_filter(default_ctx(), &a, &b, sigma_floatmap, pcl::gpu::ms_arg(ms)));
  return ms;
}

/** \brief synchronizes device execution (src/internal.h:456-457) */
inline void sync() { rgbidSafeCall(rgbid_ctx_sync(default_ctx())); }

template <class D, class Matx> D& device_cast(Matx& matx) { return (*reinterpret_cast<D*>(matx.data())); }

}  // namespace device
}  // namespace RGBID_SLAM
