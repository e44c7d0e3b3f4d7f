// keyframe_align.cpp -- see include/rgbid/keyframe_align.h; follows src/keyframe_align.cpp:115-357 of the reference.
#include "../../include/rgbid/keyframe_align.h"

#include <algorithm>
#include <cstdlib>
#include <iostream>

#include "../../include/rgbid/se3.h"

using namespace RGBID_SLAM::device;
namespace se3 = rgbid::se3;

namespace RGBID_SLAM {

KeyframeAlign::KeyframeAlign(int rows, int cols) : rows_(rows), cols_(cols), finest_level_(0) {
  // keyframe_align.cpp:34-98
  const int iters[] = {5, 5, 3, 0};
  std::copy(iters, iters + LEVELS, alignment_iterations_);
}

KeyframeAlign::~KeyframeAlign() {
  if (aligner_) rgbid_kfalign_destroy(aligner_);
  if (aligner_ctx_) rgbid_ctx_destroy(aligner_ctx_);
}

// the containers of the host-driven loop (allocated on its first use: the device-resident aligner has its own)
void KeyframeAlign::allocateHostDrivenBuffers() {
  if (!depthinvs_ini_.empty()) return;
  const int rows = rows_, cols = cols_;
  depthinvs_ini_.resize(LEVELS); depthinvs_end_.resize(LEVELS); warped_depthinvs_end_.resize(LEVELS);
  intensities_ini_.resize(LEVELS); intensities_end_.resize(LEVELS); warped_intensities_end_.resize(LEVELS);
  xGradsDepthinv_ini_.resize(LEVELS); yGradsDepthinv_ini_.resize(LEVELS); xGradsIntensity_ini_.resize(LEVELS); yGradsIntensity_ini_.resize(LEVELS);
  res_depthinvs_.resize(LEVELS); res_intensities_.resize(LEVELS);
  for (int i = 0; i < LEVELS; ++i) {
    int pr = rows >> i, pc = cols >> i;
    intensities_ini_[i].create(pr, pc); intensities_end_[i].create(pr, pc); warped_intensities_end_[i].create(pr, pc);
    depthinvs_ini_[i].create(pr, pc); depthinvs_end_[i].create(pr, pc); warped_depthinvs_end_[i].create(pr, pc);
    xGradsDepthinv_ini_[i].create(pr, pc); yGradsDepthinv_ini_[i].create(pr, pc);
    xGradsIntensity_ini_[i].create(pr, pc); yGradsIntensity_ini_[i].create(pr, pc);
    res_depthinvs_[i].create((size_t)pr * pc); res_intensities_[i].create((size_t)pr * pc);
  }
  grey_f_.resize((size_t)rows * cols);
}

bool KeyframeAlign::ensureAligner(int pairs) {
  // a context of the object's own (the per-thread default context ends with its thread, see VisodoTracker::createEngine)
  if (!aligner_ctx_ && rgbid_ctx_create(&aligner_ctx_, pcl::gpu::current_device().load(), nullptr) != RGBID_OK) return false;
  // the host-driven loop samples with the thread's default context, whose interpolation mode VisodoTracker::setInterpMode changes: the aligner's own
  // context follows it at every call, so both modes sample alike whatever the application selected
  int mode = RGBID_INTERP_TEX8;
  if (rgbid_ctx_get_interp_mode(pcl::gpu::default_ctx(), &mode) == RGBID_OK) rgbid_ctx_set_interp_mode(aligner_ctx_, mode);
  if (aligner_ && aligner_cap_ >= pairs) return true;
  if (aligner_) { rgbid_kfalign_destroy(aligner_); aligner_ = nullptr; }
  if (rgbid_kfalign_create(&aligner_, aligner_ctx_, rows_, cols_, pairs) != RGBID_OK) return false;
  aligner_cap_ = pairs;
  return true;
}

bool KeyframeAlign::alignKeyframesBatched(int pairs, const float* depthinv_ini, const unsigned char* grey_ini, const float* depthinv_end, const unsigned char* grey_end,
                                          const float* K, double* R, double* t, double* cov) {
  if (pairs < 1 || !ensureAligner(pairs)) return false;
  return rgbid_kfalign_batched_host(aligner_, pairs, depthinv_ini, grey_ini, depthinv_end, grey_end, K, R, t, cov) == RGBID_OK;
}

bool KeyframeAlign::alignKeyframes(const KeyframeImages& a, const KeyframeImages& b, Affine3d& pose, Matrix6d& cov) {
  return alignKeyframes(a, b, pose.R, pose.t, cov);
}

bool KeyframeAlign::alignKeyframes(const KeyframeImages& kf_ini, const KeyframeImages& kf_end, Matrix3ft& rotation_ini2end,
                                   Vector3ft& translation_ini2end, Matrix6d& covariance_ini2end) {
  if (!host_driven_) {
    // the 1-pair case of the batched, device-resident aligner: the same kernels in the same order, the pose algebra in per-pair kernels
    const float K[4] = {kf_ini.fx, kf_ini.fy, kf_ini.cx, kf_ini.cy};
    if (!alignKeyframesBatched(1, kf_ini.depthinv, kf_ini.grey, kf_end.depthinv, kf_end.grey, K, rotation_ini2end.m, translation_ini2end.v, covariance_ini2end.data())) {
      std::cerr << "KeyframeAlign: the device-resident aligner could not run" << std::endl; std::exit(0);   // as pcl::gpu::error()
    }
    return true;
  }
  allocateHostDrivenBuffers();
  pcl::gpu::ScopedAsyncBridge bridge_scope;   // the returned kernel times are not used here either (see include/rgbid/containers.hpp)
  Intr cam_intrinsics(kf_ini.fx, kf_ini.fy, kf_ini.cx, kf_ini.cy, 0.075f);
  // uploads (:120-129); grey_image_.convertTo(CV_32F)
  depthinvs_ini_[0].upload(kf_ini.depthinv, (size_t)cols_ * 4, rows_, cols_);
  depthinvs_end_[0].upload(kf_end.depthinv, (size_t)cols_ * 4, rows_, cols_);
  for (size_t i = 0; i < grey_f_.size(); ++i) grey_f_[i] = (float)kf_ini.grey[i];
  intensities_ini_[0].upload(grey_f_.data(), (size_t)cols_ * 4, rows_, cols_);
  for (size_t i = 0; i < grey_f_.size(); ++i) grey_f_[i] = (float)kf_end.grey[i];
  intensities_end_[0].upload(grey_f_.data(), (size_t)cols_ * 4, rows_, cols_);

  double A_total[36], b_total[6];
  Matrix3ft current_rotation = rotation_ini2end; Vector3ft current_translation = translation_ini2end;
  for (int i = 1; i < LEVELS; ++i) {  // :155-162
    pyrDownDepth(depthinvs_ini_[i - 1], depthinvs_ini_[i]);
    pyrDownDepth(depthinvs_end_[i - 1], depthinvs_end_[i]);
    pyrDownIntensity(intensities_ini_[i - 1], intensities_ini_[i]);
    pyrDownIntensity(intensities_end_[i - 1], intensities_end_[i]);
  }
  for (int i = 0; i < LEVELS; ++i) {  // :166-174
    computeGradientDepth(depthinvs_ini_[i], xGradsDepthinv_ini_[i], yGradsDepthinv_ini_[i]);
    computeGradientIntensity(intensities_ini_[i], xGradsIntensity_ini_[i], yGradsIntensity_ini_[i]);
  }
  float3 zero3 = {0.f, 0.f, 0.f};
  for (int level_index = LEVELS - 1; level_index >= finest_level_; --level_index) {
    int iter_num = alignment_iterations_[level_index];
    for (int iter = 0; iter < iter_num; ++iter) {
      // :208-231
      Matrix3ft Ri; Vector3ft ti;
      se3::m3_inv(current_rotation.m, Ri.m);
      se3::m3_mulv(Ri.m, current_translation.v, ti.v);
      for (int i = 0; i < 3; ++i) ti.v[i] = -ti.v[i];
      int div = 1 << level_index;
      float Rf[9], tf[3];
      se3::project_trafo(cam_intrinsics.fx / div, cam_intrinsics.fy / div, cam_intrinsics.cx / div, cam_intrinsics.cy / div, Ri.m, ti.v, Rf, tf);
      Mat33 Rp; float3 tp;
      for (int i = 0; i < 3; ++i) { Rp.data[i].x = Rf[i * 3]; Rp.data[i].y = Rf[i * 3 + 1]; Rp.data[i].z = Rf[i * 3 + 2]; }
      tp.x = tf[0]; tp.y = tf[1]; tp.z = tf[2];
      warpInvDepthWithTrafo3D(depthinvs_end_[level_index], warped_depthinvs_end_[level_index], depthinvs_ini_[level_index], Rp, tp, cam_intrinsics(level_index));
      // NOTE: sampled on the keyframe iD, not on the warped iD as the tracker does (:239-242)
      warpIntensityWithTrafo3DInvDepth(intensities_end_[level_index], warped_intensities_end_[level_index], depthinvs_ini_[level_index], Rp, tp, cam_intrinsics(level_index));
      computeErrorGridStride(warped_depthinvs_end_[level_index], depthinvs_ini_[level_index], res_depthinvs_[level_index], 19200);
      computeErrorGridStride(warped_intensities_end_[level_index], intensities_ini_[level_index], res_intensities_[level_index], 19200);
      float sigma_depthinv = 0.0025f, bias_depthinv = 0.f, sigma_intensity = 5.f, bias_intensity = 0.f, nu_depthinv = 5.f, nu_intensity = 5.f;
      computeNuStudent(res_depthinvs_[level_index], bias_depthinv, sigma_depthinv, nu_depthinv);
      computeNuStudent(res_intensities_[level_index], bias_intensity, sigma_intensity, nu_intensity);
      nu_intensity = std::max(nu_depthinv, nu_intensity);
      (void)nu_intensity;  // the reference passes nu_depthinv for BOTH channels (:308)
      buildSystemStudentNuGridStride(zero3, zero3, depthinvs_ini_[level_index], intensities_ini_[level_index], xGradsDepthinv_ini_[level_index],
                                     yGradsDepthinv_ini_[level_index], xGradsIntensity_ini_[level_index], yGradsIntensity_ini_[level_index],
                                     warped_depthinvs_end_[level_index], warped_intensities_end_[level_index], STUDENT, INDEPENDENT, sigma_depthinv,
                                     sigma_intensity, bias_depthinv, bias_intensity, nu_depthinv, nu_depthinv, cam_intrinsics(level_index), B_SIZE,
                                     gbuf_, sumbuf_, A_total, b_total);
      double x[6];
      se3::llt_solve6(A_total, b_total, x);
      Matrix3ft inc_inv, inc; double tinc[3], tmp[3];
      se3::expmap_rot(x + 3, inc_inv.m);
      se3::m3_inv(inc_inv.m, inc.m);
      se3::m3_mulv(inc.m, x, tinc);
      se3::m3_mulv(inc.m, current_translation.v, tmp);
      for (int i = 0; i < 3; ++i) current_translation[i] = tmp[i] - tinc[i];
      se3::m3_mul(inc.m, current_rotation.m, current_rotation.m);
    }
  }
  rotation_ini2end = current_rotation;
  translation_ini2end = current_translation;
  se3::inverse6(A_total, covariance_ini2end.data());  // :343
  return true;
}

}  // namespace RGBID_SLAM
