// keyframe_align.h -- KeyframeAlign: dense keyframe-to-keyframe alignment used by the loop closer
// (reference include/keyframe_align.h:41-104, src/keyframe_align.cpp:34-357), the second consumer of the tracking
// kernels: 4-level pyramid, iterations {5,5,3,0}, fixed sigmas (0.0025 / 5), nu by bisection (computeNuStudent),
// intensity warp sampled on the KEYFRAME inverse depth.  OpenCV-free: keyframes are passed as host arrays.
#pragma once
#include "visodo.h"

namespace RGBID_SLAM {

struct KeyframeImages {           // what alignKeyframes reads from a Keyframe (keyframe_align.cpp:118-129)
  const float* depthinv;          // rows x cols inverse depth (NaN = invalid)
  const unsigned char* grey;      // rows x cols 8-bit grey
  float fx, fy, cx, cy;           // kf->K_
};

class KeyframeAlign {
 public:
  enum { LEVELS = 4 };            // keyframe_align.h:50
  KeyframeAlign(int rows = 480, int cols = 640);
  // rotation/translation_ini2end: in = initial guess, out = aligned pose; covariance = A_final.inverse()
  bool alignKeyframes(const KeyframeImages& kf_ini, const KeyframeImages& kf_end, Matrix3ft& rotation_ini2end, Vector3ft& translation_ini2end,
                      Matrix6d& covariance_ini2end);
  bool alignKeyframes(const KeyframeImages& kf_ini, const KeyframeImages& kf_end, Affine3d& pose_ini2end, Matrix6d& covariance_ini2end);

 private:
  int rows_, cols_, finest_level_;
  int alignment_iterations_[LEVELS];
  std::vector<device::DepthMapf> depthinvs_ini_, depthinvs_end_, warped_depthinvs_end_;
  std::vector<device::IntensityMapf> intensities_ini_, intensities_end_, warped_intensities_end_;
  std::vector<device::GradientMap> xGradsDepthinv_ini_, yGradsDepthinv_ini_, xGradsIntensity_ini_, yGradsIntensity_ini_;
  std::vector<DeviceArray<float> > res_depthinvs_, res_intensities_;
  DeviceArray2D<device::float_type> gbuf_;
  DeviceArray<device::float_type> sumbuf_;
  std::vector<float> grey_f_;
};

}  // namespace RGBID_SLAM
