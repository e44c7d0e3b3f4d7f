// keyframe_align.h -- KeyframeAlign: dense keyframe-to-keyframe alignment used by the loop closer
// (reference include/keyframe_align.h:41-104, src/keyframe_align.cpp:34-357), the second consumer of the tracking
// kernels: 4-level pyramid, iterations {5,5,3,0}, fixed sigmas (0.0025 / 5), nu by bisection (computeNuStudent),
// intensity warp sampled on the KEYFRAME inverse depth.  OpenCV-free: keyframes are passed as host arrays.
#pragma once
#include "visodo.h"
#include "../rgbid_kfalign.h"

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
  ~KeyframeAlign();
  KeyframeAlign(const KeyframeAlign&) = delete;
  KeyframeAlign& operator=(const KeyframeAlign&) = delete;
  // Default since round 5: alignKeyframes runs as the 1-pair case of the batched, device-resident aligner (include/rgbid_kfalign.h: one launch sequence, poses and
  // solves on the device) instead of ~300 synchronous bridge calls; the result is the host-driven loop's bit for bit (tests/test_gpu_tracker_cpp.py).
  // setHostDriven(true) selects the reference's call sequence through the bridge.
  void setHostDriven(bool on) { host_driven_ = on; }
  // loop-closure verification is a batch of candidate pairs: pairs alignments in lock-step (rgbid_kfalign_batched_host); arrays as documented there
  bool alignKeyframesBatched(int pairs, const float* depthinv_ini, const unsigned char* grey_ini, const float* depthinv_end, const unsigned char* grey_end,
                             const float* K, double* R, double* t, double* cov);
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
  bool host_driven_ = false;
  ::rgbid_kfalign* aligner_ = nullptr;
  ::rgbid_ctx* aligner_ctx_ = nullptr;
  int aligner_cap_ = 0;
  bool ensureAligner(int pairs);
  void allocateHostDrivenBuffers();
};

}  // namespace RGBID_SLAM
