// visodo.cpp -- VisodoTracker on the HIP bridge: the per-frame driver of the reference (src/visodo.cpp) with the same
// control flow, written against include/rgbid/internal.h (same bridge function names as the reference) and
// include/rgbid/se3.h (Eigen-free SE(3) / 6x6 algebra).  Every block cites the reference lines it follows.
// This is the host-driven, one-stream-of-frames path (the drop-in for the existing pipeline); the batched
// device-resident engine (csrc/engine.hip) runs the same algorithm for many streams without host round trips.
#include "../../include/rgbid/visodo.h"

#include <chrono>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/rgbid/se3.h"
#include "../../include/rgbid_engine.h"

using namespace RGBID_SLAM::device;
namespace se3 = rgbid::se3;

namespace RGBID_SLAM {

namespace {
inline Matrix6d zero6() { Matrix6d z; z.fill(0.0); return z; }
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// K R K^-1, K t in float (Eigen float expression `K*R.cast<float>()*K.inverse()`), as Mat33 rows / float3
inline void project(const Matrix3f& K, const Matrix3ft& R, const Vector3ft& t, Mat33& Rp, float3& tp) {
  float Rf[9], tf[3];
  se3::project_trafo(K.m[0], K.m[4], K.m[2], K.m[5], R.m, t.v, Rf, tf);
  for (int i = 0; i < 3; ++i) { Rp.data[i].x = Rf[i * 3]; Rp.data[i].y = Rf[i * 3 + 1]; Rp.data[i].z = Rf[i * 3 + 2]; }
  tp.x = tf[0]; tp.y = tf[1]; tp.z = tf[2];
}
}  // namespace

VisodoTracker::VisodoTracker(int optim_dim, int Mestimator, int motion_model, int sigma_estimator, int weighting, int warping,
                             int max_odoKF_count, int finest_level, int termination, float visratio_odo, int image_filtering,
                             float visratio_integr, int max_integrKF_count, int Nsamples, int rows, int cols, int levels)
    : rows_(rows), cols_(cols), levels_(levels), global_time_(0), lost_(false), optim_dim_(optim_dim), Mestimator_(Mestimator),
      motion_model_(motion_model), sigma_estimator_(sigma_estimator), weighting_(weighting), warping_(warping),
      max_odoKF_count_(max_odoKF_count), finest_level_(finest_level), termination_(termination),
      visibility_ratio_odo_threshold_(visratio_odo), image_filtering_(image_filtering),
      visibility_ratio_integr_threshold_(visratio_integr), max_integrKF_count_(max_integrKF_count), Nsamples_(Nsamples) {
  // src/visodo.cpp:49-97
  k1_ = k2_ = k3_ = k4_ = k5_ = 0.f;
  setRGBIntrinsics(FOCAL_LENGTH, FOCAL_LENGTH, CENTER_X, CENTER_Y);
  setDepthIntrinsics(FOCAL_LENGTH, FOCAL_LENGTH_DEPTH, CENTER_X, CENTER_Y);
  init_Rcam_ = Matrix3ft::Identity();
  init_tcam_ = Vector3ft::Zero();
  custom_registration_ = 0;
  const float eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero3[3] = {0, 0, 0};
  setDepthToRGBExtrinsics(eye, zero3);  // dRc_ = I, t_dc_ = 0 (:59-60)
  const int iters[] = {10, 5, 3, 3, 3, 3, 3, 3};  // {10,5,3} for the reference's 3 levels (:65); extra levels get 3
  for (int i = 0; i < 8; ++i) visodo_iterations_[i] = iters[i];
  real_time_flag_ = false;
  compute_deltat_flag_ = false;
  exit_ = false;
  preview_ = true;
  verbose_ = false;
  keyframe_manager_ptr_ = &null_sink_;
  timestamp_rgb_curr_ = timestamp_depth_curr_ = timestamp_ini_ = 0;
  kf_time_accum_ = 0.f;
  odometry_success_ = false;
  last_info_ = LastFrameInfo();
  allocateBuffers(rows, cols);
  scene_view_.resize((size_t)cols_ * rows_);
  intensity_view_.resize((size_t)cols_ * rows_, 0.f);
  depthinv_view_.resize((size_t)cols_ * rows_);
  newKF_ = false;
  scene_view_has_changed_ = false;
  camera_pose_has_changed_ = false;
  factor_depth_ = 1.f;
  reset();
}

VisodoTracker::~VisodoTracker() {
  if (engine_) { rgbid_engine_destroy(engine_); engine_ = nullptr; }
  if (engine_ctx_) { rgbid_ctx_destroy(engine_ctx_); engine_ctx_ = nullptr; }
  if (visodo_thread_ && visodo_thread_->joinable()) {
    { std::unique_lock<std::mutex> lock(mutex_); exit_ = true; }
    new_frame_cond_.notify_one();
    visodo_thread_->join();
  }
}

void VisodoTracker::setIterations(const int* iters, int n) { for (int i = 0; i < n && i < 8; ++i) visodo_iterations_[i] = iters[i]; }
void VisodoTracker::setInterpMode(int mode) { rgbidSafeCall(rgbid_ctx_set_interp_mode(default_ctx(), mode)); interp_mode_ = mode; }

void VisodoTracker::loadCalibration(std::string const& calib_file) {
  // src/visodo.cpp:99-181 ([CALIBRATION] / [RGB_CALIBRATION]: fx fy cx cy kd factor_depth)
  std::ifstream filestream(calib_file.c_str());
  if (!filestream.is_open()) { std::cout << "Could not open configuration file " << calib_file << std::endl; return; }
  Settings settings(filestream, verbose_);
  Section calibration;
  if (settings.getSection("CALIBRATION", calibration) || settings.getSection("RGB_CALIBRATION", calibration)) {
    Entry entry;
    if (calibration.getEntry("fx", entry)) { std::stringstream ss(entry.getValue()); ss >> fx_; fxd_ = fx_; }
    if (calibration.getEntry("fy", entry)) { fy_ = (float)atof(entry.getValue().c_str()); fyd_ = fy_; }
    if (calibration.getEntry("cx", entry)) { cx_ = (float)atof(entry.getValue().c_str()); cxd_ = cx_; }
    if (calibration.getEntry("cy", entry)) { cy_ = (float)atof(entry.getValue().c_str()); cyd_ = cy_; }
    if (calibration.getEntry("kd", entry)) { std::stringstream ss(entry.getValue()); ss >> k1_ >> k2_ >> k3_ >> k4_ >> k5_; }
    if (calibration.getEntry("factor_depth", entry)) factor_depth_ = (float)atof(entry.getValue().c_str());
  }
  if (settings.getSection("DEPTH_CALIBRATION", calibration)) {  // :183-286
    Entry entry;
    if (calibration.getEntry("custom_registration", entry)) { std::stringstream ss(entry.getValue()); ss >> custom_registration_; }
    if (calibration.getEntry("fx", entry)) { std::stringstream ss(entry.getValue()); ss >> fxd_; }
    if (calibration.getEntry("fy", entry)) fyd_ = (float)atof(entry.getValue().c_str());
    if (calibration.getEntry("cx", entry)) cxd_ = (float)atof(entry.getValue().c_str());
    if (calibration.getEntry("cy", entry)) cyd_ = (float)atof(entry.getValue().c_str());
    if (calibration.getEntry("kd", entry)) { std::stringstream ss(entry.getValue()); ss >> k1d_ >> k2d_ >> k3d_ >> k4d_ >> k5d_; }
    if (calibration.getEntry("c0", entry)) { std::stringstream ss(entry.getValue()); ss >> c0_; }
    if (calibration.getEntry("c1", entry)) { std::stringstream ss(entry.getValue()); ss >> c1_; }
    if (calibration.getEntry("q0", entry)) { std::stringstream ss(entry.getValue()); for (int i = 0; i < 9; ++i) ss >> q0_[i]; }
    if (calibration.getEntry("q1", entry)) { std::stringstream ss(entry.getValue()); for (int i = 0; i < 9; ++i) ss >> q1_[i]; }
  }
  if (settings.getSection("STEREO_DEPTH2RGB", calibration)) {  // :288-316
    Entry entry;
    if (calibration.getEntry("dRc", entry)) { std::stringstream ss(entry.getValue()); for (int i = 0; i < 9; ++i) ss >> dRc_[i]; }
    if (calibration.getEntry("t_dc", entry)) { std::stringstream ss(entry.getValue()); ss >> t_dc_[0] >> t_dc_[1] >> t_dc_[2]; }
  }
}

namespace {
// Eigen Matrix3f products / inverse of prepareImagesCustomCalibration (:792-801), float, cofactor inverse
void m3f_mul(const float* A, const float* B, float* C) {
  float T[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  for (int i = 0; i < 9; ++i) C[i] = T[i];
}
void m3f_inv(const float* A, float* I) {
  float c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  float det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  float id = 1.f / det;
  float T[9] = {c00 * id, (A[2] * A[7] - A[1] * A[8]) * id, (A[1] * A[5] - A[2] * A[4]) * id,
                c01 * id, (A[0] * A[8] - A[2] * A[6]) * id, (A[2] * A[3] - A[0] * A[5]) * id,
                c02 * id, (A[1] * A[6] - A[0] * A[7]) * id, (A[0] * A[4] - A[1] * A[3]) * id};
  for (int i = 0; i < 9; ++i) I[i] = T[i];
}
}  // namespace

// K_d dRc K_c^-1, K_d t_dc and the inverse of the first (src/visodo.cpp:792-801), in float: shared by the host-driven front-end and the engine's configuration
void VisodoTracker::stereoProjections(float dRc_proj[9], float t_dc_proj[3], float cRd_proj[9]) {
  Matrix3f Kc = getCalibMatrix(0), Kd = getCalibMatrixDepth(0);
  float Kci[9], T[9];
  m3f_inv(Kc.m, Kci);
  m3f_mul(Kd.m, dRc_, T); m3f_mul(T, Kci, dRc_proj);
  for (int i = 0; i < 3; ++i) t_dc_proj[i] = Kd.m[i * 3] * t_dc_[0] + Kd.m[i * 3 + 1] * t_dc_[1] + Kd.m[i * 3 + 2] * t_dc_[2];
  m3f_inv(dRc_proj, cRd_proj);
}

void VisodoTracker::prepareImagesCustomCalibration(const DepthMap& depth_raw, const View& colors_raw) {
  // src/visodo.cpp:775-824: undistort both images, correct the depth-sensor distortion, register depth onto the RGB camera
  if (depthinv_distorted_.rows() != rows_) {  // allocateBuffers :619-624 (only needed on this path)
    depthinv_distorted_.create(rows_, cols_); intensity_distorted_.create(rows_, cols_);
    depthinv_corr_distorted_.create(rows_, cols_); depthinv_preregister_.create(rows_, cols_);
    depthinv_register_trans_.create(3 * rows_, 3 * cols_); depthinv_register_trans_as_int_.create(3 * rows_, 3 * cols_);
  }
  PtrStepSz<uchar3> colors(rows_, cols_, (uchar3*)colors_raw.ptr(), colors_raw.step());
  Intr rgb_intrinsics = intr();
  Intr depth_intrinsics(fxd_, fyd_, cxd_, cyd_, k1d_, k2d_, k3d_, k4d_, k5d_);
  DepthDist depth_spdist(c1_, c0_, q0_[0], q0_[1], q0_[2], q0_[3], q0_[4], q0_[5], q0_[6], q0_[7], q0_[8], q1_[0], q1_[1], q1_[2], q1_[3], q1_[4],
                         q1_[5], q1_[6], q1_[7], q1_[8]);
  computeIntensity(colors, intensity_distorted_);
  decomposeRGBInChannels(colors, r_curr_, g_curr_, b_curr_);
  convertDepth2InvDepth(depth_raw, depthinv_distorted_, factor_depth_);
  undistortIntensity(intensity_distorted_, intensities_curr_[0], rgb_intrinsics);
  undistortDepthInv(depthinv_distorted_, depthinv_corr_distorted_, depthinv_preregister_, depth_intrinsics, depth_spdist);
  float dRc_proj[9], cRd_proj[9], t_dc_proj[3];
  stereoProjections(dRc_proj, t_dc_proj, cRd_proj);
  Mat33 dRc_dev, cRd_dev; float3 t_dev;
  for (int i = 0; i < 3; ++i) {
    dRc_dev.data[i].x = dRc_proj[i * 3]; dRc_dev.data[i].y = dRc_proj[i * 3 + 1]; dRc_dev.data[i].z = dRc_proj[i * 3 + 2];
    cRd_dev.data[i].x = cRd_proj[i * 3]; cRd_dev.data[i].y = cRd_proj[i * 3 + 1]; cRd_dev.data[i].z = cRd_proj[i * 3 + 2];
  }
  t_dev.x = t_dc_proj[0]; t_dev.y = t_dc_proj[1]; t_dev.z = t_dc_proj[2];
  registerDepthinv(depthinv_preregister_, depthinv_register_trans_, depthinv_register_trans_as_int_, depthinvs_curr_[0], dRc_dev, t_dev, cRd_dev);
  for (int i = 1; i < levels_; ++i) {
    pyrDownIntensity(intensities_curr_[i - 1], intensities_curr_[i]);
    pyrDownDepth(depthinvs_curr_[i - 1], depthinvs_curr_[i]);
  }
}

void VisodoTracker::loadSettings(Settings& settings) {
  // src/visodo.cpp:321-435 ([VISODO])
  Section s;
  if (!settings.getSection("VISODO", s)) return;
  Entry e;
  if (s.getEntry("M_ESTIMATOR", e)) {
    std::string v = e.getValue();
    if (v == "Student") Mestimator_ = STUDENT;
    if (v == "LeastSquares") Mestimator_ = LSQ;
    if (v == "Tukey") Mestimator_ = TUKEY;
    if (v == "Huber") Mestimator_ = HUBER;
  }
  if (s.getEntry("MOTION_MODEL", e)) {
    if (e.getValue() == "none") motion_model_ = NO_MM;
    if (e.getValue() == "constVelocity") motion_model_ = CONSTANT_VELOCITY;
  }
  if (s.getEntry("WARP_ORDER", e)) {
    if (e.getValue() == "warpFirst") warping_ = WARP_FIRST;
    if (e.getValue() == "pyrFirst") warping_ = PYR_FIRST;
  }
  if (s.getEntry("IMAGE_FILTERING", e)) {
    if (e.getValue() == "none") image_filtering_ = NO_FILTERS;
    if (e.getValue() == "gradients") image_filtering_ = FILTER_GRADS;
  }
  if (s.getEntry("SIGMA_ESTIMATOR", e)) {
    if (e.getValue() == "sigmaMAD") sigma_estimator_ = SIGMA_MAD;
    if (e.getValue() == "sigmaML") sigma_estimator_ = SIGMA_PDF;
    if (e.getValue() == "sigmaConst") sigma_estimator_ = SIGMA_CONS;
  }
  if (s.getEntry("INTEGRATION_VISRATIO_THRESHOLD", e)) visibility_ratio_integr_threshold_ = (float)atof(e.getValue().c_str());
  if (s.getEntry("ODOMETRY_VISRATIO_THRESHOLD", e)) visibility_ratio_odo_threshold_ = (float)atof(e.getValue().c_str());
  if (s.getEntry("FINEST_PYR_LEVEL", e)) finest_level_ = (int)atof(e.getValue().c_str());
}

void VisodoTracker::start() {
  // src/visodo.cpp:437-445
  std::unique_lock<std::mutex> lock(created_aux_mutex_);
  visodo_thread_.reset(new std::thread([this]() { (*this)(); }));
  created_cond_.wait(lock);
}

bool VisodoTracker::operator()() {
  // src/visodo.cpp:2249-2267
  std::unique_lock<std::mutex> lock(mutex_);
  exit_ = false;
  { std::unique_lock<std::mutex> l2(created_aux_mutex_); created_cond_.notify_one(); }
  while (!exit_) {
    new_frame_cond_.wait(lock);
    if (exit_) break;
    trackNewFrame();
  }
  return true;
}

void VisodoTracker::setRGBIntrinsics(float fx, float fy, float cx, float cy, float k1, float k2, float k3, float k4, float k5) {
  // src/visodo.cpp:448-462
  fx_ = fx; fy_ = fy;
  cx_ = (cx == -1) ? cols_ / 2 - 0.5f : cx;
  cy_ = (cy == -1) ? rows_ / 2 - 0.5f : cy;
  k1_ = k1; k2_ = k2; k3_ = k3; k4_ = k4; k5_ = k5;
  lost_ = false;
}
void VisodoTracker::setDepthIntrinsics(float fxd, float fyd, float cxd, float cyd, float k1d, float k2d, float k3d, float k4d, float k5d, float c0,
                                       float c1, float q00, float q01, float q02, float q03, float q04, float q05, float q06, float q07, float q08,
                                       float q10, float q11, float q12, float q13, float q14, float q15, float q16, float q17, float q18) {
  // src/visodo.cpp:464-505
  fxd_ = fxd; fyd_ = fyd;
  cxd_ = (cxd == -1) ? cols_ / 2 - 0.5f : cxd;
  cyd_ = (cyd == -1) ? rows_ / 2 - 0.5f : cyd;
  k1d_ = k1d; k2d_ = k2d; k3d_ = k3d; k4d_ = k4d; k5d_ = k5d;
  c0_ = c0; c1_ = c1;
  const float a[9] = {q00, q01, q02, q03, q04, q05, q06, q07, q08}, b[9] = {q10, q11, q12, q13, q14, q15, q16, q17, q18};
  for (int i = 0; i < 9; ++i) { q0_[i] = a[i]; q1_[i] = b[i]; }
}
void VisodoTracker::setDepthToRGBExtrinsics(const float dRc[9], const float t_dc[3]) {
  for (int i = 0; i < 9; ++i) dRc_[i] = dRc[i];
  for (int i = 0; i < 3; ++i) t_dc_[i] = t_dc[i];
}
Matrix3f VisodoTracker::getCalibMatrixDepth(int level_index) const {
  // src/visodo.cpp:1904-1919
  int div = 1 << level_index;
  Matrix3f K = {{fxd_ / div, 0.f, cxd_ / div, 0.f, fyd_ / div, cyd_ / div, 0.f, 0.f, 1.f}};
  return K;
}
void VisodoTracker::setSharedCameraPose(const Affine3d& pose) {
  std::lock_guard<std::mutex> lock(mutex_shared_camera_pose_);
  shared_camera_pose_ = pose;
  camera_pose_has_changed_ = true;
}
Affine3d VisodoTracker::getSharedCameraPose() {
  std::lock_guard<std::mutex> lock(mutex_shared_camera_pose_);
  camera_pose_has_changed_ = false;
  return shared_camera_pose_;
}
Affine3d VisodoTracker::getCameraPose(int time) const {
  if (time > (int)rmats_.size() || time < 0) time = (int)rmats_.size() - 1;
  Affine3d a; a.R = rmats_[time]; a.t = tvecs_[time];
  return a;
}
float VisodoTracker::getVisOdoTime(int time) const {
  if (time > (int)vis_odo_times_.size() || time < 0) time = (int)vis_odo_times_.size() - 1;
  return vis_odo_times_[time];
}
int64_t VisodoTracker::getTimestamp(int time) const {
  if (timestamps_.empty()) return 0;
  if (time > (int)timestamps_.size() || time < 0) time = (int)timestamps_.size() - 1;
  return timestamps_[time];
}
Matrix3f VisodoTracker::getCalibMatrix(int level_index) const {
  // src/visodo.cpp:1885-1900
  int div = 1 << level_index;
  Matrix3f K;
  float v[9] = {fx_ / div, 0.f, cx_ / div, 0.f, fy_ / div, cy_ / div, 0.f, 0.f, 1.f};
  for (int i = 0; i < 9; ++i) K.m[i] = v[i];
  return K;
}

void VisodoTracker::getImage(std::vector<PixelRGB>& scene_view, std::vector<float>& intensity_view, std::vector<float>& depthinv_view) {
  // src/visodo.cpp:559-580
  LightSource light;
  light.number = 1;
  light.pos[0].x = (float)last_integrKF_global_translation_[0];
  light.pos[0].y = (float)last_integrKF_global_translation_[1];
  light.pos[0].z = (float)last_integrKF_global_translation_[2];
  View scene_view_dev;
  scene_view_dev.create(rows_, cols_);
  PtrStepSz<uchar3> rgb(rows_, cols_, (uchar3*)colors_integrKF_.ptr(), colors_integrKF_.step());
  PtrStepSz<uchar3> dst(rows_, cols_, (uchar3*)scene_view_dev.ptr(), scene_view_dev.step());
  generateImageRGB(vertices_integrKF_, normals_integrKF_, rgb, light, dst);
  device::sync();
  int c;
  scene_view_dev.download(scene_view, c);
  intensities_curr_[0].download(intensity_view, c);
  depthinv_integrKF_.download(depthinv_view, c);
}

void VisodoTracker::allocateBuffers(int rows, int cols) {
  // src/visodo.cpp:584-691 (only the buffers that live code touches)
  const int L = levels_;
  depthinvs_curr_.resize(L); intensities_curr_.resize(L); depthinvs_odoKF_.resize(L); intensities_odoKF_.resize(L);
  depthinvs_odoKF_filtered_.resize(L); intensities_odoKF_filtered_.resize(L);
  xGradsInt_odoKF_.resize(L); yGradsInt_odoKF_.resize(L); xGradsDepthinv_odoKF_.resize(L); yGradsDepthinv_odoKF_.resize(L);
  xGradsInt_odoKF_covOnly_.resize(L); yGradsInt_odoKF_covOnly_.resize(L); xGradsDepthinv_odoKF_covOnly_.resize(L); yGradsDepthinv_odoKF_covOnly_.resize(L);
  warped_depthinvs_curr_.resize(L); warped_intensities_curr_.resize(L);
  res_intensities_.resize(L); res_depthinvs_.resize(L);
  rgb24_.create(rows, cols); depth_.create(rows, cols);
  warped_weight_curr_.create(rows, cols);
  initialiseDeviceMemory2D<float>(warped_weight_curr_, 0.f);  // the reference leaves it uninitialised; defined as 0 here
  warped_depthinv_integr_curr_.create(rows, cols);
  depthinv_integrKF_.create(rows, cols); weight_integrKF_.create(rows, cols); overlap_mask_integrKF_.create(rows, cols);
  depthinv_integrKF_raw_.create(rows, cols);
  vertices_integrKF_.create(3 * rows, cols); normals_integrKF_.create(3 * rows, cols);
  xGradsDepthinv_integrKF_.create(rows, cols); yGradsDepthinv_integrKF_.create(rows, cols);
  colors_integrKF_.create(rows, cols);
  r_curr_.create(rows, cols); g_curr_.create(rows, cols); b_curr_.create(rows, cols);
  for (int i = 0; i < L; ++i) {
    int pr = rows >> i, pc = cols >> i;
    intensities_curr_[i].create(pr, pc); depthinvs_curr_[i].create(pr, pc);
    intensities_odoKF_[i].create(pr, pc); depthinvs_odoKF_[i].create(pr, pc);
    intensities_odoKF_filtered_[i].create(pr, pc); depthinvs_odoKF_filtered_[i].create(pr, pc);
    xGradsInt_odoKF_[i].create(pr, pc); yGradsInt_odoKF_[i].create(pr, pc);
    xGradsDepthinv_odoKF_[i].create(pr, pc); yGradsDepthinv_odoKF_[i].create(pr, pc);
    xGradsInt_odoKF_covOnly_[i].create(pr, pc); yGradsInt_odoKF_covOnly_[i].create(pr, pc);
    xGradsDepthinv_odoKF_covOnly_[i].create(pr, pc); yGradsDepthinv_odoKF_covOnly_[i].create(pr, pc);
    warped_depthinvs_curr_[i].create(pr, pc); warped_intensities_curr_[i].create(pr, pc);
    res_intensities_[i].create((size_t)pr * pc); res_depthinvs_[i].create((size_t)pr * pc);
  }
}

void VisodoTracker::reset() {
  // src/visodo.cpp:519-553
  global_time_ = 0;
  rmats_.clear(); tvecs_.clear(); vis_odo_times_.clear(); timestamps_.clear();
  odo_rmats_.clear(); odo_tvecs_.clear(); odo_covmats_.clear();
  rmats_.push_back(init_Rcam_); tvecs_.push_back(init_tcam_); vis_odo_times_.push_back(0.f);
  last_estimated_rotation_ = Matrix3ft::Identity();
  last_estimated_translation_ = Vector3ft::Zero();
  velocity_ = Vector3ft::Zero(); omega_ = Vector3ft::Zero();
  lost_ = false;
}

void VisodoTracker::prepareImages(const DepthMap& depth_raw, const View& colors_raw) {
  // src/visodo.cpp:760-773
  PtrStepSz<uchar3> colors(rows_, cols_, (uchar3*)colors_raw.ptr(), colors_raw.step());
  computeIntensity(colors, intensities_curr_[0]);
  decomposeRGBInChannels(colors, r_curr_, g_curr_, b_curr_);
  convertDepth2InvDepth(depth_raw, depthinvs_curr_[0], factor_depth_);
  for (int i = 1; i < levels_; ++i) {
    pyrDownIntensity(intensities_curr_[i - 1], intensities_curr_[i]);
    pyrDownDepth(depthinvs_curr_[i - 1], depthinvs_curr_[i]);
  }
}

void VisodoTracker::saveCurrentImagesAsOdoKeyframes() {
  // src/visodo.cpp:826-878
  const float sigma_int_ref = 3.f, sigma_depthinv_ref = 0.0025f;
  for (int i = 0; i < levels_; ++i) copyImages(depthinvs_curr_[i], intensities_curr_[i], depthinvs_odoKF_[i], intensities_odoKF_[i]);
  bilateralFilter(depthinvs_odoKF_[0], depthinvs_odoKF_filtered_[0], 2.f * sigma_depthinv_ref);
  bilateralFilter(intensities_odoKF_[0], intensities_odoKF_filtered_[0], sigma_int_ref);
  computeGradientIntensity(intensities_odoKF_filtered_[0], xGradsInt_odoKF_covOnly_[0], yGradsInt_odoKF_covOnly_[0]);
  computeGradientDepth(depthinvs_odoKF_filtered_[0], xGradsDepthinv_odoKF_covOnly_[0], yGradsDepthinv_odoKF_covOnly_[0]);
  for (int i = 1; i < levels_; ++i) {
    pyrDownDepth(depthinvs_odoKF_filtered_[i - 1], depthinvs_odoKF_filtered_[i]);
    pyrDownIntensity(intensities_odoKF_filtered_[i - 1], intensities_odoKF_filtered_[i]);
    computeGradientIntensity(intensities_odoKF_filtered_[i], xGradsInt_odoKF_covOnly_[i], yGradsInt_odoKF_covOnly_[i]);
    computeGradientDepth(depthinvs_odoKF_filtered_[i], xGradsDepthinv_odoKF_covOnly_[i], yGradsDepthinv_odoKF_covOnly_[i]);
  }
  for (int i = 0; i < levels_; ++i) {
    if (image_filtering_ == FILTER_GRADS) {
      copyImages(xGradsInt_odoKF_covOnly_[i], yGradsInt_odoKF_covOnly_[i], xGradsInt_odoKF_[i], yGradsInt_odoKF_[i]);
      copyImages(xGradsDepthinv_odoKF_covOnly_[i], yGradsDepthinv_odoKF_covOnly_[i], xGradsDepthinv_odoKF_[i], yGradsDepthinv_odoKF_[i]);
    } else {
      computeGradientIntensity(intensities_odoKF_[i], xGradsInt_odoKF_[i], yGradsInt_odoKF_[i]);
      computeGradientDepth(depthinvs_odoKF_[i], xGradsDepthinv_odoKF_[i], yGradsDepthinv_odoKF_[i]);
    }
  }
}

void VisodoTracker::saveCurrentImagesAsIntegrationKeyframes(const View& colors) {
  // src/visodo.cpp:880-893
  copyImage(depthinvs_curr_[0], depthinv_integrKF_);
  copyImage(depthinvs_curr_[0], depthinv_integrKF_raw_);
  colors.copyTo(colors_integrKF_);
  initialiseWeightKeyframe(depthinvs_curr_[0], weight_integrKF_);
  createVMap(intr()(0), depthinv_integrKF_, vertices_integrKF_);
  computeGradientDepth(depthinv_integrKF_, xGradsDepthinv_integrKF_, yGradsDepthinv_integrKF_);
  createNMapGradients(intr()(0), depthinv_integrKF_, xGradsDepthinv_integrKF_, yGradsDepthinv_integrKF_, normals_integrKF_);
}

void VisodoTracker::warpAtLevel(int level, const Matrix3ft& R, const Vector3ft& t) {
  // src/visodo.cpp:1066-1067, 1108-1126: inverse pose, projected with K(level); the intensity warp samples on the WARPED iD
  Matrix3ft Ri; Vector3ft ti;
  se3::m3_inv(R.m, Ri.m);
  se3::m3_mulv(Ri.m, t.v, ti.v);
  for (int i = 0; i < 3; ++i) ti.v[i] = -ti.v[i];
  Mat33 Rp; float3 tp;
  project(getCalibMatrix(level), Ri, ti, Rp, tp);
  warpInvDepthWithTrafo3D(depthinvs_curr_[level], warped_depthinvs_curr_[level], depthinvs_odoKF_[level], Rp, tp, intr()(level));
  warpIntensityWithTrafo3DInvDepth(intensities_curr_[level], warped_intensities_curr_[level], warped_depthinvs_curr_[level], Rp, tp, intr()(level));
}

bool VisodoTracker::estimateVisualOdometry(Matrix3ft& resulting_rotation, Vector3ft& resulting_translation, Matrix6d& resulting_covariance) {
  // src/visodo.cpp:944-1479
  if (real_time_flag_) visodo_iterations_[0] = 5;  // :961-964
  double A_total[36], b_total[6];
  const float sigma_int_ref = 5.f, sigma_depthinv_ref = 0.0025f;
  float sigma_int = 40.f, sigma_depthinv = 5.f, bias_int = 0.f, bias_depthinv = 0.f, nu_int = 5.f, nu_depthinv = 5.f;
  float chi_test = 1.f, chi_square = 1.f, Ndof = 640.f * 480.f, RMSE = 9999.f, RMSE_prev = 9999.f;
  Matrix3ft cam_rot_incremental_inv, cam_rot_incremental; Vector3ft cam_trans_incremental;
  float3 zero3 = {0.f, 0.f, 0.f};
  Matrix3ft previous_rotation = resulting_rotation; Vector3ft previous_translation = resulting_translation;
  Matrix3ft current_rotation; Vector3ft current_translation;
  if ((global_time_ > 1) && (motion_model_ == CONSTANT_VELOCITY) && (!lost_)) {  // :1016-1027
    double vt[3], wt[3], dR[9], dt[3], tmp[3];
    for (int i = 0; i < 3; ++i) { vt[i] = velocity_[i] * delta_t_; wt[i] = omega_[i] * delta_t_; }
    se3::expmap(wt, vt, dR, dt);
    se3::m3_mulv(previous_rotation.m, dt, tmp);
    for (int i = 0; i < 3; ++i) current_translation[i] = tmp[i] + previous_translation[i];
    se3::m3_mul(previous_rotation.m, dR, current_rotation.m);
  } else { current_rotation = previous_rotation; current_translation = previous_translation; }
  double t_start = now_ms();

  for (int level_index = levels_ - 1; level_index >= finest_level_; --level_index) {
    int iter_num = visodo_iterations_[level_index];
    for (int iter = 0; iter < iter_num; ++iter) {
      if (warping_ == WARP_FIRST) {  // :1078-1105
        warpAtLevel(0, current_rotation, current_translation);
        for (int i = 1; i < level_index + 1; ++i) {
          pyrDownIntensity(warped_intensities_curr_[i - 1], warped_intensities_curr_[i]);
          pyrDownDepth(warped_depthinvs_curr_[i - 1], warped_depthinvs_curr_[i]);
        }
      } else warpAtLevel(level_index, current_rotation, current_translation);
      if ((termination_ == CHI_SQUARED) && (iter != 0)) {  // :1134-1164
        computeErrorGridStride(warped_intensities_curr_[0], intensities_odoKF_[0], res_intensities_[0]);
        computeErrorGridStride(warped_depthinvs_curr_[0], depthinvs_odoKF_[0], res_depthinvs_[0]);
        computeChiSquare(res_intensities_[0], res_depthinvs_[0], sigma_int_ref, sigma_depthinv_ref, Mestimator_, chi_square, chi_test, Ndof);
        RMSE = std::sqrt(chi_square) / std::sqrt(Ndof);
        if (iter != 1) {
          if (RMSE > RMSE_prev) {  // undo the previous increment and end this level
            double d[3], tmp[3];
            for (int i = 0; i < 3; ++i) d[i] = current_translation[i] - cam_trans_incremental[i];
            se3::m3_mulv(cam_rot_incremental_inv.m, d, tmp);
            for (int i = 0; i < 3; ++i) current_translation[i] = tmp[i];
            se3::m3_mul(cam_rot_incremental_inv.m, current_rotation.m, current_rotation.m);
            break;
          }
        }
        RMSE_prev = RMSE;
      }
      sigma_int = 5.f; sigma_depthinv = 0.0025f; bias_int = 0.f; bias_depthinv = 0.f; nu_int = 5.f; nu_depthinv = 5.f;  // :1168-1173
      if (sigma_estimator_ == SIGMA_PDF) {  // :1175-1186
        computeErrorGridStride(warped_intensities_curr_[level_index], intensities_odoKF_[level_index], res_intensities_[level_index], Nsamples_);
        computeErrorGridStride(warped_depthinvs_curr_[level_index], depthinvs_odoKF_[level_index], res_depthinvs_[level_index], Nsamples_);
        computeSigmaAndNuStudent(res_intensities_[level_index], bias_int, sigma_int, nu_int, Mestimator_);
        computeSigmaAndNuStudent(res_depthinvs_[level_index], bias_depthinv, sigma_depthinv, nu_depthinv, Mestimator_);
        nu_int = std::max(nu_int, nu_depthinv);
      } else if (sigma_estimator_ == SIGMA_CONS) {
        sigma_int = (float)std::exp(std::log((double)sigma_int_ref));
        sigma_depthinv = (float)std::exp(std::log((double)sigma_depthinv_ref));
      }
      buildSystemStudentNuGridStride(zero3, zero3, depthinvs_odoKF_[level_index], intensities_odoKF_[level_index],
                                     xGradsDepthinv_odoKF_[level_index], yGradsDepthinv_odoKF_[level_index], xGradsInt_odoKF_[level_index],
                                     yGradsInt_odoKF_[level_index], warped_depthinvs_curr_[level_index], warped_intensities_curr_[level_index],
                                     Mestimator_, weighting_, sigma_depthinv, sigma_int, bias_depthinv, bias_int, nu_depthinv, nu_int,
                                     intr()(level_index), B_SIZE, gbuf_, sumbuf_, A_total, b_total);
      double x[6];
      se3::llt_solve6(A_total, b_total, x);  // A.llt().solve(b) :1249
      // :1252-1263
      se3::expmap_rot(x + 3, cam_rot_incremental_inv.m);
      se3::m3_inv(cam_rot_incremental_inv.m, cam_rot_incremental.m);
      se3::m3_mulv(cam_rot_incremental.m, x, cam_trans_incremental.v);
      for (int i = 0; i < 3; ++i) cam_trans_incremental[i] = -cam_trans_incremental[i];
      double tmp[3];
      se3::m3_mulv(cam_rot_incremental.m, current_translation.v, tmp);
      for (int i = 0; i < 3; ++i) current_translation[i] = tmp[i] + cam_trans_incremental[i];
      se3::m3_mul(cam_rot_incremental.m, current_rotation.m, current_rotation.m);
      if (se3::has_nan(current_rotation.m, current_translation.v)) {  // :1265-1274
        resulting_translation = previous_translation;
        resulting_rotation = previous_rotation;
        resulting_covariance = zero6();
        for (int i = 0; i < 6; ++i) resulting_covariance[i * 7] = 100.0;
        vis_odo_times_.push_back((float)(now_ms() - t_start));
        return false;
      }
    }
  }
  last_info_.sigma_int = sigma_int; last_info_.sigma_depthinv = sigma_depthinv; last_info_.nu_int = nu_int; last_info_.nu_depthinv = nu_depthinv;
  {
    // covariance pass :1283-1417
    int fl = finest_level_;
    warpAtLevel(fl, current_rotation, current_translation);
    sigma_int = (float)std::exp(std::log((double)sigma_int_ref));
    sigma_depthinv = (float)std::exp(std::log((double)sigma_depthinv_ref));
    buildSystemGridStride(zero3, zero3, depthinvs_odoKF_[fl], intensities_odoKF_[fl], xGradsDepthinv_odoKF_covOnly_[fl],
                          yGradsDepthinv_odoKF_covOnly_[fl], xGradsInt_odoKF_covOnly_[fl], yGradsInt_odoKF_covOnly_[fl],
                          warped_depthinvs_curr_[fl], warped_intensities_curr_[fl], STUDENT, weighting_, sigma_depthinv, sigma_int, 0.f, 0.f,
                          intr()(fl), B_SIZE, gbuf_, sumbuf_, A_total, b_total);
    resulting_rotation = current_rotation;
    resulting_translation = current_translation;
    se3::inverse6(A_total, resulting_covariance.data());
    // :1411-1415 -- full-resolution residuals + chi-square; the reference only writes locals with the result
    computeErrorGridStride(warped_intensities_curr_[fl], intensities_odoKF_[fl], res_intensities_[fl]);
    computeErrorGridStride(warped_depthinvs_curr_[fl], depthinvs_odoKF_[fl], res_depthinvs_[fl]);
    computeChiSquare(res_intensities_[fl], res_depthinvs_[fl], sigma_int_ref, sigma_depthinv_ref, Mestimator_, chi_square, chi_test, Ndof);
  }
  {
    // :1459-1468
    double pT[9], dR[9], d[3], dt[3], twist[6];
    se3::m3_T(previous_rotation.m, pT);
    se3::m3_mul(pT, current_rotation.m, dR);
    for (int i = 0; i < 3; ++i) d[i] = current_translation[i] - previous_translation[i];
    se3::m3_mulv(pT, d, dt);
    se3::logmap(dR, dt, twist);
    float inv_dt = 1.f / delta_t_;
    for (int i = 0; i < 3; ++i) { velocity_[i] = twist[i] * (double)inv_dt; omega_[i] = twist[3 + i] * (double)inv_dt; }
  }
  return true;
}

float VisodoTracker::computeCovisibility(const Matrix3ft& R_AtoB, const Vector3ft& t_AtoB, const DepthMapf& depthinvA, const DepthMapf& depthinvB) {
  // src/visodo.cpp:1481-1514
  const float geom_tol = 0.05f / (2.f * 2.f);
  float visibility_ratio_AtoB = 1.f, visibility_ratio_BtoA = 1.f;
  Matrix3f K = getCalibMatrix(0);
  Mat33 Rab, Rba; float3 tab, tba, dummy;
  project(K, R_AtoB, t_AtoB, Rab, tab);
  Matrix3ft Ri; Vector3ft zero = Vector3ft::Zero();
  se3::m3_inv(R_AtoB.m, Ri.m);
  project(K, Ri, zero, Rba, dummy);
  {
    // translation_BtoA_f = -K*Rinv.cast<float>()*t.cast<float>() in float
    float Rf[9], T[9], tf[3] = {(float)t_AtoB[0], (float)t_AtoB[1], (float)t_AtoB[2]}, out[3];
    for (int i = 0; i < 9; ++i) Rf[i] = (float)Ri.m[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i * 3 + j] = -K.m[i * 3] * Rf[j] + -K.m[i * 3 + 1] * Rf[3 + j] + -K.m[i * 3 + 2] * Rf[6 + j];
    for (int i = 0; i < 3; ++i) out[i] = T[i * 3] * tf[0] + T[i * 3 + 1] * tf[1] + T[i * 3 + 2] * tf[2];
    tba.x = out[0]; tba.y = out[1]; tba.z = out[2];
  }
  getVisibilityRatio(depthinvB, depthinvA, Rab, tab, intr()(0), visibility_ratio_BtoA, geom_tol);
  getVisibilityRatio(depthinvA, depthinvB, Rba, tba, intr()(0), visibility_ratio_AtoB, geom_tol);
  return std::min(visibility_ratio_AtoB, visibility_ratio_BtoA);
}

float VisodoTracker::computeOverlapping(const Matrix3ft& R_AtoB, const Vector3ft& t_AtoB, const DepthMapf& depthinvA, const DepthMapf& depthinvB,
                                        BinaryMap& overlap_maskB) {
  // src/visodo.cpp:1517-1539
  float ratio = 1.f;
  Mat33 Rab; float3 tab;
  project(getCalibMatrix(0), R_AtoB, t_AtoB, Rab, tab);
  getVisibilityRatioWithOverlapMask(depthinvB, depthinvA, Rab, tab, intr()(0), ratio, 0.0125f, overlap_maskB);
  return ratio;
}

static void propagate_next(Matrix3ft& next_R, Vector3ft& next_t, Matrix6d& next_cov, const Matrix3ft& dR, const Vector3ft& dt, const Matrix6d& dcov) {
  // shared head of resetOdometryKeyframe / resetIntegrationKeyframe (src/visodo.cpp:1553-1567, 1594-1607)
  double J[36], tn[3], S[9];
  se3::m6_zero(J);
  se3::m6_set_block(J, 0, 0, next_R.m, 1.0);
  se3::m6_set_block(J, 3, 3, next_R.m, 1.0);
  se3::m3_mulv(next_R.m, dt.v, tn);
  se3::skew(tn, S);
  se3::m6_set_block(J, 0, 3, S, 1.0);
  se3::m6_JCJt_add(J, dcov.data(), next_cov.data());
  for (int i = 0; i < 3; ++i) next_t[i] = tn[i] + next_t[i];
  se3::m3_mul(next_R.m, dR.m, next_R.m);
}

void VisodoTracker::resetOdometryKeyframe() {
  // src/visodo.cpp:1541-1575
  odoKF_count_ = 0;
  propagate_next(delta_rotation_odo2integr_next_, delta_translation_odo2integr_next_, delta_covariance_odo2integr_next_, delta_rotation_,
                 delta_translation_, delta_covariance_);
  last_odoKF_index_ = global_time_;
  last_odoKF_global_rotation_ = last_estimated_rotation_;
  last_odoKF_global_translation_ = last_estimated_translation_;
  delta_rotation_ = Matrix3ft::Identity();
  delta_translation_ = Vector3ft::Zero();
  delta_covariance_ = zero6();
}

void VisodoTracker::resetIntegrationKeyframe() {
  // src/visodo.cpp:1577-1672
  integrKF_count_ = 0;
  rmatsKF_.push_back(last_integrKF_global_rotation_);
  tvecsKF_.push_back(last_integrKF_global_translation_);
  propagate_next(delta_rotation_odo2integr_next_, delta_translation_odo2integr_next_, delta_covariance_odo2integr_next_, delta_rotation_,
                 delta_translation_, delta_covariance_);
  // T{k-1,k} = inv(T{odo,k-1})*T{odo,k} and its covariance (:1612-1629)
  Matrix3ft lastT, delta_rotation_kf; Vector3ft delta_translation_kf;
  se3::m3_T(delta_rotation_odo2integr_last_.m, lastT.m);
  se3::m3_mul(lastT.m, delta_rotation_odo2integr_next_.m, delta_rotation_kf.m);
  double d[3];
  for (int i = 0; i < 3; ++i) d[i] = delta_translation_odo2integr_next_[i] - delta_translation_odo2integr_last_[i];
  se3::m3_mulv(lastT.m, d, delta_translation_kf.v);
  double Jn[36], Jl[36], S[9], SR[9];
  se3::m6_zero(Jn); se3::m6_set_block(Jn, 0, 0, lastT.m, 1.0); se3::m6_set_block(Jn, 3, 3, lastT.m, 1.0);
  se3::m6_zero(Jl); se3::m6_set_block(Jl, 0, 0, lastT.m, -1.0); se3::m6_set_block(Jl, 3, 3, lastT.m, -1.0);
  se3::skew(delta_translation_kf.v, S); se3::m3_mul(S, lastT.m, SR); se3::m6_set_block(Jl, 0, 3, SR, 1.0);
  Matrix6d delta_covariance_kf = zero6();
  se3::m6_JCJt_add(Jl, delta_covariance_odo2integr_last_.data(), delta_covariance_kf.data());
  se3::m6_JCJt_add(Jn, delta_covariance_odo2integr_next_.data(), delta_covariance_kf.data());
  {
    // keyframe export record :1631-1652 (4 D2H downloads)
    std::shared_ptr<KeyframeRecord> kf(new KeyframeRecord());
    kf->K = getCalibMatrix(0);
    kf->kd[0] = k1_; kf->kd[1] = k2_; kf->kd[2] = k3_; kf->kd[3] = k4_; kf->kd[4] = k5_;
    kf->rotation = last_integrKF_global_rotation_; kf->translation = last_integrKF_global_translation_;
    kf->rotation_rel = delta_rotation_kf; kf->translation_rel = delta_translation_kf;
    kf->id = last_integrKF_index_; kf->cols = cols_; kf->rows = rows_;
    int c;
    overlap_mask_integrKF_.download(kf->overlap_mask_, c);
    colors_integrKF_.download(kf->colors_, c);
    depthinv_integrKF_.download(kf->depthinv_, c);
    normals_integrKF_.download(kf->normals_, c);
    if (keyframe_manager_ptr_->tryPushKeyframe(kf)) {
      PoseConstraint kf_constr;
      kf_constr.ini_id_ = last_integrKF_index_; kf_constr.end_id_ = global_time_; kf_constr.type_ = PoseConstraint::SEQ_KF;
      kf_constr.rotation_ = delta_rotation_kf; kf_constr.translation_ = delta_translation_kf; kf_constr.scale_ = 1.f;
      kf_constr.covariance_ = delta_covariance_kf;
      keyframe_manager_ptr_->pushConstraint(kf_constr);
    }
  }
  kf_times_.push_back(1000.f * kf_time_accum_);
  kf_time_accum_ = 0.f;
  last_integrKF_index_ = global_time_;
  last_integrKF_global_rotation_ = last_estimated_rotation_;
  last_integrKF_global_translation_ = last_estimated_translation_;
  delta_rotation_odo2integr_last_ = delta_rotation_;
  delta_translation_odo2integr_last_ = delta_translation_;
  delta_covariance_odo2integr_last_ = delta_covariance_;
  delta_rotation_odo2integr_next_ = Matrix3ft::Identity();
  delta_translation_odo2integr_next_ = Vector3ft::Zero();
  delta_covariance_odo2integr_next_ = zero6();
}

void VisodoTracker::integrateImagesIntoKeyframes(DepthMapf& depthinv_src, Matrix3ft dR, Vector3ft dt) {
  // src/visodo.cpp:1674-1764: K R K^-1 in DOUBLE, inverted, then cast to float
  double K[9] = {(double)fx_, 0, (double)cx_, 0, (double)fy_, (double)cy_, 0, 0, 1}, Ki[9], T[9], Rp[9], tp[3], Rpi[9], tpi[3];
  se3::m3_inv(K, Ki);
  se3::m3_mul(K, dR.m, T); se3::m3_mul(T, Ki, Rp);
  se3::m3_mulv(K, dt.v, tp);
  se3::m3_inv(Rp, Rpi);
  se3::m3_mulv(Rpi, tp, tpi);
  Mat33 Rd; float3 td;
  for (int i = 0; i < 3; ++i) { Rd.data[i].x = (float)Rpi[i * 3]; Rd.data[i].y = (float)Rpi[i * 3 + 1]; Rd.data[i].z = (float)Rpi[i * 3 + 2]; }
  td.x = (float)(-tpi[0]); td.y = (float)(-tpi[1]); td.z = (float)(-tpi[2]);
  warpInvDepthWithTrafo3DWeighted(depthinv_src, warped_depthinv_integr_curr_, depthinv_integrKF_, warped_weight_curr_, Rd, td, intr()(0));
  integrateWarpedFrame(warped_depthinv_integr_curr_, warped_weight_curr_, depthinv_integrKF_, weight_integrKF_);
  createVMap(intr()(0), depthinv_integrKF_, vertices_integrKF_);
  computeGradientDepth(depthinv_integrKF_, xGradsDepthinv_integrKF_, yGradsDepthinv_integrKF_);
  createNMapGradients(intr()(0), depthinv_integrKF_, xGradsDepthinv_integrKF_, yGradsDepthinv_integrKF_, normals_integrKF_);
}

float VisodoTracker::computeInterframeTime() {
  // src/visodo.cpp:1929-1964
  float dt = 0.03333f;
  if (!compute_deltat_flag_) { timestamp_rgb_curr_ = 0; timestamp_depth_curr_ = 0; return dt; }
  if (global_time_ == 0) timestamp_ini_ = (timestamp_rgb_curr_ <= timestamp_depth_curr_) ? timestamp_rgb_curr_ : timestamp_depth_curr_;
  uint64_t timestamp_rgb_curr_zeroed = timestamp_rgb_curr_ - timestamp_ini_;
  timestamps_.push_back((int64_t)timestamp_rgb_curr_zeroed);
  if (global_time_ > 0) dt = (float)(1e-9 * (double)(timestamp_rgb_curr_zeroed - (uint64_t)timestamps_[global_time_ - 1]));
  return dt;
}

// ---- engine-backed mode: the same trackNewFrame, driven through a one-lane device-resident engine (include/rgbid_engine.h) ---------------------------
// What the one-lane engine cannot take over from the host-driven loop; nullptr = nothing (round 5: CHI_SQUARED termination and custom calibration are
// engine configurations now)
const char* VisodoTracker::engineObstacle() const {
  const bool identity_start = std::memcmp(init_Rcam_.m, Matrix3ft::Identity().m, sizeof(init_Rcam_.m)) == 0 && init_tcam_[0] == 0.0 && init_tcam_[1] == 0.0 && init_tcam_[2] == 0.0;
  if (!identity_start) return "a non-identity initial camera pose";
  if (levels_ > 8) return "more than 8 pyramid levels";
  if (std::getenv("RGBID_VISODO_HOST_DRIVEN")) return "RGBID_VISODO_HOST_DRIVEN is set";
  return nullptr;
}

bool VisodoTracker::setEngineBacked(bool on) {
  if (global_time_ != 0) return false;                                  // before the first frame (or after reset())
  if (!on) { engine_backed_ = false; engine_auto_ = false; return true; }
  if (engineObstacle()) return false;                                   // stays as it was
  engine_backed_ = true; engine_auto_ = false;
  return true;
}

bool VisodoTracker::createEngine() {
  if (engine_) { rgbid_engine_destroy(engine_); engine_ = nullptr; }
  // a context of the tracker's own: the per-thread default context (containers.hpp) is destroyed when its thread ends, which may be before this object
  if (!engine_ctx_ && rgbid_ctx_create(&engine_ctx_, pcl::gpu::current_device().load(), nullptr) != RGBID_OK) return false;
  if (rgbid_engine_config_size() != sizeof(rgbid_engine_config)) {   // librgbid_hip.so built from another revision of rgbid_engine.h
    std::cerr << "VisodoTracker: librgbid_hip.so and librgbid_host.so disagree on rgbid_engine_config (" << rgbid_engine_config_size() << " vs " << sizeof(rgbid_engine_config)
              << " bytes): rebuild both" << std::endl;
    return false;
  }
  rgbid_engine_config c;
  rgbid_engine_default_config(&c);
  if (real_time_flag_) visodo_iterations_[0] = 5;   // :961-964 (estimateVisualOdometry does this on every frame; the engine's schedule is fixed at its creation)
  c.rows = rows_; c.cols = cols_; c.levels = levels_; c.lanes = 1;
  for (int i = 0; i < 8; ++i) c.iters[i] = i < levels_ ? visodo_iterations_[i] : 0;
  c.mestimator = Mestimator_; c.motion_model = motion_model_; c.sigma_estimator = sigma_estimator_; c.weighting = weighting_;
  c.max_odoKF_count = max_odoKF_count_; c.finest_level = finest_level_; c.image_filtering = image_filtering_;
  c.visratio_odo = visibility_ratio_odo_threshold_; c.visratio_integr = visibility_ratio_integr_threshold_;
  c.max_integrKF_count = max_integrKF_count_; c.nsamples = Nsamples_;
  c.fx = fx_; c.fy = fy_; c.cx = cx_; c.cy = cy_; c.factor_depth = factor_depth_;
  c.interp_mode = interp_mode_;
  c.delta_t = 0.03333f;
  c.use_graph = 0;            // eager launches read depth_ / rgb24_ in place (graph replay would add two staging copies and measured the same 1.04 ms per frame)
  c.fused_gn = 1;
  c.fast_numerics = 0;        // the compat tracker's numerics class: the IEEE evaluation of the oracle, bit for bit
  c.chi_square_stats = 0;
  c.preview = preview_ ? 1 : 0;
  c.record_capacity = 2;
  c.warping = warping_ == device::WARP_FIRST ? RGBID_WARP_FIRST : RGBID_PYR_FIRST;
  c.keyframe_capacity = 2;    // what resetIntegrationKeyframe hands to the back-end is read out right after the step that exported it
  c.termination = termination_ == device::CHI_SQUARED ? RGBID_CHI_SQUARED : RGBID_ALL_ITERS;
  c.custom_registration = custom_registration_ ? 1 : 0;
  if (custom_registration_) {   // prepareImagesCustomCalibration's constants (:775-801), formed exactly as the host-driven path forms them
    const float kc[5] = {k1_, k2_, k3_, k4_, k5_};
    for (int i = 0; i < 5; ++i) c.rgb_dist[i] = kc[i];
    c.depth_intr = rgbid_intr_k{fxd_, fyd_, cxd_, cyd_, k1d_, k2d_, k3d_, k4d_, k5d_};
    const DepthDist dd(c1_, c0_, q0_[0], q0_[1], q0_[2], q0_[3], q0_[4], q0_[5], q0_[6], q0_[7], q0_[8], q1_[0], q1_[1], q1_[2], q1_[3], q1_[4], q1_[5], q1_[6], q1_[7], q1_[8]);
    c.depth_dist = device::c_depth_dist(dd);   // the same conversion the bridge call undistortDepthInv applies (default pixel shifts included)
    float t_dc_proj[3];
    stereoProjections(c.dRc_proj, t_dc_proj, c.cRd_proj);
    for (int i = 0; i < 3; ++i) c.t_dc_proj[i] = t_dc_proj[i];
  }
  return rgbid_engine_create(&engine_, engine_ctx_, &c) == RGBID_OK;
}

void VisodoTracker::downloadKeyframeMaps(float* depthinv_host, float* weight_host) const {
  if (engine_backed_ && engine_) {
    rgbid_img d, w;
    rgbidSafeCall(rgbid_engine_keyframe_maps(engine_, 0, &d, &w, nullptr, nullptr, nullptr));
    rgbidSafeCall(rgbid_ctx_sync(engine_ctx_));
    if (depthinv_host) rgbidSafeCall(rgbid_memcpy2d_d2h(engine_ctx_, depthinv_host, (size_t)cols_ * 4, d.data, d.step, (size_t)cols_ * 4, rows_));
    if (weight_host) rgbidSafeCall(rgbid_memcpy2d_d2h(engine_ctx_, weight_host, (size_t)cols_ * 4, w.data, w.step, (size_t)cols_ * 4, rows_));
    return;
  }
  if (depthinv_host) depthinv_integrKF_.download(depthinv_host, (size_t)cols_ * 4);
  if (weight_host) weight_integrKF_.download(weight_host, (size_t)cols_ * 4);
}

void VisodoTracker::downloadCurrentMaps(float* depthinv_host, float* intensity_host) const {
  if (engine_backed_ && engine_) {
    rgbid_img d, i;
    rgbidSafeCall(rgbid_engine_current_maps(engine_, 0, &d, &i));
    rgbidSafeCall(rgbid_ctx_sync(engine_ctx_));
    if (depthinv_host) rgbidSafeCall(rgbid_memcpy2d_d2h(engine_ctx_, depthinv_host, (size_t)cols_ * 4, d.data, d.step, (size_t)cols_ * 4, rows_));
    if (intensity_host) rgbidSafeCall(rgbid_memcpy2d_d2h(engine_ctx_, intensity_host, (size_t)cols_ * 4, i.data, i.step, (size_t)cols_ * 4, rows_));
    return;
  }
  if (depthinv_host) depthinvs_curr_[0].download(depthinv_host, (size_t)cols_ * 4);
  if (intensity_host) intensities_curr_[0].download(intensity_host, (size_t)cols_ * 4);
}

bool VisodoTracker::trackNewFrameEngine() {
  // src/visodo.cpp:1967-2247 with the device work of the frame done by ONE engine step; the host keeps the bookkeeping the application and the
  // back-end see (rmats_ / tvecs_, odo_*, Pose / PoseConstraint / Keyframe pushes, shared pose, timing)
  delta_t_ = computeInterframeTime();
  kf_time_accum_ += delta_t_;
  const double t1 = now_ms();
  last_info_ = LastFrameInfo();
  TrackerSink* sink = keyframe_manager_ptr_ ? keyframe_manager_ptr_ : &null_sink_;
  keyframe_manager_ptr_ = sink;
  if (!engine_ || global_time_ == 0) {
    if (!createEngine()) { std::cerr << "VisodoTracker: the engine-backed mode could not create its engine" << std::endl; std::exit(0); }   // as pcl::gpu::error()
  }
  rgbidSafeCall(rgbid_engine_set_delta_t(engine_, delta_t_));
  rgbidSafeCall(rgbid_engine_step_strided(engine_, depth_.ptr(), depth_.step(), 0, rgb24_.ptr(), rgb24_.step(), 0));
  rgbid_pose_record rec;
  rgbidSafeCall(rgbid_engine_read_records(engine_, rgbid_engine_steps(engine_) - 1, 1, &rec));   // synchronises
  auto M3 = [](const double* p) { Matrix3ft m; std::memcpy(m.m, p, sizeof(m.m)); return m; };
  auto V3 = [](const double* p) { Vector3ft v; std::memcpy(v.v, p, sizeof(v.v)); return v; };
  auto M6 = [](const double* p) { Matrix6d m; std::memcpy(m.data(), p, sizeof(double) * 36); return m; };
  // what resetIntegrationKeyframe hands over (:1631-1652), exported by the engine inside the step
  auto push_exported_keyframe = [&]() {
    int count = 0;
    rgbidSafeCall(rgbid_engine_keyframe_counts(engine_, &count));
    rgbid_keyframe_header h;
    std::shared_ptr<KeyframeRecord> kf(new KeyframeRecord());
    kf->overlap_mask_.resize((size_t)rows_ * cols_); kf->colors_.resize((size_t)rows_ * cols_); kf->depthinv_.resize((size_t)rows_ * cols_);
    kf->normals_.resize((size_t)3 * rows_ * cols_);
    rgbidSafeCall(rgbid_engine_read_keyframe(engine_, 0, count - 1, &h, kf->overlap_mask_.data(), reinterpret_cast<unsigned char*>(kf->colors_.data()), kf->depthinv_.data(),
                                             kf->normals_.data()));
    rmatsKF_.push_back(M3(h.R)); tvecsKF_.push_back(V3(h.t));
    kf->K = getCalibMatrix(0);
    kf->kd[0] = k1_; kf->kd[1] = k2_; kf->kd[2] = k3_; kf->kd[3] = k4_; kf->kd[4] = k5_;
    kf->rotation = M3(h.R); kf->translation = V3(h.t);
    kf->rotation_rel = M3(h.R_rel); kf->translation_rel = V3(h.t_rel);
    kf->id = h.id; kf->cols = cols_; kf->rows = rows_;
    if (sink->tryPushKeyframe(kf)) {
      PoseConstraint kc;
      kc.ini_id_ = h.id; kc.end_id_ = h.end_id; kc.type_ = PoseConstraint::SEQ_KF;
      kc.rotation_ = kf->rotation_rel; kc.translation_ = kf->translation_rel; kc.scale_ = 1.f; kc.covariance_ = M6(h.cov_rel);
      sink->pushConstraint(kc);
    }
    kf_times_.push_back(1000.f * kf_time_accum_);
    kf_time_accum_ = 0.f;
    last_integrKF_index_ = h.end_id;
  };
  if (rec.status & RGBID_ST_FIRST) {   // :1994-2045
    ++global_time_;
    odo_rmats_.push_back(Matrix3ft::Identity()); odo_tvecs_.push_back(Vector3ft::Zero()); odo_covmats_.push_back(zero6());
    delta_rotation_ = Matrix3ft::Identity(); delta_translation_ = Vector3ft::Zero(); delta_covariance_ = zero6();
    kf_time_accum_ = 0.f;
    last_integrKF_index_ = 0;
    Pose pose_new; pose_new.id_ = 0; pose_new.rotation_ = Matrix3ft::Identity(); pose_new.translation_ = Vector3ft::Zero(); pose_new.scale_ = 1.f;
    sink->pushPose(pose_new);
    sink_back_pose_ = pose_new;
    setSharedCameraPose(pose_new.getAffine());
    last_info_.odo_kf_switched = last_info_.integr_kf_switched = true;
    lost_ = false;
    return false;
  }
  const bool was_lost = lost_;
  const bool tracked = (rec.status & RGBID_ST_TRACKED) != 0;
  odometry_success_ = tracked;
  if (!tracked) {
    if (was_lost) return false;                                     // still lost: keyframes re-seeded from this frame, nothing recorded (:2110-2116)
    last_info_.odo_kf_switched = last_info_.integr_kf_switched = true;
    // this frame lost tracking (:2066-2085)
    lost_ = true;
    delta_rotation_ = M3(rec.kf_R); delta_translation_ = V3(rec.kf_t); delta_covariance_ = M6(rec.kf_cov);
    last_estimated_rotation_ = M3(rec.R); last_estimated_translation_ = V3(rec.t);
    rmats_.push_back(last_estimated_rotation_); tvecs_.push_back(last_estimated_translation_);
    PoseConstraint dummy; dummy.ini_id_ = global_time_ - 1; dummy.end_id_ = global_time_; dummy.type_ = PoseConstraint::SEQ_ODO;
    dummy.rotation_ = Matrix3ft::Identity(); dummy.translation_ = Vector3ft::Zero(); dummy.scale_ = 1.f; dummy.covariance_ = zero6();
    for (int i = 0; i < 6; ++i) dummy.covariance_[i * 7] = 100.0;
    sink->pushConstraint(dummy);
    sink->backPose(sink_back_pose_);
    Pose p = sink_back_pose_; p.id_ = global_time_; p.scale_ = 1.f;
    sink->pushPose(p);
    sink_back_pose_ = p;
    if (rec.status & RGBID_ST_KF_EXPORTED) push_exported_keyframe();
    ++global_time_;
    if (verbose_) std::cout << "I am LOST!!!" << std::endl;
    return false;
  }
  lost_ = false;
  last_info_.sigma_int = rec.sigma_int; last_info_.sigma_depthinv = rec.sigma_depthinv; last_info_.nu_int = rec.nu_int; last_info_.nu_depthinv = rec.nu_depthinv;
  delta_rotation_ = M3(rec.kf_R); delta_translation_ = V3(rec.kf_t); delta_covariance_ = M6(rec.kf_cov);
  last_estimated_rotation_ = M3(rec.R); last_estimated_translation_ = V3(rec.t);
  rmats_.push_back(last_estimated_rotation_); tvecs_.push_back(last_estimated_translation_);
  {
    // sequential constraint + pose :2128-2165
    const Matrix3ft Rseq = M3(rec.odo_R); const Vector3ft tseq = V3(rec.odo_t); const Matrix6d cseq = M6(rec.odo_cov);
    odo_rmats_.push_back(Rseq); odo_tvecs_.push_back(tseq); odo_covmats_.push_back(cseq);
    PoseConstraint c; c.ini_id_ = global_time_ - 1; c.end_id_ = global_time_; c.type_ = PoseConstraint::SEQ_ODO;
    c.rotation_ = Rseq; c.translation_ = tseq; c.scale_ = 1.f; c.covariance_ = cseq;
    sink->pushConstraint(c);
    sink->backPose(sink_back_pose_);
    Pose p; p.id_ = global_time_; p.scale_ = 1.f;
    se3::m3_mul(sink_back_pose_.rotation_.m, Rseq.m, p.rotation_.m);
    double tb[3];
    se3::m3_mulv(sink_back_pose_.rotation_.m, tseq.v, tb);
    for (int i = 0; i < 3; ++i) p.translation_[i] = sink_back_pose_.translation_[i] + tb[i];
    sink->pushPose(p);
    sink_back_pose_ = p;
    setSharedCameraPose(p.getAffine());
  }
  last_info_.visratio_odo = rec.vis_odo; last_info_.visratio_integr = rec.vis_integr;
  if (rec.status & RGBID_ST_ODO_KF) {   // resetOdometryKeyframe :1541-1575
    last_info_.odo_kf_switched = true;
    delta_rotation_ = Matrix3ft::Identity(); delta_translation_ = Vector3ft::Zero(); delta_covariance_ = zero6();
  }
  if (rec.status & RGBID_ST_INTEGR_KF) {
    if (rec.status & RGBID_ST_KF_EXPORTED) push_exported_keyframe();
    newKF_ = true;
    last_info_.integr_kf_switched = true;
  }
  vis_odo_times_.push_back((float)(now_ms() - t1));
  if (preview_) {
    std::lock_guard<std::mutex> lock(mutex_scene_view_);
    rgbid_img pv;
    scene_view_.resize((size_t)rows_ * cols_); intensity_view_.resize((size_t)rows_ * cols_); depthinv_view_.resize((size_t)rows_ * cols_);
    rgbidSafeCall(rgbid_engine_preview(engine_, 0, &pv, nullptr));
    rgbidSafeCall(rgbid_memcpy2d_d2h(engine_ctx_, scene_view_.data(), (size_t)cols_ * 3, pv.data, pv.step, (size_t)cols_ * 3, rows_));
    downloadCurrentMaps(nullptr, intensity_view_.data());     // getImage :559-580: the current intensity next to the keyframe's inverse depth
    downloadKeyframeMaps(depthinv_view_.data(), nullptr);
    scene_view_has_changed_ = true;
  }
  ++global_time_;
  return true;
}

bool VisodoTracker::trackNewFrame() {
  // src/visodo.cpp:1967-2247
  if (global_time_ == 0 && engine_backed_) {
    // the mode of a run is settled at its first frame, with the configuration as it is NOW (settings / calibration files may have been loaded after
    // setEngineBacked): what the engine cannot take over runs host-driven -- same results, ~330 bridge calls per frame instead of one launch sequence
    if (const char* why = engineObstacle()) {
      std::cerr << "VisodoTracker: host-driven frame loop (" << why << "; the device-resident engine is used otherwise)" << std::endl;
      engine_backed_ = false;
    }
  }
  if (engine_backed_) return trackNewFrameEngine();
  pcl::gpu::ScopedAsyncBridge bridge_scope(async_bridge_);
  delta_t_ = computeInterframeTime();
  kf_time_accum_ += delta_t_;
  double t1 = now_ms();
  last_info_ = LastFrameInfo();
  if (custom_registration_) prepareImagesCustomCalibration(depth_, rgb24_);  // :1982-1989
  else prepareImages(depth_, rgb24_);
  device::sync();
  TrackerSink* sink = keyframe_manager_ptr_ ? keyframe_manager_ptr_ : &null_sink_;
  keyframe_manager_ptr_ = sink;
  if (global_time_ == 0) {  // :1994-2045
    ++global_time_;
    odoKF_count_ = 0; last_odoKF_index_ = 0;
    last_odoKF_global_rotation_ = rmats_[0]; last_odoKF_global_translation_ = tvecs_[0];
    integrKF_count_ = 0; last_integrKF_index_ = 0;
    last_integrKF_global_rotation_ = Matrix3ft::Identity(); last_integrKF_global_translation_ = Vector3ft::Zero();
    delta_rotation_ = Matrix3ft::Identity(); delta_translation_ = Vector3ft::Zero(); delta_covariance_ = zero6();
    odo_rmats_.push_back(delta_rotation_); odo_tvecs_.push_back(delta_translation_); odo_covmats_.push_back(delta_covariance_);
    delta_rotation_odo2integr_last_ = Matrix3ft::Identity(); delta_translation_odo2integr_last_ = Vector3ft::Zero(); delta_covariance_odo2integr_last_ = zero6();
    delta_rotation_odo2integr_next_ = Matrix3ft::Identity(); delta_translation_odo2integr_next_ = Vector3ft::Zero(); delta_covariance_odo2integr_next_ = zero6();
    saveCurrentImagesAsOdoKeyframes();
    saveCurrentImagesAsIntegrationKeyframes(rgb24_);
    initialiseDeviceMemory2D<unsigned char>(overlap_mask_integrKF_, 0);
    kf_time_accum_ = 0.f;
    Pose pose_new; pose_new.id_ = 0; pose_new.rotation_ = Matrix3ft::Identity(); pose_new.translation_ = Vector3ft::Zero(); pose_new.scale_ = 1.f;
    sink->pushPose(pose_new);
    sink_back_pose_ = pose_new;
    setSharedCameraPose(pose_new.getAffine());
    last_info_.odo_kf_switched = last_info_.integr_kf_switched = true;
    return false;
  }
  Matrix3ft delta_rotation_prev = delta_rotation_; Vector3ft delta_translation_prev = delta_translation_; Matrix6d delta_covariance_prev = delta_covariance_;
  auto compose_global = [&]() {
    double tmp[3];
    se3::m3_mulv(last_odoKF_global_rotation_.m, delta_translation_.v, tmp);
    for (int i = 0; i < 3; ++i) last_estimated_translation_[i] = last_odoKF_global_translation_[i] + tmp[i];
    se3::m3_mul(last_odoKF_global_rotation_.m, delta_rotation_.m, last_estimated_rotation_.m);
    rmats_.push_back(last_estimated_rotation_); tvecs_.push_back(last_estimated_translation_);
  };
  if (!lost_) {
    odometry_success_ = estimateVisualOdometry(delta_rotation_, delta_translation_, delta_covariance_);
    compose_global();
    if (!odometry_success_) {  // :2066-2117
      lost_ = true;
      PoseConstraint dummy; dummy.ini_id_ = global_time_ - 1; dummy.end_id_ = global_time_; dummy.type_ = PoseConstraint::SEQ_ODO;
      dummy.rotation_ = Matrix3ft::Identity(); dummy.translation_ = Vector3ft::Zero(); dummy.scale_ = 1.f; dummy.covariance_ = zero6();
      for (int i = 0; i < 6; ++i) dummy.covariance_[i * 7] = 100.0;
      sink->pushConstraint(dummy);
      sink->backPose(sink_back_pose_);
      Pose p = sink_back_pose_; p.id_ = global_time_; p.scale_ = 1.f;   // repeats the back-end's last pose (:2077-2078)
      sink->pushPose(p);
      sink_back_pose_ = p;
      resetOdometryKeyframe();
      resetIntegrationKeyframe();
      saveCurrentImagesAsOdoKeyframes();
      saveCurrentImagesAsIntegrationKeyframes(rgb24_);
      ++global_time_;
      last_info_.odo_kf_switched = last_info_.integr_kf_switched = true;
      if (verbose_) std::cout << "I am LOST!!!" << std::endl;
      return false;
    }
  } else {
    odometry_success_ = estimateVisualOdometry(delta_rotation_, delta_translation_, delta_covariance_);
    if (odometry_success_) { lost_ = false; compose_global(); }
    else {
      saveCurrentImagesAsOdoKeyframes();
      saveCurrentImagesAsIntegrationKeyframes(rgb24_);
      return false;
    }
  }
  device::sync();
  odoKF_count_++; integrKF_count_++;
  {
    // sequential constraint + covariance :2128-2165
    Matrix3ft pT, Rseq; Vector3ft tseq; double d[3], Jn[36], Jl[36], S[9], SR[9];
    se3::m3_T(delta_rotation_prev.m, pT.m);
    se3::m3_mul(pT.m, delta_rotation_.m, Rseq.m);
    for (int i = 0; i < 3; ++i) d[i] = delta_translation_[i] - delta_translation_prev[i];
    se3::m3_mulv(pT.m, d, tseq.v);
    se3::m6_zero(Jn); se3::m6_set_block(Jn, 0, 0, pT.m, 1.0); se3::m6_set_block(Jn, 3, 3, pT.m, 1.0);
    se3::m6_zero(Jl); se3::m6_set_block(Jl, 0, 0, pT.m, -1.0); se3::m6_set_block(Jl, 3, 3, pT.m, -1.0);
    se3::skew(tseq.v, S); se3::m3_mul(S, pT.m, SR); se3::m6_set_block(Jl, 0, 3, SR, 1.0);
    Matrix6d cseq = zero6();
    se3::m6_JCJt_add(Jl, delta_covariance_prev.data(), cseq.data());
    se3::m6_JCJt_add(Jn, delta_covariance_.data(), cseq.data());
    odo_rmats_.push_back(Rseq); odo_tvecs_.push_back(tseq); odo_covmats_.push_back(cseq);
    PoseConstraint c; c.ini_id_ = global_time_ - 1; c.end_id_ = global_time_; c.type_ = PoseConstraint::SEQ_ODO;
    c.rotation_ = Rseq; c.translation_ = tseq; c.scale_ = 1.f; c.covariance_ = cseq;
    sink->pushConstraint(c);
    // the new pose continues the back-end's last pose, not last_estimated_* (:2161-2162)
    sink->backPose(sink_back_pose_);
    Pose p; p.id_ = global_time_; p.scale_ = 1.f;
    se3::m3_mul(sink_back_pose_.rotation_.m, Rseq.m, p.rotation_.m);
    double tb[3];
    se3::m3_mulv(sink_back_pose_.rotation_.m, tseq.v, tb);
    for (int i = 0; i < 3; ++i) p.translation_[i] = sink_back_pose_.translation_[i] + tb[i];
    sink->pushPose(p);
    sink_back_pose_ = p;
    setSharedCameraPose(p.getAffine());
  }
  // odometry keyframe :2172-2180
  float visibility_ratio_odo = computeCovisibility(delta_rotation_, delta_translation_, depthinvs_odoKF_[0], depthinvs_curr_[0]);
  last_info_.visratio_odo = visibility_ratio_odo;
  if ((odoKF_count_ >= max_odoKF_count_) || (visibility_ratio_odo < visibility_ratio_odo_threshold_)) {
    resetOdometryKeyframe();
    saveCurrentImagesAsOdoKeyframes();
    last_info_.odo_kf_switched = true;
  }
  // integration keyframe :2182-2211
  Matrix3ft iRi, delta_integr_rotation; Vector3ft delta_integr_translation; double d[3];
  se3::m3_inv(last_integrKF_global_rotation_.m, iRi.m);
  se3::m3_mul(iRi.m, last_estimated_rotation_.m, delta_integr_rotation.m);
  for (int i = 0; i < 3; ++i) d[i] = last_estimated_translation_[i] - last_integrKF_global_translation_[i];
  se3::m3_mulv(iRi.m, d, delta_integr_translation.v);
  float visibility_ratio_integr = computeCovisibility(delta_integr_rotation, delta_integr_translation, depthinv_integrKF_raw_, depthinvs_curr_[0]);
  last_info_.visratio_integr = visibility_ratio_integr;
  if ((integrKF_count_ >= max_integrKF_count_) || (visibility_ratio_integr < visibility_ratio_integr_threshold_)) {
    resetIntegrationKeyframe();
    computeOverlapping(delta_integr_rotation, delta_integr_translation, depthinv_integrKF_raw_, depthinvs_curr_[0], overlap_mask_integrKF_);
    saveCurrentImagesAsIntegrationKeyframes(rgb24_);
    newKF_ = true;
    last_info_.integr_kf_switched = true;
  } else {
    integrateImagesIntoKeyframes(depthinvs_curr_[0], delta_integr_rotation, delta_integr_translation);
  }
  vis_odo_times_.push_back((float)(now_ms() - t1));
  if (preview_) {
    std::lock_guard<std::mutex> lock(mutex_scene_view_);
    getImage(scene_view_, intensity_view_, depthinv_view_);
    scene_view_has_changed_ = true;
  }
  device::sync();
  ++global_time_;
  return true;
}

}  // namespace RGBID_SLAM
