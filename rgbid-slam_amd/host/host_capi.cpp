// host_capi.cpp -- C-ABI over the host-side C++ (VisodoTracker, SE(3) helpers, settings) so tests and other languages can
// drive it (declared in include/rgbid_host.h).
#include "../../include/rgbid_host.h"

#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <vector>
#include <fstream>
#include <new>

#include "../../include/rgbid/se3.h"
#include "../../include/rgbid/evaluation.h"
#include "../../include/rgbid/keyframe_align.h"
#include "../../include/rgbid/visodo.h"

namespace RGBID_SLAM { namespace device { DeviceProp dev_prop; int dev_id = 0; } }

using namespace RGBID_SLAM;
namespace se3 = rgbid::se3;

// KeyframeManager's three containers (include/keyframe_manager.h:77-104) behind TrackerSink
struct CollectSink : TrackerSink {
  std::mutex m;
  std::vector<Pose> poses;
  std::vector<PoseConstraint> constraints;
  std::deque<std::shared_ptr<KeyframeRecord>> keyframes;
  size_t capacity = 100;
  bool backPose(Pose& p) override { std::lock_guard<std::mutex> l(m); if (poses.empty()) return false; p = poses.back(); return true; }
  void pushPose(const Pose& p) override { std::lock_guard<std::mutex> l(m); poses.push_back(p); }
  void pushConstraint(const PoseConstraint& c) override { std::lock_guard<std::mutex> l(m); constraints.push_back(c); }
  bool tryPushKeyframe(std::shared_ptr<KeyframeRecord> k) override {
    std::lock_guard<std::mutex> l(m);
    if (keyframes.size() >= capacity) return false;
    keyframes.push_back(k);
    return true;
  }
};

struct rgbid_tracker { VisodoTracker* t; CollectSink* sink = nullptr; };

extern "C" {

void rgbid_expmap_rot(const double w[3], double R[9]) { se3::expmap_rot(w, R); }
void rgbid_expmap(const double w[3], const double v[3], double R[9], double t[3]) { se3::expmap(w, v, R, t); }
void rgbid_logmap(const double R[9], const double t[3], double twist[6]) { se3::logmap(R, t, twist); }
void rgbid_force_orthogonal(const double M[9], double R[9]) { se3::force_orthogonal(M, R); }
void rgbid_llt_solve6(const double A[36], const double b[6], double x[6]) { se3::llt_solve6(A, b, x); }
void rgbid_inverse6(const double A[36], double Ainv[36]) { se3::inverse6(A, Ainv); }

int rgbid_settings_get(const char* path, const char* section, const char* key, char* out, int cap) {
  std::ifstream f(path);
  if (!f.is_open()) return -1;
  Settings s(f, false);
  Section sec;
  if (!s.getSection(section, sec)) return -2;
  Entry e;
  if (!sec.getEntry(key, e)) return -3;
  std::string v = e.getValue();
  if ((int)v.size() + 1 > cap) return -4;
  std::memcpy(out, v.c_str(), v.size() + 1);
  return (int)v.size();
}

void rgbid_tracker_default_config(rgbid_tracker_config* c) {
  std::memset(c, 0, sizeof(*c));
  c->rows = 480; c->cols = 640; c->levels = 3;
  c->iters[0] = 10; c->iters[1] = 5; c->iters[2] = 3;
  c->mestimator = RGBID_STUDENT; c->motion_model = RGBID_CONSTANT_VELOCITY; c->sigma_estimator = RGBID_SIGMA_PDF;
  c->weighting = RGBID_INDEPENDENT; c->warping = RGBID_PYR_FIRST; c->max_odoKF_count = 9999999; c->finest_level = 0;
  c->termination = RGBID_ALL_ITERS; c->visratio_odo = 0.9f; c->image_filtering = RGBID_NO_FILTERS; c->visratio_integr = 0.7f;
  c->max_integrKF_count = 9999999; c->nsamples = 10000;
  c->fx = 525.f; c->fy = 525.f; c->cx = 319.5f; c->cy = 239.5f; c->factor_depth = 1.f;
  c->interp_mode = RGBID_INTERP_TEX8; c->preview = 0;
}

int rgbid_tracker_create(rgbid_tracker** out, const rgbid_tracker_config* c, int device) {
  if (!out || !c) return RGBID_E_INVALID;
  int n = 0;
  if (rgbid_device_count(&n) != RGBID_OK || device < 0 || device >= n) return RGBID_E_NODEV;
  pcl::gpu::setDevice(device);
  rgbid_tracker* h = new (std::nothrow) rgbid_tracker();
  if (!h) return RGBID_E_NOMEM;
  h->t = new VisodoTracker(6, c->mestimator, c->motion_model, c->sigma_estimator, c->weighting, c->warping, c->max_odoKF_count, c->finest_level,
                           c->termination, c->visratio_odo, c->image_filtering, c->visratio_integr, c->max_integrKF_count, c->nsamples, c->rows,
                           c->cols, c->levels);
  h->t->setRGBIntrinsics(c->fx, c->fy, c->cx, c->cy);
  h->t->setFactorDepth(c->factor_depth);
  h->t->setIterations(c->iters, c->levels);
  h->t->setInterpMode(c->interp_mode);
  h->t->setPreview(c->preview != 0);
  *out = h;
  return RGBID_OK;
}

int rgbid_tracker_destroy(rgbid_tracker* h) { if (h) { delete h->t; delete h->sink; delete h; } return RGBID_OK; }
int rgbid_tracker_set_async_bridge(rgbid_tracker* h, int on) { if (!h) return RGBID_E_INVALID; h->t->setAsyncBridge(on != 0); return RGBID_OK; }
int rgbid_tracker_set_engine_backed(rgbid_tracker* h, int on) {
  if (!h) return RGBID_E_INVALID;
  return h->t->setEngineBacked(on != 0) ? RGBID_OK : RGBID_E_INVALID;
}
int rgbid_tracker_get_engine_backed(const rgbid_tracker* h, int* on) { if (!h || !on) return RGBID_E_INVALID; *on = h->t->engineBacked() ? 1 : 0; return RGBID_OK; }
int rgbid_tracker_reset(rgbid_tracker* h) { if (!h) return RGBID_E_INVALID; h->t->reset(); return RGBID_OK; }
int rgbid_tracker_load_settings(rgbid_tracker* h, const char* ini_path) {
  if (!h || !ini_path) return RGBID_E_INVALID;
  std::ifstream f(ini_path);
  if (!f.is_open()) return RGBID_E_INVALID;
  Settings s(f, false);
  h->t->loadSettings(s);
  return RGBID_OK;
}
int rgbid_tracker_load_calibration(rgbid_tracker* h, const char* ini_path) {
  if (!h || !ini_path) return RGBID_E_INVALID;
  h->t->loadCalibration(ini_path);
  return RGBID_OK;
}
int rgbid_tracker_track(rgbid_tracker* h, const uint16_t* depth_mm_host, const uint8_t* rgb_host, int* tracked) {
  if (!h || !depth_mm_host || !rgb_host) return RGBID_E_INVALID;
  VisodoTracker& t = *h->t;
  // what the application's grabber does (tools/RGBID_SLAMapp.cpp:187-188): H2D upload of depth_ and rgb24_
  t.depth_.upload(depth_mm_host, (size_t)t.cols() * 2, t.rows(), t.cols());
  t.rgb24_.upload(rgb_host, (size_t)t.cols() * 3, t.rows(), t.cols());
  bool ok = t.trackNewFrame();
  if (tracked) *tracked = ok ? 1 : 0;
  return RGBID_OK;
}
int rgbid_tracker_num_poses(const rgbid_tracker* h) { return h ? (int)h->t->getNumberOfPoses() : 0; }
int rgbid_tracker_get_pose(const rgbid_tracker* h, int i, double R[9], double tv[3]) {
  if (!h || i < 0 || i >= (int)h->t->getNumberOfPoses()) return RGBID_E_INVALID;
  Affine3d a = h->t->getCameraPose(i);
  std::memcpy(R, a.R.m, sizeof(a.R.m)); std::memcpy(tv, a.t.v, sizeof(a.t.v));
  return RGBID_OK;
}
int rgbid_tracker_num_odo(const rgbid_tracker* h) { return h ? (int)h->t->odoRotations().size() : 0; }
int rgbid_tracker_get_odo(const rgbid_tracker* h, int i, double R[9], double tv[3], double cov[36]) {
  if (!h || i < 0 || i >= (int)h->t->odoRotations().size()) return RGBID_E_INVALID;
  std::memcpy(R, h->t->odoRotations()[i].m, 72); std::memcpy(tv, h->t->odoTranslations()[i].v, 24);
  std::memcpy(cov, h->t->odoCovariances()[i].data(), 288);
  return RGBID_OK;
}
int rgbid_tracker_last_info(const rgbid_tracker* h, rgbid_tracker_info* info) {
  if (!h || !info) return RGBID_E_INVALID;
  VisodoTracker::LastFrameInfo l = h->t->lastInfo();
  info->lost = h->t->visOdoIsLost() ? 1 : 0;
  info->odo_kf_switched = l.odo_kf_switched; info->integr_kf_switched = l.integr_kf_switched;
  info->visratio_odo = l.visratio_odo; info->visratio_integr = l.visratio_integr;
  info->sigma_int = l.sigma_int; info->sigma_depthinv = l.sigma_depthinv; info->nu_int = l.nu_int; info->nu_depthinv = l.nu_depthinv;
  return RGBID_OK;
}
int rgbid_tracker_keyframe_maps(rgbid_tracker* h, float* depthinv_host, float* weight_host) {
  if (!h) return RGBID_E_INVALID;
  h->t->downloadKeyframeMaps(depthinv_host, weight_host);
  return RGBID_OK;
}

int rgbid_tracker_scene_view(rgbid_tracker* h, uint8_t* rgb, float* intensity, float* depthinv, int* changed) {
  if (!h) return RGBID_E_INVALID;
  VisodoTracker& t = *h->t;
  std::lock_guard<std::mutex> lock(t.mutex_scene_view_);
  const size_t n = (size_t)t.rows() * t.cols();
  if (t.scene_view_.size() != n || t.intensity_view_.size() != n || t.depthinv_view_.size() != n) return RGBID_E_INVALID;
  if (rgb) std::memcpy(rgb, t.scene_view_.data(), n * 3);
  if (intensity) std::memcpy(intensity, t.intensity_view_.data(), n * sizeof(float));
  if (depthinv) std::memcpy(depthinv, t.depthinv_view_.data(), n * sizeof(float));
  if (changed) *changed = t.scene_view_has_changed_ ? 1 : 0;
  t.scene_view_has_changed_ = false;
  return RGBID_OK;
}

int rgbid_tracker_current_maps(const rgbid_tracker* h, float* depthinv, float* intensity) {
  if (!h) return RGBID_E_INVALID;
  h->t->downloadCurrentMaps(depthinv, intensity);
  return RGBID_OK;
}

int rgbid_tracker_collect(rgbid_tracker* h, int keyframe_capacity) {
  if (!h) return RGBID_E_INVALID;
  if (!h->sink) h->sink = new CollectSink();
  h->sink->capacity = keyframe_capacity > 0 ? (size_t)keyframe_capacity : 100;
  h->t->keyframe_manager_ptr_ = h->sink;
  return RGBID_OK;
}
int rgbid_tracker_num_sink_poses(const rgbid_tracker* h) {
  if (!h || !h->sink) return 0;
  std::lock_guard<std::mutex> l(h->sink->m);
  return (int)h->sink->poses.size();
}
int rgbid_tracker_get_sink_pose(const rgbid_tracker* h, int i, int* id, double R[9], double tv[3]) {
  if (!h || !h->sink) return RGBID_E_INVALID;
  std::lock_guard<std::mutex> l(h->sink->m);
  if (i < 0 || i >= (int)h->sink->poses.size()) return RGBID_E_INVALID;
  const Pose& p = h->sink->poses[i];
  if (id) *id = p.id_;
  if (R) for (int k = 0; k < 9; ++k) R[k] = p.rotation_.m[k];
  if (tv) for (int k = 0; k < 3; ++k) tv[k] = p.translation_[k];
  return RGBID_OK;
}
int rgbid_tracker_set_sink_pose(rgbid_tracker* h, int i, const double R[9], const double tv[3]) {
  if (!h || !h->sink || !R || !tv) return RGBID_E_INVALID;
  std::lock_guard<std::mutex> l(h->sink->m);
  if (i < 0 || i >= (int)h->sink->poses.size()) return RGBID_E_INVALID;
  Pose& p = h->sink->poses[i];
  for (int k = 0; k < 9; ++k) p.rotation_.m[k] = R[k];
  for (int k = 0; k < 3; ++k) p.translation_[k] = tv[k];
  return RGBID_OK;
}
int rgbid_tracker_num_constraints(const rgbid_tracker* h) {
  if (!h || !h->sink) return 0;
  std::lock_guard<std::mutex> l(h->sink->m);
  return (int)h->sink->constraints.size();
}
int rgbid_tracker_get_constraint(const rgbid_tracker* h, int i, int* ini_id, int* end_id, int* type, double R[9], double tv[3], double cov[36]) {
  if (!h || !h->sink) return RGBID_E_INVALID;
  std::lock_guard<std::mutex> l(h->sink->m);
  if (i < 0 || i >= (int)h->sink->constraints.size()) return RGBID_E_INVALID;
  const PoseConstraint& c = h->sink->constraints[i];
  if (ini_id) *ini_id = c.ini_id_;
  if (end_id) *end_id = c.end_id_;
  if (type) *type = c.type_;
  if (R) for (int k = 0; k < 9; ++k) R[k] = c.rotation_.m[k];
  if (tv) for (int k = 0; k < 3; ++k) tv[k] = c.translation_[k];
  if (cov) for (int k = 0; k < 36; ++k) cov[k] = c.covariance_[k];
  return RGBID_OK;
}
int rgbid_tracker_num_keyframes(const rgbid_tracker* h) {
  if (!h || !h->sink) return 0;
  std::lock_guard<std::mutex> l(h->sink->m);
  return (int)h->sink->keyframes.size();
}
int rgbid_tracker_peek_keyframe(const rgbid_tracker* h, int i, rgbid_keyframe_info* info, unsigned char* overlap_mask, unsigned char* colors,
                                float* depthinv, float* normals) {
  if (!h || !h->sink) return RGBID_E_INVALID;
  std::lock_guard<std::mutex> l(h->sink->m);
  if (i < 0 || i >= (int)h->sink->keyframes.size()) return RGBID_E_INVALID;
  const KeyframeRecord& k = *h->sink->keyframes[i];
  if (info) {
    info->id = k.id; info->rows = k.rows; info->cols = k.cols;
    for (int j = 0; j < 9; ++j) { info->K[j] = k.K.m[j]; info->R[j] = k.rotation.m[j]; info->R_rel[j] = k.rotation_rel.m[j]; }
    for (int j = 0; j < 5; ++j) info->kd[j] = k.kd[j];
    for (int j = 0; j < 3; ++j) { info->t[j] = k.translation[j]; info->t_rel[j] = k.translation_rel[j]; }
  }
  if (overlap_mask) std::memcpy(overlap_mask, k.overlap_mask_.data(), k.overlap_mask_.size());
  if (colors) std::memcpy(colors, k.colors_.data(), k.colors_.size() * sizeof(PixelRGB));
  if (depthinv) std::memcpy(depthinv, k.depthinv_.data(), k.depthinv_.size() * sizeof(float));
  if (normals) std::memcpy(normals, k.normals_.data(), k.normals_.size() * sizeof(float));
  return RGBID_OK;
}
int rgbid_tracker_pop_keyframe(rgbid_tracker* h) {
  if (!h || !h->sink) return RGBID_E_INVALID;
  std::lock_guard<std::mutex> l(h->sink->m);
  if (h->sink->keyframes.empty()) return RGBID_E_INVALID;
  h->sink->keyframes.pop_front();
  return RGBID_OK;
}

int rgbid_keyframe_align_mode(int device, int rows, int cols, const float* depthinv_ini, const unsigned char* grey_ini, const float* depthinv_end,
                              const unsigned char* grey_end, float fx, float fy, float cx, float cy, double R[9], double t[3], double cov[36], int host_driven) {
  if (!depthinv_ini || !grey_ini || !depthinv_end || !grey_end || !R || !t || !cov) return RGBID_E_INVALID;
  int n = 0;
  if (rgbid_device_count(&n) != RGBID_OK || device < 0 || device >= n) return RGBID_E_NODEV;
  pcl::gpu::setDevice(device);
  KeyframeAlign ka(rows, cols);
  ka.setHostDriven(host_driven != 0);
  KeyframeImages a = {depthinv_ini, grey_ini, fx, fy, cx, cy}, b = {depthinv_end, grey_end, fx, fy, cx, cy};
  Matrix3ft Rm; Vector3ft tv; Matrix6d c;
  std::memcpy(Rm.m, R, 72); std::memcpy(tv.v, t, 24);
  ka.alignKeyframes(a, b, Rm, tv, c);
  std::memcpy(R, Rm.m, 72); std::memcpy(t, tv.v, 24); std::memcpy(cov, c.data(), 288);
  return RGBID_OK;
}
int rgbid_default_ctx_set_interp_mode(int mode) { return rgbid_ctx_set_interp_mode(pcl::gpu::default_ctx(), mode); }
int rgbid_keyframe_align(int device, int rows, int cols, const float* depthinv_ini, const unsigned char* grey_ini, const float* depthinv_end,
                         const unsigned char* grey_end, float fx, float fy, float cx, float cy, double R[9], double t[3], double cov[36]) {
  return rgbid_keyframe_align_mode(device, rows, cols, depthinv_ini, grey_ini, depthinv_end, grey_end, fx, fy, cx, cy, R, t, cov, 0);
}

struct rgbid_dataset { Evaluation* e; };

int rgbid_png_info(const char* path, int* rows, int* cols, int* channels, int* bit_depth) {
  if (!path) return RGBID_E_INVALID;
  try {
    PngImage im = read_png(path);
    if (rows) *rows = im.rows;
    if (cols) *cols = im.cols;
    if (channels) *channels = im.channels;
    if (bit_depth) *bit_depth = im.bit_depth;
  } catch (const std::exception& ex) { std::fprintf(stderr, "%s\n", ex.what()); return RGBID_E_INVALID; }
  return RGBID_OK;
}
int rgbid_png_read(const char* path, void* dst, size_t dst_bytes) {
  if (!path || !dst) return RGBID_E_INVALID;
  try {
    PngImage im = read_png(path);
    if (im.bytes.size() > dst_bytes) return RGBID_E_INVALID;
    std::memcpy(dst, im.bytes.data(), im.bytes.size());
  } catch (const std::exception& ex) { std::fprintf(stderr, "%s\n", ex.what()); return RGBID_E_INVALID; }
  return RGBID_OK;
}
int rgbid_png_write(const char* path, const void* data, int rows, int cols, int channels, int bit_depth) {
  if (!path || !data) return RGBID_E_INVALID;
  try { write_png(path, data, rows, cols, channels, bit_depth); }
  catch (const std::exception& ex) { std::fprintf(stderr, "%s\n", ex.what()); return RGBID_E_INVALID; }
  return RGBID_OK;
}
int rgbid_dataset_open(rgbid_dataset** out, const char* folder, const char* match_file) {
  if (!out || !folder) return RGBID_E_INVALID;
  *out = nullptr;
  try { *out = new rgbid_dataset{new Evaluation(folder, match_file ? match_file : "")}; }
  catch (const std::exception&) { return RGBID_E_INVALID; }
  return RGBID_OK;
}
void rgbid_dataset_close(rgbid_dataset* d) { if (d) { delete d->e; delete d; } }
int rgbid_dataset_size(const rgbid_dataset* d) { return d ? (int)d->e->size() : 0; }
double rgbid_dataset_stamp(const rgbid_dataset* d, int i) { return (d && i >= 0 && i < (int)d->e->size()) ? d->e->stamp(i) : 0.0; }
int rgbid_dataset_grab(rgbid_dataset* d, int i, uint16_t* depth_mm, uint8_t* rgb, int rows, int cols, int* grabbed) {
  if (!d || !depth_mm || !rgb || !grabbed) return RGBID_E_INVALID;
  ImageWrapper<unsigned short> dw; ImageWrapper<PixelRGB> cw;
  *grabbed = 0;
  try { if (!d->e->grab(i, dw, cw)) return RGBID_OK; }
  catch (const std::exception& ex) { std::fprintf(stderr, "%s\n", ex.what()); return RGBID_E_INVALID; }
  if (dw.rows != rows || dw.cols != cols || cw.rows != rows || cw.cols != cols) return RGBID_E_INVALID;
  std::memcpy(depth_mm, dw.data, (size_t)rows * cols * 2);
  std::memcpy(rgb, cw.data, (size_t)rows * cols * 3);
  *grabbed = 1;
  return RGBID_OK;
}
int rgbid_format_pose_line(double stamp, const double R[9], const double t[3], char* dst, size_t dst_bytes) {
  if (!R || !t || !dst) return RGBID_E_INVALID;
  std::string s = format_pose_line(stamp, R, t);
  if (s.size() + 1 > dst_bytes) return RGBID_E_INVALID;
  std::memcpy(dst, s.c_str(), s.size() + 1);
  return (int)s.size();
}
int rgbid_tracker_save_poses(const rgbid_tracker* t, const rgbid_dataset* d, int frame_number, const char* poses_logfile, const char* misc_logfile) {
  if (!t || !d || !poses_logfile || !misc_logfile) return RGBID_E_INVALID;
  d->e->saveAllPoses(*t->t, frame_number, poses_logfile, misc_logfile);
  return RGBID_OK;
}
int rgbid_tracker_save_kf_times(const rgbid_tracker* t, const rgbid_dataset* d, const char* kftimes_logfile) {
  if (!t || !d || !kftimes_logfile) return RGBID_E_INVALID;
  d->e->saveTimeLogFiles(*t->t, std::vector<float>(), kftimes_logfile);
  return RGBID_OK;
}

}  // extern "C"
