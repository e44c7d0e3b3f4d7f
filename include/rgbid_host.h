/*
 * rgbid_host.h -- C-ABI of the host-side library (librgbid_host.so): the VisodoTracker mirror
 * (include/rgbid/visodo.h; reference include/visodo.h + src/visodo.cpp), the SE(3) helpers
 * (reference src/util_funcs.cpp:31-155) and the INI settings parser (reference src/settings.cpp).
 * The C++ classes themselves are the drop-in surface; this C layer exists so tests and other languages
 * can drive them without a C++ compiler.
 */
#ifndef RGBID_HOST_H_
#define RGBID_HOST_H_

#include "rgbid.h"

#ifdef __cplusplus
extern "C" {
#endif

/* util_funcs.cpp: expMapRot :124-148, expMap :86-122, logMap :31-83 (twist = v, omega), forceOrthogonalisation :150-155 */
void rgbid_expmap_rot(const double w[3], double R[9]);
void rgbid_expmap(const double w[3], const double v[3], double R[9], double t[3]);
void rgbid_logmap(const double R[9], const double t[3], double twist[6]);
void rgbid_force_orthogonal(const double M[9], double R[9]);
/* Eigen A.llt().solve(b) (visodo.cpp:1249) and A.inverse() (visodo.cpp:1409) */
void rgbid_llt_solve6(const double A[36], const double b[6], double x[6]);
void rgbid_inverse6(const double A[36], double Ainv[36]);

/* Settings::getSection + Section::getEntry (settings.cpp); returns the value length or <0 */
int rgbid_settings_get(const char* ini_path, const char* section, const char* key, char* out, int cap);

typedef struct rgbid_tracker rgbid_tracker;
/* the VisodoTracker constructor arguments (include/visodo.h:54-68) + calibration */
typedef struct rgbid_tracker_config {
  int rows, cols, levels;
  int iters[8];
  int mestimator, motion_model, sigma_estimator, weighting, warping;
  int max_odoKF_count, finest_level, termination;
  float visratio_odo;
  int image_filtering;
  float visratio_integr;
  int max_integrKF_count, nsamples;
  float fx, fy, cx, cy, factor_depth;
  int interp_mode, preview;
} rgbid_tracker_config;
typedef struct rgbid_tracker_info {
  int lost, odo_kf_switched, integr_kf_switched;
  float visratio_odo, visratio_integr, sigma_int, sigma_depthinv, nu_int, nu_depthinv;
} rgbid_tracker_info;

void rgbid_tracker_default_config(rgbid_tracker_config* c);  /* ctor defaults + shipped ini + factory calibration */
int rgbid_tracker_create(rgbid_tracker** t, const rgbid_tracker_config* c, int device);
int rgbid_tracker_destroy(rgbid_tracker* t);
/* 1 (default): trackNewFrame enqueues its device calls without per-call timing events / synchronisation (it ignores the returned milliseconds);
 * 0: every bridge call synchronous and timed, as in the reference.  Results are identical. */
int rgbid_tracker_set_async_bridge(rgbid_tracker* t, int on);
/* 1 (the DEFAULT since round 5): trackNewFrame runs the frame as ONE step of a one-lane device-resident engine (rgbid_engine.h: the same kernels in the
 * bit-exact numerics class, ~100 launches and no host round trip per Gauss-Newton iteration) instead of ~330 synchronous bridge calls; the tracker's
 * observable state (poses, odometry constraints, keyframe stream, lastInfo) is maintained from the step's pose record -- bit-identical to the host-driven
 * loop, every shipped configuration (CHI_SQUARED termination, custom_registration = 1 included).  0: the host-driven loop.  Only before the first frame or
 * after reset() (else RGBID_E_INVALID).  1 also returns RGBID_E_INVALID when the engine cannot take the run over (non-identity initial pose, > 8 levels);
 * left at its default such a run falls back to the host-driven loop by itself, with one line on stderr.  VisodoTracker::setEngineBacked. */
int rgbid_tracker_set_engine_backed(rgbid_tracker* t, int on);
/* the current mode (VisodoTracker::engineBacked): settled for good once the first frame has been taken */
int rgbid_tracker_get_engine_backed(const rgbid_tracker* t, int* on);
int rgbid_tracker_reset(rgbid_tracker* t);                                     /* VisodoTracker::reset (src/visodo.cpp:519-553): the next frame is a first frame */
int rgbid_tracker_load_settings(rgbid_tracker* t, const char* ini_path);      /* VisodoTracker::loadSettings */
int rgbid_tracker_load_calibration(rgbid_tracker* t, const char* ini_path);   /* VisodoTracker::loadCalibration */
/* uploads depth (u16 mm, rows x cols) and rgb (u8 r,g,b) from HOST memory and runs trackNewFrame */
int rgbid_tracker_track(rgbid_tracker* t, const uint16_t* depth_mm_host, const uint8_t* rgb_host, int* tracked);
int rgbid_tracker_num_poses(const rgbid_tracker* t);
int rgbid_tracker_get_pose(const rgbid_tracker* t, int i, double R[9], double tv[3]);
int rgbid_tracker_num_odo(const rgbid_tracker* t);
int rgbid_tracker_get_odo(const rgbid_tracker* t, int i, double R[9], double tv[3], double cov[36]);
int rgbid_tracker_last_info(const rgbid_tracker* t, rgbid_tracker_info* info);
int rgbid_tracker_keyframe_maps(rgbid_tracker* t, float* depthinv_host, float* weight_host);

/* what the application's viewer reads after a tracked frame when the preview is on (scene_view_, intensity_view_, depthinv_view_ under mutex_scene_view_;
 * getImage, src/visodo.cpp:559-580, 2237-2241): the Phong-shaded keyframe (rows x cols x 3 bytes), the current intensity and the keyframe's inverse depth
 * (rows x cols floats each).  Any output may be NULL; *changed = scene_view_has_changed_ (then cleared).  RGBID_E_INVALID before the first tracked frame */
int rgbid_tracker_scene_view(rgbid_tracker* t, uint8_t* rgb, float* intensity, float* depthinv, int* changed);

/* level-0 inverse depth / intensity of the last prepared frame (after undistortion + registration when custom_registration=1), to host */
int rgbid_tracker_current_maps(const rgbid_tracker* t, float* depthinv, float* intensity);

/* ---- what trackNewFrame hands to the back-end.  The reference writes into its KeyframeManager
 * (include/keyframe_manager.h:77-104): poses_ (src/visodo.cpp:2033-2038, 2073-2079, 2157-2164), constraints_ (:1646-1650, 2071-2076,
 * 2155-2160) and the bounded buffer_keyframes_ (try_push, :1632-1644; capacity 100, src/keyframe_manager.cpp:45).  rgbid_tracker_collect
 * attaches a sink with the same three containers to the tracker; a maintainer's own back-end derives RGBID_SLAM::TrackerSink instead. ---- */
enum { RGBID_SEQ_ODO = 0, RGBID_SEQ_KF = 1 };   /* PoseConstraint::SEQ_ODO / SEQ_KF (include/pose_graph_manager.h) */
typedef struct rgbid_keyframe_info {
  int id, rows, cols;
  float K[9], kd[5];                  /* calibration of the exported images (Keyframe ctor, include/keyframe.h:45-70) */
  double R[9], t[3];                  /* global pose of the keyframe */
  double R_rel[9], t_rel[3];          /* to the next keyframe */
} rgbid_keyframe_info;
int rgbid_tracker_collect(rgbid_tracker* t, int keyframe_capacity);   /* call before the first frame; capacity <= 0 -> 100 */
int rgbid_tracker_num_sink_poses(const rgbid_tracker* t);
int rgbid_tracker_get_sink_pose(const rgbid_tracker* t, int i, int* id, double R[9], double tv[3]);
int rgbid_tracker_set_sink_pose(rgbid_tracker* t, int i, const double R[9], const double tv[3]);  /* the back-end's optimiser moving a pose */
int rgbid_tracker_num_constraints(const rgbid_tracker* t);
int rgbid_tracker_get_constraint(const rgbid_tracker* t, int i, int* ini_id, int* end_id, int* type, double R[9], double tv[3], double cov[36]);
int rgbid_tracker_num_keyframes(const rgbid_tracker* t);   /* keyframes waiting in the bounded buffer */
int rgbid_tracker_peek_keyframe(const rgbid_tracker* t, int i, rgbid_keyframe_info* info, unsigned char* overlap_mask /* rows*cols */,
                                unsigned char* colors /* rows*cols*3 */, float* depthinv /* rows*cols */, float* normals /* 3*rows*cols planar */);
int rgbid_tracker_pop_keyframe(rgbid_tracker* t);          /* buffer_keyframes_.try_pop: drops the oldest; RGBID_E_INVALID when empty */

/* KeyframeAlign::alignKeyframes (src/keyframe_align.cpp:115-357): host inputs, R/t in-out (initial guess -> result) */
int rgbid_keyframe_align(int device, int rows, int cols, const float* depthinv_ini, const unsigned char* grey_ini,
                         const float* depthinv_end, const unsigned char* grey_end, float fx, float fy, float cx, float cy,
                         double R[9], double t[3], double cov[36]);
/* the same with the loop chosen (host_driven = 1: the reference's call sequence through the bridge; 0: the class default, the device-resident aligner) */
int rgbid_keyframe_align_mode(int device, int rows, int cols, const float* depthinv_ini, const unsigned char* grey_ini,
                              const float* depthinv_end, const unsigned char* grey_end, float fx, float fy, float cx, float cy,
                              double R[9], double t[3], double cov[36], int host_driven);

/* the interpolation mode of the calling thread's default bridge context -- what VisodoTracker::setInterpMode sets; KeyframeAlign samples with it in both of its loops */
int rgbid_default_ctx_set_interp_mode(int mode);

/* ---- dataset I/O (tools/evaluation.cpp:122-351,380-439): PNG codec, TUM/ICL association files, trajectory writer ---- */
int rgbid_png_info(const char* path, int* rows, int* cols, int* channels, int* bit_depth);
int rgbid_png_read(const char* path, void* dst, size_t dst_bytes);             /* interleaved, host-endian samples */
int rgbid_png_write(const char* path, const void* data, int rows, int cols, int channels, int bit_depth);
typedef struct rgbid_dataset rgbid_dataset;
/* folder with depth_associated.txt + rgb_associated.txt, or match_file (may be NULL/"") = 4-column association file */
int rgbid_dataset_open(rgbid_dataset** out, const char* folder, const char* match_file);
void rgbid_dataset_close(rgbid_dataset* d);
int rgbid_dataset_size(const rgbid_dataset* d);
double rgbid_dataset_stamp(const rgbid_dataset* d, int i);
/* Evaluation::grab(i, depth, rgb): depth = PNG x 0.2 (mm, u16), rgb = r,g,b bytes; *grabbed = 0 if the pair is unreadable */
int rgbid_dataset_grab(rgbid_dataset* d, int i, uint16_t* depth_mm, uint8_t* rgb, int rows, int cols, int* grabbed);
/* "stamp tx ty tz qx qy qz qw" (fixed, 6 decimals; Eigen::Quaternionf of the float rotation); returns length or <0 */
int rgbid_format_pose_line(double stamp, const double R[9], const double t[3], char* dst, size_t dst_bytes);
/* Evaluation::saveAllPoses / saveTimeLogFiles for a tracker driven over the dataset */
int rgbid_tracker_save_poses(const rgbid_tracker* t, const rgbid_dataset* d, int frame_number, const char* poses_logfile,
                             const char* misc_logfile);
int rgbid_tracker_save_kf_times(const rgbid_tracker* t, const rgbid_dataset* d, const char* kftimes_logfile);

#ifdef __cplusplus
}
#endif
#endif
