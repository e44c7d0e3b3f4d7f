/*
 * rgbid_engine.h -- C-ABI of the batched, device-resident tracker ("engine").
 *
 * The reference's VisodoTracker::trackNewFrame (src/visodo.cpp:1967-2247) drives ~40-70 synchronous
 * launches per Gauss-Newton iteration from the host.  The engine runs the SAME per-frame algorithm for
 * `lanes` independent trackers (sequence chunks / frame pairs) in lock-step with all
 * state in HBM: every kernel is batched over the lanes, the 6x6 solve + SE(3) update + keyframe decisions
 * run on the device, and one step is a fixed launch sequence (optionally one hipGraph) with no host
 * synchronisation.  Per lane the results are those of the single-image bridge functions of rgbid.h.
 *
 * Inputs of a step live in device memory: depth u16 millimetres [lanes][rows][cols] and packed RGB u8
 * [lanes][rows][cols][3] (the layouts VisodoTracker::depth_ / rgb24_ have after upload, tools/
 * RGBID_SLAMapp.cpp:187-188).  Outputs are fixed-size pose records.
 */
#ifndef RGBID_ENGINE_H_
#define RGBID_ENGINE_H_

#include "rgbid.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rgbid_engine rgbid_engine;

/* mirrors the VisodoTracker constructor arguments (include/visodo.h:54-68) + calibration */
typedef struct rgbid_engine_config {
  int rows, cols, levels, lanes;
  int iters[8];              /* Gauss-Newton iterations per level, level 0 first ({10,5,3}: visodo.cpp:65) */
  int mestimator, motion_model, sigma_estimator, weighting;
  int max_odoKF_count, finest_level, image_filtering;
  float visratio_odo, visratio_integr;
  int max_integrKF_count, nsamples;
  float fx, fy, cx, cy, factor_depth;
  int interp_mode;
  float delta_t;             /* inter-frame time (0.03333 in eval mode, visodo.cpp:1931) */
  int use_graph;             /* replay each step as one hipGraph */
  int fused_gn;              /* warp + residual + normal equations in one kernel (W1/I1 never stored) */
  int chi_square_stats;      /* also run the (unused-by-the-reference) full-res chi-square of visodo.cpp:1411-1415 */
  int preview;               /* also render the Phong preview (getImage, visodo.cpp:559-580) each step */
  int record_capacity;       /* steps of pose records kept on the device (ring) */
  int warping;               /* RGBID_PYR_FIRST (default) or RGBID_WARP_FIRST: warp at level 0 and pyrDown the warped maps (visodo.cpp:1078-1105) */
} rgbid_engine_config;

#define RGBID_ST_TRACKED    1   /* trackNewFrame returned true */
#define RGBID_ST_LOST       2
#define RGBID_ST_ODO_KF     4   /* odometry keyframe was (re)created from this frame */
#define RGBID_ST_INTEGR_KF  8   /* integration keyframe was (re)created from this frame */
#define RGBID_ST_FIRST     16   /* first frame of the lane */

typedef struct rgbid_pose_record {
  int frame, status;
  float vis_odo, vis_integr;
  float sigma_int, sigma_depthinv, nu_int, nu_depthinv;
  double R[9], t[3];              /* global camera pose (rmats_/tvecs_), row-major */
  double odo_R[9], odo_t[3];      /* frame-to-frame odometry (odo_rmats_/odo_tvecs_) */
  double odo_cov[36];             /* its 6x6 covariance (odo_covmats_) */
  double kf_R[9], kf_t[3];        /* keyframe-relative pose estimateVisualOdometry returned */
  double kf_cov[36];
} rgbid_pose_record;

void rgbid_engine_default_config(rgbid_engine_config* cfg);   /* ctor defaults + shipped ini + factory calibration */
int rgbid_engine_create(rgbid_engine** e, rgbid_ctx* ctx, const rgbid_engine_config* cfg);
/* an engine borrows its context's stream: destroy the engine BEFORE rgbid_ctx_destroy(ctx) */
int rgbid_engine_destroy(rgbid_engine* e);
/* VisodoTracker::reset() for every lane */
int rgbid_engine_reset(rgbid_engine* e);
/* one trackNewFrame for every lane; depth/rgb are device pointers laid out as described above.
 * Asynchronous on the context's stream (sync with rgbid_ctx_sync or a record read). */
int rgbid_engine_step(rgbid_engine* e, const void* depth_dev, const void* rgb_dev);
/* number of steps taken since reset */
int rgbid_engine_steps(const rgbid_engine* e);
/* copies records of steps [first_step, first_step+n_steps) for all lanes to host: out[n_steps][lanes]. Synchronises. */
int rgbid_engine_read_records(rgbid_engine* e, int first_step, int n_steps, rgbid_pose_record* out);
/* device pointer of the record ring (rgbid_pose_record[capacity][lanes]) for zero-copy gathers */
int rgbid_engine_records_dev(rgbid_engine* e, void** ptr, int* capacity);
/* debugging / parity access to a lane's fused keyframe maps (device pointers + geometry) */
/* device views of a lane's Phong preview (cfg.preview = 1; getImage, visodo.cpp:559-580) and of the keyframe colours it shades */
int rgbid_engine_preview(rgbid_engine* e, int lane, rgbid_img* preview_u8x3, rgbid_img* keyframe_colors_u8x3);
int rgbid_engine_keyframe_maps(rgbid_engine* e, int lane, rgbid_img* depthinv, rgbid_img* weight, rgbid_img* vmap,
                               rgbid_img* nmap, rgbid_img* overlap_mask);
/* Event timing of the dominant kernel: while profiling is on, every launch of the level-0 (full resolution)
 * residual + normal-equation kernel is bracketed by a hipEvent pair on the context's stream (steps run eagerly,
 * not as a graph).  profile_end synchronises and returns the summed kernel time, the number of launches and the
 * algorithmic bytes one launch processes (32 B/px x rows x cols x lanes, SURVEY.md section 8d unit U1). */
int rgbid_engine_profile_begin(rgbid_engine* e, int max_launches);
int rgbid_engine_profile_end(rgbid_engine* e, double* total_ms, int* n_launches, double* bytes_per_launch);
/* total HBM bytes the engine allocated */
int rgbid_engine_bytes(const rgbid_engine* e, size_t* bytes);
/* name + launch count of every kernel enqueued by the last step (for DESIGN.md / profiling); returns #launches */
int rgbid_engine_launches_per_step(const rgbid_engine* e);

#ifdef __cplusplus
}
#endif
#endif
