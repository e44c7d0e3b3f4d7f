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
  int fused_gn;              /* warp + residual + normal equations in ONE kernel per Gauss-Newton iteration: the warped maps W1 / I1 are
                              * produced in registers and never stored (32 instead of 52 B/px of HBM traffic per iteration); default 1.
                              * 0 = the reference's kernel sequence (warp pair, sigma/nu on the stored maps, normal equations).  Same results
                              * bit for bit.  Ignored (0) with RGBID_WARP_FIRST, whose pyramid of WARPED maps must exist in memory. */
  int chi_square_stats;      /* also run the (unused-by-the-reference) full-res chi-square of visodo.cpp:1411-1415 */
  int preview;               /* also render the Phong preview (getImage, visodo.cpp:559-580) each step */
  int record_capacity;       /* steps of pose records kept on the device (ring) */
  int warping;               /* RGBID_PYR_FIRST (default) or RGBID_WARP_FIRST: warp at level 0 and pyrDown the warped maps (visodo.cpp:1078-1105) */
  int keyframe_capacity;     /* keyframes exported to the back-end kept per lane on the device (ring); 0 = no export (see below) */
  int fast_numerics;         /* gather kernels (warp pair, covisibility, keyframe fusion) in the reference BUILD's class of arithmetic --
                              * hardware reciprocal + FMA contraction, what nvcc --prec-div=false + default fmad give the reference
                              * (CMakeLists.txt:105) -- instead of the IEEE evaluation of the oracle (default 1; 0 = the bit-exact
                              * kernels of the compat bridge).  The float VALUES are then the cheap ones; every discrete decision (point-sampled
                              * source pixel, validity, covisibility counts, fusion gate) is the oracle's (rgbid.h rgbid_ctx_set_numerics,
                              * csrc/guard_band.h) */
  int defer_keyframe_maps;   /* 0 (default): the vertex / normal maps of the fused keyframe are recomputed after every fusion step, as the reference does
                              * (visodo.cpp:1758-1762).  1: they are computed only where something consumes them -- right before a keyframe is exported
                              * (its normals), for the preview (cfg.preview = 1 keeps the per-frame schedule) and on rgbid_engine_keyframe_maps -- from the
                              * same fused map: every output (pose records, exported keyframes, preview, accessor) is bit-identical, a step writes 24 B/px less */
  /* ---- round 5: the two configurations that were host-driven only ---- */
  int termination;           /* RGBID_ALL_ITERS (default) or RGBID_CHI_SQUARED (visodo.cpp:1134-1164): from its second iteration on a level re-evaluates the
                              * full-lattice chi-square RMSE of the level-0 warped maps; when it grows the last increment is undone and the level ends -- per lane,
                              * as a flag that masks the rest of the level.  The test reads the STORED warped maps, so this mode runs the reference's kernel
                              * sequence (fused_gn is ignored, as with RGBID_WARP_FIRST) */
  int custom_registration;   /* 1: prepareImagesCustomCalibration (visodo.cpp:775-824) instead of prepareImages -- undistort the intensity, correct + undistort
                              * the inverse depth, register it onto the colour camera -- with the calibration below (config_data/calibration_custom*.ini) */
  float rgb_dist[5];         /* k1 .. k5 of the colour camera (its fx .. cy are the engine's) */
  rgbid_intr_k depth_intr;   /* the depth camera: fx, fy, cx, cy, k1 .. k5 */
  rgbid_depth_dist depth_dist;   /* c1, c0, q0[9], q1[9], xshift, yshift */
  float dRc_proj[9], t_dc_proj[3], cRd_proj[9];   /* K_d dRc K_c^-1, K_d t_dc, its inverse: float products as visodo.cpp:792-801 forms them (the caller's) */
} rgbid_engine_config;

#define RGBID_ST_TRACKED    1   /* trackNewFrame returned true */
#define RGBID_ST_LOST       2
#define RGBID_ST_ODO_KF     4   /* odometry keyframe was (re)created from this frame */
#define RGBID_ST_INTEGR_KF  8   /* integration keyframe was (re)created from this frame */
#define RGBID_ST_FIRST     16   /* first frame of the lane */
#define RGBID_ST_KF_EXPORTED 32 /* the outgoing integration keyframe went to the export ring this step (cfg.keyframe_capacity > 0) */

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
/* sizeof(rgbid_engine_config) as the LIBRARY was built: a caller compiled against another revision of this header (fields are only ever appended) compares it
 * with its own sizeof before it hands a config over -- librgbid_host.so, librgbid_dist.so and the Python binding do */
size_t rgbid_engine_config_size(void);
/* custom calibration: fills cfg->dRc_proj / t_dc_proj / cRd_proj from the depth -> colour extrinsics (rotation dRc row-major, translation t_dc: [STEREO_DEPTH2RGB] of
 * config_data/calibration_custom*.ini) and the config's two intrinsics (fx .. cy, depth_intr), with the float arithmetic of src/visodo.cpp:792-801 */
int rgbid_engine_config_set_stereo(rgbid_engine_config* cfg, const float dRc[9], const float t_dc[3]);
int rgbid_engine_create(rgbid_engine** e, rgbid_ctx* ctx, const rgbid_engine_config* cfg);
/* an engine borrows its context's stream: destroy the engine BEFORE rgbid_ctx_destroy(ctx) */
int rgbid_engine_destroy(rgbid_engine* e);
/* VisodoTracker::reset() for every lane */
int rgbid_engine_reset(rgbid_engine* e);
/* VisodoTracker::reset() for ONE lane (a stream that ends while the others go on): its next frame is a first frame -- pose record
 * RGBID_ST_FIRST, identity pose, new keyframes, export count back to 0.  Asynchronous on the context's stream.  Until the lane is fed
 * again (rgbid_engine_set_active) its pose records read all-zero. */
int rgbid_engine_reset_lane(rgbid_engine* e, int lane);
/* Lanes fed by the following steps: active[lane] != 0 (host array of `lanes` ints; NULL = all, the default).  A lane that is not fed sits the
 * step out -- no tracker state of it changes (keyframes, poses, motion model, current-frame pyramids), every kernel of the step is
 * predicated off for it, and its pose record repeats the last pose with status 0 -- so streams of different frame rates or lengths can
 * share an engine.  The input buffers still carry `lanes` frames; the slots of inactive lanes are not read. */
int rgbid_engine_set_active(rgbid_engine* e, const int* active);
/* one trackNewFrame for every lane; depth/rgb are device pointers laid out as described above.
 * Asynchronous on the context's stream (sync with rgbid_ctx_sync or a record read).  With use_graph = 0 the step's kernels read the two
 * buffers IN PLACE (no staging copy): keep them valid and unmodified until the step has executed; with use_graph = 1 they are copied into
 * the engine's own staging buffers first (a captured graph needs fixed addresses) and may be reused once that copy has run. */
int rgbid_engine_step(rgbid_engine* e, const void* depth_dev, const void* rgb_dev);
/* the same for PITCHED inputs (cudaMallocPitch-style containers: rgb24_ / depth_ of the compat VisodoTracker): row steps and lane strides in bytes
 * (lane strides ignored with one lane) */
int rgbid_engine_step_strided(rgbid_engine* e, const void* depth_dev, size_t depth_step, size_t depth_lane_stride, const void* rgb_dev, size_t rgb_step,
                              size_t rgb_lane_stride);
/* the inter-frame time of the constant-velocity model for the steps that follow (computeInterframeTime, visodo.cpp:1902-1965, when the caller
 * measures it per frame).  Ordered on the context's stream; the kernels read it through a device pointer, so it also works with use_graph = 1 */
int rgbid_engine_set_delta_t(rgbid_engine* e, float delta_t);
/* device views of a lane's current-frame level-0 maps (inverse depth, intensity) */
int rgbid_engine_current_maps(rgbid_engine* e, int lane, rgbid_img* depthinv, rgbid_img* intensity);
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
/* ---- keyframe export (resetIntegrationKeyframe, src/visodo.cpp:1610-1652; include/keyframe.h:45-70).  With cfg.keyframe_capacity > 0 every
 * integration-keyframe switch of a lane (resetIntegrationKeyframe: status bit RGBID_ST_KF_EXPORTED) copies, on the device and
 * inside the step, what the reference downloads for its back-end into the lane's ring slot (seq % capacity): the header below and one
 * packed block  overlap mask u8[N] | colours u8[3N] | inverse depth f32[N] | normals f32[3N planar]  (N = rows*cols, the layouts of
 * Keyframe::overlap_mask_/colors_/depthinv_/normals_).  The frame-to-frame SEQ_ODO constraints and the poses are the pose records. ---- */
typedef struct rgbid_keyframe_header {
  int id, end_id;                 /* last_integrKF_index_ and global_time_: the ids of the SEQ_KF constraint (:1646) */
  int lane, seq;                  /* exporting lane and the running number of its exports */
  double R[9], t[3];              /* global pose of the exported keyframe */
  double R_rel[9], t_rel[3];      /* to the next keyframe = the SEQ_KF constraint ... */
  double cov_rel[36];             /* ... and its covariance (:1617-1629) */
} rgbid_keyframe_header;
/* exports so far per lane: counts[lanes] (host).  Synchronises. */
int rgbid_engine_keyframe_counts(rgbid_engine* e, int* counts);
/* export `seq` of `lane` to host memory through one pinned staging buffer (a single asynchronous D2H of header + packed block, then a
 * stream wait).  Any output may be NULL.  RGBID_E_INVALID when seq is not (or no longer) in the lane's ring. */
int rgbid_engine_read_keyframe(rgbid_engine* e, int lane, int seq, rgbid_keyframe_header* header, unsigned char* overlap_mask,
                               unsigned char* colors, float* depthinv, float* normals);
/* device views for zero-copy consumers: header ring [lanes][capacity] and packed blocks [lanes][capacity][20 N bytes] */
int rgbid_engine_keyframes_dev(rgbid_engine* e, void** headers, void** blocks, size_t* block_bytes);

/* ---- the per-frame record ranks exchange when a sequence is sharded over GPUs (rgbid_dist.h): what
 * trackNewFrame appends to odo_rmats_/odo_tvecs_/odo_covmats_ (src/visodo.cpp:2150-2152) plus frame id and status.  392 bytes. ---- */
typedef struct rgbid_gather_record {
  int32_t frame_id;      /* global_time_ of the frame inside its lane's (chunk's) run: 0 = first frame */
  int32_t status;        /* RGBID_ST_* bits */
  double  R[9], t[3];    /* frame-to-frame odometry dT_k: identity on a first frame and on lost frames */
  double  cov[36];       /* its 6x6 covariance (100 I on lost frames, visodo.cpp:2070-2071; 0 on a first frame) */
} rgbid_gather_record;
/* packs steps [first_step, first_step + n_steps) of the pose-record ring into out_dev[lanes][n_steps] (lane-major: a lane's frames are
 * contiguous, the layout rgbid_dist_compose_trajectory reads) with one kernel on the context's stream; out_dev is DEVICE memory */
int rgbid_engine_pack_gather_records(rgbid_engine* e, int first_step, int n_steps, rgbid_gather_record* out_dev);

/* Event timing of the dominant kernel: while profiling is on, every launch of the level-0 (full resolution)
 * residual + normal-equation kernel is bracketed by a hipEvent pair on the context's stream (steps run eagerly,
 * not as a graph; with use_graph = 1 the inputs still go through the staging copy, so the buffer-lifetime rule of
 * rgbid_engine_step does not change).  profile_end synchronises and returns the summed kernel time, the number of launches and the
 * algorithmic bytes one launch processes (32 B/px x rows x cols x lanes, SURVEY.md section 8d unit U1). */
int rgbid_engine_profile_begin(rgbid_engine* e, int max_launches);
int rgbid_engine_profile_end(rgbid_engine* e, double* total_ms, int* n_launches, double* bytes_per_launch);
/* total HBM bytes the engine allocated */
int rgbid_engine_bytes(const rgbid_engine* e, size_t* bytes);
/* name + launch count of every kernel enqueued by the last step (for DESIGN.md / profiling); returns #launches */
int rgbid_engine_launches_per_step(const rgbid_engine* e);
/* Algorithmic HBM bytes PER LANE of the launch list of the last (non-first) step, summed over the engine's own launches as each launcher's
 * DESIGN.md figure (bytes every kernel must read / write once): out[0] every tracked frame; out[1] in addition per odometry-keyframe switch;
 * out[2] per integration-keyframe switch; out[3] per frame fused into the integration keyframe.  bytes/frame = out[0] + p_odo out[1] +
 * p_int out[2] + (1 - p_int) out[3] with the switch rates of the records -- what the engine actually has to move, against which frames/s
 * is a fraction of the HBM roofline (the unfused SURVEY budget U3 = 275.6 MB no longer applies to the fused path). */
int rgbid_engine_step_bytes(const rgbid_engine* e, double out[4]);

#ifdef __cplusplus
}
#endif
#endif
