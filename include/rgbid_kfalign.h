/*
 * rgbid_kfalign.h -- C-ABI of the batched, device-resident keyframe-to-keyframe alignment.
 *
 * The reference's KeyframeAlign::alignKeyframes (src/keyframe_align.cpp:115-357, include/keyframe_align.h:41-104) is the loop closer's
 * dense verification step: a 4-level pyramid of two keyframes, {5,5,3,0} Gauss-Newton iterations with fixed sigmas (0.0025 / 5), nu by
 * bisection (computeNuStudent), the intensity warp sampled on the KEYFRAME inverse depth -- ~300 synchronous bridge calls per pair from
 * the host.  Loop-closure verification is naturally a batch of candidate pairs; here `pairs` alignments advance in lock-step as ONE
 * launch sequence on the context's stream with the poses, the 6x6 solves and the update on the device: the same kernels as the tracker,
 * every pair its own intrinsics and initial guess, no host round trip until the results are read.
 *
 * RGBID_SLAM::KeyframeAlign (include/rgbid/keyframe_align.h) rides on the 1-pair case; its result is the host-driven loop's bit for bit.
 */
#ifndef RGBID_KFALIGN_H_
#define RGBID_KFALIGN_H_

#include "rgbid.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rgbid_kfalign rgbid_kfalign;

/* buffers for up to max_pairs pairs of rows x cols keyframes (16.3 MB per pair at 640 x 480); the aligner borrows the context's stream:
 * destroy it before the context */
int rgbid_kfalign_create(rgbid_kfalign** a, rgbid_ctx* ctx, int rows, int cols, int max_pairs);
int rgbid_kfalign_destroy(rgbid_kfalign* a);
/* Aligns pairs <= max_pairs keyframe pairs.  DEVICE inputs, pair-major and dense: depthinv_* float [pairs][rows][cols] (NaN = invalid: Keyframe::depthinv_),
 * grey_* u8 [pairs][rows][cols] (Keyframe::grey_image_).  HOST arrays: K [pairs][4] = fx, fy, cx, cy of each pair (kf_ini->K_); R [pairs][9] row-major and
 * t [pairs][3] in-out -- the initial guess pose_ini2end, then the aligned pose; cov [pairs][36] out = the inverse of the last iteration's normal matrix
 * (keyframe_align.cpp:343).  Synchronous on return.  Results of a pair do not depend on the other pairs of the call; a 1-pair call is bit-identical to the
 * host-driven KeyframeAlign, larger batches sum the normal equations in another (fixed) order and agree to rounding. */
int rgbid_kfalign_batched(rgbid_kfalign* a, int pairs, const float* depthinv_ini_dev, const unsigned char* grey_ini_dev, const float* depthinv_end_dev,
                          const unsigned char* grey_end_dev, const float* K, double* R, double* t, double* cov);
/* the same with HOST image arrays (uploaded through the aligner's own buffers) */
int rgbid_kfalign_batched_host(rgbid_kfalign* a, int pairs, const float* depthinv_ini, const unsigned char* grey_ini, const float* depthinv_end,
                               const unsigned char* grey_end, const float* K, double* R, double* t, double* cov);
/* kernel launches of the last call (for DESIGN.md / profiling) and the HBM bytes the aligner allocated */
int rgbid_kfalign_launches(const rgbid_kfalign* a);
int rgbid_kfalign_bytes(const rgbid_kfalign* a, size_t* bytes);

#ifdef __cplusplus
}
#endif
#endif
