/*
 * rgbid_batched.h -- batched C-ABI of the hot-path kernels ("batch variants (`..._batched` with arrays of
 * images)").
 *
 * rgbid.h exports one entry point per reference bridge function, one image per call.  The batched engine (rgbid_engine.h) runs the
 * same per-frame algorithm for many independent trackers and does so with kernels of its own -- the fused Gauss-Newton evaluation, the
 * lattice pre-pass + sigma / nu pair, the one-pass keyframe fusion, vertex + normal maps, the two-direction covisibility and the
 * one-pass frame preparation -- plus natively batched forms of the bridge kernels.  This header exports every one of them as a
 * single call over `lanes` images, so that a host that schedules its own frame pairs (SURVEY.md section 8e "U-pair") can call them
 * directly, and so that each can be held to the CPU oracle on its own (tests/test_gpu_batched.py): what the engine times is what these
 * functions launch.
 *
 * Conventions (rgbid.h applies: device pointers, fp32 maps, quiet NaN = invalid, row-major R_proj, 0 / hipError_t / negative RGBID_E_*):
 *  - rgbid_imgb = `lanes` images of identical geometry in ONE allocation, lane l at data + l * lane_stride bytes (the engine's
 *    structure-of-lanes layout); lane_stride is ignored when lanes == 1.
 *  - per-lane parameters (transforms, noise scales) are HOST arrays of `lanes` entries; results (A, b, counts, scales) are HOST arrays.
 *  - numerics: RGBID_NUMERICS_EXACT = the IEEE evaluation of the oracle (bit-exact values and selection); RGBID_NUMERICS_FAST = the
 *    reference build's class of arithmetic for the values with the oracle's pixel SELECTION, validity and gates (rgbid.h
 *    rgbid_ctx_set_numerics, which also states the class's DOMAIN: projected inverse depths in [2^-14, 2^14] or invalid).  FAST needs rows of
 *    whole 4-pixel groups, 16-byte aligned rows and lanes, and one pitch for the six keyframe-side maps: a FAST call on any other geometry
 *    returns RGBID_E_INVALID (it never silently runs the other class).
 *  - the one-pass kernels marked "16-byte geometry" need cols % 4 == 0 and 16-byte aligned data / step / lane_stride and return
 *    RGBID_E_INVALID otherwise; a caller with odd geometry uses the kernel sequence of rgbid.h, as the engine does.
 *  - synchronous on return unless the context is asynchronous (rgbid_ctx_set_async: the `_async` form of every call below; results
 *    that travel to HOST arrays always synchronise).  `ms` (nullable) = elapsed device time of the call's kernels.
 */
#ifndef RGBID_BATCHED_H_
#define RGBID_BATCHED_H_

#include "rgbid.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rgbid_imgb {
  void*  data;          /* device pointer of lane 0 */
  size_t step;          /* bytes between rows */
  size_t lane_stride;   /* bytes between lanes */
  int    rows, cols;
} rgbid_imgb;

/* per-lane inputs of constraintsHandler (src/cuda/estimate_VO.cu:95-139): what buildSystem(StudentNu)GridStride take as scalars */
typedef struct rgbid_sys_params {
  float sigma_depthinv, sigma_int, bias_depthinv, bias_int, nu_depthinv, nu_int;
  int mestimator, weighting;
  int student_nu;        /* 1: buildSystemStudentNuGridStride (:649-789), 0: buildSystemGridStride (:505-645, fixed-nu M-estimators) */
  int nu_int_from_max;   /* 1: nu_int := max(nu_int, nu_depthinv) first, as src/visodo.cpp:1186 does after its two sigma calls */
} rgbid_sys_params;

/* kernel variant of the fused Gauss-Newton evaluation: what the caller guarantees about EVERY lane's parameters so that the
 * per-pixel code carries no configuration branches (same arithmetic in every variant) */
enum {
  RGBID_WM_AUTO = -1,     /* pick 1 / 2 when every lane qualifies, else 0 */
  RGBID_WM_GENERIC = 0,   /* decided per pixel from the lane's parameters */
  RGBID_WM_STUDENT_NU = 1,/* student_nu set, weighting != MIN_WEIGHT, RGBID_INTERP_TEX8: the Gauss-Newton iterations of the shipped ini */
  RGBID_WM_STUDENT_FIXED = 2 /* student_nu clear, mestimator STUDENT, weighting != MIN_WEIGHT: the covariance pass (visodo.cpp:1349-1365) */
};

/* ---- Gauss-Newton normal equations ------------------------------------------------------------------------------------------------ */
/* One fused Gauss-Newton evaluation per lane: trafo3DKernelInvDepthGridStride + trafo3DKernelIntensityWithInvDepthGridStride
 * (src/cuda/warping_registration.cu:465-546) + computeStudentNuSystemGridStride / computeSystemGridStride + FinalReductionKernel
 * (src/cuda/estimate_VO.cu:354-500) in ONE kernel -- W1 = warp(Wcur onto the W0 grid), I1 = warp(Icur sampled at W1) are formed in
 * registers and never stored.  A[lane]: 6x6 row-major symmetric, b[lane]: 6 (host).  What the engine launches once per iteration. */
int rgbid_gn_fused_batched(rgbid_ctx*, int lanes, const rgbid_imgb* W0, const rgbid_imgb* I0, const rgbid_imgb* gradW0_x,
                           const rgbid_imgb* gradW0_y, const rgbid_imgb* gradI0_x, const rgbid_imgb* gradI0_y,
                           const rgbid_imgb* Wcur, const rgbid_imgb* Icur, const float* R_proj /* [lanes][9] */,
                           const float* t_proj /* [lanes][3] */, rgbid_intr intr, const rgbid_sys_params* params /* [lanes] */,
                           int numerics, int weight_mode, double* A /* [lanes][36] */, double* b /* [lanes][6] */, float* ms);
/* buildSystem(StudentNu)GridStride (estimate_VO.cu:505-789) on stored W1 / I1 for `lanes` frame pairs (the unfused kernel sequence) */
int rgbid_build_system_batched(rgbid_ctx*, int lanes, const rgbid_imgb* W0, const rgbid_imgb* I0, const rgbid_imgb* gradW0_x,
                               const rgbid_imgb* gradW0_y, const rgbid_imgb* gradI0_x, const rgbid_imgb* gradI0_y,
                               const rgbid_imgb* W1, const rgbid_imgb* I1, rgbid_intr intr, const rgbid_sys_params* params,
                               double* A, double* b, float* ms);
/* both warps of one Gauss-Newton iteration for `lanes` pairs (rgbid_warp_pair of rgbid.h, batched) */
int rgbid_warp_pair_batched(rgbid_ctx*, int lanes, const rgbid_imgb* src_iD, const rgbid_imgb* src_I, const rgbid_imgb* grid_iD,
                            const rgbid_imgb* dst_iD, const rgbid_imgb* dst_I, const float* R_proj, const float* t_proj,
                            int numerics, float* ms);

/* ---- residual lattice + scale estimation of the fused path (src/cuda/sigmaFuncs.cu:701-765, 858-1066) -------------------------------- */
/* keyframe side of a level's lattice packed once per keyframe: out_dev[lane][2][n] = W0 | I0 at the n lattice points
 * (n = rgbid_error_lattice_size(rows, cols, min_nsamples)); out_lane_stride in floats, >= 2 n */
int rgbid_lattice_pack_batched(rgbid_ctx*, int lanes, const rgbid_imgb* W0, const rgbid_imgb* I0, int min_nsamples, float* out_dev,
                               size_t out_lane_stride, float* ms);
/* computeErrorGridStride of BOTH channels with the warped maps produced on the fly: res_dev[lane][2][n] = (W1 - W0) | (I1 - I0) at
 * the lattice points.  kf_lat_dev (nullable) = the packed keyframe side from rgbid_lattice_pack_batched. */
int rgbid_lattice_residuals_batched(rgbid_ctx*, int lanes, const rgbid_imgb* Wcur, const rgbid_imgb* W0, const rgbid_imgb* Icur,
                                    const rgbid_imgb* I0, const float* R_proj, const float* t_proj, int min_nsamples, int numerics,
                                    const float* kf_lat_dev, size_t kf_lat_lane_stride, float* res_dev, size_t res_lane_stride,
                                    float* ms);
/* computeSigmaAndNuStudent (:858-1066) of both channels of every lane from res_dev[lane][2][n], with the start values the tracker sets
 * before every iteration (bias 0, sigma 0.0025 / 5, nu 5; src/visodo.cpp:1168-1173); out (host) [lanes] */
typedef struct rgbid_scale_pair { float bias_depthinv, sigma_depthinv, nu_depthinv, bias_int, sigma_int, nu_int; } rgbid_scale_pair;
int rgbid_sigma_pair_batched(rgbid_ctx*, int lanes, const float* res_dev, size_t res_lane_stride, int n, int mestimator,
                             rgbid_scale_pair* out, float* ms);

/* ---- keyframe fusion / maps / covisibility -------------------------------------------------------------------------------------------- */
/* warpInvDepthWithTrafo3DWeighted + integrateWarpedFrame (warping_registration.cu:549-669) in one pass: kf_depthinv / kf_weight updated
 * in place; EXACT: warped_weight left exactly as the two kernels leave it; FAST: warped_weight is neither read nor written (it only exists between
 * the reference's two kernels) -- a warped value whose weight is not positive, i.e. an infinite intermediate, fuses with weight 0.  16-byte geometry. */
int rgbid_fuse_frame_batched(rgbid_ctx*, int lanes, const rgbid_imgb* cur_depthinv, const rgbid_imgb* kf_depthinv,
                             const rgbid_imgb* kf_weight, const rgbid_imgb* warped_weight, const float* R_proj, const float* t_proj,
                             int numerics, float* ms);
/* createVMap + computeGradientDepth + createNMapGradients (maps.cu:63-179, misc.cu:176-220) of one map in one pass; vmap / nmap planar
 * 3*rows x cols.  Planes 1 / 2 of an invalid pixel (plane 0 = NaN) are written as NaN (the reference leaves them untouched).
 * 16-byte geometry. */
int rgbid_kf_maps_batched(rgbid_ctx*, int lanes, rgbid_intr intr, const rgbid_imgb* depthinv, const rgbid_imgb* vmap,
                          const rgbid_imgb* nmap, float* ms);
/* both directions of computeCovisibility (visodo.cpp:1481-1514; partialVisibilityKernel warping_registration.cu:297-360) between maps a
 * and b: counts (host) [lanes][4] = {visible a->b, valid a, visible b->a, valid b} -- exact integer counts */
int rgbid_visibility_pair_batched(rgbid_ctx*, int lanes, const rgbid_imgb* a, const rgbid_imgb* b, const float* R_ab, const float* t_ab,
                                  const float* R_ba, const float* t_ba, int numerics, unsigned int* counts, float* ms);

/* ---- frame preparation ---------------------------------------------------------------------------------------------------------------- */
/* convertDepth2InvDepth + computeIntensity + decomposeRGBInChannels (misc.cu:105-172) in one pass (any geometry) */
int rgbid_prep_frame_batched(rgbid_ctx*, int lanes, const rgbid_imgb* depth_u16, const rgbid_imgb* rgb_u8x3, const rgbid_imgb* depthinv,
                             const rgbid_imgb* intensity, const rgbid_imgb* r, const rgbid_imgb* g, const rgbid_imgb* b,
                             float factor_depth, float* ms);
/* pyrDownIntensity / pyrDownDepth (pyrdown.cu:194-242): dst lanes are (rows/2) x (cols/2) */
int rgbid_pyr_down_batched(rgbid_ctx*, int lanes, const rgbid_imgb* src, const rgbid_imgb* dst, float* ms);
/* computeGradientIntensity / computeGradientDepth (misc.cu:400-441) */
int rgbid_compute_gradient_batched(rgbid_ctx*, int lanes, const rgbid_imgb* src, const rgbid_imgb* dst_hor, const rgbid_imgb* dst_vert,
                                   float* ms);
/* the same Sobel pair + a copy of src into `keep` in ONE pass: what the engine runs at a keyframe switch, where copyImages (visodo.cpp:837-841) is followed
 * by computeGradient* of the copied map (:862-877).  Takes the 16-byte path only: returns RGBID_E_INVALID for a geometry it does not cover (cols % 4 != 0,
 * rows / lanes not 16-byte aligned) -- the engine then copies and takes the gradient as two calls */
int rgbid_gradient_keep_batched(rgbid_ctx*, int lanes, const rgbid_imgb* src, const rgbid_imgb* dst_hor, const rgbid_imgb* dst_vert,
                                const rgbid_imgb* keep, float* ms);
/* bilateralFilter (filters.cu:139-162); numerics FAST = the engine's v_exp_f32 kernel */
int rgbid_bilateral_filter_batched(rgbid_ctx*, int lanes, const rgbid_imgb* src, const rgbid_imgb* dst, float sigma_floatmap,
                                   int numerics, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* RGBID_BATCHED_H_ */
