/*
 * rgbid.h -- C-ABI of the MI355X-native dense RGB-iD alignment front-end.
 *
 * This is the drop-in boundary: every entry point replaces one host
 * wrapper of the reference's bridge API `namespace RGBID_SLAM::device` (src/internal.h:187-453)
 * and is what a replacement shared library must export.  Signatures are plain C: device
 * pointers + pitch + sizes + scalars, no C++/torch/Eigen types.  include/rgbid/internal.h wraps
 * these back into the reference's own C++ signatures (DeviceArray2D<T>&, Intr, Mat33, float3).
 *
 * Conventions
 *  - rgbid_img.data is DEVICE memory, row-major, `step` bytes between rows (the reference's
 *    PtrStep convention, ThirdParty/pcl_gpu_containers/include/kernel_containers.h:78-92).
 *  - invalid pixels are quiet NaN (src/cuda/utils.hpp:76-77); all maps are fp32.
 *  - R_proj is row-major 3x3 (= reference Mat33, three float3 rows, src/internal.h:166-169).
 *  - Every function returns 0 on success, a positive hipError_t, or a negative RGBID_E_* code;
 *    nothing throws.  `ms` (nullable) receives the elapsed device time in milliseconds, the
 *    reference wrappers' return value (cudaTimer, src/cuda/device.hpp:83-106).
 *  - Calls are synchronous on return (the reference does cudaStreamSynchronize after every
 *    launch) unless the context was switched to asynchronous mode with rgbid_ctx_set_async().
 *  - A context owns a HIP stream + scratch arena; use one context per host thread (the
 *    reference relies on nvcc --default-stream per-thread, CMakeLists.txt:97).
 */
#ifndef RGBID_H_
#define RGBID_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGBID_OK 0
#define RGBID_E_INVALID (-1)   /* bad argument (null pointer, mismatched sizes) */
#define RGBID_E_NOMEM   (-2)
#define RGBID_E_NODEV   (-3)   /* no HIP device / HIP runtime unusable */

typedef struct rgbid_ctx rgbid_ctx;

typedef struct rgbid_img {
  void*  data;   /* device pointer */
  size_t step;   /* bytes between consecutive rows */
  int    rows;
  int    cols;
} rgbid_img;

/* src/internal.h:119-140 Intr (distortion coefficients are unused on this path) */
typedef struct rgbid_intr { float fx, fy, cx, cy; } rgbid_intr;

/* enums of src/internal.h:66-72 */
enum { RGBID_LSQ = 0, RGBID_HUBER, RGBID_TUKEY, RGBID_STUDENT };
enum { RGBID_NO_MM = 0, RGBID_CONSTANT_VELOCITY };
enum { RGBID_SIGMA_MAD = 0, RGBID_SIGMA_PDF, RGBID_SIGMA_CONS };
enum { RGBID_INDEPENDENT = 0, RGBID_MIN_WEIGHT, RGBID_GEOM_ONLY, RGBID_PHOT_ONLY };
enum { RGBID_WARP_FIRST = 0, RGBID_PYR_FIRST };
enum { RGBID_CHI_SQUARED = 0, RGBID_ALL_ITERS };
enum { RGBID_NO_FILTERS = 0, RGBID_FILTER_GRADS };
/* bilinear filter of the intensity warp: exact fp32 weights, or the CUDA texture unit's 1.8
 * fixed-point weights (what the reference's tex2D computes; default) */
enum { RGBID_INTERP_EXACT = 0, RGBID_INTERP_TEX8 = 1 };
/* arithmetic class of the gather / filter kernels, see rgbid_ctx_set_numerics and rgbid_warp_pair */
enum { RGBID_NUMERICS_EXACT = 0, RGBID_NUMERICS_FAST = 1 };

/* ---- library / context ------------------------------------------------------------------- */
const char* rgbid_version(void);
const char* rgbid_error_string(int err);
int rgbid_device_count(int* n);
/* what the reference reads from cudaDeviceProp (tools/RGBID_SLAMapp.cpp:385-387, src/cuda/device.hpp:166-187) */
typedef struct rgbid_device_prop {
  char name[256];
  int multiProcessorCount;          /* compute units */
  int maxThreadsPerMultiProcessor;
  int warpSize;                     /* 64 on CDNA */
  int clockRateKHz;
  size_t totalGlobalMem;
  size_t sharedMemPerBlock;         /* LDS per workgroup */
  char gcnArchName[64];
} rgbid_device_prop;
int rgbid_get_device_prop(int device, rgbid_device_prop* prop);
int rgbid_set_device(int device);   /* pcl::gpu::setDevice, ThirdParty/pcl_gpu_containers/src/initialization.cpp */
/* stream: a hipStream_t to launch on (e.g. torch's current stream) or NULL for a private stream */
int rgbid_ctx_create(rgbid_ctx** ctx, int device, void* stream);
int rgbid_ctx_destroy(rgbid_ctx* ctx);
int rgbid_ctx_set_stream(rgbid_ctx* ctx, void* stream);
int rgbid_ctx_set_async(rgbid_ctx* ctx, int async_on);       /* default 0: synchronous on return */
int rgbid_ctx_get_async(rgbid_ctx* ctx, int* async_on);
int rgbid_ctx_set_interp_mode(rgbid_ctx* ctx, int mode);     /* default RGBID_INTERP_TEX8 */
int rgbid_ctx_get_interp_mode(rgbid_ctx* ctx, int* mode);
int rgbid_ctx_sync(rgbid_ctx* ctx);                          /* internal.h:456-457 sync() */
/* Arithmetic class of the bridge calls that have two implementations (today: rgbid_bilateral_filter; rgbid_warp_pair takes it per call):
 * RGBID_NUMERICS_EXACT (default) = the IEEE evaluation of the oracle, bit for bit; RGBID_NUMERICS_FAST = the reference BUILD's class of
 * arithmetic for the float VALUES (hardware reciprocal / exp2, FMA contraction -- what nvcc --prec-div=false, default fmad and __expf give
 * the reference's own kernels, CMakeLists.txt:105, filters.cu:124) with the oracle's SELECTION: which source pixel a warp point-samples,
 * whether a projection is inside the image, the sign test of a warped value, the covisibility lattice point and the covisibility / fusion
 * gates are decided exactly as the IEEE evaluation decides them (a pixel whose cheap coordinate lies within a proven error bound of a
 * decision threshold is recomputed with the exact instruction sequence: csrc/guard_band.h).  The batched engine selects the class with
 * rgbid_engine_config.fast_numerics.
 * DOMAIN of the FAST class: non-NaN inverse depths of a PROJECTED map (the `grid` / keyframe map of a warp, both maps of a covisibility pair)
 * inside [2^-14, 2^14] m^-1 -- a value outside (zero, negative, infinite, denormal) is treated as invalid, like NaN, where the oracle would
 * project it; non-NaN values of a SAMPLED map 0 or of magnitude in [2^-60, 2^60]; transforms with finite entries below 2^20.  A lane whose
 * motion defeats the sign analysis of the bound (rotations towards 90 degrees, translations of the order of 2^14 scene depths) is not an error:
 * all its pixels take the exact path.  Maps this library builds from 16-bit depth are inside the domain by construction.
 * VALUE tolerance of the class: a selected sample's inverse depth within ~2e-6 relative; a bilinear intensity sample blends the oracle's four texels
 * with weights that may sit one 1.8 fixed-point step (1/256) off the oracle's -- 1 % of the samples at 640x480, at most 2/256 of the local contrast
 * (that rounding has a boundary every 1/256 px, far inside any provable coordinate bound, so it is not guarded: DESIGN.md section 4.1).
 * WHAT "SELECTION-EXACT" MEANS, AND WHAT IT DOES NOT: the statement holds PER CALL, on identical inputs -- every pixel of one warp / gate /
 * covisibility evaluation takes the decision the IEEE evaluation takes for the same maps and the same transform.  It is not a statement about a
 * TRAJECTORY: the two classes' poses differ in their last bits (the values above), the next call's transform therefore differs in its last bits
 * too, and a projected coordinate that sits on a pixel boundary for one of the two transforms then selects another source pixel -- both correctly.
 * Across a depth edge that one sample can move an unconverged iterate visibly: a 6 384-case fuzz campaign found one such engine case (seed 284 of
 * tests/test_gpu_fuzz.py's engine campaign: a 43 x 56 image at a focal length of 46 px, one level, six iterations), 5.3e-5 rad / 1.15e-4 m against
 * the oracle tracker where the EXACT class stays at 2.5e-9 -- a tenth of the estimate's own standard deviation.  The engine fuzz therefore scales the
 * 640 x 480 pose bar of 1e-4 rad / 1e-4 m with the angular size of a pixel below 320 columns (1e-4 * 320 / cols); at and above 320 columns the bar is
 * 1e-4 and every recorded run is one to two orders inside it. */
int rgbid_ctx_set_numerics(rgbid_ctx* ctx, int numerics);
/* orders the context's stream after a hipEvent_t recorded on another stream (interop with the caller's framework streams) */
int rgbid_ctx_wait_event(rgbid_ctx* ctx, void* hip_event);
/* exhaustive device self-test of the kernels' exact reciprocal (csrc/common.h rcp_exact) against IEEE 1.0f/x over all 2^32 float
 * bit patterns; *mismatches must come back 0 */
int rgbid_selftest_rcp(rgbid_ctx* ctx, unsigned long long* mismatches);
/* Same kind of proof for the bilateral filter's per-tap division by its range sigma: the 3-instruction sequence (multiply by the rounded
 * reciprocal + two FMA corrections) is compared with IEEE x / divisor for all 2^32 x; *mismatches counts the x whose short result is
 * flagged usable and differs (0 for the tracker's two sigmas, 2*0.0025 and 3 -- the only ones the filter uses it for: *used_by_filter). */
int rgbid_selftest_div_const(rgbid_ctx* ctx, float divisor, unsigned long long* mismatches, int* used_by_filter);
/* v_cvt_flr_i32_f32 (one-instruction floor + saturating convert of the engine's fast-numerics gather kernels) against v_floor_f32 +
 * v_cvt_i32_f32 over every `stride`-th of the 2^32 float bit patterns (stride 1 = exhaustive; NaN excluded: the two differ there and
 * every index is clamped before it addresses memory); *mismatches must come back 0 */
int rgbid_selftest_cvt_flr(rgbid_ctx* ctx, unsigned stride, unsigned long long* mismatches);
/* The three hardware facts the FAST class's guard band (csrc/guard_band.h) rests on, over all 2^32 float bit patterns: v_rcp_f32 within 1 ulp of
 * 1 / x for every normal x with a normal reciprocal (a denormal x reads as zero); v_med3_f32(x, lo, hi) == x exactly for lo <= x <= hi and a
 * value of [lo, hi] otherwise, NaN included (the sanitising clamp of the domain rule above); v_fract_f32(x) == min(x - floor(x), 1 - 2^-24) for
 * |x| < 2^23.  *mismatches must come back 0. */
int rgbid_selftest_fast_primitives(rgbid_ctx* ctx, unsigned long long* mismatches);
/* the guard-band constants of one projection (host-side evaluation of csrc/guard_band.h make_guard; no device needed; ctx may be NULL):
 * out[0..9] = d1, c2, d2, q0, q1, db, g0, g1, e0, e1; *zsafe = the per-lane verdict of the sign analysis.  For tests and for hosts that want
 * to know in advance whether a transform runs in the guard band's regular regime. */
int rgbid_fast_guard(const float R_proj[9], const float t_proj[3], int cols, int rows, float out[10], int* zsafe);
/* the lane-constant forms the kernels run since round 5 (guard_band.h (3'), (3'')): out[0..3] = bL, cL, kL, wcore -- the point sample is taken at
 * floor(xs'), xs' = Y_0 wc + bL, and is the oracle's whenever max3(fract(xs'), fract(ys'), |wc| kL) < cL; a bilinear sample whose coordinate lies in
 * [0, cols - 1] x [0, rows - 1] with |wc| <= wcore is inside the image for the oracle too.  cL = -1: every pixel of the lane takes the exact path. */
int rgbid_fast_guard_lane(const float R_proj[9], const float t_proj[3], int cols, int rows, float out[4]);
/* the context's hipStream_t */
int rgbid_ctx_get_stream(rgbid_ctx* ctx, void** hip_stream);
/* showGPUMemoryUsage(), src/cuda/misc.cu:526-540 */
int rgbid_mem_info(size_t* free_bytes, size_t* total_bytes);

/* ---- memory (pcl_gpu_containers: src/device_memory.cpp:107-321) -------------------------- */
int rgbid_malloc(void** ptr, size_t bytes);
int rgbid_malloc_pitch(void** ptr, size_t* step, size_t width_bytes, size_t rows); /* step is 256-B aligned */
int rgbid_free(void* ptr);
/* pinned (page-locked) host memory: uploads from it run asynchronously on a stream (frame staging of the batched drivers) */
int rgbid_malloc_host(void** ptr, size_t bytes);
int rgbid_free_host(void* ptr);
int rgbid_memcpy_h2d(rgbid_ctx* ctx, void* dst, const void* src, size_t bytes);
int rgbid_memcpy_d2h(rgbid_ctx* ctx, void* dst, const void* src, size_t bytes);
int rgbid_memcpy_d2d(rgbid_ctx* ctx, void* dst, const void* src, size_t bytes);
int rgbid_memcpy2d_h2d(rgbid_ctx* ctx, void* dst, size_t dstep, const void* src, size_t sstep, size_t width_bytes, size_t rows);
int rgbid_memcpy2d_d2h(rgbid_ctx* ctx, void* dst, size_t dstep, const void* src, size_t sstep, size_t width_bytes, size_t rows);
int rgbid_memcpy2d_d2d(rgbid_ctx* ctx, void* dst, size_t dstep, const void* src, size_t sstep, size_t width_bytes, size_t rows);

/* ---- frame preparation (src/cuda/misc.cu) ------------------------------------------------- */
/* convertDepth2InvDepth misc.cu:365-374: u16 mm -> m^-1, 0 -> NaN */
int rgbid_depth_to_invdepth(rgbid_ctx*, const rgbid_img* depth_u16, const rgbid_img* dst, float factor_depth);
/* computeIntensity misc.cu:377-386: packed r,g,b bytes -> luma */
int rgbid_compute_intensity(rgbid_ctx*, const rgbid_img* rgb_u8x3, const rgbid_img* dst);
/* decomposeRGBInChannels misc.cu:388-397 */
int rgbid_decompose_rgb(rgbid_ctx*, const rgbid_img* rgb_u8x3, const rgbid_img* r, const rgbid_img* g, const rgbid_img* b);
/* computeGradientIntensity / computeGradientDepth misc.cu:400-441 (same kernel) */
int rgbid_compute_gradient(rgbid_ctx*, const rgbid_img* src, const rgbid_img* dst_hor, const rgbid_img* dst_vert, float* ms);
/* copyImages :445-455, copyImage :458-467, copyImageRGB :469-478 */
int rgbid_copy_images(rgbid_ctx*, const rgbid_img* src_depth, const rgbid_img* src_int, const rgbid_img* dst_depth, const rgbid_img* dst_int);
int rgbid_copy_image(rgbid_ctx*, const rgbid_img* src, const rgbid_img* dst);
int rgbid_copy_image_rgb(rgbid_ctx*, const rgbid_img* src_u8x3, const rgbid_img* dst_u8x3);
/* initialiseWeightKeyframe misc.cu:480-489 */
int rgbid_init_weight_keyframe(rgbid_ctx*, const rgbid_img* src_depth, const rgbid_img* dst_weight);
/* initialiseDeviceMemory2D<T> misc.cu:491-512; elem_size in {1,4}; value is the raw bit pattern */
int rgbid_fill_2d(rgbid_ctx*, const rgbid_img* img, int elem_size, uint32_t value_bits);

/* ---- pyramid (src/cuda/pyrdown.cu:194-242): dst must be (rows/2) x (cols/2) --------------- */
int rgbid_pyr_down(rgbid_ctx*, const rgbid_img* src, const rgbid_img* dst, float* ms);

/* ---- bilateral (src/cuda/filters.cu:139-162) --------------------------------------------- */
int rgbid_bilateral_filter(rgbid_ctx*, const rgbid_img* src, const rgbid_img* dst, float sigma_floatmap, float* ms);

/* ---- custom-calibration front-end (src/cuda/undistortion.cu, warping_registration.cu:720-822); custom_registration=1 only */
typedef struct rgbid_intr_k { float fx, fy, cx, cy, k1, k2, k3, k4, k5; } rgbid_intr_k;            /* Intr, src/internal.h:119-140 */
typedef struct rgbid_depth_dist { float c1, c0, q0[9], q1[9]; int xshift, yshift; } rgbid_depth_dist; /* DepthDist, :142-161 */
/* undistortIntensity undistortion.cu:214-243 (bilinear fetch, filter model = ctx interp mode) */
int rgbid_undistort_intensity(rgbid_ctx*, const rgbid_img* src, const rgbid_img* dst, const rgbid_intr_k* intr, float* ms);
/* undistortDepthInv undistortion.cu:246-312: src_corr receives the depth-distortion-corrected map, dst its undistortion */
int rgbid_undistort_depthinv(rgbid_ctx*, const rgbid_img* src, const rgbid_img* src_corr, const rgbid_img* dst,
                             const rgbid_intr_k* intr_depth, const rgbid_depth_dist* dp, float* ms);
/* registerDepthinv warping_registration.cu:720-822: intermediate (float) and intermediate_as_int (int32) are the enlarged
 * (3 rows x 3 cols in the reference) scratch images of the translation splat */
int rgbid_register_depthinv(rgbid_ctx*, const rgbid_img* src, const rgbid_img* intermediate, const rgbid_img* intermediate_as_int,
                            const rgbid_img* dst, const float dRc_proj[9], const float t_dc_proj[3], const float cRd_proj[9], float* ms);

/* ---- warps, fusion, visibility (src/cuda/warping_registration.cu) ------------------------ */
/* warpInvDepthWithTrafo3D :971-1019 */
int rgbid_warp_invdepth(rgbid_ctx*, const rgbid_img* src, const rgbid_img* dst, const rgbid_img* depthinv_prev,
                        const float R_proj[9], const float t_proj[3], float* ms);
/* warpIntensityWithTrafo3DInvDepth :920-967 */
int rgbid_warp_intensity(rgbid_ctx*, const rgbid_img* src, const rgbid_img* dst, const rgbid_img* depthinv_prev,
                         const float R_proj[9], const float t_proj[3], float* ms);
/* Both warps of one Gauss-Newton iteration in one launch (the pair the tracker issues back to back, src/visodo.cpp:1094-1100): dst_iD =
 * warpInvDepthWithTrafo3D(src_iD, grid), dst_I = warpIntensityWithTrafo3DInvDepth(src_I, dst_iD) with the warped inverse depth consumed from
 * registers.  numerics EXACT: bit-identical to the two calls above (IEEE evaluation of the oracle); FAST: the reference BUILD's class of
 * arithmetic for the values -- hardware reciprocal + FMA contraction, what nvcc --prec-div=false and default fmad give the reference's own
 * kernels (CMakeLists.txt:105) -- with the oracle's pixel selection and validity at every pixel (rgbid_ctx_set_numerics above; what the batched
 * engine runs by default, rgbid_engine_config.fast_numerics). */
int rgbid_warp_pair(rgbid_ctx*, const rgbid_img* src_iD, const rgbid_img* src_I, const rgbid_img* grid_iD, const rgbid_img* dst_iD, const rgbid_img* dst_I,
                    const float R_proj[9], const float t_proj[3], int numerics, float* ms);
/* warpInvDepthWithTrafo3DWeighted :1021-1069 */
int rgbid_warp_invdepth_weighted(rgbid_ctx*, const rgbid_img* src, const rgbid_img* dst, const rgbid_img* depthinv_prev,
                                 const rgbid_img* weight_warped, const float R_proj[9], const float t_proj[3], float* ms);
/* integrateWarpedFrame :1072-1095 */
int rgbid_integrate_warped_frame(rgbid_ctx*, const rgbid_img* warped_depthinv, const rgbid_img* warped_weight,
                                 const rgbid_img* depthinv_dst, const rgbid_img* weight_dst, float* ms);
/* getVisibilityRatio :825-869 / getVisibilityRatioWithOverlapMask :873-913 (overlap_mask nullable, u8) */
int rgbid_visibility_ratio(rgbid_ctx*, const rgbid_img* depthinv_src, const rgbid_img* depthinv_dst,
                           const float R_proj[9], const float t_proj[3], const rgbid_img* overlap_mask,
                           float* visibility_ratio, float* ms);

/* ---- bridge functions the reference defines but its tracker no longer calls (kept so the C++ surface is complete) ---- */
/* convertDepth2Float misc.cu:353-362 (u16 mm -> metres) ; convertFloat2RGB misc.cu:514-523 (grey visualisation, NaN/inf colour coded) */
int rgbid_depth_to_float(rgbid_ctx*, const rgbid_img* depth_u16, const rgbid_img* dst);
int rgbid_float_to_rgb(rgbid_ctx*, const rgbid_img* src, const rgbid_img* dst_u8x3);
/* createNMap maps.cu:346-393 (normals = normalised cross product of forward differences of the vertex map) */
int rgbid_create_nmap(rgbid_ctx*, const rgbid_img* vmap, const rgbid_img* nmap);
/* integrateWarpedRGB warping_registration.cu:1097-1129 (inverse-depth + colour fusion, gate 0.0075 each way) */
int rgbid_integrate_warped_rgb(rgbid_ctx*, const rgbid_img* warped_depthinv, const rgbid_img* r, const rgbid_img* g, const rgbid_img* b,
                               const rgbid_img* warped_weight, const rgbid_img* depthinv_dst, const rgbid_img* colors_dst_u8x3,
                               const rgbid_img* weight_dst, float* ms);

/* ---- vertex / normal maps (src/cuda/maps.cu:300-344, 396-443); planar 3*rows x cols ------ */
int rgbid_create_vmap(rgbid_ctx*, rgbid_intr intr, const rgbid_img* depthinv, const rgbid_img* vmap);
int rgbid_create_nmap_gradients(rgbid_ctx*, rgbid_intr intr, const rgbid_img* depthinv, const rgbid_img* grad_x,
                                const rgbid_img* grad_y, const rgbid_img* nmap);
/* generateImageRGB image_generator.cu:206-224 (rgb nullable -> generateImage :187-203) */
int rgbid_generate_image(rgbid_ctx*, const rgbid_img* vmap, const rgbid_img* nmap, const rgbid_img* rgb_u8x3,
                         const float light_pos[3], const rgbid_img* dst_u8x3);

/* ---- residual lattice + scale estimation (src/cuda/sigmaFuncs.cu) ------------------------ */
/* lattice geometry of computeErrorGridStride :701-765 (host-side, no device work) */
int rgbid_error_lattice_size(int rows, int cols, int min_nsamples, int* n_samples, int* lat_rows, int* lat_cols, int* stride);
/* computeErrorGridStride: error must hold n_samples floats (device) */
int rgbid_compute_error(rgbid_ctx*, const rgbid_img* im1, const rgbid_img* im0, float* error_dev, int min_nsamples,
                        int* n_samples, float* ms);
/* computeSigmaAndNuStudent :858-1066; bias/sigma/nu are host in/out */
int rgbid_sigma_nu_student(rgbid_ctx*, const float* error_dev, int n, float* bias, float* sigma, float* nu,
                           int mestimator, float* ms);
/* computeNuStudent :1068-1222 */
int rgbid_nu_student(rgbid_ctx*, const float* error_dev, int n, float bias, float sigma, float* nu, float* ms);
/* computeSigmaPdf :773-854 */
int rgbid_sigma_pdf(rgbid_ctx*, const float* error_dev, int n, float* bias, float* sigma, int mestimator, float* ms);
/* computeChiSquare :1225-1297 */
int rgbid_chi_square(rgbid_ctx*, const float* error_int_dev, const float* error_depth_dev, int n, float sigma_int,
                     float sigma_depth, int mestimator, float* chi_square, float* chi_test, float* ndof, float* ms);

/* ---- normal equations (src/cuda/estimate_VO.cu) ------------------------------------------ */
/* buildSystemGridStride :505-645.  A: 6x6 row-major symmetric (host), b: 6 (host). */
int rgbid_build_system(rgbid_ctx*, const rgbid_img* W0, const rgbid_img* I0, const rgbid_img* gradW0_x,
                       const rgbid_img* gradW0_y, const rgbid_img* gradI0_x, const rgbid_img* gradI0_y,
                       const rgbid_img* W1, const rgbid_img* I1, int mestimator, int weighting,
                       float sigma_depthinv, float sigma_int, float bias_depthinv, float bias_int,
                       rgbid_intr intr, double A[36], double b[6], float* ms);
/* buildSystemStudentNuGridStride :649-789 */
int rgbid_build_system_student_nu(rgbid_ctx*, const rgbid_img* W0, const rgbid_img* I0, const rgbid_img* gradW0_x,
                                  const rgbid_img* gradW0_y, const rgbid_img* gradI0_x, const rgbid_img* gradI0_y,
                                  const rgbid_img* W1, const rgbid_img* I1, int mestimator, int weighting,
                                  float sigma_depthinv, float sigma_int, float bias_depthinv, float bias_int,
                                  float nu_depthinv, float nu_int, rgbid_intr intr, double A[36], double b[6], float* ms);

#ifdef __cplusplus
}
#endif
#endif /* RGBID_H_ */
