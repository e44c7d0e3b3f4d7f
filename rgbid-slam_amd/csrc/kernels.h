// kernels.h -- internal (C++) launch API shared by the C-ABI wrappers (c_api.hip) and the batched
// engine (engine.hip).  All launchers are asynchronous on `stream`; `B` is the number of lanes.
#pragma once
#include "common.h"

namespace rgbid {

// ---- per-lane parameter blocks ---------------------------------------------------------------
struct SysParams {     // inputs of constraintsHandler (estimate_VO.cu:95-139)
  float fx, fy, cx, cy;
  float sigma_d, sigma_i, bias_d, bias_i, nu_d, nu_i;
  int mestimator, weighting, student_nu;
  int nu_i_max;   // engine: nu_i := max(nu_i, nu_d) as visodo.cpp:1186 does after the two sigma calls
};
struct SigmaIO { float bias, sigma, nu; };
struct IntrP { float fx, fy, cx, cy; };
struct LightP { float x, y, z; };
struct IntrK { float fx, fy, cx, cy, k1, k2, k3, k4, k5; };                      // Intr with distortion, src/internal.h:119-140
struct DepthDistP { float c1, c0, q0[9], q1[9]; int xshift, yshift; };          // DepthDist, src/internal.h:142-161

enum { SYS_TERMS = 27 };
static constexpr float TH_HUBER = 1.345f, TH_TUKEY = 4.685f, STUDENT_DOF = 5.f;   // computeWeight estimate_VO.cu:141-167, sigmaFuncs.cu:179-255

// ---- custom-calibration front-end (kernels_calib.hip) -------------------------------------------
void launch_undistort(hipStream_t s, int B, ImgB src, ImgB dst, IntrK k, bool linear, int interp_mode, LaneMask m);
void launch_depthinv_correction(hipStream_t s, int B, ImgB src, ImgB dst, IntrK k, DepthDistP dp, LaneMask m);
void launch_register_depthinv(hipStream_t s, int B, ImgB src, ImgB inter_f, ImgB inter_i, ImgB dst, const float dRc_proj[9], const float t_dc_proj[3],
                              const float cRd_proj[9], LaneMask m);

// ---- prep (kernels_prep.hip) -------------------------------------------------------------------
void launch_depth_to_invdepth(hipStream_t s, int B, ImgB src_u16, ImgB dst, float factor_depth, LaneMask m);
void launch_intensity(hipStream_t s, int B, ImgB rgb, ImgB dst, LaneMask m);
void launch_depth_to_float(hipStream_t s, int B, ImgB src_u16, ImgB dst, LaneMask m);
void launch_float_to_rgb(hipStream_t s, int B, ImgB src, ImgB dst_rgb, LaneMask m);
// engine: iD warp + intensity warp (sampled on the warped iD) of one GN iteration in one launch, bit-identical to the two kernels
void launch_warp_pair(hipStream_t s, int B, ImgB src_iD, ImgB src_I, ImgB grid, ImgB dst_iD, ImgB dst_I, const WarpParams* lane_params, int interp_mode, LaneMask m,
                      const WarpParams* host_p = nullptr);
// same in the reference build's class of arithmetic (engine, fast_numerics); false: geometry outside the tile map's range, nothing launched
bool launch_warp_pair_fast(hipStream_t s, int B, ImgB src_iD, ImgB src_I, ImgB grid, ImgB dst_iD, ImgB dst_I, const WarpParams* host_p, const WarpParams* lane_params,
                           int interp_mode, LaneMask m);
void launch_selftest_cvt_flr(hipStream_t s, unsigned long long* mismatches_dev, unsigned stride);
void launch_nmap_cross(hipStream_t s, int B, ImgB vmap, ImgB nmap, LaneMask m);
void launch_integrate_warped_rgb(hipStream_t s, int B, ImgB warped, ImgB r, ImgB g, ImgB b, ImgB wweight, ImgB kf, ImgB colors, ImgB kfw, LaneMask m);
void launch_decompose_rgb(hipStream_t s, int B, ImgB rgb, ImgB r, ImgB g, ImgB b, LaneMask m);
void launch_gradient(hipStream_t s, int B, ImgB src, ImgB gx, ImgB gy, LaneMask m);
bool launch_gradient_keep(hipStream_t s, int B, ImgB src, ImgB gx, ImgB gy, ImgB keep, LaneMask m);   // Sobel pair + copy of src into keep; false: not launched
// depth->iD + rgb->luma + rgb->r,g,b planes in one pass (engine); falls back to the three kernels when not 16-byte aligned
void launch_prep_frame(hipStream_t s, int B, ImgB depth_u16, ImgB rgb, ImgB iD, ImgB I, ImgB r, ImgB g, ImgB b, float factor_depth, LaneMask m);
void launch_copy_bytes(hipStream_t s, int B, ImgB src, ImgB dst, int elem_size, LaneMask m);  // row-wise D2D copy kernel
void launch_fill(hipStream_t s, int B, ImgB dst, int elem_size, uint32_t bits, LaneMask m);
void launch_pyr_down(hipStream_t s, int B, ImgB src, ImgB dst, LaneMask m);
// Two maps of the same geometry in ONE launch (the intensity / inverse-depth pair of a pyramid level; the two covisibility checks of a frame; every
// level's lattice pack): below ~100 lanes a step is bound by its number of dependent launches.  Same kernels, same per-map results.
void launch_pyr_down2(hipStream_t s, int B, ImgB src0, ImgB dst0, ImgB src1, ImgB dst1, LaneMask m);
void launch_gradient2(hipStream_t s, int B, ImgB src0, ImgB gx0, ImgB gy0, ImgB src1, ImgB gx1, ImgB gy1, LaneMask m);
bool launch_gradient_keep2(hipStream_t s, int B, ImgB src0, ImgB gx0, ImgB gy0, ImgB keep0, ImgB src1, ImgB gx1, ImgB gy1, ImgB keep1, LaneMask m);
void launch_bilateral2(hipStream_t s, int B, ImgB src0, ImgB dst0, float sigma0, ImgB src1, ImgB dst1, float sigma1, LaneMask m, bool fast = false);
void launch_lattice_pack_levels(hipStream_t s, int B, int n_levels, const ImgB* W0, const ImgB* I0, int min_nsamples, float* const* out, size_t out_lane_stride, LaneMask m);
void launch_visibility_pair2(hipStream_t s, int B, ImgB a, ImgB b0, const WarpParams* p_ab0, const WarpParams* p_ba0, unsigned int* counts_ab0, unsigned int* counts_ba0,
                             ImgB b1, const WarpParams* p_ab1, const WarpParams* p_ba1, unsigned int* counts_ab1, unsigned int* counts_ba1, LaneMask m, bool fast);
void launch_bilateral(hipStream_t s, int B, ImgB src, ImgB dst, float sigma_floatmap, LaneMask m, bool fast = false);
bool div_const_verified(float c);   // constants for which the bilateral filter uses the short exact division
void launch_selftest_div_const(hipStream_t s, float c, unsigned long long* mismatches_dev);

// ---- warps / fusion / maps (kernels_warp.hip) ---------------------------------------------------
// params: host pointer (by value) when lane_params == nullptr, else device array [B]
void launch_warp_invdepth(hipStream_t s, int B, ImgB src, ImgB grid, ImgB dst, const WarpParams* host_p, const WarpParams* lane_p, LaneMask m);
void launch_warp_intensity(hipStream_t s, int B, ImgB src, ImgB grid, ImgB dst, const WarpParams* host_p, const WarpParams* lane_p, int interp_mode, LaneMask m);
void launch_warp_invdepth_weighted(hipStream_t s, int B, ImgB src, ImgB grid, ImgB dst, ImgB weight, const WarpParams* host_p, const WarpParams* lane_p, LaneMask m);
void launch_integrate_warped(hipStream_t s, int B, ImgB warped, ImgB wweight, ImgB kf, ImgB kfweight, LaneMask m);
// counts[lane*2+0] = visible, [lane*2+1] = valid (float, must be zeroed by the caller); mask.base nullable
void launch_visibility(hipStream_t s, int B, ImgB src, ImgB dst, ImgB mask, const WarpParams* host_p, const WarpParams* lane_p, unsigned int* counts, LaneMask m);
// engine: both directions of computeCovisibility between two maps in one pass (counts_ab: a projected into b; counts_ba: b into a)
void launch_visibility_pair(hipStream_t s, int B, ImgB a, ImgB b, const WarpParams* p_ab, const WarpParams* p_ba, unsigned int* counts_ab,
                            unsigned int* counts_ba, LaneMask m, bool fast = false);
void launch_vmap(hipStream_t s, int B, ImgB depthinv, ImgB vmap, IntrP k, LaneMask m);
void launch_nmap_gradients(hipStream_t s, int B, ImgB depthinv, ImgB gx, ImgB gy, ImgB nmap, IntrP k, LaneMask m);
// engine: warpInvDepthWithTrafo3DWeighted + integrateWarpedFrame in one pass (false: layout not 16-byte friendly, nothing launched)
bool launch_fuse_frame(hipStream_t s, int B, ImgB src, ImgB kf, ImgB kfw, ImgB wweight, const WarpParams* lane_params, LaneMask m, bool fast = false);
// engine: createVMap + computeGradientDepth + createNMapGradients of one map in one pass (false: layout not 16-byte friendly, nothing launched)
bool launch_kf_maps(hipStream_t s, int B, ImgB depthinv, ImgB vmap, ImgB nmap, IntrP k, LaneMask m);
void launch_generate_image(hipStream_t s, int B, ImgB vmap, ImgB nmap, ImgB rgb, ImgB dst, const LightP* host_l, const LightP* lane_l, LaneMask m);

// ---- residual lattice + sigma/nu (kernels_sigma.hip) --------------------------------------------
void lattice_geometry(int rows, int cols, int min_nsamples, int* n, int* lrows, int* lcols, int* stride);
void launch_error_lattice(hipStream_t s, int B, ImgB im1, ImgB im0, float* err, size_t err_lane_stride, int lrows, int lcols, int stride, LaneMask m);
// mode 0: computeSigmaAndNuStudent, 1: computeNuStudent, 2: computeSigmaPdf.  io: device [B]
void launch_sigma(hipStream_t s, int B, int mode, const float* err, size_t err_lane_stride, int n, SigmaIO* io, int mestimator, LaneMask m);
// engine: lattice sampling + computeSigmaAndNuStudent for both channels of every lane, results into sp[lane]
void launch_sigma_pair(hipStream_t s, int B, ImgB W1, ImgB W0, ImgB I1, ImgB I0, int min_nsamples, SysParams* sp, int mestimator, LaneMask m);
// same, but W1 / I1 are produced on the fly from the current frame (fused engine path: they are never stored)
void launch_sigma_pair_fused(hipStream_t s, int B, ImgB Wcur, ImgB W0, ImgB Icur, ImgB I0, const WarpParams* lane_wp, int interp_mode,
                             int min_nsamples, SysParams* sp, int mestimator, LaneMask m, bool fast = false, float* res = nullptr, size_t res_lane_stride = 0,
                             const float* kf_lat = nullptr, size_t kf_lat_lane_stride = 0);
// the two halves of launch_sigma_pair_fused on their own (C-ABI rgbid_lattice_residuals_batched / rgbid_sigma_pair_batched)
void launch_lattice_residuals_fused(hipStream_t s, int B, ImgB Wcur, ImgB W0, ImgB Icur, ImgB I0, const WarpParams* lane_wp, int interp_mode, int min_nsamples,
                                    LaneMask m, bool fast, float* res, size_t res_lane_stride, const float* kf_lat, size_t kf_lat_lane_stride);
void launch_sigma_pair_arrays(hipStream_t s, int B, const float* res, size_t res_lane_stride, int n, SysParams* sp, int mestimator, LaneMask m);
// the keyframe side of a level's residual lattice packed as [lane][2][n] = W0 | I0 (once per keyframe; see k_lattice_residuals_fused)
void launch_lattice_pack(hipStream_t s, int B, ImgB W0, ImgB I0, int min_nsamples, float* out, size_t out_lane_stride, LaneMask m);
// samples of the residual lattice computeErrorGridStride takes for this geometry (scratch sizing: 2 * n floats per lane for `res` above)
int lattice_samples(int rows, int cols, int min_nsamples);
// out: device [B][3] = chi_square, chi_test, ndof
void launch_chi_square(hipStream_t s, int B, const float* err_int, const float* err_depth, size_t err_lane_stride, int n,
                       float sigma_int, float sigma_depth, int mestimator, float* out, LaneMask m);

// ---- normal equations (kernels_system.hip) ------------------------------------------------------
// number of partial-sum blocks per lane the build-system kernel will use for this geometry
int system_blocks_per_lane(int rows, int cols, int B);
// partials: device double [B][nblk][27] (size it with system_blocks_per_lane); returns the nblk used.
// sums: device double [B][27] (packed upper-tri + b, reference order estimate_VO.cu:774-786)
int launch_build_system(hipStream_t s, int B, ImgB W0, ImgB I0, ImgB gWx, ImgB gWy, ImgB gIx, ImgB gIy, ImgB W1, ImgB I1,
                        const SysParams* host_p, const SysParams* lane_p, double* partials, LaneMask m, int level_tag = 0);
// fused Gauss-Newton evaluation (engine): warp of the current frame + residual rows + 27-term reduction in one kernel.
// fast: the resolved class of the level -- must imply gn_fast_supported() (returns -1 and launches nothing otherwise).
// weight_mode: what the caller guarantees about EVERY lane's parameters (kernel variants without per-pixel configuration branches; same
// arithmetic): 1 = student_nu set, 2 = student_nu clear and mestimator STUDENT, both with weighting != MIN_WEIGHT; 0 = nothing
int launch_gn_fused(hipStream_t s, int B, ImgB W0, ImgB I0, ImgB gWx, ImgB gWy, ImgB gIx, ImgB gIy, ImgB Wcur, ImgB Icur,
                    const WarpParams* lane_wp, int interp_mode, const SysParams* lane_p, double* partials, LaneMask m, int level_tag, bool fast = false,
                    int weight_mode = 0);
// THE predicate for "this level's gather kernels run in the fast numerics class" (engine / C-ABI decide it once per level and hand the
// resolved bool to the lattice pre-pass, the warp pair and the fused normal equations, so the three always agree): rows of whole 4-pixel
// groups, a 2 x 2 neighbourhood for the paired bilinear taps, six keyframe maps of one geometry with 16-byte aligned rows
bool gn_fast_supported(ImgB W0, ImgB I0, ImgB gWx, ImgB gWy, ImgB gIx, ImgB gIy, ImgB Icur);
// the next launch_build_system / launch_gn_fused on this host thread is bracketed by these events (kernel duration)
void set_system_kernel_events(hipEvent_t start, hipEvent_t stop);
void launch_reduce_system(hipStream_t s, int B, const double* partials, int nblk, double* sums, LaneMask m);

}  // namespace rgbid
