// kernels_system.hip -- per-pixel photometric + inverse-depth residual/Jacobian rows and the 21+6 term
// Gauss-Newton normal-equation reduction for gfx950 (the headline kernel).  Replaces
// src/cuda/estimate_VO.cu: constraintsHandler (:95-441), FinalReductionKernel (:459-500) and the host
// wrappers buildSystemGridStride (:505-645) / buildSystemStudentNuGridStride (:649-789); in its FUSED forms also the two warps of
// a Gauss-Newton iteration (src/cuda/warping_registration.cu:465-546).
//
// Reference geometry (NOT reproduced): 32x2-thread blocks, grid capped at 120 blocks (tuned for a 5-SM
// GTX 850M), 27 sequential __syncthreads-bracketed shared-memory tree reductions, a second kernel and
// two stream syncs + a 216-byte D2H per call.
//
// CDNA4 design: one template, k_build_system<params source, 16-byte path, level tag, FUSED, weight variant>.
//  * FUSED = 0 -- the normal equations on stored W1 / I1: a pure 8-stream read (32 B/px, no reuse -> no LDS staging), 0.83 of the HBM peak.
//  * FUSED = 2 -- what the engine runs: W1 / I1 are formed in registers from the current frame (fast-numerics gathers), so an iteration
//    moves 32 instead of 52 B/px and is one launch instead of two.  0.72-0.75 of the HBM peak at 761 VALU instructions per 4-pixel unit:
//    memory (6.5 TB/s streaming ceiling), VALU issue (~2.4 ms of the 3.4 ms launch) and the L1 address path (~1.5 ms) all run at 45-90 %
//    under 4 waves per SIMD -- what is left is their imperfect overlap (profiles/r03_experiments/).  FUSED = 1: the same with the
//    exact-numerics warps (0.44).
// Each lane of a wave64 owns 4 consecutive pixels (16-byte loads, 1 KiB per wave per map, fully coalesced); 27 fp32 accumulators live in
// VGPRs; the workgroup (4 waves) reduces with DPP wave reductions -> 4x27 floats of LDS -> doubles, and writes ONE 27-double partial row
// per workgroup.  Partials are summed in a fixed order by a second tiny kernel (or by the batched engine's solve kernel), so results are
// deterministic -- no floating-point atomics.  blockIdx is remapped so that the 8 XCDs each stream a contiguous slab.
#define RGBID_ROW_PTR_MUL64   // common.h row_ptr: this file forms row addresses with the 64-bit multiply (its scalar unit does them; the 24-bit VALU form costs VGPRs here)
#include "kernels.h"
#include <atomic>
#include <type_traits>
#include "warp_device.h"
#include <hip/hip_ext.h>

namespace rgbid {

static constexpr int SYS_T = 256;
static constexpr int FUSED_WAVES = 4;   // waves per SIMD the fused fast kernel's register allocation must allow (<= 128 VGPRs; a 96-VGPR schedule spills 25 registers)
// a scheduling fence between the four pixels of a unit: each pixel's tap loads are waited for where its rows are built, not all at the top
#define RGBID_SYS_PIXEL_FENCE __builtin_amdgcn_sched_barrier(0)
static constexpr float TH_HUBER = 1.345f, TH_TUKEY = 4.685f, STUDENT_DOF = 5.f;

struct SysConst {  // per-thread derived constants
  float inv_fx, inv_fy, cx_f, cy_f, inv_sd, inv_si, be_d, be_i, wmul_d, wmul_i, nud1, nui1;
  float rho2;            // (sigma_d / sigma_i)^2: the intensity channel's weight relative to the common factor 1 / sigma_d^2
  float nud1_m, nui1_s;  // (nu_d + 1) * wmul_d  and  (nu_i + 1) * wmul_i * rho2: Student-t numerators with the channel switches folded in
  float wmul_i_s;        // wmul_i * rho2
};

__device__ __forceinline__ float m_weight(float e, int mest) {  // computeWeight estimate_VO.cu:141-167
  float weight = 1.f;
  if (mest == 1) { if (fabsf(e) > TH_HUBER) weight = TH_HUBER / fabsf(e); }
  else if (mest == 2) {
    if (fabsf(e) < TH_TUKEY) { float a = (e / TH_TUKEY) * (e / TH_TUKEY); weight = (1.f - a) * (1.f - a); }
    else weight = 0.f;
  } else if (mest == 3) weight = (STUDENT_DOF + 1.f) * __builtin_amdgcn_rcpf(STUDENT_DOF + e * e);
  return weight;
}

// One pixel: invDepthConstraint (:214-262) + intensityConstraint (:176-212) + the 27-term update (:408-418).
// VALU issue is a co-limiter of every variant (see the file header), so the row algebra is arranged for the fewest instructions, not for the
// reference's order of operations (the sums agree with the oracle to ~1e-6 relative; tolerance 2e-5):
//  * both rows are accumulated WITHOUT their 1/sigma factors: A = (1/sigma_d^2) sum[ w_d nfac Jd Jd' + (w_i rho2) Ji Ji' ] with
//    rho2 = (sigma_d/sigma_i)^2 folded into the intensity weight's numerator; the common factor multiplies the 27 workgroup sums once
//    (in double, block_reduce_store) -- 8 multiplies per pixel less;
//  * nfac = |n^ . p^| with n = g/w0 + e_z: n . p == 1 identically (g_z = -(g_x p_x + g_y p_y)), so nfac = |w0| rsqrt(|m|^2 |p|^2) with
//    m = n w0 = (g_x, g_y, g_z + w0): no reciprocal of w0, no normal -- 6 operations less (and without the reference's cancellation noise);
//  * explicit FMAs with shared sub-expressions (the naive `acc += a*J + d*J'` costs three operations per term without reassociation).
// Invalid constraints are neutralised by sanitising their INPUTS (so every row entry stays finite) and zeroing their weight: they
// contribute exactly 0, as in the reference (weight 0 times a stale finite row).
// WM: what the LAUNCHER knows about every lane's configuration, so that the per-pixel code has no wave-uniform branches and the four
// pixels of a unit schedule as one block: 1 = Student-t weights with estimated nu (the Gauss-Newton iterations of the shipped configuration),
// 2 = the covariance pass's fixed-nu Student-t weights, both with a weighting other than MIN_WEIGHT; 0 = decided per pixel from P (every
// other configuration).  Same arithmetic either way.
template <int WM>
__device__ __forceinline__ void accumulate_pixel(float acc[SYS_TERMS], float px_, float py_, float pp_y, float w0, float i0, float gwx, float gwy,
                                                 float gix, float giy, float w1, float i1, const SysParams& P, const SysConst& C) {
  const bool snu = WM == 1 ? true : WM == 2 ? false : (P.student_nu != 0);
  const bool minw = WM != 0 ? false : (P.weighting == 1);
  const int mest = WM == 2 ? 3 : P.mestimator;
  const bool v0 = !isnan(w0);
  const bool vd = v0 && !(isnan(w1) || isnan(gwx) || isnan(gwy));
  const bool vi = v0 && !(isnan(i0) || isnan(i1) || isnan(gix) || isnan(giy));
  w0 = v0 ? w0 : 1.f;
  w1 = vd ? w1 : w0; gwx = vd ? gwx : 0.f; gwy = vd ? gwy : 0.f;
  gix = vi ? gix : 0.f; giy = vi ? giy : 0.f;
  // ---- inverse-depth row (times sigma_d)
  float gx = gwx * P.fx, gy = gwy * P.fy;
  float gz = -fmaf(gx, px_, gy * py_);
  float gz0 = gz + w0, gz1 = gz + w1;
  float mm = fmaf(gx, gx, fmaf(gy, gy, gz0 * gz0)), pp = fmaf(px_, px_, pp_y);
  float nfac = fabsf(w0) * __builtin_amdgcn_rsqf(mm * pp);             // |n^ . p^|
  float Jd[6];
  Jd[0] = gx * w0;
  Jd[1] = gy * w0;
  Jd[2] = gz1 * w0;                                                    // gz*w0 + w0*w1
  Jd[3] = fmaf(gz1, py_, -gy);                                         // -(g' x p)
  Jd[4] = fmaf(-gz1, px_, gx);
  Jd[5] = fmaf(gy, px_, -(gx * py_));
  float ed = w0 - w1;
  float eu = fmaf(ed, C.inv_sd, -C.be_d);
  float wd = snu ? C.nud1_m * __builtin_amdgcn_rcpf(fmaf(eu, eu, P.nu_d)) : m_weight(eu, mest) * C.wmul_d;
  wd = vd ? wd : 0.f;
  // ---- intensity row (times sigma_i; its weight carries rho2)
  float hx = gix * P.fx, hy = giy * P.fy;
  float hz = -fmaf(hx, px_, hy * py_);
  float Ji[6];
  Ji[0] = hx * w0;
  Ji[1] = hy * w0;
  Ji[2] = hz * w0;
  Ji[3] = fmaf(hz, py_, -hy);
  Ji[4] = fmaf(-hz, px_, hx);
  Ji[5] = fmaf(hy, px_, -(hx * py_));
  float ei = i0 - i1;
  ei = vi ? ei : 0.f;
  float eiu = fmaf(ei, C.inv_si, -C.be_i);
  float wi;
  if (minw) {  // MIN_WEIGHT (:403-406): the minimum is taken on the true weights
    wi = snu ? C.nui1 * __builtin_amdgcn_rcpf(fmaf(eiu, eiu, P.nu_i)) : m_weight(eiu, mest);
    wi = vi ? wi * C.wmul_i : 0.f;
    wi = fminf(wd, wi) * C.rho2;
  } else {
    wi = snu ? C.nui1_s * __builtin_amdgcn_rcpf(fmaf(eiu, eiu, P.nu_i)) : m_weight(eiu, mest) * C.wmul_i_s;
    wi = vi ? wi : 0.f;
  }
  float sd = nfac * wd;
  int s = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    float a = wi * Ji[r], d = sd * Jd[r];
#pragma unroll
    for (int c = r; c < 6; ++c) { acc[s] = fmaf(a, Ji[c], acc[s]); acc[s] = fmaf(d, Jd[c], acc[s]); ++s; }
    acc[s] = fmaf(a, ei, acc[s]); acc[s] = fmaf(d, ed, acc[s]); ++s;
  }
}

__device__ __forceinline__ SysConst make_const(const SysParams& P) {
  SysConst C;
  C.inv_fx = 1.f / P.fx; C.inv_fy = 1.f / P.fy;
  C.cx_f = P.cx; C.cy_f = P.cy;
  C.inv_sd = 1.f / P.sigma_d; C.inv_si = 1.f / P.sigma_i;
  C.be_d = P.bias_d / P.sigma_d; C.be_i = P.bias_i / P.sigma_i;
  C.wmul_d = (float)(1 - (P.weighting == 3));  // PHOT_ONLY
  C.wmul_i = (float)(1 - (P.weighting == 2));  // GEOM_ONLY
  C.nud1 = P.nu_d + 1.f; C.nui1 = P.nu_i + 1.f;
  const float rho = P.sigma_d / P.sigma_i;
  C.rho2 = rho * rho;
  C.nud1_m = C.nud1 * C.wmul_d; C.nui1_s = C.nui1 * C.wmul_i * C.rho2; C.wmul_i_s = C.wmul_i * C.rho2;
  return C;
}

// workgroup reduction of 27 per-thread fp32 sums -> one row of doubles in `out`, times the common factor 1 / sigma_d^2
__device__ __forceinline__ void block_reduce_store(float acc[SYS_TERMS], double* out, double scale) {
  __shared__ float sm[SYS_T / 64][SYS_TERMS + 1];
  int wid = threadIdx.x >> 6, lid = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < SYS_TERMS; ++k) {
    float v = wave_sum_l63(acc[k]);
    if (lid == 63) sm[wid][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < SYS_TERMS) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < SYS_T / 64; ++w) t += (double)sm[w][threadIdx.x];
    out[threadIdx.x] = t * scale;
  }
}

// 16-byte streaming load: every map is read exactly once per launch, so bypass-friendly (non-temporal) loads
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream4(const float* p) {
  f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// XCD-aware logical block id: hardware places block b on XCD b % 8; give every XCD a contiguous slab
__device__ __forceinline__ int xcd_slab_block(int b, int n) {
  int per = n >> 3;
  if (per == 0 || b >= (per << 3)) return b;  // tail blocks keep their id
  return (b & 7) * per + (b >> 3);
}

// LEVEL is only a tag: it gives each pyramid level its own kernel symbol, so profilers (rocprofv3 --stats) and the
// benchmark's event timing report the 640x480 level-0 evaluation (unit U1 of SURVEY 8d) separately.
// FUSED: W1 / I1 are not read from memory but produced in registers by the per-pixel inverse warps of
// warp_device.h from the CURRENT frame's inverse-depth and intensity maps (passed in the W1 / I1 slots), with the
// lane's WarpParams: one Gauss-Newton iteration then moves 24 B/px of keyframe maps + cache-resident gathers
// instead of 56 B/px (12+12 for the two warp kernels, 32 for this one) and two launches disappear.
struct FusedArgs { const WarpParams* wp; int interp_mode; };

// Work decomposition of the 16-byte path: the image is cut into tiles of TW x TH units (a unit = 4 consecutive pixels; TW = 2^tw_log2 units wide,
// TH = SYS_T / TW rows high: 128 px x 8 rows at 640 px), one tile per workgroup per step, thread <-> (unit column, row) inside the tile.  Tiles are
// numbered DOWN a strip of TW units, then strip by strip, and a workgroup takes `upt` consecutive tiles.  A workgroup therefore reads the current
// frame's maps (fused variants: the gathers) over 8 neighbouring rows AT THE SAME TIME -- the bilinear taps' lower row is the next pixel row's upper
// row, and with a row-major walk (1 024 consecutive pixels per step) that second use came one step later, after the L2 had been flushed by
// ~40 MB of streamed keyframe maps: the intensity map was fetched ~1.6 x, now 9 rows per 8.
struct SysTiles { int tw_log2, tiles_y, ntiles; };

// optional event pair that brackets exactly the next normal-equation kernel dispatch (hipExtLaunchKernelGGL:
// the events carry the dispatch's own start / end timestamps, i.e. the kernel duration a profiler reports)
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
void set_system_kernel_events(hipEvent_t start, hipEvent_t stop) { g_ev_start = start; g_ev_stop = stop; }

template <class PS, bool VEC, int LEVEL, int FUSED, int WMK = 0>
__global__ __launch_bounds__(SYS_T) __attribute__((amdgpu_waves_per_eu(FUSED == 2 ? FUSED_WAVES : 1, 8))) void k_build_system(ImgB W0, ImgB I0, ImgB gWx, ImgB gWy, ImgB gIx, ImgB gIy, ImgB W1, ImgB I1,
                                                        PS ps, double* partials, int nblk, int upt, LaneMask m, FusedArgs fa, SysTiles tp) {
  int gb = xcd_slab_block(blockIdx.x, gridDim.x);
  int lane = gb / nblk, blk = gb - lane * nblk;
  double* out = partials + ((size_t)lane * nblk + blk) * SYS_TERMS;
  if (!m.on(lane)) return;
  SysParams P = ps.get(lane);
  if (P.nu_i_max) P.nu_i = fmaxf(P.nu_i, P.nu_d);
  const SysConst C = make_const(P);
  WarpParams WP;
  if (FUSED) WP = fa.wp[lane];
  const FMap Wc(W1, lane), Ic(I1, lane);  // FUSED: the current frame's maps travel in the W1 / I1 slots
  float acc[SYS_TERMS];
#pragma unroll
  for (int k = 0; k < SYS_TERMS; ++k) acc[k] = 0.f;
  const int rows = W0.rows, cols = W0.cols;
  auto pixel_loop = [&](auto wm_tag) {
  constexpr int WM = decltype(wm_tag)::value;
  if (VEC) {
    const int upr = cols >> 2;  // float4 units per row
    const int L = tp.tw_log2, TH = SYS_T >> L;
    const int lx = threadIdx.x & ((1 << L) - 1), ly = threadIdx.x >> L;
    const int T0 = blk * upt;                                  // first tile of the workgroup (wave-uniform)
    const int nt = min(upt, tp.ntiles - T0);                   // its tiles
    int sx = T0 / tp.tiles_y, ty = T0 - sx * tp.tiles_y;       // strip and tile-in-strip: one scalar division, then count-and-wrap
    int xu = (sx << L) + lx, y = ty * TH + ly;
    if (FUSED == 2) {
      // A unit needs three dependent memory round trips (keyframe inverse depth -> point-sampled current inverse depth -> bilinear taps).  The
      // unit's inverse depth w0 is loaded one unit AHEAD (4 VGPRs); hipcc issues the other five 16-byte streams behind the eight tap loads and the
      // next w0 behind them (ISA of this build: 4 gathers, counted waits, 8 taps, 5 streams, next w0, then the row updates with vmcnt(12) / (11) /
      // (2)).  Where the streams are issued does not matter -- forced to the top the in-order vmcnt makes the gather wait include them, and
      // fetched a whole unit ahead through LDS by LDS-DMA the kernel gets 3.5 - 8 % slower (profiles/r03_experiments/gn_lds_dma_prefetch.md): it is
      // not waiting for HBM round trips, it is co-limited by VALU issue, streaming bandwidth and the L1 address path (file header).
      // The six keyframe maps of a level share their geometry (checked by the launcher): ONE 32-bit byte offset per unit on six wave-uniform
      // lane bases (global_load ... saddr) instead of six 64-bit row pointers -- the kernel holds ~100 wave-uniform values (8 image
      // descriptors, intrinsics, scale constants, the warp) and whatever does not fit the 102 SGPRs lives in VGPRs and costs occupancy.
      const char* const bW0 = static_cast<const char*>(W0.base) + (size_t)lane * W0.lane_stride;
      const char* const bI0 = static_cast<const char*>(I0.base) + (size_t)lane * I0.lane_stride;
      const char* const bA = static_cast<const char*>(gWx.base) + (size_t)lane * gWx.lane_stride;
      const char* const bB = static_cast<const char*>(gWy.base) + (size_t)lane * gWy.lane_stride;
      const char* const bC = static_cast<const char*>(gIx.base) + (size_t)lane * gIx.lane_stride;
      const char* const bD = static_cast<const char*>(gIy.base) + (size_t)lane * gIy.lane_stride;
      const unsigned pitch_b = (unsigned)W0.pitch;
      auto unit_off = [&](int yy, int xx) { return __umul24((unsigned)yy, pitch_b) + ((unsigned)xx << 2); };
      const int im = WM == 1 ? 1 : fa.interp_mode;   // variant 1 also fixes the 1.8 fixed-point bilinear weights (launcher)
      const fastnum::Guard G = fastnum::lane_guard(WP, Wc.cols, Wc.rows);
      const fastnum::BorderBand BB = fastnum::border_band(G, Ic.cols, Ic.rows);
      bool live = nt > 0 && xu < upr && y < rows;               // ragged right / bottom tiles
      float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) w0 = ld_stream4(reinterpret_cast<const float*>(bW0 + unit_off(y, xu << 2)));
#pragma unroll 1
      for (int j = 0; j < nt; ++j) {
        int tyn = ty + 1, sxn = sx;
        if (tyn == tp.tiles_y) { tyn = 0; ++sxn; }
        const int xn = (sxn << L) + lx, yn = tyn * TH + ly;
        const bool live_n = (j + 1 < nt) && xn < upr && yn < rows;
        float4 w0n = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
        const int x = xu << 2;
        const unsigned off = unit_off(y, x);
        const float4 i0 = ld_stream4(reinterpret_cast<const float*>(bI0 + off)), a = ld_stream4(reinterpret_cast<const float*>(bA + off)),
                     b = ld_stream4(reinterpret_cast<const float*>(bB + off)), c = ld_stream4(reinterpret_cast<const float*>(bC + off)),
                     d = ld_stream4(reinterpret_cast<const float*>(bD + off));
        // fast values, the oracle's selection (warp_device.h fastnum, guard_band.h): the four projections, then -- for the few pixels per thousand
        // whose coordinates lie inside the guard band -- the oracle's coordinates under a wave-level branch, then the four gathers
        const fastnum::RowRay rr = fastnum::row_ray(WP, (float)y);
        const float xf = (float)x;
        const fastnum::Ray r0 = fastnum::ray_at(WP, rr, xf), r1 = fastnum::ray_at(WP, rr, xf + 1.f), r2 = fastnum::ray_at(WP, rr, xf + 2.f),
                           r3 = fastnum::ray_at(WP, rr, xf + 3.f);
        // per pixel: projection -> [the oracle's coordinates if inside the guard band] -> gather issued; the flags die with the pixel (SGPR pairs)
        bool fc, f0, f1, f2, f3;
        fastnum::IdProj p0 = fastnum::id_project(r0, w0.x, WP, G, Wc.cols, Wc.rows, fc);
        if (__builtin_expect(fc, 0)) fastnum::id_fix_coords(p0, x, y, WP, Wc.cols, Wc.rows);
        const float s0 = Wc.at(p0.iy, p0.ix);   // unclamped (warp_device.h)
        fastnum::IdProj p1 = fastnum::id_project(r1, w0.y, WP, G, Wc.cols, Wc.rows, fc);
        if (__builtin_expect(fc, 0)) fastnum::id_fix_coords(p1, x + 1, y, WP, Wc.cols, Wc.rows);
        const float s1 = Wc.at(p1.iy, p1.ix);
        fastnum::IdProj p2 = fastnum::id_project(r2, w0.z, WP, G, Wc.cols, Wc.rows, fc);
        if (__builtin_expect(fc, 0)) fastnum::id_fix_coords(p2, x + 2, y, WP, Wc.cols, Wc.rows);
        const float s2 = Wc.at(p2.iy, p2.ix);
        fastnum::IdProj p3 = fastnum::id_project(r3, w0.w, WP, G, Wc.cols, Wc.rows, fc);
        if (__builtin_expect(fc, 0)) fastnum::id_fix_coords(p3, x + 3, y, WP, Wc.cols, Wc.rows);
        const float s3 = Wc.at(p3.iy, p3.ix);
        float4 w1;
        w1.x = fastnum::id_finish(p0, s0, WP, G, f0); w1.y = fastnum::id_finish(p1, s1, WP, G, f1);
        w1.z = fastnum::id_finish(p2, s2, WP, G, f2); w1.w = fastnum::id_finish(p3, s3, WP, G, f3);
        if (__builtin_expect(f0 | f1 | f2 | f3, 0)) {   // the sign of the oracle's value is not implied by the fast one (never on data of the stated domain and sane motion)
          if (f0) w1.x = warp_invdepth_px(Wc, x, y, w0.x, WP);
          if (f1) w1.y = warp_invdepth_px(Wc, x + 1, y, w0.y, WP);
          if (f2) w1.z = warp_invdepth_px(Wc, x + 2, y, w0.z, WP);
          if (f3) w1.w = warp_invdepth_px(Wc, x + 3, y, w0.w, WP);
        }
        bool bd;
        fastnum::IntensityTaps t0 = fastnum::intensity_taps(Ic, r0, w1.x, WP, BB, im, bd);
        if (__builtin_expect(bd, 0)) t0.ok = fastnum::intensity_fix_border(Ic, r0, x, y, w1.x, WP, G);   // not safely inside the image: surely outside, or the oracle's predicate
        fastnum::IntensityTaps t1 = fastnum::intensity_taps(Ic, r1, w1.y, WP, BB, im, bd);
        if (__builtin_expect(bd, 0)) t1.ok = fastnum::intensity_fix_border(Ic, r1, x + 1, y, w1.y, WP, G);
        fastnum::IntensityTaps t2 = fastnum::intensity_taps(Ic, r2, w1.z, WP, BB, im, bd);
        if (__builtin_expect(bd, 0)) t2.ok = fastnum::intensity_fix_border(Ic, r2, x + 2, y, w1.z, WP, G);
        fastnum::IntensityTaps t3 = fastnum::intensity_taps(Ic, r3, w1.w, WP, BB, im, bd);
        if (__builtin_expect(bd, 0)) t3.ok = fastnum::intensity_fix_border(Ic, r3, x + 3, y, w1.w, WP, G);
        if (live_n) w0n = ld_stream4(reinterpret_cast<const float*>(bW0 + unit_off(yn, xn << 2)));
        float py_ = ((float)y - C.cy_f) * C.inv_fy, pp_y = fmaf(py_, py_, 1.f);
        float px0 = ((float)x - C.cx_f) * C.inv_fx;
        // each pixel's taps are waited for where its rows are built (RGBID_SYS_PIXEL_FENCE keeps the scheduler from hoisting all four
        // waits to the top of the unit when the per-pixel code is branch-free): the later gathers land under the earlier pixels' updates
        bool nt;
        float i1v = fastnum::intensity_finish(t0, nt);
        if (__builtin_expect(nt, 0)) i1v = warp_intensity_px(Ic, x, y, w1.x, WP, im);   // a NaN tap (corner pixels of levels >= 1): the oracle's texel pair decides
        accumulate_pixel<WM>(acc, px0, py_, pp_y, w0.x, i0.x, a.x, b.x, c.x, d.x, w1.x, i1v, P, C);
        RGBID_SYS_PIXEL_FENCE;
        i1v = fastnum::intensity_finish(t1, nt);
        if (__builtin_expect(nt, 0)) i1v = warp_intensity_px(Ic, x + 1, y, w1.y, WP, im);
        accumulate_pixel<WM>(acc, px0 + C.inv_fx, py_, pp_y, w0.y, i0.y, a.y, b.y, c.y, d.y, w1.y, i1v, P, C);
        RGBID_SYS_PIXEL_FENCE;
        i1v = fastnum::intensity_finish(t2, nt);
        if (__builtin_expect(nt, 0)) i1v = warp_intensity_px(Ic, x + 2, y, w1.z, WP, im);
        accumulate_pixel<WM>(acc, fmaf(2.f, C.inv_fx, px0), py_, pp_y, w0.z, i0.z, a.z, b.z, c.z, d.z, w1.z, i1v, P, C);
        RGBID_SYS_PIXEL_FENCE;
        i1v = fastnum::intensity_finish(t3, nt);
        if (__builtin_expect(nt, 0)) i1v = warp_intensity_px(Ic, x + 3, y, w1.w, WP, im);
        accumulate_pixel<WM>(acc, fmaf(3.f, C.inv_fx, px0), py_, pp_y, w0.w, i0.w, a.w, b.w, c.w, d.w, w1.w, i1v, P, C);
        } else if (live_n) w0n = ld_stream4(reinterpret_cast<const float*>(bW0 + unit_off(yn, xn << 2)));
        w0 = w0n; live = live_n; y = yn; xu = xn; ty = tyn; sx = sxn;
      }
    } else
#pragma unroll 1
    for (int j = 0; j < nt; ++j) {
      if (j) {
        if (++ty == tp.tiles_y) { ty = 0; ++sx; }
        xu = (sx << L) + lx; y = ty * TH + ly;
      }
      if (xu < upr && y < rows) {
        const int x = xu << 2;
        float4 w0 = ld_stream4(row_ptr<float>(W0, lane, y) + x);
        float4 i0 = ld_stream4(row_ptr<float>(I0, lane, y) + x);
        float4 a = ld_stream4(row_ptr<float>(gWx, lane, y) + x);
        float4 b = ld_stream4(row_ptr<float>(gWy, lane, y) + x);
        float4 c = ld_stream4(row_ptr<float>(gIx, lane, y) + x);
        float4 d = ld_stream4(row_ptr<float>(gIy, lane, y) + x);
        float4 w1, i1;
        if (FUSED) {
          w1.x = warp_invdepth_px(Wc, x, y, w0.x, WP);     i1.x = warp_intensity_px(Ic, x, y, w1.x, WP, fa.interp_mode);
          w1.y = warp_invdepth_px(Wc, x + 1, y, w0.y, WP); i1.y = warp_intensity_px(Ic, x + 1, y, w1.y, WP, fa.interp_mode);
          w1.z = warp_invdepth_px(Wc, x + 2, y, w0.z, WP); i1.z = warp_intensity_px(Ic, x + 2, y, w1.z, WP, fa.interp_mode);
          w1.w = warp_invdepth_px(Wc, x + 3, y, w0.w, WP); i1.w = warp_intensity_px(Ic, x + 3, y, w1.w, WP, fa.interp_mode);
        } else {
          w1 = ld_stream4(row_ptr<float>(W1, lane, y) + x);
          i1 = ld_stream4(row_ptr<float>(I1, lane, y) + x);
        }
        float py_ = ((float)y - C.cy_f) * C.inv_fy, pp_y = fmaf(py_, py_, 1.f);
        float px0 = ((float)x - C.cx_f) * C.inv_fx;
        accumulate_pixel<WM>(acc, px0, py_, pp_y, w0.x, i0.x, a.x, b.x, c.x, d.x, w1.x, i1.x, P, C);
        accumulate_pixel<WM>(acc, px0 + C.inv_fx, py_, pp_y, w0.y, i0.y, a.y, b.y, c.y, d.y, w1.y, i1.y, P, C);
        accumulate_pixel<WM>(acc, fmaf(2.f, C.inv_fx, px0), py_, pp_y, w0.z, i0.z, a.z, b.z, c.z, d.z, w1.z, i1.z, P, C);
        accumulate_pixel<WM>(acc, fmaf(3.f, C.inv_fx, px0), py_, pp_y, w0.w, i0.w, a.w, b.w, c.w, d.w, w1.w, i1.w, P, C);
      }
    }
  } else {
    const int units = rows * cols;
    int u0 = blk * (SYS_T * upt) + threadIdx.x;
#pragma unroll 1
    for (int j = 0; j < upt; ++j) {
      int u = u0 + j * SYS_T;
      if (u < units) {
        int y = u / cols, x = u - y * cols;
        float w0 = px<float>(W0, lane, y, x), w1, i1;
        if (FUSED == 2) {
          const fastnum::Guard G = fastnum::lane_guard(WP, Wc.cols, Wc.rows);
          const fastnum::Ray r = fastnum::ray(WP, (float)x, (float)y);
          w1 = fastnum::warp_invdepth_px(Wc, r, x, y, w0, WP, G); i1 = fastnum::warp_intensity_px(Ic, r, x, y, w1, WP, G, fa.interp_mode);
        }
        else if (FUSED) { w1 = warp_invdepth_px(Wc, x, y, w0, WP); i1 = warp_intensity_px(Ic, x, y, w1, WP, fa.interp_mode); }
        else { w1 = px<float>(W1, lane, y, x); i1 = px<float>(I1, lane, y, x); }
        float py_ = ((float)y - C.cy_f) * C.inv_fy, pp_y = fmaf(py_, py_, 1.f);
        accumulate_pixel<WM>(acc, ((float)x - C.cx_f) * C.inv_fx, py_, pp_y, w0, px<float>(I0, lane, y, x), px<float>(gWx, lane, y, x),
                         px<float>(gWy, lane, y, x), px<float>(gIx, lane, y, x), px<float>(gIy, lane, y, x), w1, i1, P, C);
      }
    }
  }
  };
  pixel_loop(std::integral_constant<int, WMK>{});
  block_reduce_store(acc, out, (double)C.inv_sd * (double)C.inv_sd);
}

static inline bool vec_ok(const ImgB& a) { return ((a.pitch & 15) == 0) && ((a.lane_stride & 15) == 0) && ((((uintptr_t)a.base) & 15) == 0); }

// compute units of the current device (hipDeviceProp_t::multiProcessorCount), cached per device id.  The library is re-entrant (two host threads
// may plan launches at once): the cache entries are atomics, and two threads that both miss store the same value
static int device_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cus[dev].load(std::memory_order_relaxed);
  if (!v) {
    int n = 0;
    v = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    cus[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}

// tile shape of the 16-byte path (SysTiles): the widest of 32 / 16 / 8 units that divides the row, else 32 with a ragged last strip
static SysTiles system_tiles(int rows, int cols) {
  const int upr = cols / 4;
  int L = 5;
  if (upr % 32 != 0) { if (upr % 16 == 0) L = 4; else if (upr % 8 == 0) L = 3; }
  const int tw = 1 << L, th = SYS_T >> L;
  SysTiles t;
  t.tw_log2 = L;
  t.tiles_y = (rows + th - 1) / th;
  t.ntiles = ((upr + tw - 1) / tw) * t.tiles_y;
  return t;
}

// launch plan: steps per thread (upt; a step = one tile of SYS_T units on the 16-byte path, SYS_T single pixels otherwise) and workgroups per
// lane.  16-byte path: the grid is a whole multiple of 20 workgroups per compute unit -- whole rounds of the chip's resident capacity both for
// the kernels that hold 5 four-wave workgroups per CU (<= 102 VGPRs) and for the fused fast kernel's 4 (<= 128), so there is no partially
// filled tail round -- and as few workgroups as that allows with at most 60 tiles (240 px per thread in fp32 partial sums) each: the epilogue of a
// workgroup (27-term DPP + LDS reduction, one row of partials) is not free -- 15 -> 60 tiles per workgroup at 2 048 lanes: -2.2 % per launch.
// The plan depends on the geometry and the device only -- never on the kernel variant -- so that every variant sums the same pixels in the same
// order (fused and unfused results are bit-identical).
static void system_plan(int rows, int cols, int B, bool vec, int* upt, int* nblk) {
  long long steps = vec ? (long long)system_tiles(rows, cols).ntiles : ((long long)rows * cols + SYS_T - 1) / SYS_T;   // workgroup-steps per lane
  if (steps < 1) steps = 1;
  const long long capacity = (long long)device_cus() * (vec ? 20 : 5);
  const long long max_upt = vec ? 60 : 64;
  long long rounds = (steps * B + capacity * max_upt - 1) / (capacity * max_upt);
  long long nb = (rounds * capacity) / B;                    // workgroups per lane
  if (nb < 1) nb = 1;
  if (nb > steps) nb = steps;
  long long u = (steps + nb - 1) / nb;
  if (vec) {
    // A workgroup walks `u` consecutive tiles DOWN a strip, and the workgroups of a lane start together.  When u divides the tiles of a strip (or is
    // a multiple of it), the workgroups of neighbouring strips work on the same image rows at the same time and the cache lines their gathers share
    // across the strip border are fetched once; with u = 17 on 60-tile strips (2 048 lanes at 640x480) neighbours are ~9 tile steps = ~50 us apart, the
    // shared lines are long gone from the XCD's L2, and the launch moved 1.07 x its algorithmic bytes (1.01 x at 512 lanes, where u = 15).
    const long long ty = system_tiles(rows, cols).tiles_y;
    long long best = 0;
    for (long long d = 1; d <= u; ++d) if (ty % d == 0 || d % ty == 0) best = d;
    if (best * 4 >= u * 3) u = best;                         // at most a third more workgroups
  }
  nb = (steps + u - 1) / u;                                  // drop workgroups that would get no step
  *upt = (int)u;
  *nblk = (int)nb;
}

int system_blocks_per_lane(int rows, int cols, int B) {
  // worst case over both code paths so scratch sized with this is always sufficient
  int upt, nb1, nb2;
  system_plan(rows, cols, B, true, &upt, &nb1);
  system_plan(rows, cols, B, false, &upt, &nb2);
  return nb1 > nb2 ? nb1 : nb2;
}

static int launch_system_impl(hipStream_t s, int B, ImgB W0, ImgB I0, ImgB gWx, ImgB gWy, ImgB gIx, ImgB gIy, ImgB W1, ImgB I1,
                              const SysParams* hp, const SysParams* lp, double* partials, LaneMask m, int level_tag, int fused, FusedArgs fa, int wm = 0) {
  bool vec = (W0.cols % 4 == 0) && vec_ok(W0) && vec_ok(I0) && vec_ok(gWx) && vec_ok(gWy) && vec_ok(gIx) && vec_ok(gIy) && (fused || (vec_ok(W1) && vec_ok(I1)));
  int upt, nblk;
  system_plan(W0.rows, W0.cols, B, vec, &upt, &nblk);
  dim3 g(nblk * B), b(SYS_T);
  const SysTiles tp = vec ? system_tiles(W0.rows, W0.cols) : SysTiles{0, 1, 0};
#define RGBID_SYS_LAUNCH(PSV, V, T, F) hipExtLaunchKernelGGL((k_build_system<decltype(PSV), V, T, F>), g, b, 0, s, g_ev_start, g_ev_stop, 0, W0, I0, gWx, gWy, gIx, gIy, W1, I1, PSV, partials, nblk, upt, m, fa, tp)
#define RGBID_SYS_LEVELS(PSV, V, F) do { if (level_tag == 0) RGBID_SYS_LAUNCH(PSV, V, 0, F); else if (level_tag == 1) RGBID_SYS_LAUNCH(PSV, V, 1, F); else RGBID_SYS_LAUNCH(PSV, V, 2, F); } while (0)
  if (lp) {
    ByLane<SysParams> p{lp};
    if (fused == 2 && vec && (wm == 2 || (wm == 1 && fa.interp_mode == 1))) {
#define RGBID_SYS_LAUNCH_WM(T, W) hipExtLaunchKernelGGL((k_build_system<ByLane<SysParams>, true, T, 2, W>), g, b, 0, s, g_ev_start, g_ev_stop, 0, W0, I0, gWx, gWy, gIx, gIy, W1, I1, p, partials, nblk, upt, m, fa, tp)
      if (wm == 1) { if (level_tag == 0) RGBID_SYS_LAUNCH_WM(0, 1); else if (level_tag == 1) RGBID_SYS_LAUNCH_WM(1, 1); else RGBID_SYS_LAUNCH_WM(2, 1); }
      else { if (level_tag == 0) RGBID_SYS_LAUNCH_WM(0, 2); else if (level_tag == 1) RGBID_SYS_LAUNCH_WM(1, 2); else RGBID_SYS_LAUNCH_WM(2, 2); }
#undef RGBID_SYS_LAUNCH_WM
    } else if (fused == 2) { if (vec) RGBID_SYS_LEVELS(p, true, 2); else RGBID_SYS_LAUNCH(p, false, 0, 2); }
    else if (fused) { if (vec) RGBID_SYS_LEVELS(p, true, 1); else RGBID_SYS_LAUNCH(p, false, 0, 1); }
    else { if (vec) RGBID_SYS_LEVELS(p, true, 0); else RGBID_SYS_LEVELS(p, false, 0); }
  } else {
    ByValue<SysParams> p{*hp};
    if (vec) RGBID_SYS_LEVELS(p, true, 0); else RGBID_SYS_LAUNCH(p, false, 0, 0);
  }
#undef RGBID_SYS_LEVELS
#undef RGBID_SYS_LAUNCH
  g_ev_start = g_ev_stop = nullptr;
  return nblk;
}

int launch_build_system(hipStream_t s, int B, ImgB W0, ImgB I0, ImgB gWx, ImgB gWy, ImgB gIx, ImgB gIy, ImgB W1, ImgB I1,
                        const SysParams* hp, const SysParams* lp, double* partials, LaneMask m, int level_tag) {
  return launch_system_impl(s, B, W0, I0, gWx, gWy, gIx, gIy, W1, I1, hp, lp, partials, m, level_tag, 0, FusedArgs{nullptr, 0});
}

bool gn_fast_supported(ImgB W0, ImgB I0, ImgB gWx, ImgB gWy, ImgB gIx, ImgB gIy, ImgB Icur) {
  // the pipelined kernel addresses the six keyframe maps through one shared 32-bit row offset; the paired bilinear taps read (x, x + 1) of rows (y, y + 1)
  const bool same_geom = W0.pitch == I0.pitch && W0.pitch == gWx.pitch && W0.pitch == gWy.pitch && W0.pitch == gIx.pitch && W0.pitch == gIy.pitch &&
                         (unsigned long long)W0.rows * W0.pitch < (1ull << 32) && W0.pitch < (1u << 24) && W0.rows < (1 << 24);
  const bool vec = (W0.cols % 4 == 0) && W0.cols >= 4 && vec_ok(W0) && vec_ok(I0) && vec_ok(gWx) && vec_ok(gWy) && vec_ok(gIx) && vec_ok(gIy);
  return same_geom && vec && Icur.cols >= 2 && Icur.rows >= 2;
}

int launch_gn_fused(hipStream_t s, int B, ImgB W0, ImgB I0, ImgB gWx, ImgB gWy, ImgB gIx, ImgB gIy, ImgB Wcur, ImgB Icur,
                    const WarpParams* lane_wp, int interp_mode, const SysParams* lane_p, double* partials, LaneMask m, int level_tag, bool fast, int weight_mode) {
  if (fast && !gn_fast_supported(W0, I0, gWx, gWy, gIx, gIy, Icur)) return -1;   // the caller resolves the class once per level (kernels.h)
  return launch_system_impl(s, B, W0, I0, gWx, gWy, gIx, gIy, Wcur, Icur, nullptr, lane_p, partials, m, level_tag, fast ? 2 : 1, FusedArgs{lane_wp, interp_mode}, weight_mode);
}

// FinalReductionKernel estimate_VO.cu:459-500 (all-double here; the reference's tree is fp32).
// 8 slices x 32 threads: slice s sums blocks s, s+8, ... of one term, then the 8 slice sums are added in
// a fixed order -> deterministic for a given launch plan.
__global__ __launch_bounds__(256) void k_reduce_system(const double* partials, int nblk, double* sums, LaneMask m) {
  int lane = blockIdx.x;
  if (!m.on(lane)) return;
  __shared__ double sm[8][32];
  int k = threadIdx.x & 31, sl = threadIdx.x >> 5;
  double t = 0.0;
  if (k < SYS_TERMS) {
    const double* p = partials + (size_t)lane * nblk * SYS_TERMS + k;
    for (int b = sl; b < nblk; b += 8) t += p[(size_t)b * SYS_TERMS];
  }
  sm[sl][k] = t;
  __syncthreads();
  if (threadIdx.x < SYS_TERMS) {
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += sm[i][threadIdx.x];
    sums[lane * SYS_TERMS + threadIdx.x] = r;
  }
}
void launch_reduce_system(hipStream_t s, int B, const double* partials, int nblk, double* sums, LaneMask m) {
  hipLaunchKernelGGL(k_reduce_system, dim3(B), dim3(256), 0, s, partials, nblk, sums, m);
}

}  // namespace rgbid
