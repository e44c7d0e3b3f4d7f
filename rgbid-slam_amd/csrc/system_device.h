// system_device.h -- device side of the normal-equation kernels (kernels_system.hip) shared with the per-lane persistent Gauss-Newton kernel
// (tools/experiments/persistent_gn_level/kernels_gnlevel.hip: a measured experiment, profiles/r04_experiments/persistent_gn_level.md): the per-pixel row algebra, the workgroup reduction and the body of k_build_system as a function of a (virtual) block.
// Moved here unchanged from kernels_system.hip (round 4); see that file's header for the design.
#pragma once
#include "kernels.h"
#include <type_traits>
#include "warp_device.h"

namespace rgbid {

static constexpr int SYS_T = 256;
#ifndef RGBID_FUSED_WAVES
#define RGBID_FUSED_WAVES 4
#endif
static constexpr int FUSED_WAVES = RGBID_FUSED_WAVES;   // waves per SIMD the fused fast kernel's register allocation must allow (<= 128 VGPRs; a 96-VGPR schedule spills 25 registers)
// a scheduling fence between the four pixels of a unit: each pixel's tap loads are waited for where its rows are built, not all at the top
#define RGBID_SYS_PIXEL_FENCE __builtin_amdgcn_sched_barrier(0)

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

// ---- packed accumulation: the upper triangle of the 7-column row update [J | e] tiled by DOMINOES --------------------------
// v_pk_fma_f32 does two IEEE fp32 FMAs for ~1.45 x the issue cost of one v_fma_f32 on gfx950 (tools/experiments/valu_rate5.hip), so a packed update only
// pays when no half is wasted.  With the row vector in aligned pairs V = (J0 J1 | J2 J3 | J4 J5) the 27 products tile as 9 horizontal dominoes
// {(r,c),(r,c+1)}, c even  [scalar (w J_r) broadcast through op_sel times a pair of V], 3 vertical dominoes {(r,6),(r+1,6)}, r even  [the pair
// (w J_r, w J_r+1) times the broadcast residual] and the 3 odd diagonal terms (1,1), (3,3), (5,5) as plain FMAs: 12 packed + 3 scalar issues per row
// instead of 27, the 6 weighted-row multiplies as 3 packed ones.  (Round 4's packing -- horizontal pairs only, 9 wasted halves, 18 issues per row --
// had no issue-cost advantage at all: profiles/r04_experiments/packed_fma_accumulate.md.)  Every sum is the same fmaf(w J_r, J_c, acc) chain, intensity
// row first, as the scalar form: bit-identical partial sums.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct AccPk { f32x2 p[12]; float s[3]; };
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void acc_pk_zero(AccPk& A) {
#pragma unroll
  for (int k = 0; k < 12; ++k) A.p[k] = f32x2{0.f, 0.f};
  A.s[0] = A.s[1] = A.s[2] = 0.f;
}
// one weighted row: V = the row in pairs, e = its residual, w = its weight
__device__ __forceinline__ void acc_pk_row(AccPk& A, const f32x2 V[3], float e, float w) {
  const f32x2 ww = {w, w}, ee = {e, e};
  const f32x2 W0 = V[0] * ww, W1 = V[1] * ww, W2 = V[2] * ww;
  const f32x2 a0 = {W0.x, W0.x}, a1 = {W0.y, W0.y}, a2 = {W1.x, W1.x}, a3 = {W1.y, W1.y}, a4 = {W2.x, W2.x};
  A.p[0] = pk_fma(a0, V[0], A.p[0]); A.p[1] = pk_fma(a0, V[1], A.p[1]); A.p[2] = pk_fma(a0, V[2], A.p[2]); A.p[3] = pk_fma(W0, ee, A.p[3]);
  A.s[0] = fmaf(W0.y, V[0].y, A.s[0]); A.p[4] = pk_fma(a1, V[1], A.p[4]); A.p[5] = pk_fma(a1, V[2], A.p[5]);
  A.p[6] = pk_fma(a2, V[1], A.p[6]); A.p[7] = pk_fma(a2, V[2], A.p[7]); A.p[8] = pk_fma(W1, ee, A.p[8]);
  A.s[1] = fmaf(W1.y, V[1].y, A.s[1]); A.p[9] = pk_fma(a3, V[2], A.p[9]);
  A.p[10] = pk_fma(a4, V[2], A.p[10]); A.p[11] = pk_fma(W2, ee, A.p[11]);
  A.s[2] = fmaf(W2.y, V[2].y, A.s[2]);
}
// the 27 sums in kernels.h order (row r: columns r..5, then the right-hand side)
__device__ __forceinline__ void acc_pk_unpack(const AccPk& A, float acc[SYS_TERMS]) {
  acc[0] = A.p[0].x; acc[1] = A.p[0].y; acc[2] = A.p[1].x; acc[3] = A.p[1].y; acc[4] = A.p[2].x; acc[5] = A.p[2].y; acc[6] = A.p[3].x;
  acc[7] = A.s[0]; acc[8] = A.p[4].x; acc[9] = A.p[4].y; acc[10] = A.p[5].x; acc[11] = A.p[5].y; acc[12] = A.p[3].y;
  acc[13] = A.p[6].x; acc[14] = A.p[6].y; acc[15] = A.p[7].x; acc[16] = A.p[7].y; acc[17] = A.p[8].x;
  acc[18] = A.s[1]; acc[19] = A.p[9].x; acc[20] = A.p[9].y; acc[21] = A.p[8].y;
  acc[22] = A.p[10].x; acc[23] = A.p[10].y; acc[24] = A.p[11].x;
  acc[25] = A.s[2]; acc[26] = A.p[11].y;
}

#ifndef RGBID_PK_ACC
#define RGBID_PK_ACC 1
#endif
#ifndef RGBID_PK_JROWS
#define RGBID_PK_JROWS 0   // the fused fast kernel's Jacobian rows built as packed pairs (needs RGBID_PK_ACC)
#endif
#if RGBID_PK_ACC
using AccM = AccPk;
#else
struct AccM { float a[SYS_TERMS]; };
#endif
// both weighted rows of a pixel into the sums: intensity row (weight wi, residual ei) first, then the inverse-depth row (weight sd, residual ed)
__device__ __forceinline__ void accumulate_rows(AccM& accm, const float Ji[6], float ei, float wi, const float Jd[6], float ed, float sd) {
#if RGBID_PK_ACC
  const f32x2 Vi[3] = {{Ji[0], Ji[1]}, {Ji[2], Ji[3]}, {Ji[4], Ji[5]}}, Vd[3] = {{Jd[0], Jd[1]}, {Jd[2], Jd[3]}, {Jd[4], Jd[5]}};
  acc_pk_row(accm, Vi, ei, wi);
  acc_pk_row(accm, Vd, ed, sd);
#else
  float* acc = accm.a;
  int s = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    float a = wi * Ji[r], d = sd * Jd[r];
#pragma unroll
    for (int c = r; c < 6; ++c) { acc[s] = fmaf(a, Ji[c], acc[s]); acc[s] = fmaf(d, Jd[c], acc[s]); ++s; }
    acc[s] = fmaf(a, ei, acc[s]); acc[s] = fmaf(d, ed, acc[s]); ++s;
  }
#endif
}
__device__ __forceinline__ void acc_zero(AccM& A) {
#if RGBID_PK_ACC
  acc_pk_zero(A);
#else
#pragma unroll
  for (int k = 0; k < SYS_TERMS; ++k) A.a[k] = 0.f;
#endif
}
__device__ __forceinline__ void acc_unpack(const AccM& A, float acc[SYS_TERMS]) {
#if RGBID_PK_ACC
  acc_pk_unpack(A, acc);
#else
#pragma unroll
  for (int k = 0; k < SYS_TERMS; ++k) acc[k] = A.a[k];
#endif
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
__device__ __forceinline__ void accumulate_pixel(AccM& accm, float px_, float py_, float pp_y, float w0, float i0, float gwx, float gwy,
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
  accumulate_rows(accm, Ji, ei, wi, Jd, ed, sd);
}

// The same pixel for the fused fast kernel, whose warps hand over validity as MASKS instead of NaN values: w0s = the keyframe inverse depth sanitised to the
// FAST domain (finite, positive: fastnum::sanitised), w1 finite, okd = the warped inverse depth is valid (implies a valid w0), oki = the warped intensity is
// valid (implies okd).  Saves the w0 / w1 / i1 NaN tests and four selects per pixel; the sums are the same bit for bit (an invalid row has weight 0 either way).
template <int WM>
__device__ __forceinline__ void accumulate_pixel_m(AccM& accm, float px_, float py_, float pp_y, float w0, float i0, float gwx, float gwy,
                                                   float gix, float giy, float w1, float i1, bool okd, bool oki, const SysParams& P, const SysConst& C) {
  const bool snu = WM == 1 ? true : WM == 2 ? false : (P.student_nu != 0);
  const bool minw = WM != 0 ? false : (P.weighting == 1);
  const int mest = WM == 2 ? 3 : P.mestimator;
  const bool vd = okd && !(isnan(gwx) || isnan(gwy));
  const bool vi = oki && !(isnan(i0) || isnan(gix) || isnan(giy));
  gwx = vd ? gwx : 0.f; gwy = vd ? gwy : 0.f;
  gix = vi ? gix : 0.f; giy = vi ? giy : 0.f;
  // ---- inverse-depth row (times sigma_d)
  float gx = gwx * P.fx, gy = gwy * P.fy;
  float gz = -fmaf(gx, px_, gy * py_);
  float gz0 = gz + w0, gz1 = gz + w1;
  float mm = fmaf(gx, gx, fmaf(gy, gy, gz0 * gz0)), pp = fmaf(px_, px_, pp_y);
  float nfac = fabsf(w0) * __builtin_amdgcn_rsqf(mm * pp);             // |n^ . p^|
#if RGBID_PK_JROWS
  // the row directly in the pair layout acc_pk_row consumes (round 6): the same products and FMAs, two per issue.  fmaf(a, b, -0.f) == a * b bit for bit
  // (the sum of a product and -0 is the product, signed zeros included)
  const f32x2 w0w0 = {w0, w0}, pxpx = {px_, px_};
  f32x2 Vd[3];
  Vd[0] = f32x2{gx, gy} * w0w0;
  Vd[1] = pk_fma(f32x2{gz1, gz1}, f32x2{w0, py_}, f32x2{-0.f, -gy});
  Vd[2] = pk_fma(f32x2{-gz1, gy}, pxpx, f32x2{gx, -(gx * py_)});
#else
  float Jd[6];
  Jd[0] = gx * w0;
  Jd[1] = gy * w0;
  Jd[2] = gz1 * w0;
  Jd[3] = fmaf(gz1, py_, -gy);
  Jd[4] = fmaf(-gz1, px_, gx);
  Jd[5] = fmaf(gy, px_, -(gx * py_));
#endif
  float ed = w0 - w1;
  float eu = fmaf(ed, C.inv_sd, -C.be_d);
  float wd = snu ? C.nud1_m * __builtin_amdgcn_rcpf(fmaf(eu, eu, P.nu_d)) : m_weight(eu, mest) * C.wmul_d;
  wd = vd ? wd : 0.f;
  // ---- intensity row (times sigma_i; its weight carries rho2)
  float hx = gix * P.fx, hy = giy * P.fy;
  float hz = -fmaf(hx, px_, hy * py_);
#if RGBID_PK_JROWS
  f32x2 Vi[3];
  Vi[0] = f32x2{hx, hy} * w0w0;
  Vi[1] = pk_fma(f32x2{hz, hz}, f32x2{w0, py_}, f32x2{-0.f, -hy});
  Vi[2] = pk_fma(f32x2{-hz, hy}, pxpx, f32x2{hx, -(hx * py_)});
#else
  float Ji[6];
  Ji[0] = hx * w0;
  Ji[1] = hy * w0;
  Ji[2] = hz * w0;
  Ji[3] = fmaf(hz, py_, -hy);
  Ji[4] = fmaf(-hz, px_, hx);
  Ji[5] = fmaf(hy, px_, -(hx * py_));
#endif
  float ei = i0 - i1;
  ei = vi ? ei : 0.f;
  float eiu = fmaf(ei, C.inv_si, -C.be_i);
  float wi;
  if (minw) {
    wi = snu ? C.nui1 * __builtin_amdgcn_rcpf(fmaf(eiu, eiu, P.nu_i)) : m_weight(eiu, mest);
    wi = vi ? wi * C.wmul_i : 0.f;
    wi = fminf(wd, wi) * C.rho2;
  } else {
    wi = snu ? C.nui1_s * __builtin_amdgcn_rcpf(fmaf(eiu, eiu, P.nu_i)) : m_weight(eiu, mest) * C.wmul_i_s;
    wi = vi ? wi : 0.f;
  }
  float sd = nfac * wd;
#if RGBID_PK_JROWS
  acc_pk_row(accm, Vi, ei, wi);
  acc_pk_row(accm, Vd, ed, sd);
#else
  accumulate_rows(accm, Ji, ei, wi, Jd, ed, sd);
#endif
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
__device__ __forceinline__ void block_reduce_store(float acc[SYS_TERMS], double* out, double scale, int tid, float (*sm)[SYS_TERMS + 1]) {
  int wid = tid >> 6, lid = tid & 63;
#pragma unroll
  for (int k = 0; k < SYS_TERMS; ++k) {
    float v = wave_sum_l63(acc[k]);
    if (lid == 63) sm[wid][k] = v;
  }
  __syncthreads();
  if (tid < SYS_TERMS) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < SYS_T / 64; ++w) t += (double)sm[w][tid];
    out[tid] = t * scale;
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

// One (virtual) workgroup's share of a lane: `blk` of the launch plan's nblk blocks, threads tid = 0 .. SYS_T - 1, one row of partial sums.  The body of
// k_build_system, callable with a virtual block id / thread id so that the per-lane persistent kernel (tools/experiments/persistent_gn_level/kernels_gnlevel.hip: a measured experiment, profiles/r04_experiments/persistent_gn_level.md) sums the same pixels in the
// same order as the one-launch-per-iteration path (bit-identical records).  sm: SYS_T / 64 rows of LDS of this (virtual) workgroup.
template <class PS, bool VEC, int FUSED, int WMK>
__device__ __forceinline__ void build_system_block(const ImgB& W0, const ImgB& I0, const ImgB& gWx, const ImgB& gWy, const ImgB& gIx, const ImgB& gIy, const ImgB& W1, const ImgB& I1,
                                                   const PS& ps, double* partials, int nblk, int upt, const FusedArgs& fa, const SysTiles& tp, int lane, int blk, int tid,
                                                   float (*sm)[SYS_TERMS + 1]) {
  double* out = partials + ((size_t)lane * nblk + blk) * SYS_TERMS;
  SysParams P = ps.get(lane);
  if (P.nu_i_max) P.nu_i = fmaxf(P.nu_i, P.nu_d);
  const SysConst C = make_const(P);
  WarpParams WP;
  if (FUSED) WP = fa.wp[lane];
  const FMap Wc(W1, lane), Ic(I1, lane);  // FUSED: the current frame's maps travel in the W1 / I1 slots
  AccM accm;   // the 27 sums (packed pairs + three scalars: acc_pk_row)
  acc_zero(accm);
  const int rows = W0.rows, cols = W0.cols;
  auto pixel_loop = [&](auto wm_tag) {
  constexpr int WM = decltype(wm_tag)::value;
  if (VEC) {
    const int upr = cols >> 2;  // float4 units per row
    const int L = tp.tw_log2, TH = SYS_T >> L;
    const int lx = tid & ((1 << L) - 1), ly = tid >> L;
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
        // validity travels as masks (SGPR pairs), not as NaN values: w1 is finite at every pixel, k0..k3 say where it is the warped inverse depth
        float4 w1;
        bool k0, k1, k2, k3;
        w1.x = fastnum::id_finish_m(p0, s0, WP, G, k0, f0); w1.y = fastnum::id_finish_m(p1, s1, WP, G, k1, f1);
        w1.z = fastnum::id_finish_m(p2, s2, WP, G, k2, f2); w1.w = fastnum::id_finish_m(p3, s3, WP, G, k3, f3);
        if (__builtin_expect(f0 | f1 | f2 | f3, 0)) {   // the sign of the oracle's value is not implied by the fast one (never on data of the stated domain and sane motion)
          if (f0) { const float e = warp_invdepth_px(Wc, x, y, w0.x, WP); k0 = e == e; w1.x = k0 ? e : p0.ws; }
          if (f1) { const float e = warp_invdepth_px(Wc, x + 1, y, w0.y, WP); k1 = e == e; w1.y = k1 ? e : p1.ws; }
          if (f2) { const float e = warp_invdepth_px(Wc, x + 2, y, w0.z, WP); k2 = e == e; w1.z = k2 ? e : p2.ws; }
          if (f3) { const float e = warp_invdepth_px(Wc, x + 3, y, w0.w, WP); k3 = e == e; w1.w = k3 ? e : p3.ws; }
        }
        bool bd;
        fastnum::IntensityTaps t0 = fastnum::intensity_taps(Ic, r0, w1.x, WP, G, im, bd, k0);
        if (__builtin_expect(bd, 0)) t0.ok = fastnum::intensity_fix_border(Ic, r0, x, y, w1.x, WP, G);   // not in the core of the image: surely inside, surely outside, or the oracle's predicate
        fastnum::IntensityTaps t1 = fastnum::intensity_taps(Ic, r1, w1.y, WP, G, im, bd, k1);
        if (__builtin_expect(bd, 0)) t1.ok = fastnum::intensity_fix_border(Ic, r1, x + 1, y, w1.y, WP, G);
        fastnum::IntensityTaps t2 = fastnum::intensity_taps(Ic, r2, w1.z, WP, G, im, bd, k2);
        if (__builtin_expect(bd, 0)) t2.ok = fastnum::intensity_fix_border(Ic, r2, x + 2, y, w1.z, WP, G);
        fastnum::IntensityTaps t3 = fastnum::intensity_taps(Ic, r3, w1.w, WP, G, im, bd, k3);
        if (__builtin_expect(bd, 0)) t3.ok = fastnum::intensity_fix_border(Ic, r3, x + 3, y, w1.w, WP, G);
        if (live_n) w0n = ld_stream4(reinterpret_cast<const float*>(bW0 + unit_off(yn, xn << 2)));
        float py_ = ((float)y - C.cy_f) * C.inv_fy, pp_y = fmaf(py_, py_, 1.f);
        float px0 = ((float)x - C.cx_f) * C.inv_fx;
        // each pixel's taps are waited for where its rows are built (RGBID_SYS_PIXEL_FENCE keeps the scheduler from hoisting all four
        // waits to the top of the unit when the per-pixel code is branch-free): the later gathers land under the earlier pixels' updates
        bool nt;
        float i1v = fastnum::intensity_finish_m(t0, nt);
        if (__builtin_expect(nt, 0)) i1v = warp_intensity_px(Ic, x, y, w1.x, WP, im);   // a NaN tap (corner pixels of levels >= 1): the oracle's texel pair decides (never NaN inside the image: fminf)
        accumulate_pixel_m<WM>(accm, px0, py_, pp_y, p0.ws, i0.x, a.x, b.x, c.x, d.x, w1.x, i1v, k0, t0.ok, P, C);
        RGBID_SYS_PIXEL_FENCE;
        i1v = fastnum::intensity_finish_m(t1, nt);
        if (__builtin_expect(nt, 0)) i1v = warp_intensity_px(Ic, x + 1, y, w1.y, WP, im);
        accumulate_pixel_m<WM>(accm, px0 + C.inv_fx, py_, pp_y, p1.ws, i0.y, a.y, b.y, c.y, d.y, w1.y, i1v, k1, t1.ok, P, C);
        RGBID_SYS_PIXEL_FENCE;
        i1v = fastnum::intensity_finish_m(t2, nt);
        if (__builtin_expect(nt, 0)) i1v = warp_intensity_px(Ic, x + 2, y, w1.z, WP, im);
        accumulate_pixel_m<WM>(accm, fmaf(2.f, C.inv_fx, px0), py_, pp_y, p2.ws, i0.z, a.z, b.z, c.z, d.z, w1.z, i1v, k2, t2.ok, P, C);
        RGBID_SYS_PIXEL_FENCE;
        i1v = fastnum::intensity_finish_m(t3, nt);
        if (__builtin_expect(nt, 0)) i1v = warp_intensity_px(Ic, x + 3, y, w1.w, WP, im);
        accumulate_pixel_m<WM>(accm, fmaf(3.f, C.inv_fx, px0), py_, pp_y, p3.ws, i0.w, a.w, b.w, c.w, d.w, w1.w, i1v, k3, t3.ok, P, C);
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
        accumulate_pixel<WM>(accm, px0, py_, pp_y, w0.x, i0.x, a.x, b.x, c.x, d.x, w1.x, i1.x, P, C);
        accumulate_pixel<WM>(accm, px0 + C.inv_fx, py_, pp_y, w0.y, i0.y, a.y, b.y, c.y, d.y, w1.y, i1.y, P, C);
        accumulate_pixel<WM>(accm, fmaf(2.f, C.inv_fx, px0), py_, pp_y, w0.z, i0.z, a.z, b.z, c.z, d.z, w1.z, i1.z, P, C);
        accumulate_pixel<WM>(accm, fmaf(3.f, C.inv_fx, px0), py_, pp_y, w0.w, i0.w, a.w, b.w, c.w, d.w, w1.w, i1.w, P, C);
      }
    }
  } else {
    const int units = rows * cols;
    int u0 = blk * (SYS_T * upt) + tid;
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
        accumulate_pixel<WM>(accm, ((float)x - C.cx_f) * C.inv_fx, py_, pp_y, w0, px<float>(I0, lane, y, x), px<float>(gWx, lane, y, x),
                         px<float>(gWy, lane, y, x), px<float>(gIx, lane, y, x), px<float>(gIy, lane, y, x), w1, i1, P, C);
      }
    }
  }
  };
  pixel_loop(std::integral_constant<int, WMK>{});
  float acc[SYS_TERMS];
  acc_unpack(accm, acc);
  block_reduce_store(acc, out, (double)C.inv_sd * (double)C.inv_sd, tid, sm);
}

void system_plan_vec(int rows, int cols, int B, int* upt, int* nblk, SysTiles* tp);   // kernels_system.hip

}  // namespace rgbid
