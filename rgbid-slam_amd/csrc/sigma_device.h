// sigma_device.h -- device side of the robust scale estimation (kernels_sigma.hip) shared with the per-lane persistent Gauss-Newton kernel
// (tools/experiments/persistent_gn_level/kernels_gnlevel.hip: a measured experiment, profiles/r04_experiments/persistent_gn_level.md): the sample containers, the IRLS / bisection passes of computeSigmaAndNuStudent (sigmaFuncs.cu:858-1066) and the lattice
// getter that warps a residual sample on the fly.  Moved here unchanged from kernels_sigma.hip (round 4).
#pragma once
#include "kernels.h"
#include "warp_device.h"
#include <type_traits>

// no FMA contraction in any function of this header: RGBID_FP_STRICT (se3.h / common.h) opens every body, so the including file's own state is untouched

namespace rgbid {

struct NuTable { float t[17][4]; };  // nu = 2 + k/2: {-psi(nu/2), ln(nu/2), psi((nu+1)/2), ln((nu+1)/2)}
const NuTable& sigma_nu_table();   // host: tabulated once (kernels_sigma.hip)

// One workgroup per (lane, channel).  512 threads x <= 40 samples in registers: four workgroups fit a CU, so the 1 024 workgroups of a 512-lane
// launch are resident in ONE round (1 024 threads x 24 samples needed two rounds of 512 and paid the ~10 block reductions of a pass sequence
// twice: 81 us; 768 x 26: 57; 512 x 40: 54; 384 x 50: 64; 256 x 76: 59).
static constexpr int SIG_T = 512, SIG_MAXPT = 40, SIG_W = SIG_T / 64;

// block-wide sum of 4 per-thread fp32 partials: DPP wave reduction in fp32, then the SIG_W wave totals are
// added in double in a fixed order and broadcast.  Two alternating LDS buffers (the passes are strictly sequential) make the
// write-after-read barrier of a single buffer unnecessary (round 5: and every thread forms the block totals itself): 1 barrier per pass.  sm: 2 * (SIG_W*4 + 4) doubles of LDS.
static constexpr int SIG_SM = 2 * (SIG_W * 4 + 4);
struct BlockSum {
  double* sm;
  int phase;
  __device__ __forceinline__ explicit BlockSum(double* p) : sm(p), phase(0) {}
};
__device__ __forceinline__ void block_sum4(const float in[4], double out[4], BlockSum& bs) { RGBID_FP_STRICT
  float w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = wave_sum_l63(in[k]);
  int wid = threadIdx.x >> 6, lid = threadIdx.x & 63;
  double* sm = bs.sm + (bs.phase & 1) * (SIG_W * 4 + 4);
  bs.phase++;
  if (lid == 63) {
#pragma unroll
    for (int k = 0; k < 4; ++k) sm[wid * 4 + k] = (double)w[k];
  }
  __syncthreads();
  // every thread adds the SIG_W wave totals itself, in the same fixed order (broadcast LDS reads; the same doubles as a designated thread would form):
  // ONE barrier per pass instead of two.  The two alternating buffers keep that safe: a buffer is rewritten two passes later, and every thread that
  // gets there has passed the barrier of the pass in between, i.e. all reads of this pass are done.
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < SIG_W; ++i) t += sm[i * 4 + k];  // fixed order: deterministic
    out[k] = t;
  }
}

// Per-thread residual samples, produced by a getter get(i) (a plain array for the bridge calls; a lattice
// sample of two maps for the batched engine).  REG: <= SIG_MAXPT samples per thread held in VGPRs; otherwise the getter is
// re-evaluated on every pass.  Samples are kept SANITISED (an invalid residual -- NaN, infinite or a slot beyond n -- is stored as 0) so
// the passes run branch-free.  Every sum a pass forms is LINEAR in the validity flag: sum_valid f(e_i) = sum_all f(e_i) - n_invalid f(0).
// The register path therefore keeps no per-sample flag at all: it sums f over all its slots with flag 1 and then calls f once more on the
// value an invalid slot holds with flag -n_invalid (SIG_MAXPT VGPRs and one multiply per sample, sum and pass less).
// Round 5: the register path keeps its slots in PAIRS (f32x2) and the passes that dominate the kernel -- the Student-t moments and the nu
// bisection's weight sums -- run two samples per instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two IEEE fp32 operations for ~1.45 x the
// issue cost of one on gfx950, tools/experiments/valu_rate5.hip; the kernel is VALU-bound at every lane count).  A thread's even and odd slots then sum
// into the two halves of a pair, added once per pass; the slot count is rounded up to a multiple of 8 (one uniform branch per four pairs), padding slots counted as invalid.
typedef float sig_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sig_f32x2 sig_pk_fma(sig_f32x2 a, sig_f32x2 b, sig_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
template <bool REG, class Getter>
struct Samples {
  static constexpr bool reg = REG;
  // register path: <= SIG_MAXPT fp32 adds per thread; streaming path (up to a full frame per thread-stride): double
  using Acc = typename std::conditional<REG, float, double>::type;
  sig_f32x2 e2[REG ? SIG_MAXPT / 2 : 1];
  float zero_slot;   // what an invalid slot currently holds (0, or its image under to_squared_normalised)
  float neg_ninv;    // -(number of invalid slots among this thread's cnt slots)
  Getter get;
  int n, tid, cnt;   // cnt: the thread's slots that the passes visit (register path: a multiple of 8)
  __device__ __forceinline__ Samples(const Getter& g, int n_, int tid_) : zero_slot(0.f), neg_ninv(0.f), get(g), n(n_), tid(tid_) { RGBID_FP_STRICT
    cnt = (n + SIG_T - 1) / SIG_T;
    if constexpr (REG) {
      static_assert(SIG_MAXPT % 8 == 0, "slots are visited in groups of four pairs");
      cnt = (cnt + 7) & ~7;   // one wave-uniform branch per four pairs (a branch per pair is if-converted into two selects per pair and pass)
      // the thread's samples tid, tid + SIG_T, ... are visited through the getter's cursor (seek once, then fixed strides): a lattice
      // getter turns that into one integer division per thread instead of one per sample
      Getter cur = get;
      cur.seek(tid);
#pragma unroll
      for (int j = 0; j < SIG_MAXPT; ++j) {
        int i = tid + j * SIG_T;
        float v = (i < n) ? cur.load() : qnan();
        cur.step();
        bool ok = fabsf(v) < __builtin_inff();  // !isinf && !isnan
        if (j & 1) e2[j >> 1].y = ok ? v : 0.f; else e2[j >> 1].x = ok ? v : 0.f;
        if (j < cnt) neg_ninv -= ok ? 0.f : 1.f;
      }
    }
  }
  // number of slots the passes visit, as the float the count sums add up to (exact)
  __device__ __forceinline__ float slots() const { return (float)cnt; }
  // After the last moments pass the residuals themselves are no longer needed: the nu bisection only uses en^2 = ((e - bias)/sigma)^2,
  // the same for every candidate nu, so the register copy is overwritten with it once (no extra VGPRs, 3 instructions less per
  // sample and pass).  The streaming path recomputes it on the fly.
  // en^2 is capped at 2^60 (|en| ~ 1e9: residuals sit within a few 1e4 sigma of the bias on any data with noise): the product of two nu + en^2 then stays
  // finite, which func_weights_nu's shared reciprocal / logarithm of a PAIR needs; 1 / (nu + en^2) of such a sample is < 1e-18 either way
  static constexpr float EN2_MAX = 0x1p60f;
  __device__ __forceinline__ void to_squared_normalised(float bias, float inv_sigma) { RGBID_FP_STRICT
    if constexpr (REG) {
      const sig_f32x2 b2 = {bias, bias}, s2 = {inv_sigma, inv_sigma};
#pragma unroll
      for (int j = 0; j < SIG_MAXPT / 2; ++j) {
        const sig_f32x2 en = (e2[j] - b2) * s2, q = en * en;
        e2[j].x = q.x > EN2_MAX ? EN2_MAX : q.x; e2[j].y = q.y > EN2_MAX ? EN2_MAX : q.y;   // NaN stays NaN
      }
      float en = (zero_slot - bias) * inv_sigma;
      en = en * en;
      zero_slot = en > EN2_MAX ? EN2_MAX : en;
    }
  }
  // register path: f2(pair of slots) over the visited slots, then f1(value of an invalid slot, -number of invalid slots)
  template <class F2, class F1>
  __device__ __forceinline__ void for_each_pair(F2&& f2, F1&& f1) const { RGBID_FP_STRICT
    static_assert(REG, "register path only");
#pragma unroll
    for (int g = 0; g < SIG_MAXPT / 8; ++g)
      if (8 * g < cnt) {  // wave-uniform
        f2(e2[4 * g]); f2(e2[4 * g + 1]); f2(e2[4 * g + 2]); f2(e2[4 * g + 3]);
      }
    f1(zero_slot, neg_ninv);
  }
  template <class F>
  __device__ __forceinline__ void for_each_en2(float bias, float inv_sigma, F&& f) const { RGBID_FP_STRICT  // f(en^2, validity flag); f linear in the flag
    if constexpr (REG) {
#pragma unroll
      for (int j = 0; j < SIG_MAXPT / 2; ++j)
        if (2 * j < cnt) { f(e2[j].x, 1.f); f(e2[j].y, 1.f); }
      f(zero_slot, neg_ninv);
    } else {
      for (int i = tid; i < n; i += SIG_T) {
        float v = get(i);
        bool ok = fabsf(v) < __builtin_inff();
        float en = ((ok ? v : 0.f) - bias) * inv_sigma;
        f(en * en, ok ? 1.f : 0.f);
      }
    }
  }
  template <class F>
  __device__ __forceinline__ void for_each(F&& f) const { RGBID_FP_STRICT  // f(residual, validity flag); f linear in the flag
    if constexpr (REG) {
#pragma unroll
      for (int j = 0; j < SIG_MAXPT / 2; ++j)
        if (2 * j < cnt) { f(e2[j].x, 1.f); f(e2[j].y, 1.f); }  // wave-uniform
      f(zero_slot, neg_ninv);
    } else {
      for (int i = tid; i < n; i += SIG_T) {
        float v = get(i);
        bool ok = fabsf(v) < __builtin_inff();
        f(ok ? v : 0.f, ok ? 1.f : 0.f);
      }
    }
  }
};

// x * y + a in the accumulator's precision (fp32 for the register path, double for the streaming path)
__device__ __forceinline__ float mad_acc(float x, float y, float a) { RGBID_FP_STRICT return fmaf(x, y, a); }
__device__ __forceinline__ double mad_acc(float x, float y, double a) { RGBID_FP_STRICT return fma((double)x, (double)y, a); }

// one moments pass: partialBiasAndSigmaStudent (:258-332) when student_variant, else partialBiasAndSigma (:179-255).
// Per-sample divisions are reciprocal multiplies (<= 1 ulp) and the per-thread partial sums are fp32: the pass
// is VALU-bound, and the moments only feed a 10%-tolerance fixed point.
template <class SM>
__device__ __forceinline__ void pass_moments(const SM& S, float bias, float sigma, float nu, int mest, bool student_variant,
                                             BlockSum& sm, float& swsr, float& swr, float& sw, float& nel) { RGBID_FP_STRICT
  typename SM::Acc a[4] = {0, 0, 0, 0};
  const float inv_sigma = 1.f / sigma, nup1 = nu + 1.f;
  auto acc = [&](float er, float weight, float valid) {  // weight is already 0 for an invalid sample
    float wr = er * weight;
    float wsr = wr * er;
    a[0] += wsr; a[1] += wr; a[2] += weight; a[3] += valid;
  };
  if (student_variant) {
    if constexpr (SM::reg) {
      // two samples per instruction; the even / odd slots sum into the halves of a pair.  The count sums are known: slots - invalid ones.
      sig_f32x2 A0 = {0.f, 0.f}, A1 = {0.f, 0.f}, A2 = {0.f, 0.f};
      float t0, t1, t2;   // the invalid slots' correction (a value times -n_invalid)
      if (mest == 0) {
        S.for_each_pair([&](sig_f32x2 er) { A0 += er * er; A1 += er; },
                        [&](float er, float mv) { const float wr = er * mv; t0 = wr * er; t1 = wr; t2 = mv; });
        a[0] = (A0.x + A0.y) + t0; a[1] = (A1.x + A1.y) + t1; a[2] = S.slots() + t2; a[3] = a[2];
      } else {
        // w = (nu + 1) / (nu + en^2): the constant numerator is applied to the three weighted sums once, after the loop, and the
        // normalisation is one FMA -- per PAIR of samples 3 packed FMAs, 3 packed multiplies / adds and 2 reciprocals
        const float nb = -bias * inv_sigma;
        const sig_f32x2 is2 = {inv_sigma, inv_sigma}, nb2 = {nb, nb}, nu2 = {nu, nu};
        S.for_each_pair([&](sig_f32x2 er) {
                          const sig_f32x2 en = sig_pk_fma(er, is2, nb2);
                          const sig_f32x2 t = sig_pk_fma(en, en, nu2);
                          const sig_f32x2 r = {__builtin_amdgcn_rcpf(t.x), __builtin_amdgcn_rcpf(t.y)};
                          const sig_f32x2 wr = er * r;
                          A0 = sig_pk_fma(wr, er, A0); A1 += wr; A2 += r;
                        },
                        [&](float er, float mv) {
                          const float en = fmaf(er, inv_sigma, nb);
                          const float r = __builtin_amdgcn_rcpf(fmaf(en, en, nu)) * mv;
                          const float wr = er * r;
                          t0 = wr * er; t1 = wr; t2 = r;
                        });
        a[0] = ((A0.x + A0.y) + t0) * nup1; a[1] = ((A1.x + A1.y) + t1) * nup1; a[2] = ((A2.x + A2.y) + t2) * nup1;
        a[3] = S.slots() + S.neg_ninv;
      }
    } else {
    if (mest == 0) S.for_each([&](float er, float mv) { acc(er, mv, mv); });
    else {
      // w = (nu + 1) / (nu + en^2): the constant numerator is applied to the three weighted sums once, after the loop, and the
      // normalisation is one FMA -- 7 VALU + 1 reciprocal per sample (the pass is instruction-bound; the sums only feed a 10 % fixed point)
      const float nb = -bias * inv_sigma;
      S.for_each([&](float er, float mv) {
        const float en = fmaf(er, inv_sigma, nb);
        const float r = __builtin_amdgcn_rcpf(fmaf(en, en, nu)) * mv;
        const float wr = er * r;
        a[0] = mad_acc(wr, er, a[0]); a[1] += wr; a[2] += r; a[3] += mv;
      });
      a[0] *= nup1; a[1] *= nup1; a[2] *= nup1;
    }
    }
  } else {
    S.for_each([&](float er, float mv) {
      float weight = 1.f, is_valid = 1.f;
      float en = (er - bias) * inv_sigma;
      if ((mest == 1) && (fabsf(en) > TH_HUBER)) weight = TH_HUBER / fabsf(en);
      else if (mest == 2) {
        if (fabsf(en) < TH_TUKEY) { float aux1 = (en / TH_TUKEY) * (en / TH_TUKEY); weight = (1.f - aux1) * (1.f - aux1); }
        else { weight = 0.f; is_valid = 0.f; }
      } else if (mest == 3) weight = (STUDENT_DOF + 1.f) * __builtin_amdgcn_rcpf(STUDENT_DOF + en * en);
      acc(er, weight * mv, is_valid * mv);
    });
  }
  double t[4];
  float af[4] = {(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
  block_sum4(af, t, sm);
  swsr = (float)t[0]; swr = (float)t[1]; sw = (float)t[2]; nel = (float)t[3];
}

// finalReductionBiasAndSigma :361-407
__device__ __forceinline__ void final_bias_sigma(float swsr, float swr, float sw, float nel, float& bias, float& sigma) { RGBID_FP_STRICT
  float b = swr / sw;
  bias = b;
  sigma = sqrtf((swsr - 2.f * b * swr + b * b * sw) / nel);
}

// partialFuncWeightsNu :410-468 + finalReductionFuncWeightsNu :471-512 (S holds en^2 after to_squared_normalised)
template <class SM>
__device__ __forceinline__ float func_weights_nu(const SM& S, float bias, float inv_sigma, float nu, BlockSum& sm) { RGBID_FP_STRICT
  typename SM::Acc a[4] = {0, 0, 0, 0};
  const float nup1 = nu + 1.f;
  // sum ln w = N ln(nu+1) + ln 2 * sum log2 r  and  sum w = (nu+1) sum r  with r = 1 / (nu + en^2) (finite and positive also for a
  // sanitised sample): 3 VALU + reciprocal + log2 per sample, the constants once per thread
  if constexpr (SM::reg) {
    // The transcendental unit is what bounds this kernel when the chip is full, so a PAIR of samples shares its two transcendentals: with t = nu + en^2,
    //   log2 r1 + log2 r2 = -log2(t1 t2)   and   r1 + r2 = (t1 + t2) / (t1 t2)
    // -- one v_log_f32 and one v_rcp_f32 of the product t1 t2 per pair instead of two of each per pair (2 <= t <= 2 + 2^60: the product is a normal
    // fp32 number).  The sums feed the sign of C(nu) on the bisection grid; the pairing changes them by a few ulp.
    float A0 = 0.f, A1 = 0.f, t0, t1;
    const sig_f32x2 nu2 = {nu, nu};
    S.for_each_pair([&](sig_f32x2 en2) {
                      const sig_f32x2 t = en2 + nu2;
                      const float p = t.x * t.y, ts = t.x + t.y;
                      A0 -= __builtin_amdgcn_logf(p);
                      A1 = fmaf(ts, __builtin_amdgcn_rcpf(p), A1);
                    },
                    [&](float en2, float mv) {
                      const float r = __builtin_amdgcn_rcpf(nu + en2);
                      t0 = __builtin_amdgcn_logf(r) * mv; t1 = r * mv;
                    });
    a[0] = A0 + t0; a[1] = A1 + t1; a[2] = S.slots() + S.neg_ninv;
  } else {
  S.for_each_en2(bias, inv_sigma, [&](float en2, float mv) {
    const float r = __builtin_amdgcn_rcpf(nu + en2);
    a[0] = mad_acc(__builtin_amdgcn_logf(r), mv, a[0]); a[1] = mad_acc(r, mv, a[1]); a[2] += mv;
  });
  }
  a[0] = (typename SM::Acc)0.69314718055994531 * a[0] + (typename SM::Acc)__logf(nup1) * a[2];
  a[1] *= nup1;
  double t[4];
  float af[4] = {(float)a[0], (float)a[1], (float)a[2], 0.f};
  block_sum4(af, t, sm);
  return ((float)t[0] - (float)t[1]) / (float)t[2];
}

// C(nu) = -psi(nu/2) + ln(nu/2) + mean(ln w - w) + 1 + psi((nu+1)/2) - ln((nu+1)/2)   (sigmaFuncs.cu:951)
__device__ __forceinline__ float C_nu(const NuTable& T, float nu, float fw) { RGBID_FP_STRICT
  int k = (int)((nu - 2.f) * 2.f);  // exact: nu is a multiple of 0.5 in [2,10]
  return T.t[k][0] + T.t[k][1] + fw + 1.f + T.t[k][2] - T.t[k][3];
}

// bisection of C(nu) on [2,10]: sigmaFuncs.cu:934-1039 == :1100-1205
template <class SM>
__device__ __forceinline__ float estimate_nu(SM& S, const NuTable& T, float bias, float sigma_, BlockSum& sm) { RGBID_FP_STRICT
  const float sigma = 1.f / sigma_;  // inv_sigma; S.e becomes en^2 (the residuals are not used after this point)
  S.to_squared_normalised(bias, sigma);
  float nu_up = 10.f, nu_down = 2.f, nu_new = 0.f, nu;
  float C_down = C_nu(T, nu_down, func_weights_nu(S, bias, sigma, nu_down, sm));
  float C_up = C_nu(T, nu_up, func_weights_nu(S, bias, sigma, nu_up, sm));
  if (C_up * C_down > 0) {
    nu = (C_down <= 0.f) ? nu_down : nu_up;
  } else {
    for (int j = 0; j < 5; j++) {
      nu_new = (nu_up + nu_down) / 2;
      if ((nu_up - nu_down) < 1.f) break;
      float C_new = C_nu(T, nu_new, func_weights_nu(S, bias, sigma, nu_new, sm));
      if (C_new * C_up > 0) { C_up = C_new; nu_up = nu_new; }
      else { C_down = C_new; nu_down = nu_new; }
    }
    nu = nu_new;
  }
  return nu;
}

// the three host wrappers of the reference as one device routine over a sample set
template <class SM>
__device__ __forceinline__ void sigma_core(SM& S, const NuTable& T, int mode, int mestimator, float& bias, float& sigma, float& nu, BlockSum& sm) { RGBID_FP_STRICT
  float swsr, swr, sw, nel;
  if (mode == 0) {
    // computeSigmaAndNuStudent :858-1066
    float sh_sigma = sigma, sh_bias = bias, sh_nu = 5.f, sigma_prev;
    int sh_mest = 0;
    for (int i = 0; i < 10; i++) {
      pass_moments(S, sh_bias, sh_sigma, sh_nu, sh_mest, true, sm, swsr, swr, sw, nel);
      final_bias_sigma(swsr, swr, sw, nel, bias, sigma);
      sigma_prev = sh_sigma;
      sh_bias = bias; sh_sigma = sigma; sh_mest = mestimator;
      if ((i > 0) && ((fabsf(sigma - sigma_prev) / sigma_prev) < 0.1f)) break;
    }
    nu = estimate_nu(S, T, sh_bias, sh_sigma, sm);
  } else if (mode == 1) {
    // computeNuStudent :1068-1222
    nu = estimate_nu(S, T, bias, sigma, sm);
  } else {
    // computeSigmaPdf :773-854
    float sh_sigma = sigma, sh_bias = bias;
    int sh_mest = 0;
    for (int i = 0; i < 10; i++) {
      pass_moments(S, sh_bias, sh_sigma, 5.f, sh_mest, false, sm, swsr, swr, sw, nel);
      final_bias_sigma(swsr, swr, sw, nel, bias, sigma);
      if ((i > 0) && ((fabsf(sigma - sh_sigma) / sh_sigma) < 0.1f)) break;
      sh_bias = bias; sh_sigma = sigma; sh_mest = mestimator;
    }
  }
}

// getters: operator()(i) = sample i (streaming path); seek / load / step = cursor over samples i, i + SIG_T, ... (register path)
struct ArrayGetter {
  const float* p;
  int pos;
  __device__ __forceinline__ float operator()(int i) const { RGBID_FP_STRICT return p[i]; }
  __device__ __forceinline__ void seek(int i) { RGBID_FP_STRICT pos = i; }
  __device__ __forceinline__ float load() const { RGBID_FP_STRICT return p[pos]; }
  __device__ __forceinline__ void step() { RGBID_FP_STRICT pos += SIG_T; }
};

// fused engine path: the lattice residuals are warped on the fly (W1, I1 are never materialised)
struct FusedLatticeGetter {
  ImgB cur_iD, cur_I, W0, I0;
  WarpParams P;
  int lane, stride, interp_mode;
  int fast;            // the same arithmetic as the normal-equation kernel that follows (warp_device.h fastnum)
  // both channels of one lattice sample (the inverse-depth warp is shared)
  __device__ __forceinline__ void both(int ly, int lx, float& rd, float& ri) const { RGBID_FP_STRICT
    both_given(ly, lx, px<float>(W0, lane, ly * stride, lx * stride), px<float>(I0, lane, ly * stride, lx * stride), rd, ri);
  }
  __device__ __forceinline__ void both_given(int ly, int lx, float w0, float i0v, float& rd, float& ri) const { RGBID_FP_STRICT
    int y = ly * stride, x = lx * stride;
    float w1, i1;
    if (fast) {
      // the same functions of the pixel as the normal-equation kernel that follows: identical W1 / I1, identical selection
      const fastnum::Guard G = fastnum::lane_guard(P, cur_iD.cols, cur_iD.rows);
      const fastnum::Ray r = fastnum::ray(P, (float)x, (float)y);
      w1 = fastnum::warp_invdepth_px(FMap(cur_iD, lane), r, x, y, w0, P, G);
      i1 = fastnum::warp_intensity_px(FMap(cur_I, lane), r, x, y, w1, P, G, interp_mode);
    } else {
      w1 = warp_invdepth_px(FMap(cur_iD, lane), x, y, w0, P);
      i1 = warp_intensity_px(FMap(cur_I, lane), x, y, w1, P, interp_mode);
    }
    rd = w1 - w0; ri = i1 - i0v;
  }
};

}  // namespace rgbid

