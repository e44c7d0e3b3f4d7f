// guard_band.h -- the PROVEN error bounds that make the FAST numerics class SELECTION-EXACT.
//
// The FAST gather kernels (warp_device.h, namespace fastnum) evaluate registerPixel (warping_registration.cu:129-146) in a cheaper arithmetic than
// the scalar oracle (v_rcp_f32, FMAs, the scaled point q + w t instead of q / w + t).  Their float VALUES may differ from the oracle's in the last
// bits -- every parity test allows that -- but every DISCRETE decision taken from a projected coordinate must be the oracle's:
//   * the source pixel of the point sample,   floor(xs), floor(ys)                        (trafo3DKernelInvDepth*, :505-594)
//   * the in-image predicate,                 0 <= floor(xs) < cols, 0 <= floor(ys) < rows (:486-487, :526-527, :570-571)
//   * the sign test of the warped value,      res > 0                                     (:541, :588)
//   * the covisibility lattice point,         rint(xd), rint(yd), 0 < xd < cols - 1, ...  (partialVisibilityKernel, :297-437)
//   * the covisibility and fusion gates,      |w' - D| < 0.020,  |w_s - w_KF| < 0.0225    (:340, :653)
// Method: a guard band.  A FAST kernel computes the coordinate its cheap way and tests, with the constants below, whether the ORACLE's result could
// fall on the other side of the decision; if so the pixel is recomputed with the exact instruction sequence (register_pixel, common.h) and ITS
// decision is used.  This header holds the bounds; kernels and tests share it (tests/test_cpu_guard_band.py checks them numerically on millions of
// emulated pixels, tests/test_gpu_*.py check the outcome: zero differing selections).
//
// Notation: u = 2^-24 (unit roundoff, round to nearest), every fp32 operation fl(a op b) = (a op b)(1 + e), |e| <= u (no underflow: see DOMAIN).
// Pixel (x, y), 0 <= x <= xm = cols - 1, 0 <= y <= ym = rows - 1 (exact in fp32), inverse depth w, Z = 1 / w, R = K R K^-1 (9 floats), t = K t.
//   q_r*  = R[3r] x + R[3r+1] y + R[3r+2]                         (r = 0, 1, 2; exact real arithmetic on the float inputs)
//   X_r*  = q_r* Z + t_r,   Y_r* = w X_r* = q_r* + w t_r,   x* = X_0* / X_2* = Y_0* / Y_2*,   rho = Z / |X_2*| = 1 / |Y_2*|
//   Q_r   = |R[3r]| xm + |R[3r+1]| ym + |R[3r+2]|  >= |q_r*|,     M_r = 6 |R[3r]| xm + 6 |R[3r+1]| ym + 4 |R[3r+2]|
//
// (1) ORACLE, register_pixel: zd = fl(1/w); Xd = (fl(x zd), fl(y zd), zd); X_r = fl(fl(fl(fl(R0 Xd0) + fl(R1 Xd1)) + fl(R2 Xd2)) + t_r).
//     The term R0 x Z passes through 6 roundings (zd, x zd, the product, three sums), R1 y Z through 6, R2 Z through 4, t_r through 1:
//         |X_r - X_r*| <= u' (Z M_r + |t_r|),                                    u' = u (1 + 2^-10) absorbs the second-order terms.
//     wc = fl(1 / X_2), xc = fl(X_0 wc), xs = fl(xc + 0.5):
//         |X_0 / X_2 - x*| <= (|dX_0| + |x*| |dX_2|) / |X_2|  <= u' [ rho (M_0 + |x*| M_2) + w rho (|t_0| + |x*| |t_2|) ]
//     and w |t_0| <= |Y_0*| + |q_0*| = |x*| / rho + |q_0*|,  w |t_2| <= 1 / rho + |q_2*|, so
//         |xs_oracle - (x* + 0.5)| <= u' [ rho (M_0 + Q_0 + |x*| (M_2 + Q_2)) + 5 |x*| + 1 ]                                              (E)
//     (2 |x*| from the line above, 2 |x*| for wc and xc, |x*| + 0.5 for the last sum).
// (2) FAST, fastnum::project: c = fl(R1 y + R2) (one FMA), q_r = fl(R0 x + c) (one FMA): |q_r - q_r*| <= 2 u Q_r;  Y_r = fl(t_r w + q_r):
//     |Y_r - Y_r*| <= 2 u Q_r + u |Y_r*|;  wc = v_rcp_f32(Y_2) (1 ulp: relative 2 u);  xs = fl(Y_0 wc + 0.5) (one FMA):
//         |xs_fast - (x* + 0.5)| <= u' [ rho (2 Q_0 + 2 |x*| Q_2) + 5 |x*| + 1 ]                                                         (F)
// (3) GUARD.  Decisions are only open for |x*| <= Xb := 1.01 max(cols, rows) + 2 (beyond, (E) and (F) are relative errors below 2^-9 while
//     rho <= 2^9, and both evaluations are far outside the image; rho > 2^9 makes the band wider than a pixel, i.e. the pixel is recomputed).  With
//         d1 = u' (max(M_0, M_1) + 3 max(Q_0, Q_1) + Xb (M_2 + 3 Q_2)),      d2 = u' (10 Xb + 2),       delta = |wc| d1 (1 + 2^-8) + d2
//     |xs_fast - xs_oracle| <= delta (the factor 1 + 2^-8 covers |wc| against rho).  floor() of the two agrees when no integer lies within
//     delta of xs_fast:  | frac(xs) - 1/2 | <= 1/2 - delta;  rint() agrees when | frac(xd) - 1/2 | >= delta.
// (3') GUARD AS A LANE CONSTANT (round 5: the form the kernels run).  (3) costs two v_fract, two subtractions, a max, an FMA and a compare per
//     projection.  The same verdict for 2 v_fract + 3: fix WCM = 2 (the largest |wc| the lane-constant band is priced for: Y_2 = q_2 + w t_z is 1 +- 0.3 on
//     indoor data) and dL >= WCM d1 + d2, bias the FAST coordinate DOWN by it -- xs' = fl(Y_0 wc + bL), bL = fl(0.5 - dL), dLa := 0.5 - bL (exact: bL in
//     [1/4, 1/2]) -- and take the source pixel from floor(xs').  (F) holds for xs' against x* + bL unchanged (the addend is smaller), so for |wc| <= WCM
//         xs' <= xs_oracle - (dLa - delta) ... xs_oracle <= xs' + dLa + delta,       i.e.   xs_oracle in [xs', xs' + 2 dLa],
//     and floor(xs_oracle) = floor(xs') whenever xs' - floor(xs') < 1 - 2 dLa.  v_fract_f32(x) = min(fl(x - floor x), 1 - 2^-24) is exact except for x in
//     (-1, 0), where 1 + x may round by 2^-25 -- and is never above 1 - 2^-24.  With cL = 1 - 2 dLa - 2^-22 and kL >= cL / WCM (1 + 2^-21):
//         safe  <=>  max3( fract(xs'), fract(ys'), |wc| kL ) < cL          (one multiply, one v_max3_f32, one compare)
//     implies |wc| <= WCM (so delta <= dLa, and rho <= 2^9 as (3) requires) and both floor() agreements; anything non-finite fails it (|wc| = inf gives inf;
//     a lane that is not `sane` / not `zsafe` gets cL = -1: never safe).  Pixels with |wc| > WCM -- points closer than half their keyframe depth along the
//     optical axis -- take the exact path: correct, slower.  Price: the band is 2 dLa wide for every pixel instead of 2 delta(wc): at 640 x 480, |wc| ~ 1,
//     2 (2 d1 + d2) = 3.2e-3 instead of 2.0e-3 px per axis, i.e. 6.4e-3 instead of 4.0e-3 of the pixels are recomputed.
// (3'') INSIDE PREDICATE OF THE BILINEAR WARP (:486-487: 0 <= floor(xB + 0.5) < cols).  A FAST coordinate with 0 <= xB <= cols - 1 has the oracle's
//     xB within delta(wc) of it, hence strictly inside (-0.5, cols - 0.5), whenever delta(wc) <= 1/4: |wc| <= WCORE := (1/4 - d2) / d1.  The hot path takes
//     "inside" from  xB == med3(xB, 0, hx) (the clamp its tap addresses need anyway), the same in y, and |wc| <= WCORE; every other pixel of the domain -- the half-pixel
//     ring around the image, projections outside it, |wc| > WCORE -- is classified in the cold path: farther than db inside, farther than db outside
//     (both at |wc| <= RHO_BORDER), or by the oracle's own coordinates.
// (4) SIGN of the warped inverse depth res = v / (1 - w2 t_z) * w2, v = (1/w3 - t_z) w  (oracle) = q_2 (FAST; (X_2 - t_z) w = q_2 exactly):
//     the oracle's 1/w3 - t_z = q_2* Z + e, |e| <= u' (Z M_2 + |t_2|) + 2 u |X_2| (+ the rounding of the difference, relative): it has the sign of
//     q_2* with at least half its magnitude when |q_2*| >= 2 u' (M_2 + Q_2 + 3 |Y_2*|); FAST's q_2 needs |q_2*| > 2 u Q_2.  Required by the guard:
//         |q_2| >= q0 + q1 |Y_2|,          q0 = 4 u' (M_2 + 5 Q_2),  q1 = 8 u'.
//     1 - w2 t_z: fl(1 - fl(w2 t_z)) and fl(1 - w2 t_z) (FMA) are non-zero of the same sign when the FMA's |value| >= 2^-20 (the inner product's
//     rounding error is below 2 u when the difference is below 1, below u (1 + |difference|) otherwise).  Required: |rcp(1 - w2 t_z)| <= 2^19.
//     Then both results are finite, non-zero and of sign(q_2) sign(1 - w2 t_z) sign(w2) for every w2 of the DOMAIN.
//     The first requirement is decided ONCE PER LANE: q_2* is affine in (x, y), so over the image |q_2*| >= qmin := the smallest corner value (0 when
//     the corners differ in sign), |Y_2| <= Q_2 + W_HI |t_2| for every grid inverse depth of the domain, and `zsafe` := qmin - 4 u Q_2 >= q0 + q1 (Q_2 +
//     W_HI |t_2|) holds for any sane motion (q_2 ~ 1; with W_HI = 2^14 up to |t_z| ~ 100 m).  A lane that fails it runs every pixel through the exact path.
// (5) GATES on values.  Relative distance between the oracle's and FAST's inverse depth in the other frame, w' = 1 / X_2:
//         eps_w <= u' (rho (M_2 + 3 Q_2) + 6);
//     of the warped value res:  eps_res <= u' ((M_2 + 7 Q_2 + 3 |Y_2|) / |q_2| + 2 |rcp(1 - w2 t_z)| + 12)
//                                        <= e0 + e1 |rcp(1 - w2 t_z)|   with the per-lane bounds |q_2| >= qmin, |Y_2| <= Q_2 + W_HI |t_2| of (4).
//     A gate |a - b| < th is open when | |a - b| - th | <= eps |a| + 2 u th: recomputed.
//
// DOMAIN of the FAST class (documented in include/rgbid_batched.h): non-NaN inverse depths of the projected (grid) map inside [2^-14, 2^14] -- values
// outside are treated as invalid (NaN), the same in every FAST kernel; non-NaN values of the sampled map 0 or of magnitude in [2^-60, 2^60]; |R|, |t|
// entries finite and below 2^20, 2^-10 <= Q_2 <= 2^10 (otherwise d1 = inf: every pixel of the lane takes the exact path).  Maps made by this library's own kernels
// from 16-bit depth (convertDepth2InvDepth: [0.1, 1000] / factor_depth, pyramids and fused means of such values) are inside.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define RGBID_GB_HD __host__ __device__
#else
#define RGBID_GB_HD
#endif

namespace rgbid {
namespace fastnum {

constexpr float W_LO = 0x1p-14f, W_HI = 0x1p14f;   // grid inverse depths outside are invalid in the FAST class
constexpr float RHO_BORDER = 4.f;                  // the border band of the intensity warp is priced at |wc| <= RHO_BORDER (else: recomputed)

struct Guard {
  float d1, c2;     // coordinates: safe  <=>  max(|frac(xs) - .5|, |frac(ys) - .5|) <= c2 - |wc| d1          (c2 = 0.5 - d2)
  float d2;         // rint():      safe  <=>  min(|frac(xd) - .5|, |frac(yd) - .5|) >= |wc| d1 + d2
  float q0, q1;     // sign:        safe  <=>  |q_2| >= q0 + q1 |Y_2|  (per pixel; the kernels use the per-lane verdict `zsafe`)
  int zsafe;        // 1: the sign analysis (4) holds for every pixel of the lane
  float db;         // intensity warp, in-image predicate: band around the image border at |wc| <= RHO_BORDER
  float g0, g1;     // gates: eps_w = |wc| g1 + g0 (relative distance of the inverse depth in the other frame)
  float e0, e1;     //        eps_res = e0 + e1 |rcp(1 - w2 t_z)| (relative distance of the warped inverse depth)
  float bL, cL, kL; // (3'): xs' = Y_0 wc + bL;  safe  <=>  max3(fract(xs'), fract(ys'), |wc| kL) < cL
  float wcore;      // (3''): a projection with 0 <= xB <= cols - 1, 0 <= yB <= rows - 1 and |wc| <= wcore has the oracle's inside verdict
};
#ifndef RGBID_WCM
#define RGBID_WCM 2.f
#endif
constexpr float WCM = RGBID_WCM;   // (3')

RGBID_GB_HD inline Guard make_guard(const float R[9], const float t[3], int cols, int rows) {
  const float u1 = 0x1p-24f * (1.f + 0x1p-10f);
  const float xm = (float)(cols - 1), ym = (float)(rows - 1);
  const float a0 = fabsf(R[0]), a1 = fabsf(R[1]), a2 = fabsf(R[2]), a3 = fabsf(R[3]), a4 = fabsf(R[4]), a5 = fabsf(R[5]), a6 = fabsf(R[6]), a7 = fabsf(R[7]),
              a8 = fabsf(R[8]);
  const float Q0 = a0 * xm + a1 * ym + a2, Q1 = a3 * xm + a4 * ym + a5, Q2 = a6 * xm + a7 * ym + a8;
  const float M0 = 6.f * (a0 * xm + a1 * ym) + 4.f * a2, M1 = 6.f * (a3 * xm + a4 * ym) + 4.f * a5, M2 = 6.f * (a6 * xm + a7 * ym) + 4.f * a8;
  const float Xb = 1.01f * (float)(cols > rows ? cols : rows) + 2.f;
  const float Mx = M0 > M1 ? M0 : M1, Qx = Q0 > Q1 ? Q0 : Q1;
  Guard g;
  g.d1 = u1 * (1.f + 0x1p-8f) * (Mx + 3.f * Qx + Xb * (M2 + 3.f * Q2));
  g.d2 = u1 * (10.f * Xb + 2.f);
  g.c2 = 0.5f - g.d2;
  g.q0 = 4.f * u1 * (M2 + 5.f * Q2);
  g.q1 = 8.f * u1;
  g.db = RHO_BORDER * g.d1 + g.d2;
  g.g1 = u1 * (1.f + 0x1p-8f) * (M2 + 3.f * Q2);
  g.g0 = 6.f * u1;
  // a lane whose transform is not finite or absurdly large: every pixel is recomputed exactly (all comparisons against inf / NaN fail safe)
  float big = fabsf(t[0]);
  big = fmaxf(big, fabsf(t[1])); big = fmaxf(big, fabsf(t[2]));
  big = fmaxf(big, fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)), fmaxf(fmaxf(a4, a5), fmaxf(fmaxf(a6, a7), a8))));
  bool sane = big < 0x1p20f && Q2 >= 0x1p-10f && Q2 <= 0x1p10f;
  for (int i = 0; i < 9; ++i) sane = sane && (R[i] == R[i]);
  for (int i = 0; i < 3; ++i) sane = sane && (t[i] == t[i]);
  // the smallest |q_2*| over the image: q_2* is affine, so it is attained at a corner unless the corners differ in sign
  const float c00 = R[8], c10 = R[6] * xm + R[8], c01 = R[7] * ym + R[8], c11 = R[6] * xm + R[7] * ym + R[8];
  const bool one_sign = (c00 > 0.f && c10 > 0.f && c01 > 0.f && c11 > 0.f) || (c00 < 0.f && c10 < 0.f && c01 < 0.f && c11 < 0.f);
  const float qmin = one_sign ? fminf(fminf(fabsf(c00), fabsf(c10)), fminf(fabsf(c01), fabsf(c11))) - 8.f * u1 * Q2 : 0.f;
  g.zsafe = sane && (qmin >= g.q0 + g.q1 * (Q2 + W_HI * fabsf(t[2]))) ? 1 : 0;
  g.e1 = 2.f * u1;
  g.e0 = g.zsafe ? u1 * ((M2 + 7.f * Q2 + 3.f * (Q2 + W_HI * fabsf(t[2]))) / qmin + 12.f) : INFINITY;
  if (!sane || !g.zsafe) { g.d1 = INFINITY; g.db = INFINITY; }
  if (!sane) { g.q0 = INFINITY; g.g1 = INFINITY; }
  // (3') / (3''): the lane-constant forms
  const float dL = WCM * g.d1 + g.d2 + 0x1p-24f;
  if (dL < 0.125f) {                                   // also false for inf / NaN
    g.bL = 0.5f - dL;
    const float dLa = 0.5f - g.bL;                     // exact
    g.cL = 1.f - 2.f * dLa - 0x1p-22f;
    g.kL = (g.cL / WCM) * (1.f + 0x1p-20f);
    g.wcore = fminf((0.25f - g.d2) / g.d1 * (1.f - 0x1p-20f), 0x1p9f);
  } else {                                             // every pixel of the lane is recomputed the oracle's way
    g.bL = 0.5f; g.cL = -1.f; g.kL = 1.f; g.wcore = 0.f;
  }
  return g;
}

}  // namespace fastnum
}  // namespace rgbid
