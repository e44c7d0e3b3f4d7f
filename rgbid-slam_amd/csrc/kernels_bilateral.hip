// kernels_bilateral.hip -- NaN-aware 5x5 bilateral filter for gfx950 (filters.cu:86-135 of the reference): the IEEE-exact kernels of the bridge and the
// FAST-class kernel of the engine, which computes every pair weight once (round 6).  Its own translation unit: built with -fno-slp-vectorize (csrc/Makefile) --
// the automatic pairing of the three partial sums of k_bilateral_shared costs more v_mov shuffles than the packed instructions save.
#include "kernels.h"
#include "warp_device.h"

#pragma clang fp contract(off)

namespace rgbid {

static constexpr int TX = 64, TY = 4;  // one wave per tile row (kernels_prep.hip)
static inline bool same_geometry(const ImgB& a, const ImgB& b) { return a.rows == b.rows && a.cols == b.cols; }
template <int CTRL>
__device__ __forceinline__ float dpp_shift(float v) {   // 0x138 = wave_shr:1 (lane l reads lane l - 1), 0x130 = wave_shl:1 (lane l reads lane l + 1); edge lanes get 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// ---- bilateralKernel (filters.cu:86-135), clipped 5x5 window -------------------------------------
static constexpr int BR = 2, BIL_TILES = 4;
// the 24 off-centre taps of one pixel.  FAST: the per-tap division by sigma (a constant of the launch) as the 3-instruction exact sequence of
// common.h div_const_fast; the caller recomputes the pixel with the IEEE division if any tap left its verified range.  The centre tap is the
// pixel itself: its weight is expf(-0) = 1 exactly, so it enters the sums as (value, 1) without arithmetic -- at its place in the tap order.
template <bool FAST>
__device__ __forceinline__ float bilateral_px(const float (*tile)[TX + 2 * BR + 1], int ty, int tx, float value, float sigma_floatmap, DivConst dc, float s2ih,
                                              bool& all_ok) {
  float sum1 = 0.f, sum2 = 0.f;
  // clipped window == full window over the NaN-padded tile (NaN taps are skipped either way); tap order cy outer, cx inner
#pragma unroll
  for (int dy = -BR; dy <= BR; ++dy)
#pragma unroll
    for (int dx = -BR; dx <= BR; ++dx) {
      if (dx == 0 && dy == 0) { sum1 = sum1 + value; sum2 = sum2 + 1.f; continue; }   // value * 1.f, weight 1.f
      const float tmp = tile[ty + dy][tx + dx];
      const float space2 = (float)(dx * dx + dy * dy);
      float fn;
      if (FAST) { bool ok_; fn = div_const_fast(value - tmp, dc, ok_); all_ok = all_ok && (ok_ || isnan(tmp)); }
      else fn = (value - tmp) / sigma_floatmap;
      // the source mixes float and double here (`0.5*fn*fn`): keep the double evaluation
      const double arg = (double)(s2ih * space2) + (0.5 * (double)fn) * (double)fn;
      const float weight = expf((float)(-arg));
      const bool ok = !isnan(tmp);
      sum1 = ok ? sum1 + tmp * weight : sum1;
      sum2 = ok ? sum2 + weight : sum2;
    }
  return sum1 / sum2;
}
// reference-build-class numerics (engine fast_numerics / rgbid_ctx_set_numerics): the reference's own tap weight is
// __expf(-(space2 / 50 + 0.5 fn^2)) = ex2.approx(log2e * arg) (filters.cu:124 under nvcc's fast exp); here the same exponent is formed with
// the constants folded -- arg2 = c_space[dy][dx] + k d^2, k = 0.5 log2e / sigma^2 -- and handed to v_exp_f32: 8 instructions per tap
// instead of ~50 (exact division, double-precision exponent, full-range expf), results within a few 1e-7 relative of the exact kernel.
// Invalid taps (NaN in the map, window positions outside the image) are stored in the tile as BIL_SENTINEL, a large FINITE value: the range
// term k d^2 of such a tap is >= 1e36, v_exp_f32 returns exactly 0 and 0 * sentinel adds exactly 0 to both sums -- the tap drops out without a
// compare and two selects per tap (a third of the kernel's issue time).  So that no VALID value can collide with the sentinel or overflow the
// range term on its own, this class treats |v| >= BIL_MAXABS (and +-inf) as invalid when the tile is loaded -- one compare per loaded value,
// nothing per tap; inverse depths (<= 1e3 m^-1) and intensities (<= 255) are six orders below it.  Stated domain of the FAST filter: the
// oracle's result on the map with such values replaced by NaN (tests/test_gpu_fuzz.py).
static constexpr float BIL_SENTINEL = 1e19f, BIL_MAXABS = 1e9f;
__device__ __forceinline__ float bilateral_px_fast(const float (*tile)[TX + 2 * BR + 1], int ty, int tx, float value, float k, float cs) {
  float sum1 = value, sum2 = 1.f;   // centre tap: weight exp(-0) = 1
#pragma unroll
  for (int dy = -BR; dy <= BR; ++dy)
#pragma unroll
    for (int dx = -BR; dx <= BR; ++dx) {
      if (dx == 0 && dy == 0) continue;
      const float tmp = tile[ty + dy][tx + dx];
      const float d = value - tmp;
      const float w = __builtin_amdgcn_exp2f(-__builtin_fmaf(k * d, d, cs * (float)(dx * dx + dy * dy)));
      sum1 = __builtin_fmaf(tmp, w, sum1);
      sum2 += w;
    }
  return sum1 * __builtin_amdgcn_rcpf(sum2);
}
struct BilSet { ImgB src, dst; float sigma; DivConst dc; };
// ny: tile rows of one map; blockIdx.y >= ny: the second map of the launch (same geometry, its own range sigma -- the keyframe's inverse depth and
// intensity: one launch instead of two)
template <int MODE>   // 0: IEEE division per tap, 1: the verified 3-instruction exact division, 2: reference-build-class numerics
__global__ __launch_bounds__(256) void k_bilateral(BilSet b0, BilSet b1, int ny, LaneMask m) {
  constexpr bool FAST = MODE == 1;
  // XCD-contiguous tile order inside the lane (common.h xcd_lane_local_tile; the launch is predicated): neighbouring tiles share halo columns / rows (a
  // 68-float row segment spans 6 cache lines, 4 of them its own), and with the natural order the neighbours of a tile always run on other XCDs (round 4
  // counted 1.68 x the algorithmic traffic)
  const TileId tid_ = xcd_lane_local_tile();
  const int lane = tid_.lane;
  if (!m.on(lane)) return;
  const bool second = tid_.by >= ny;
  const BilSet& S = second ? b1 : b0;
  const ImgB& src = S.src; const ImgB& dst = S.dst;
  const float sigma_floatmap = S.sigma;
  const DivConst dc = S.dc;
  const int tile_y = tid_.by - (second ? ny : 0);
  // one halo tile of BIL_TILES x TY rows per workgroup (16 + 4 rows x 68 columns: 1.33 loads per output, ONE barrier and ONE exposed memory round
  // trip per four outputs of a thread; four separate 4-row tiles cost 2.1 loads per output and four round trips)
  __shared__ float tile[TY * BIL_TILES + 2 * BR][TX + 2 * BR + 1];
  const int x0 = tid_.bx * TX;
  const float sigma_space = 5.f;
  const float s2ih = (float)(0.5 / (double)(sigma_space * sigma_space));
  const int y0 = tile_y * (TY * BIL_TILES);
  for (int ty = threadIdx.y; ty < TY * BIL_TILES + 2 * BR; ty += TY) {
    const int cy = y0 + ty - BR;
    const bool row_in = cy >= 0 && cy < src.rows;
    const float* rp = row_ptr<float>(src, lane, row_in ? cy : 0);
    for (int tx = threadIdx.x; tx < TX + 2 * BR; tx += TX) {
      const int cx = x0 + tx - BR;
      float v = (row_in && cx >= 0 && cx < src.cols) ? rp[cx] : qnan();
      if (MODE == 2) v = fabsf(v) < BIL_MAXABS ? v : BIL_SENTINEL;   // NaN fails the compare too
      tile[ty][tx] = v;
    }
  }
  __syncthreads();
  const int x = x0 + threadIdx.x;
  if (x >= src.cols) return;
#pragma unroll
  for (int it_ = 0; it_ < BIL_TILES; ++it_) {
    const int ly = it_ * TY + threadIdx.y, y = y0 + ly;
    if (y >= src.rows) break;
    const float value = tile[ly + BR][threadIdx.x + BR];
    if (MODE == 2 ? value == BIL_SENTINEL : isnan(value)) { px<float>(dst, lane, y, x) = qnan(); continue; }
    float res;
    if (MODE == 2) {
      const float log2e = 1.44269504088896341f;
      res = bilateral_px_fast(tile, ly + BR, threadIdx.x + BR, value, 0.5f * log2e / (sigma_floatmap * sigma_floatmap), s2ih * log2e);
    } else if (FAST) {
      bool all_ok = true;
      res = bilateral_px<true>(tile, ly + BR, threadIdx.x + BR, value, sigma_floatmap, dc, s2ih, all_ok);
      if (__builtin_expect(!all_ok, 0)) res = bilateral_px<false>(tile, ly + BR, threadIdx.x + BR, value, sigma_floatmap, dc, s2ih, all_ok);
    } else {
      bool unused = true;
      res = bilateral_px<false>(tile, ly + BR, threadIdx.x + BR, value, sigma_floatmap, dc, s2ih, unused);
    }
    px<float>(dst, lane, y, x) = res;
  }
}
// ---- FAST class, round 6: the pair weight computed ONCE ------------------------------------------------------------------------------------
// The range weight of a pair of pixels is the same seen from either end, bit for bit: with d = value_p - value_q the other end forms -d exactly,
// k * (-d) = -(k * d) exactly, fmaf(-(k d), -d, c) == fmaf(k d, d, c), and the spatial constant c depends on dx^2 + dy^2 only.  k_bilateral<2> evaluates it
// at both ends: 24 v_exp_f32 (2.6 issue slots each) per pixel, 917 VALU instructions per 4-pixel thread -- 0.18 of the HBM roofline, pure issue.  Here a
// pixel evaluates only its 12 FORWARD taps (the rest of its row, the two rows below); its 12 backward taps are the forward weights of the pixels above /
// to the left, which arrive from the neighbouring LANES:
//  * lane <-> column, a wave walks DOWN a strip of 64 columns (60 outputs, two halo-provider lanes per side as in k_pyr_down_dpp); the window columns of a
//    row are DPP wave shifts of the one value a lane loads per row (no LDS, no barriers, one coalesced 256-byte load and store per wave and row);
//  * step t evaluates the forward weights F(t) of row t and spends them at once: output row t + 2 is STARTED (centre tap, then its dy = -2 taps = the
//    dy = +2 forward weights of row t in the lanes x + dx), output row t + 1 gets its dy = -1 taps, output row t its dy = 0, +1, +2 taps and is stored.
//    Three partial sums are in flight, no weight outlives its step, and every output adds its taps in the oracle's raster order (centre first, as
//    k_bilateral<2> does): the result is k_bilateral<2>'s BIT FOR BIT (tests/test_gpu_kernels.py::test_bilateral_shared_weights_equals_the_two_sided_kernel).
//  * a weight from lane x +- 1 is a DPP operand of the v_fmac / v_add that consumes it (free), one from lane x +- 2 costs one v_mov_dpp: 9 moves,
//    48 instructions for the 12 weights and 48 for the 24 taps per pixel instead of 24 x 6 + 24 LDS reads.
// The sentinel scheme is unchanged (an invalid pixel is 1e19 in the window: its pair weights are exactly 0 from either end).
static constexpr int BS_OUT = 60;     // output columns per wave (lanes 2 .. 61)
// output rows a wave walks (two lead-in steps each): 30 while the launch is small (480 = 16 x 30: twice the waves), 60 when lanes fill the chip anyway
static inline int bil_rows_per_wave(int B) { return B >= 32 ? 60 : 30; }
__device__ __forceinline__ float bil_pair_weight(float v0, float tmp, float k, float c) {
  const float d = v0 - tmp;
  return __builtin_amdgcn_exp2f(-__builtin_fmaf(k * d, d, c));
}
// hipcc folds a v_mov_dpp into a v_add_f32 that is its only user, but not into a v_fmac_f32 and not when the shifted value has two users: written out.
// A DPP operand must not have been written by the VALU in the two preceding wait states (the compiler's hazard recogniser does not look into inline
// assembly): the s_nop is part of the statement; it occupies no VALU slot.
#ifndef RGBID_BIL_FUSED_DPP
#define RGBID_BIL_FUSED_DPP 1
#endif
template <int CTRL>
__device__ __forceinline__ void bil_tap_from(float& s1, float& s2, float w_there, float tmp) {   // the weight lives one lane away (CTRL: 0x138 from lane - 1, 0x130 from lane + 1)
#if RGBID_BIL_FUSED_DPP
  static_assert(CTRL == 0x138 || CTRL == 0x130, "wave_shr:1 / wave_shl:1");
  if (CTRL == 0x138)
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %1, %2, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(s1), "+v"(s2) : "v"(w_there), "v"(tmp));
  else
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %2, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %1, %2, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(s1), "+v"(s2) : "v"(w_there), "v"(tmp));
#else
  const float w = dpp_shift<CTRL>(w_there);
  s1 = __builtin_fmaf(w, tmp, s1);
  s2 += w;
#endif
}
__device__ __forceinline__ void bil_tap(float& s1, float& s2, float w, float tmp) { s1 = __builtin_fmaf(w, tmp, s1); s2 += w; }
__global__ __launch_bounds__(256) void k_bilateral_shared(BilSet b0, BilSet b1, int ny, int rpw, LaneMask m) {
  const TileId tid_ = xcd_lane_local_tile();
  const int lane = tid_.lane;
  if (!m.on(lane)) return;
  const bool second = tid_.by >= ny;
  const BilSet& S = second ? b1 : b0;
  const ImgB& src = S.src; const ImgB& dst = S.dst;
  const int rows = src.rows, cols = src.cols;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;   // the wave index as a SCALAR: every row computation below is SALU work
  const int y0 = ((tid_.by - (second ? ny : 0)) * 4 + wave) * rpw;   // the four waves of a workgroup: four row blocks (rpw rows each) of one strip
  if (y0 >= rows) return;
  const int x = tid_.bx * BS_OUT + l - 2;
  const bool col_in = x >= 0 && x < cols;
  const bool writer = l >= 2 && l < 2 + BS_OUT && col_in;
  const int xc = min(max(x, 0), cols - 1);
  const float log2e = 1.44269504088896341f;
  const float sigma_space = 5.f;
  const float s2ih = (float)(0.5 / (double)(sigma_space * sigma_space));
  const float k = 0.5f * log2e / (S.sigma * S.sigma), cs = s2ih * log2e;
  const float c1 = cs * 1.f, c2 = cs * 2.f, c4 = cs * 4.f, c5 = cs * 5.f, c8 = cs * 8.f;   // cs * (dx^2 + dy^2), as k_bilateral<2> forms them
  // a lane's column is one 32-bit byte offset on wave-uniform row bases (global_load ... saddr); the row index is scalar arithmetic
  const char* const src_lane = static_cast<const char*>(src.base) + (size_t)lane * src.lane_stride;
  const unsigned xoff = (unsigned)xc << 2;
  auto issue = [&](int y) { return *reinterpret_cast<const float*>(src_lane + (size_t)min(max(y, 0), rows - 1) * src.pitch + xoff); };
  auto finish = [&](float raw, int y) {   // outside the image, NaN, inf and |v| >= BIL_MAXABS: the sentinel (k_bilateral<2>'s tile load)
    const bool inside = col_in && y >= 0 && y < rows;
    return (inside && fabsf(raw) < BIL_MAXABS) ? raw : BIL_SENTINEL;
  };
  // vs[slot][dx + 2]: the window row of slot `slot` (row t + j lives in slot (phase + j) % 3), column x + dx
  float vs[3][5], s1[3], s2[3];
  auto spread = [&](float v, float r[5]) {
    r[2] = v; r[1] = dpp_shift<0x138>(v); r[0] = dpp_shift<0x138>(r[1]); r[3] = dpp_shift<0x130>(v); r[4] = dpp_shift<0x130>(r[3]);
  };
  spread(finish(issue(y0 - 2), y0 - 2), vs[0]);
  spread(finish(issue(y0 - 1), y0 - 1), vs[1]);
#pragma unroll
  for (int j = 0; j < 3; ++j) { s1[j] = 0.f; s2[j] = 1.f; }
#pragma unroll
  for (int j = 0; j < 5; ++j) vs[2][j] = 0.f;
  float pend = issue(y0);
  const int nsteps = min(rpw, rows - y0) + 2;   // two lead-in steps: the rows above the block only provide their forward weights
  char* const dst_lane = static_cast<char*>(dst.base) + (size_t)lane * dst.lane_stride;
  for (int i0 = 0; i0 < nsteps; i0 += 3) {
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
      const int i = i0 + ph;
      if (i >= nsteps) break;
      const int t = y0 - 2 + i;
      const int a = ph, b = (ph + 1) % 3, c = (ph + 2) % 3;
      // the window's new bottom row (loaded during the previous step); the row after it starts its round trip
      spread(finish(pend, t + 2), vs[c]);
      pend = issue(t + 3);
      const float v0 = vs[a][2];
      // forward weights of pixel (x, t): F0[dx] = w((x, t), (x + dx, t)), F1 / F2: the rows below
      const float F01 = bil_pair_weight(v0, vs[a][3], k, c1), F02 = bil_pair_weight(v0, vs[a][4], k, c4);
      float F1[5], F2[5];
      F1[0] = bil_pair_weight(v0, vs[b][0], k, c5); F1[1] = bil_pair_weight(v0, vs[b][1], k, c2); F1[2] = bil_pair_weight(v0, vs[b][2], k, c1);
      F1[3] = bil_pair_weight(v0, vs[b][3], k, c2); F1[4] = bil_pair_weight(v0, vs[b][4], k, c5);
      F2[0] = bil_pair_weight(v0, vs[c][0], k, c8); F2[1] = bil_pair_weight(v0, vs[c][1], k, c5); F2[2] = bil_pair_weight(v0, vs[c][2], k, c4);
      F2[3] = bil_pair_weight(v0, vs[c][3], k, c5); F2[4] = bil_pair_weight(v0, vs[c][4], k, c8);
      // output row t + 2 starts: centre tap, then its dy = -2 taps (x + dx, t): the weight is F2[-dx] of lane x + dx
      s1[c] = vs[c][2]; s2[c] = 1.f;
      bil_tap_from<0x138>(s1[c], s2[c], dpp_shift<0x138>(F2[4]), vs[a][0]);
      bil_tap_from<0x138>(s1[c], s2[c], F2[3], vs[a][1]);
      bil_tap(s1[c], s2[c], F2[2], vs[a][2]);
      bil_tap_from<0x130>(s1[c], s2[c], F2[1], vs[a][3]);
      bil_tap_from<0x130>(s1[c], s2[c], dpp_shift<0x130>(F2[0]), vs[a][4]);
      // output row t + 1: its dy = -1 taps
      bil_tap_from<0x138>(s1[b], s2[b], dpp_shift<0x138>(F1[4]), vs[a][0]);
      bil_tap_from<0x138>(s1[b], s2[b], F1[3], vs[a][1]);
      bil_tap(s1[b], s2[b], F1[2], vs[a][2]);
      bil_tap_from<0x130>(s1[b], s2[b], F1[1], vs[a][3]);
      bil_tap_from<0x130>(s1[b], s2[b], dpp_shift<0x130>(F1[0]), vs[a][4]);
      // output row t: the rest of its own row, then the two rows below with its own forward weights
      bil_tap_from<0x138>(s1[a], s2[a], dpp_shift<0x138>(F02), vs[a][0]);
      bil_tap_from<0x138>(s1[a], s2[a], F01, vs[a][1]);
      bil_tap(s1[a], s2[a], F01, vs[a][3]);
      bil_tap(s1[a], s2[a], F02, vs[a][4]);
#pragma unroll
      for (int j = 0; j < 5; ++j) bil_tap(s1[a], s2[a], F1[j], vs[b][j]);
#pragma unroll
      for (int j = 0; j < 5; ++j) bil_tap(s1[a], s2[a], F2[j], vs[c][j]);
      if (i >= 2 && writer) {
        const float res = s1[a] * __builtin_amdgcn_rcpf(s2[a]);
        *reinterpret_cast<float*>(dst_lane + (size_t)t * dst.pitch + xoff) = v0 == BIL_SENTINEL ? qnan() : res;
      }
    }
  }
}
static bool bilateral_two_sided() { static const bool v = getenv("RGBID_BILATERAL_TWO_SIDED") != nullptr; return v; }   // A/B and the bit-identity test: k_bilateral<2>
static void launch_bilateral_shared(hipStream_t s, int B, const BilSet& b0, const BilSet& b1, int maps, LaneMask m) {
  const int rpw = bil_rows_per_wave(B), ny = div_up(div_up(b0.src.rows, rpw), 4);
  hipLaunchKernelGGL(k_bilateral_shared, dim3(div_up(b0.src.cols, BS_OUT), maps * ny, B), dim3(256), 0, s, b0, b1, ny, rpw, m);
}
// constants whose 3-instruction division has been verified exhaustively (rgbid_selftest_div_const in the GPU tests): the tracker's two
// range sigmas, 2 * 0.0025 (inverse depth) and 3 (intensity), visodo.cpp:843-844
bool div_const_verified(float c) { return c == 2.f * 0.0025f || c == 3.f; }
void launch_bilateral(hipStream_t s, int B, ImgB src, ImgB dst, float sigma_floatmap, LaneMask m, bool fast) {
  const BilSet bs{src, dst, sigma_floatmap, DivConst{sigma_floatmap, 1.0f / sigma_floatmap}};
  const int ny = div_up(src.rows, TY * BIL_TILES);
  const dim3 g(div_up(src.cols, TX), ny, B), b(TX, TY);
  if (fast && !bilateral_two_sided()) launch_bilateral_shared(s, B, bs, bs, 1, m);
  else if (fast) hipLaunchKernelGGL(k_bilateral<2>, g, b, 0, s, bs, bs, ny, m);
  else if (div_const_verified(sigma_floatmap)) hipLaunchKernelGGL(k_bilateral<1>, g, b, 0, s, bs, bs, ny, m);
  else hipLaunchKernelGGL(k_bilateral<0>, g, b, 0, s, bs, bs, ny, m);
}
void launch_bilateral2(hipStream_t s, int B, ImgB src0, ImgB dst0, float sigma0, ImgB src1, ImgB dst1, float sigma1, LaneMask m, bool fast) {
  const bool both_verified = div_const_verified(sigma0) && div_const_verified(sigma1);
  if (!same_geometry(src0, src1) || (!fast && !both_verified && (div_const_verified(sigma0) || div_const_verified(sigma1)))) {   // the two maps need different kernels
    launch_bilateral(s, B, src0, dst0, sigma0, m, fast); launch_bilateral(s, B, src1, dst1, sigma1, m, fast);
    return;
  }
  const BilSet b0{src0, dst0, sigma0, DivConst{sigma0, 1.0f / sigma0}}, b1{src1, dst1, sigma1, DivConst{sigma1, 1.0f / sigma1}};
  const int ny = div_up(src0.rows, TY * BIL_TILES);
  const dim3 g(div_up(src0.cols, TX), 2 * ny, B), b(TX, TY);
  if (fast && !bilateral_two_sided()) launch_bilateral_shared(s, B, b0, b1, 2, m);
  else if (fast) hipLaunchKernelGGL(k_bilateral<2>, g, b, 0, s, b0, b1, ny, m);
  else if (both_verified) hipLaunchKernelGGL(k_bilateral<1>, g, b, 0, s, b0, b1, ny, m);
  else hipLaunchKernelGGL(k_bilateral<0>, g, b, 0, s, b0, b1, ny, m);
}

}  // namespace rgbid
