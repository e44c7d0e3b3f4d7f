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
//    moves 32 instead of 52 B/px and is one launch instead of two.  0.72-0.75 of the HBM peak at 761 VALU instructions per 4-pixel unit in round 3;
//    0.65-0.68 since round 4, when the kernel began to take the oracle's discrete decisions at every pixel (guard_band.h: ~ +36 instructions per unit):
//    memory (6.5 TB/s streaming ceiling), VALU issue (~2.4 ms of the 3.4 ms launch) and the L1 address path (~1.5 ms) all run at 45-90 %
//    under 4 waves per SIMD -- what is left is their imperfect overlap (profiles/r03_experiments/).  Round 5: the guard as lane constants (guard_band.h (3'),
//    (3'')) and the 27-term update as domino-tiled PACKED FMAs (system_device.h acc_pk_row: 12 v_pk_fma_f32 + 3 v_fma_f32 per row instead of 27; a packed
//    instruction is two IEEE operations for ~1.45 issue slots on gfx950) bring the unit to 626 VALU instructions: 0.70-0.73 of the peak = 0.89-0.92 of the
//    chip's measured copy ceiling (6.29 TB/s) -- further instruction trimming no longer moves the launch (profiles/r05_experiments/packed_fp32.md).
//    FUSED = 1: the same with the exact-numerics warps (0.44).
// Each lane of a wave64 owns 4 consecutive pixels (16-byte loads, 1 KiB per wave per map, fully coalesced); 27 fp32 accumulators live in
// VGPRs; the workgroup (4 waves) reduces with DPP wave reductions -> 4x27 floats of LDS -> doubles, and writes ONE 27-double partial row
// per workgroup.  Partials are summed in a fixed order by a second tiny kernel (or by the batched engine's solve kernel), so results are
// deterministic -- no floating-point atomics.  blockIdx is remapped so that the 8 XCDs each stream a contiguous slab.
#define RGBID_ROW_PTR_MUL64   // common.h row_ptr: this file forms row addresses with the 64-bit multiply (its scalar unit does them; the 24-bit VALU form costs VGPRs here)
#include "kernels.h"
#include <atomic>
#include "system_device.h"
#include <hip/hip_ext.h>

namespace rgbid {


// optional event pair that brackets exactly the next normal-equation kernel dispatch (hipExtLaunchKernelGGL:
// the events carry the dispatch's own start / end timestamps, i.e. the kernel duration a profiler reports)
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
void set_system_kernel_events(hipEvent_t start, hipEvent_t stop) { g_ev_start = start; g_ev_stop = stop; }

template <class PS, bool VEC, int LEVEL, int FUSED, int WMK = 0>
__global__ __launch_bounds__(SYS_T) __attribute__((amdgpu_waves_per_eu(FUSED == 2 ? FUSED_WAVES : 1, 8))) void k_build_system(ImgB W0, ImgB I0, ImgB gWx, ImgB gWy, ImgB gIx, ImgB gIy, ImgB W1, ImgB I1,
                                                        PS ps, double* partials, int nblk, int upt, LaneMask m, FusedArgs fa, SysTiles tp) {
  int gb = xcd_slab_block(blockIdx.x, gridDim.x);
  int lane = gb / nblk, blk = gb - lane * nblk;
  if (!m.on(lane)) return;
  __shared__ float sm[SYS_T / 64][SYS_TERMS + 1];
  build_system_block<PS, VEC, FUSED, WMK>(W0, I0, gWx, gWy, gIx, gIy, W1, I1, ps, partials, nblk, upt, fa, tp, lane, blk, (int)threadIdx.x, sm);
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
    // Few lanes (round 5): the plan above makes thousands of short workgroups (32 lanes: 4 800 of 2 tiles, 4.7 rounds of the chip's resident capacity), each with
    // its own start-up and epilogue.  Up to 32 lanes the strip-aligned value with the shortest schedule is taken instead -- rounds of 4 four-wave workgroups per
    // compute unit times the length of a workgroup (u tiles + an epilogue of ~0.6 tile): ONE round of up to 1 024 long workgroups (32 lanes: u = 10).  Measured in
    // the bench: +10 % frames/s at 16 lanes, +2 % at 31 (the level-0 launch alone: -12 % at 8 lanes, -9 % at 32); at 64 lanes the level-0 launch is 6 % SLOWER in
    // the bench with it (3 % faster alone) and at 128 lanes 4.7 % slower -- a second round of long workgroups balances worse than six of short ones -- so it stops at 32.
    if (B <= 32) {
      const long long slots = (long long)device_cus() * 4;
      long long pick = u;
      double pick_cost = 1e300;
      for (long long d = 1; d <= max_upt && d <= steps; ++d) {
        if (!(ty % d == 0 || d % ty == 0)) continue;
        const long long wgs = ((steps + d - 1) / d) * (long long)B;
        const double cost = (double)((wgs + slots - 1) / slots) * ((double)d + 0.6);
        if (cost < pick_cost - 1e-9 || (cost < pick_cost + 1e-9 && d > pick)) { pick_cost = cost; pick = d; }
      }
      u = pick;
    }
  }
  nb = (steps + u - 1) / u;                                  // drop workgroups that would get no step
  *upt = (int)u;
  *nblk = (int)nb;
}

// the launch plan of the 16-byte path for this geometry and lane count, for callers that walk its blocks themselves (tools/experiments/persistent_gn_level/kernels_gnlevel.hip: a measured experiment, profiles/r04_experiments/persistent_gn_level.md)
void system_plan_vec(int rows, int cols, int B, int* upt, int* nblk, SysTiles* tp) {
  system_plan(rows, cols, B, true, upt, nblk);
  *tp = system_tiles(rows, cols);
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
