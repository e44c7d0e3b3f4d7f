// kernels_gnlevel.hip -- ALL Gauss-Newton iterations of one pyramid level as ONE launch, for engines with few lanes (the loop of
// estimateVisualOdometry, src/visodo.cpp:1041-1281).
//
// The many-lane engine runs an iteration as four dependent launches (lattice residuals -> sigma / nu -> fused normal equations -> 6x6 solve + pose
// update): 72 of the ~116 launches of a step.  With thousands of lanes the ~8 us between two dependent launches are noise; with 64 lanes they are a
// third of the step, with one lane nine tenths.  Lanes are independent, so an iteration only needs synchronisation among the
// workgroups that work on THE SAME lane: here `wpl` workgroups of 512 threads own a lane for the whole level and meet at a per-lane barrier (one
// atomic counter per lane, agent-scope release / acquire) between the four phases -- 4 barriers of ~2 us per iteration instead of 4 launches.  All
// workgroups of the grid must be resident for the spin-waits to be safe: the launcher sizes wpl from the kernel's measured occupancy and refuses
// (the engine then takes the launch-per-phase path) when lanes x 2 workgroups do not fit.
//
// Every phase is the SAME device code as its stand-alone kernel -- FusedLatticeGetter / sigma_core (sigma_device.h), build_system_block
// (system_device.h: the launch plan's blocks walked as virtual 256-thread workgroups, two per workgroup side by side), solve_update_block
// (engine_device.h) -- so records are bit-identical to the launch-per-phase path of the same engine (tests/test_gpu_engine.py).
#define RGBID_ROW_PTR_MUL64   // as kernels_system.hip (common.h row_ptr)
#include "kernels.h"
#include "system_device.h"
#include "engine_device.h"
#include "sigma_device.h"
#include "gnlevel.h"
#include <cstdlib>

namespace rgbid {

namespace {

constexpr int GL_T = 512;
static_assert(GL_T == SIG_T && GL_T == 2 * SYS_T, "a workgroup is one sigma / nu workgroup and two normal-equation workgroups");

// barrier among the workgroups of one lane: every workgroup adds 1 to the lane's counter and waits until it reaches `target` (monotone within a
// step: the counters are zeroed by k_step_begin).  Release before the add, acquire after the wait: what a phase wrote is visible to every
// workgroup of the lane in the next phase, across XCDs.
__device__ __forceinline__ void lane_barrier(unsigned* counter, unsigned target, bool fences = true) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                            // ONE write-back of what this workgroup wrote
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);   // relaxed polls: no cache maintenance per poll
    if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                            // ONE invalidate before the next phase reads
  }
  __syncthreads();
}

// The four phases are separate NON-INLINED functions that read the (static) argument block from device memory: inlined into one kernel body, the
// normal-equation loop inherited the register pressure of everything that lives across the level's iteration loop and spilled ~95 scratch
// accesses per 4-pixel unit (measured: the persistent kernel 4 x slower than the launches it replaces).  Each phase now gets the register
// allocation its stand-alone kernel has; the calls cost four per iteration.
__device__ __forceinline__ void phase_lattice(const GnLevelArgs* __restrict__ ap, int lane, int w, int fast) {
  const GnLevelArgs& a = *ap;
  const int tid = threadIdx.x;
  FusedLatticeGetter g{a.Wcur, a.Icur, a.W0, a.I0, a.wp[lane], lane, a.lat_stride_px, a.interp_mode, fast};
  float* r = a.lat_res + (size_t)lane * a.lat_res_lane_stride;
  const float* k = a.kf_lat ? a.kf_lat + (size_t)lane * a.kf_lat_lane_stride : nullptr;
  for (int i = w * GL_T + tid; i < a.n_lat; i += a.wpl * GL_T) {
    const int ly = i / a.lat_cols, lx = i - ly * a.lat_cols;
    float rd, ri;
    if (k) g.both_given(ly, lx, k[i], k[a.n_lat + i], rd, ri);
    else g.both(ly, lx, rd, ri);
    r[i] = rd; r[a.n_lat + i] = ri;
  }
}
__device__ __forceinline__ void phase_sigma(const GnLevelArgs* __restrict__ ap, int lane, int w, double* sm_sig) {
  const GnLevelArgs& a = *ap;
  const int tid = threadIdx.x;
  BlockSum sm(sm_sig);
  Samples<true, ArrayGetter> S(ArrayGetter{a.lat_res + (size_t)lane * a.lat_res_lane_stride + (size_t)w * a.n_lat, 0}, a.n_lat, tid);
  float bias = 0.f, sigma = w == 0 ? 0.0025f : 5.f, nu = 5.f;
  sigma_core(S, a.T, 0, a.mestimator, bias, sigma, nu, sm);
  if (tid == 0) {
    if (w == 0) { a.sp[lane].bias_d = bias; a.sp[lane].sigma_d = sigma; a.sp[lane].nu_d = nu; }
    else { a.sp[lane].bias_i = bias; a.sp[lane].sigma_i = sigma; a.sp[lane].nu_i = nu; }
  }
}
template <int FUSED, int WM>
__device__ __forceinline__ void phase_system(const GnLevelArgs* __restrict__ ap, int lane, int w, float (*sm_sys)[SYS_T / 64][SYS_TERMS + 1]) {
  const GnLevelArgs& a = *ap;
  const int tid = threadIdx.x;
  const ByLane<SysParams> ps{a.sp};
  const FusedArgs fa{a.wp, a.interp_mode};
  const int half = tid >> 8, vtid = tid & (SYS_T - 1);
  for (int vb0 = 2 * w; vb0 < a.nblk; vb0 += 2 * a.wpl) {
    const int vb = vb0 + half;
    if (vb < a.nblk)
      build_system_block<ByLane<SysParams>, true, FUSED, WM>(a.W0, a.I0, a.gWx, a.gWy, a.gIx, a.gIy, a.Wcur, a.Icur, ps, a.partials, a.nblk, a.upt, fa, a.tp, lane, vb, vtid, sm_sys[half]);
    else
      __syncthreads();   // the one barrier of the other half's block_reduce_store
    __syncthreads();     // sm_sys is reused by the next pair of blocks
  }
}
__device__ __forceinline__ void phase_solve(const GnLevelArgs* __restrict__ ap, int lane, int next_level, double (*sm_red)[32], double* sm_sums) {
  const GnLevelArgs& a = *ap;
  eng::solve_update_block(a.partials, a.nblk, a.st, a.f, a.wp, a.c, next_level, lane, (int)threadIdx.x, sm_red, sm_sums);
}

template <int FUSED, int WM>
__global__ __launch_bounds__(GL_T) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_gn_level(const GnLevelArgs* __restrict__ ap, int debug_skip) {
  const int wpl = ap->wpl, iters = ap->iters;
  const int lane = blockIdx.x / wpl, w = blockIdx.x - lane * wpl;
  __shared__ float sm_sys[2][SYS_T / 64][SYS_TERMS + 1];
  __shared__ double sm_sig[SIG_SM];
  __shared__ double sm_red[8][32];
  __shared__ double sm_sums[SYS_TERMS];
  unsigned* const counter = ap->barrier + (size_t)lane * GN_BARRIER_STRIDE;   // one 256-byte line per lane: the lanes' pollers do not meet at one memory channel
  unsigned target = ap->barrier_base;
  const bool fences = !(debug_skip & 16);
  // the level's constants (k_set_sys), then everybody sees them
  if (w == 0 && threadIdx.x == 0) eng::set_sys_lane(ap->sp, ap->st, ap->f.track, ap->c, ap->level, 0, lane);
  target += wpl; lane_barrier(counter, target, fences);
  for (int it = 0; it < iters; ++it) {
    const bool on = __hip_atomic_load(&ap->f.gn[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // lanes that lost tracking keep meeting at the barriers
    // ---- phase 1: the residual lattice at the current pose (k_lattice_residuals_fused)
    if (on && !(debug_skip & 1)) phase_lattice(ap, lane, w, FUSED == 2 ? 1 : 0);
    target += wpl; lane_barrier(counter, target, fences);
    // ---- phase 2: bias, sigma, nu of both channels (k_sigma_pair_arrays): workgroup 0 the inverse depth, workgroup 1 the intensity
    if (on && w < 2 && !(debug_skip & 2)) phase_sigma(ap, lane, w, sm_sig);
    target += wpl; lane_barrier(counter, target, fences);
    // ---- phase 3: the fused normal equations (k_build_system): the plan's nblk blocks as virtual 256-thread workgroups, two side by side
    if (on && !(debug_skip & 4)) phase_system<FUSED, WM>(ap, lane, w, sm_sys);
    target += wpl; lane_barrier(counter, target, fences);
    // ---- phase 4: 6x6 solve, pose update, next warp (k_solve_update)
    if (on && w == 0 && !(debug_skip & 8)) phase_solve(ap, lane, it == iters - 1 ? ap->next_level_after : ap->level_for_warp, sm_red, sm_sums);
    target += wpl; lane_barrier(counter, target, fences);
  }
}

template <int FUSED, int WM>
int occupancy_blocks() {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_gn_level<FUSED, WM>, GL_T, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

}  // namespace

int gn_level_barriers(int iters) { return 1 + 4 * iters; }

// workgroups per lane the persistent kernel would run with `lanes` lanes, or 0 when the grid cannot be resident with at least 2 per lane
int gn_level_workgroups_per_lane(int lanes, bool fast, int weight_mode) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) { (void)hipGetLastError(); return 0; }
  const int per_cu = fast ? (weight_mode == 1 ? occupancy_blocks<2, 1>() : occupancy_blocks<2, 0>()) : occupancy_blocks<1, 0>();
  const long long capacity = (long long)per_cu * cus;
  long long wpl = capacity / lanes;
  if (wpl < 2) return 0;
  if (wpl > 64) wpl = 64;       // a lane has at most a few hundred blocks of work per phase
  if (const char* e = getenv("RGBID_GN_WPL")) { const int v = atoi(e); if (v >= 2 && v <= wpl) wpl = v; }   // experiments (tools/experiments)
  return (int)wpl;
}

bool gn_level_supported(const GnLevelArgs& a) { return a.wpl >= 2 && a.n_lat <= SIG_T * SIG_MAXPT; }

// a_dev: the argument block in device memory (static per engine and level: the engine uploads it once); a: its host copy
int launch_gn_level(hipStream_t s, int lanes, const GnLevelArgs& a, const GnLevelArgs* a_dev, bool fast, int weight_mode) {
  if (!gn_level_supported(a) || !a_dev) return -1;
  int debug_skip = 0;
  if (const char* e = getenv("RGBID_GN_DEBUG_SKIP")) debug_skip = atoi(e);   // timing experiments only: phases switched off (results are garbage)
  const dim3 g((unsigned)lanes * (unsigned)a.wpl), b(GL_T);
  if (!fast) hipLaunchKernelGGL((k_gn_level<1, 0>), g, b, 0, s, a_dev, debug_skip);
  else if (weight_mode == 1 && a.interp_mode == 1) hipLaunchKernelGGL((k_gn_level<2, 1>), g, b, 0, s, a_dev, debug_skip);
  else hipLaunchKernelGGL((k_gn_level<2, 0>), g, b, 0, s, a_dev, debug_skip);
  return 0;
}

}  // namespace rgbid
