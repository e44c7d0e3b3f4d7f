// engine.hip -- batched, device-resident VisodoTracker (C-ABI in include/rgbid_engine.h).
//
// One `step` == VisodoTracker::trackNewFrame (src/visodo.cpp:1967-2247) for `lanes` independent trackers
// in lock-step.  Every image kernel is the batched kernel of kernels.h (blockIdx.z / block-id = lane); the
// host-side double-precision logic of the reference (constant-velocity prediction, LLT solve, exp-map
// update, covariance, keyframe decisions, pose composition) runs in small per-lane kernels below, so the
// whole step is a static launch sequence: no host round trip, capturable as one hipGraph.  Data-dependent
// control flow (lost tracking, keyframe switches, fuse-vs-reset) is expressed as per-lane flag arrays that
// predicate whole workgroups (LaneMask).
#include "../../include/rgbid_engine.h"
#include "ctx.h"
#include "kernels.h"
#include "../../include/rgbid/se3.h"
#include "engine_device.h"
#include "sigma_device.h"   // FusedLatticeGetter: the lattice pre-pass that carries the few-lane plan's update prologue

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

// The per-lane scalar kernels hold 6x6 / 3x3 double matrices in registers: one wave per SIMD may take the whole register file (the default budget of 128
// VGPRs spilled 660 - 1 250 bytes per thread to scratch, and dependent scratch round trips were most of these kernels' run time)
#define RGBID_SCALAR_KERNEL __attribute__((amdgpu_waves_per_eu(1, 1)))

#pragma clang fp contract(off)   // the per-lane scalar kernels below (double-precision tracker logic): no FMA contraction, as se3.h / engine_device.h

using namespace rgbid;
using namespace rgbid::eng;

constexpr int MAXL = 8;

namespace {

__device__ void reset_odometry_keyframe(LaneState& s) {
  // resetOdometryKeyframe visodo.cpp:1541-1575
  s.odoKF_count = 0;
  double J[36], tn[3], S[9];
  se3::m6_zero(J);
  se3::m6_set_block(J, 0, 0, s.o2i_next_R, 1.0);
  se3::m6_set_block(J, 3, 3, s.o2i_next_R, 1.0);
  se3::m3_mulv(s.o2i_next_R, s.delta_t, tn);
  se3::skew(tn, S);
  se3::m6_set_block(J, 0, 3, S, 1.0);
  se3::m6_JCJt_add(J, s.delta_cov, s.o2i_next_cov);
  for (int i = 0; i < 3; ++i) s.o2i_next_t[i] = tn[i] + s.o2i_next_t[i];
  se3::m3_mul(s.o2i_next_R, s.delta_R, s.o2i_next_R);
  s.last_odoKF_index = s.global_time;
  se3::m3_copy(s.last_est_R, s.odoKF_R);
  for (int i = 0; i < 3; ++i) s.odoKF_t[i] = s.last_est_t[i];
  se3::m3_id(s.delta_R);
  for (int i = 0; i < 3; ++i) s.delta_t[i] = 0.0;
  se3::m6_zero(s.delta_cov);
}

__device__ void reset_integration_keyframe(LaneState& s, const StepCfg& c, const Flags& f, int lane) {
  // resetIntegrationKeyframe visodo.cpp:1577-1672
  s.integrKF_count = 0;
  double J[36], tn[3], S[9];
  se3::m6_zero(J);
  se3::m6_set_block(J, 0, 0, s.o2i_next_R, 1.0);
  se3::m6_set_block(J, 3, 3, s.o2i_next_R, 1.0);
  se3::m3_mulv(s.o2i_next_R, s.delta_t, tn);
  se3::skew(tn, S);
  se3::m6_set_block(J, 0, 3, S, 1.0);
  se3::m6_JCJt_add(J, s.delta_cov, s.o2i_next_cov);
  for (int i = 0; i < 3; ++i) s.o2i_next_t[i] = tn[i] + s.o2i_next_t[i];
  se3::m3_mul(s.o2i_next_R, s.delta_R, s.o2i_next_R);
  if (c.kf_hdr) {
    // the Keyframe record + SEQ_KF constraint for the back-end (:1610-1652): header here, the four images by k_export_keyframe
    int slot = s.kf_exported % c.kf_cap;
    rgbid_keyframe_header& h = c.kf_hdr[(size_t)lane * c.kf_cap + slot];
    double lastT[9], d[3], Jn[36], Jl[36], S2[9], SR[9];
    se3::m3_T(s.o2i_last_R, lastT);
    se3::m3_mul(lastT, s.o2i_next_R, h.R_rel);
    for (int i = 0; i < 3; ++i) d[i] = s.o2i_next_t[i] - s.o2i_last_t[i];
    se3::m3_mulv(lastT, d, h.t_rel);
    se3::m6_zero(Jn); se3::m6_set_block(Jn, 0, 0, lastT, 1.0); se3::m6_set_block(Jn, 3, 3, lastT, 1.0);
    se3::m6_zero(Jl); se3::m6_set_block(Jl, 0, 0, lastT, -1.0); se3::m6_set_block(Jl, 3, 3, lastT, -1.0);
    se3::skew(h.t_rel, S2); se3::m3_mul(S2, lastT, SR); se3::m6_set_block(Jl, 0, 3, SR, 1.0);
    se3::m6_zero(h.cov_rel);
    se3::m6_JCJt_add(Jl, s.o2i_last_cov, h.cov_rel);
    se3::m6_JCJt_add(Jn, s.o2i_next_cov, h.cov_rel);
    h.id = s.last_integrKF_index; h.end_id = s.global_time; h.lane = lane; h.seq = s.kf_exported;
    se3::m3_copy(s.integrKF_R, h.R);
    for (int i = 0; i < 3; ++i) h.t[i] = s.integrKF_t[i];
    f.kf_slot[lane] = slot;
    s.kf_exported++;
  }
  s.last_integrKF_index = s.global_time;
  se3::m3_copy(s.last_est_R, s.integrKF_R);
  for (int i = 0; i < 3; ++i) s.integrKF_t[i] = s.last_est_t[i];
  se3::m3_copy(s.delta_R, s.o2i_last_R);
  for (int i = 0; i < 3; ++i) s.o2i_last_t[i] = s.delta_t[i];
  for (int i = 0; i < 36; ++i) s.o2i_last_cov[i] = s.delta_cov[i];
  se3::m3_id(s.o2i_next_R);
  for (int i = 0; i < 3; ++i) s.o2i_next_t[i] = 0.0;
  se3::m6_zero(s.o2i_next_cov);
}

// ---- step begin: first-frame initialisation or GN start (visodo.cpp:1994-2045, 1012-1035) ------------------
// Also what two tiny launches used to do: the lane's eight covisibility counters are zeroed (was a memset node), and the SysParams of the first
// Gauss-Newton stage are set (was the first k_set_sys; the later stages' are set by the k_solve_update that ends the stage before them).
__global__ __launch_bounds__(64) RGBID_SCALAR_KERNEL void k_step_begin(LaneState* st, Flags f, WarpParams* wp, SysParams* sp, StepCfg c, int B, unsigned int* counts, int sys_level, int sys_cov) {
  int lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= B) return;
  LaneState& s = st[lane];
  for (int which = 0; which < 4; ++which) { counts[(which * B + lane) * 2 + 0] = 0u; counts[(which * B + lane) * 2 + 1] = 0u; }
  f.vis[lane] = 0; f.overlap[lane] = 0; f.fuse[lane] = 0; f.kf_slot[lane] = -1;   // per-step launch predicates, not tracker state
  if (c.active && !c.active[lane]) {
    // no frame for this lane: nothing of its tracker state moves (only the per-step status word reads 0), every kernel of the step is
    // predicated off for it, and its record repeats the last pose / covisibility figures with status 0
    f.first[lane] = 0; f.track[lane] = 0; f.gn[lane] = 0; f.sw_odo[lane] = 0; f.sw_int[lane] = 0; f.maps[lane] = 0;
    s.status = 0;
    return;
  }
  s.status = 0; s.gn_failed = 0; s.vis_odo = 0.f; s.vis_int = 0.f;
  s.rmse_prev = 9999.f;   // RMSE_prev of estimateVisualOdometry (:1134): per frame, carried across the levels
  if (s.global_time == 0) {
    f.first[lane] = 1; f.track[lane] = 0; f.gn[lane] = 0; f.sw_odo[lane] = 1; f.sw_int[lane] = 1; f.maps[lane] = 1;
    s.global_time = 1;
    s.odoKF_count = 0; s.last_odoKF_index = 0;
    se3::m3_id(s.odoKF_R); se3::m3_id(s.integrKF_R); se3::m3_id(s.delta_R); se3::m3_id(s.last_est_R);
    se3::m3_id(s.o2i_last_R); se3::m3_id(s.o2i_next_R);
    for (int i = 0; i < 3; ++i) { s.odoKF_t[i] = s.integrKF_t[i] = s.delta_t[i] = s.last_est_t[i] = s.o2i_last_t[i] = s.o2i_next_t[i] = 0.0; s.velocity[i] = s.omega[i] = 0.0; }
    se3::m6_zero(s.delta_cov); se3::m6_zero(s.o2i_last_cov); se3::m6_zero(s.o2i_next_cov);
    s.integrKF_count = 0; s.last_integrKF_index = 0;
    s.lost = 0;
    s.status = RGBID_ST_FIRST | RGBID_ST_ODO_KF | RGBID_ST_INTEGR_KF;
    return;
  }
  f.first[lane] = 0; f.track[lane] = 1; f.gn[lane] = 1; f.sw_odo[lane] = 0; f.sw_int[lane] = 0; f.maps[lane] = 0;
  se3::m3_copy(s.delta_R, s.dprev_R);
  for (int i = 0; i < 3; ++i) s.dprev_t[i] = s.delta_t[i];
  for (int i = 0; i < 36; ++i) s.dprev_cov[i] = s.delta_cov[i];
  se3::m3_copy(s.delta_R, s.prev_R);
  for (int i = 0; i < 3; ++i) s.prev_t[i] = s.delta_t[i];
  if ((s.global_time > 1) && (c.motion_model == RGBID_CONSTANT_VELOCITY) && (!s.lost)) {
    double vt[3], wt[3], dR[9], dt[3], tmp[3];
    const float dt_frame = *c.delta_t;
    for (int i = 0; i < 3; ++i) { vt[i] = s.velocity[i] * dt_frame; wt[i] = s.omega[i] * dt_frame; }
    se3::expmap(wt, vt, dR, dt);
    se3::m3_mulv(s.prev_R, dt, tmp);
    for (int i = 0; i < 3; ++i) s.cur_t[i] = tmp[i] + s.prev_t[i];
    se3::m3_mul(s.prev_R, dR, s.cur_R);
  } else {
    se3::m3_copy(s.prev_R, s.cur_R);
    for (int i = 0; i < 3; ++i) s.cur_t[i] = s.prev_t[i];
  }
  set_warp_from_pose(c, c.start_warp_level, s.cur_R, s.cur_t, wp[lane]);
  if (sys_level >= 0) set_sys_lane(sp, st, f.track, c, sys_level, sys_cov, lane);
}

// ---- one GN update: reduce partials, LLT solve, exp-map, pose update, next warp (visodo.cpp:1242-1274) -------
// sys_level >= 0: this is the last update of a stage -- the per-level constants of the lane's SysParams for the stage that follows (the next level's
// iterations, or the covariance pass: sys_cov) are set here (every lane, as the separate k_set_sys launch did; nothing of the solve reads them)
// Register budget: the 256-thread form takes the whole register file (RGBID_SCALAR_KERNEL: the first builds spilled under the default budget); the kernel needs 147
// VGPRs, so the one-wave form may share a SIMD three ways (170 VGPRs each): 12 lanes per compute unit at a time, 2 048 lanes in one round.
template <int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, NT == 64 ? 3 : 1))) void k_solve_update(const double* partials, int nblk, LaneState* st, Flags f, WarpParams* wp,
                                                      StepCfg c, int next_level, SysParams* sp, int sys_level, int sys_cov, int pin) {
  int lane = blockIdx.x;
  if (sys_level >= 0 && threadIdx.x == NT / 4) set_sys_lane(sp, st, f.track, c, sys_level, sys_cov, lane);   // NT = 256: a wave of its own, beside thread 0's solve
  if (!f.lvl[lane]) return;   // lvl == gn unless CHI_SQUARED termination has ended this lane's level early
  __shared__ double sm[8][32];
  __shared__ double sums[SYS_TERMS];
  solve_update_block<NT>(partials, nblk, st, f, wp, c, next_level, lane, (int)threadIdx.x, sm, sums, pin);
}

// ---- few-lane plan (round 6): the update of iteration k as the PROLOGUE of iteration k + 1's lattice pre-pass -----------------------------------------------
// With a handful of lanes a step is a chain of ~96 dependent launches of 4.5 - 5 us each whatever they do; the 6x6 solve + pose update is a
// launch of its own only because the NEXT kernel needs its result.  Here every workgroup of that next kernel -- the residual-lattice pre-pass -- reduces the
// lane's partial sums in the fixed order and runs the update itself, redundantly (the same doubles in every workgroup: bit-identical to the separate launch),
// keeps the new warp in LDS for its own samples, and workgroup 0 of the lane stores what the later launches read (pose, increment, warp, flags, the next
// stage's SysParams).  The kernel boundary gives the visibility that sank the update as the TAIL of the normal-equation kernel (device-scope fences:
// profiles/r04_experiments/gn_solve_tail.md).  The pose a workgroup reads must be the one the launch began with even if workgroup 0 has finished: the updates
// alternate between the lane's two pose buffers (pin: this launch reads alt_*, writes cur_*; else the other way round).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 3))) void k_lattice_after_update(ImgB Wcur, ImgB W0, ImgB Icur, ImgB I0, int interp_mode, int n, int lcols, int stride, float* res,
                                                                                  size_t res_lane_stride, const float* kf_lat, size_t kf_lat_lane_stride, int fast,
                                                                                  const double* partials, int nblk, LaneState* st, Flags f, WarpParams* wp, StepCfg c, int level,
                                                                                  SysParams* sp, int sys_level, int sys_cov, int pin) {
  const int lane = blockIdx.y, tid = (int)threadIdx.x;
  if (sys_level >= 0 && blockIdx.x == 0 && tid == 64) set_sys_lane(sp, st, f.track, c, sys_level, sys_cov, lane);
  if (!f.lvl[lane]) return;
  __shared__ double sm[8][32];
  __shared__ double sums[SYS_TERMS];
  __shared__ WarpParams next_warp;
  __shared__ int update_ok;
  reduce_partials<256>(partials, nblk, lane, tid, sm, sums);
  if (tid == 0) {
    LaneState& s = st[lane];
    GnUpdate u;
    gn_update(sums, pin ? s.alt_R : s.cur_R, pin ? s.alt_t : s.cur_t, u);
    if (!u.failed) set_warp_from_pose(c, level, u.R, u.t, next_warp);
    update_ok = u.failed ? 0 : 1;
    if (blockIdx.x == 0) gn_commit(u, s, pin ? s.cur_R : s.alt_R, pin ? s.cur_t : s.alt_t, f, wp, &next_warp, c, level, lane);
  }
  __syncthreads();
  if (!update_ok) return;   // the lane's Gauss-Newton has failed (NaN): workgroup 0 has cleared its flags, nothing reads these residuals
  const int i = blockIdx.x * 256 + tid;
  if (i >= n) return;
  FusedLatticeGetter g{Wcur, Icur, W0, I0, next_warp, lane, stride, interp_mode, fast};
  const int ly = i / lcols, lx = i - ly * lcols;
  float rd, ri;
  if (kf_lat) {
    const float* k = kf_lat + (size_t)lane * kf_lat_lane_stride;
    g.both_given(ly, lx, k[i], k[n + i], rd, ri);
  } else {
    g.both(ly, lx, rd, ri);
  }
  float* r = res + (size_t)lane * res_lane_stride;
  r[i] = rd; r[n + i] = ri;
}
// lanes up to which the update rides in the next launch (0: never).  The kernel may take 170 VGPRs (the update needs 147): three workgroups per compute unit, 768
// on the chip = the ~48 lattice workgroups of 16 lanes at 640x480 in ONE round.  Measured (profiles/r06_experiments/update_prologue.md): 1 - 2 % per frame at 1 and 8
// lanes, eager and as a hipGraph -- a dependent launch's ~4.5 us floor is mostly the short kernel's own run time, so folding 17 launches away returns 10 - 16 us, not
// 17 x 4.5 --, 5 % SLOWER at 16 lanes and 9 % at 32 (the redundant updates start to queue): on up to 8 lanes
inline int update_prologue_max_lanes() {
  const char* e = getenv("RGBID_ENGINE_UPDATE_PROLOGUE_LANES");   // read when a step is enqueued / captured (tests and A/B runs switch it per engine)
  return e ? atoi(e) : 8;
}
// workgroup size of the per-lane reduce-and-solve kernels (engine_device.h reduce_partials): one wave per lane once there are more lanes than compute units
inline int scalar_block_threads(int B) { return B > 256 ? 64 : 256; }

// ---- CHI_SQUARED termination (visodo.cpp:1134-1164) ------------------------------------------------------------------------------------
// a level begins: every lane whose Gauss-Newton is alive iterates it
__global__ void k_level_begin(Flags f, int B) {
  int lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane < B) f.lvl[lane] = f.gn[lane];
}
// before iteration `iter` >= 1 of a level, after the warp: RMSE of the full-lattice chi-square (chi_out[lane] = {chi_square, chi_test, Ndof}); from the
// third iteration on a growing RMSE undoes the previous increment and ends the level FOR THIS LANE (the flag masks the level's remaining launches); the
// next stage's first warp is projected from the restored pose
__global__ __launch_bounds__(64) RGBID_SCALAR_KERNEL void k_chi_decide(LaneState* st, Flags f, const float* chi_out, WarpParams* wp, StepCfg c, int iter, int after_level, int B) {
  int lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= B || !f.lvl[lane]) return;
  LaneState& s = st[lane];
  const float RMSE = sqrtf(chi_out[3 * lane + 0]) / sqrtf(chi_out[3 * lane + 2]);
  if (iter != 1 && RMSE > s.rmse_prev) {
    double d[3], tmp[3];
    for (int i = 0; i < 3; ++i) d[i] = s.cur_t[i] - s.inc_t[i];
    se3::m3_mulv(s.inc_inv_R, d, tmp);
    for (int i = 0; i < 3; ++i) s.cur_t[i] = tmp[i];
    se3::m3_mul(s.inc_inv_R, s.cur_R, s.cur_R);
    f.lvl[lane] = 0;
    set_warp_from_pose(c, after_level, s.cur_R, s.cur_t, wp[lane]);
    return;
  }
  s.rmse_prev = RMSE;
}

// ---- end of estimateVisualOdometry + pose bookkeeping of trackNewFrame (visodo.cpp:1367-1468, 2051-2170) -----
template <int NT>
__global__ __launch_bounds__(NT) RGBID_SCALAR_KERNEL void k_frame_finish(const double* partials, int nblk, LaneState* st, Flags f, const SysParams* sp,
                                                      WarpParams* vis_ab, WarpParams* vis_ba, WarpParams* ivis_ab, WarpParams* ivis_ba,
                                                      rgbid_pose_record* rec, StepCfg c) {
  int lane = blockIdx.x;
  if (!f.track[lane]) return;
  __shared__ double sm[8][32];
  __shared__ double sums[SYS_TERMS];
  LaneState& s = st[lane];
  bool ok = !s.gn_failed;
  reduce_partials<NT>(partials, ok ? nblk : 0, lane, (int)threadIdx.x, sm, sums);   // a failed lane's sums are not used (zeros, as before)
  if (threadIdx.x != 0) return;
  rgbid_pose_record& R = rec[lane];
  R.frame = s.global_time;
  R.sigma_int = s.rec_sigma_i; R.sigma_depthinv = s.rec_sigma_d; R.nu_int = s.rec_nu_i; R.nu_depthinv = s.rec_nu_d;
  if (ok) {
    double A[36];
    int shift = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 7; ++j) {
        double v = sums[shift++];
        if (j != 6) A[j * 6 + i] = A[i * 6 + j] = v;
      }
    se3::m3_copy(s.cur_R, s.delta_R);
    for (int i = 0; i < 3; ++i) s.delta_t[i] = s.cur_t[i];
    se3::inverse6(A, s.delta_cov);  // resulting_covariance = A_final.inverse() :1409
    // velocity / omega :1459-1468
    double pT[9], dR[9], d[3], dt[3], twist[6];
    se3::m3_T(s.prev_R, pT);
    se3::m3_mul(pT, s.cur_R, dR);
    for (int i = 0; i < 3; ++i) d[i] = s.cur_t[i] - s.prev_t[i];
    se3::m3_mulv(pT, d, dt);
    se3::logmap(dR, dt, twist);
    float inv_dt = 1.f / *c.delta_t;
    for (int i = 0; i < 3; ++i) { s.velocity[i] = twist[i] * (double)inv_dt; s.omega[i] = twist[3 + i] * (double)inv_dt; }
  } else {
    // resulting = previous, covariance = 100 I :1267-1269
    se3::m3_copy(s.prev_R, s.delta_R);
    for (int i = 0; i < 3; ++i) s.delta_t[i] = s.prev_t[i];
    se3::m6_zero(s.delta_cov);
    for (int i = 0; i < 6; ++i) s.delta_cov[i * 7] = 100.0;
  }
  // ---- trackNewFrame :2051-2117
  bool proceed = false;
  if (!s.lost) {
    double tmp[3];
    se3::m3_mulv(s.odoKF_R, s.delta_t, tmp);
    for (int i = 0; i < 3; ++i) s.last_est_t[i] = s.odoKF_t[i] + tmp[i];
    se3::m3_mul(s.odoKF_R, s.delta_R, s.last_est_R);
    if (!ok) {
      s.lost = 1;
      reset_odometry_keyframe(s);
      reset_integration_keyframe(s, c, f, lane);
      f.sw_odo[lane] = 1; f.sw_int[lane] = 1; f.maps[lane] = 1;
      s.status = RGBID_ST_LOST | RGBID_ST_ODO_KF | RGBID_ST_INTEGR_KF;
      ++s.global_time;
    } else proceed = true;
  } else {
    if (ok) {
      s.lost = 0;
      double tmp[3];
      se3::m3_mulv(s.odoKF_R, s.delta_t, tmp);
      for (int i = 0; i < 3; ++i) s.last_est_t[i] = s.odoKF_t[i] + tmp[i];
      se3::m3_mul(s.odoKF_R, s.delta_R, s.last_est_R);
      proceed = true;
    } else {
      f.sw_odo[lane] = 1; f.sw_int[lane] = 1; f.maps[lane] = 1;
      s.status = RGBID_ST_LOST | RGBID_ST_ODO_KF | RGBID_ST_INTEGR_KF;
    }
  }
  se3::m3_copy(s.last_est_R, R.R);
  for (int i = 0; i < 3; ++i) R.t[i] = s.last_est_t[i];
  se3::m3_copy(s.delta_R, R.kf_R);
  for (int i = 0; i < 3; ++i) R.kf_t[i] = s.delta_t[i];
  for (int i = 0; i < 36; ++i) R.kf_cov[i] = s.delta_cov[i];
  if (!proceed) {
    se3::m3_id(R.odo_R);
    for (int i = 0; i < 3; ++i) R.odo_t[i] = 0.0;
    for (int i = 0; i < 36; ++i) R.odo_cov[i] = (i % 7 == 0) ? 100.0 : 0.0;  // dummy constraint :2070-2071
    return;
  }
  s.odoKF_count++; s.integrKF_count++;
  {
    // sequential odometry + covariance :2128-2152
    double pT[9], Rseq[9], d[3], tseq[3], Jn[36], Jl[36], S[9], SR[9];
    se3::m3_T(s.dprev_R, pT);
    se3::m3_mul(pT, s.delta_R, Rseq);
    for (int i = 0; i < 3; ++i) d[i] = s.delta_t[i] - s.dprev_t[i];
    se3::m3_mulv(pT, d, tseq);
    se3::m6_zero(Jn); se3::m6_set_block(Jn, 0, 0, pT, 1.0); se3::m6_set_block(Jn, 3, 3, pT, 1.0);
    se3::m6_zero(Jl); se3::m6_set_block(Jl, 0, 0, pT, -1.0); se3::m6_set_block(Jl, 3, 3, pT, -1.0);
    se3::skew(tseq, S); se3::m3_mul(S, pT, SR); se3::m6_set_block(Jl, 0, 3, SR, 1.0);
    se3::m6_zero(R.odo_cov);
    se3::m6_JCJt_add(Jl, s.dprev_cov, R.odo_cov);
    se3::m6_JCJt_add(Jn, s.delta_cov, R.odo_cov);
    se3::m3_copy(Rseq, R.odo_R);
    for (int i = 0; i < 3; ++i) R.odo_t[i] = tseq[i];
  }
  f.vis[lane] = 1;
  s.status = RGBID_ST_TRACKED;
  // covisibility transforms (computeCovisibility visodo.cpp:1481-1514) for the odometry keyframe ...
  {
    double Ri[9], zero[3] = {0, 0, 0};
    float dummy[3];
    se3::project_trafo(c.fx, c.fy, c.cx, c.cy, s.delta_R, s.delta_t, vis_ab[lane].R, vis_ab[lane].t);
    se3::m3_inv(s.delta_R, Ri);
    se3::project_trafo(c.fx, c.fy, c.cx, c.cy, Ri, zero, vis_ba[lane].R, dummy);
    // translation_BtoA_f = -K*Rinv.cast<float>()*t.cast<float>() in float
    float K[9] = {c.fx, 0.f, c.cx, 0.f, c.fy, c.cy, 0.f, 0.f, 1.f}, Rf[9], T[9];
    float tf[3] = {(float)s.delta_t[0], (float)s.delta_t[1], (float)s.delta_t[2]};
    for (int i = 0; i < 9; ++i) Rf[i] = (float)Ri[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i * 3 + j] = -K[i * 3] * Rf[j] + -K[i * 3 + 1] * Rf[3 + j] + -K[i * 3 + 2] * Rf[6 + j];
    for (int i = 0; i < 3; ++i) vis_ba[lane].t[i] = T[i * 3] * tf[0] + T[i * 3 + 1] * tf[1] + T[i * 3 + 2] * tf[2];
  }
  // ... and for the integration keyframe (:2182-2186)
  {
    double iRi[9], d[3], Ri[9], zero[3] = {0, 0, 0};
    float dummy[3];
    se3::m3_inv(s.integrKF_R, iRi);
    se3::m3_mul(iRi, s.last_est_R, s.dI_R);
    for (int i = 0; i < 3; ++i) d[i] = s.last_est_t[i] - s.integrKF_t[i];
    se3::m3_mulv(iRi, d, s.dI_t);
    se3::project_trafo(c.fx, c.fy, c.cx, c.cy, s.dI_R, s.dI_t, ivis_ab[lane].R, ivis_ab[lane].t);
    se3::m3_inv(s.dI_R, Ri);
    se3::project_trafo(c.fx, c.fy, c.cx, c.cy, Ri, zero, ivis_ba[lane].R, dummy);
    float K[9] = {c.fx, 0.f, c.cx, 0.f, c.fy, c.cy, 0.f, 0.f, 1.f}, Rf[9], T[9];
    float tf[3] = {(float)s.dI_t[0], (float)s.dI_t[1], (float)s.dI_t[2]};
    for (int i = 0; i < 9; ++i) Rf[i] = (float)Ri[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i * 3 + j] = -K[i * 3] * Rf[j] + -K[i * 3 + 1] * Rf[3 + j] + -K[i * 3 + 2] * Rf[6 + j];
    for (int i = 0; i < 3; ++i) ivis_ba[lane].t[i] = T[i * 3] * tf[0] + T[i * 3 + 1] * tf[1] + T[i * 3 + 2] * tf[2];
  }
}

// counts: [4][B][2] = {odo B->A, odo A->B, integr B->A, integr A->B} x {visible, valid}
// The lane's first counter pair is handed back zeroed: computeOverlapping counts into it next (was a memset node).
__global__ __launch_bounds__(64) RGBID_SCALAR_KERNEL void k_decide(LaneState* st, Flags f, unsigned int* counts, WarpParams* fuse_wp, StepCfg c, int B) {
  int lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= B || !f.vis[lane]) return;
  LaneState& s = st[lane];
  float cnt[4][2];
  for (int which = 0; which < 4; ++which) { cnt[which][0] = (float)counts[(which * B + lane) * 2 + 0]; cnt[which][1] = (float)counts[(which * B + lane) * 2 + 1]; }
  counts[lane * 2 + 0] = 0u; counts[lane * 2 + 1] = 0u;
  auto ratio = [&](int which) {
    float vis = cnt[which][0], val = cnt[which][1];
    return (val < 1.f) ? 0.f : vis / val;  // warping_registration.cu:862-865
  };
  // odometry keyframe :2172-2180.  NOTE: evaluated with the pre-switch delta pose, exactly as the reference.
  s.vis_odo = fminf(ratio(1), ratio(0));
  if ((s.odoKF_count >= c.max_odoKF_count) || (s.vis_odo < c.visratio_odo)) {
    reset_odometry_keyframe(s);
    f.sw_odo[lane] = 1;
    s.status |= RGBID_ST_ODO_KF;
  }
  // integration keyframe :2188-2211
  s.vis_int = fminf(ratio(3), ratio(2));
  if ((s.integrKF_count >= c.max_integrKF_count) || (s.vis_int < c.visratio_integr)) {
    reset_integration_keyframe(s, c, f, lane);
    f.sw_int[lane] = 1; f.overlap[lane] = 1; f.maps[lane] = 1;
    s.status |= RGBID_ST_INTEGR_KF;
  } else {
    f.fuse[lane] = 1; f.maps[lane] = 1;
    // integrateImagesIntoKeyframes :1674-1700: K R K^-1 in DOUBLE, inverted, then cast to float
    double K[9] = {(double)c.fx, 0, (double)c.cx, 0, (double)c.fy, (double)c.cy, 0, 0, 1};
    double Ki[9], T[9], Rp[9], tp[3], Rpi[9], tpi[3];
    se3::m3_inv(K, Ki);
    se3::m3_mul(K, s.dI_R, T); se3::m3_mul(T, Ki, Rp);
    se3::m3_mulv(K, s.dI_t, tp);
    se3::m3_inv(Rp, Rpi);
    se3::m3_mulv(Rpi, tp, tpi);
    for (int i = 0; i < 9; ++i) fuse_wp[lane].R[i] = (float)Rpi[i];
    for (int i = 0; i < 3; ++i) fuse_wp[lane].t[i] = (float)(-tpi[i]);
  }
  ++s.global_time;
}

// ---- saveCurrentImagesAsIntegrationKeyframes (visodo.cpp:880-893) + the overlap-mask reset of a first frame (:2021) in ONE launch ---------------
// lanes that switch integration keyframe: current inverse depth -> keyframe inverse depth and its raw copy (one read, two writes), current colours ->
// keyframe colours, keyframe weight = 1; lanes on their first frame: overlap mask = 0.  (Five row-copy / fill launches before, each dispatching its grid
// for the 98 % of lanes that do nothing in a step.)
__device__ __forceinline__ bool rows16(const ImgB& a, int row_bytes) { return (row_bytes & 15) == 0 && (a.pitch & 15) == 0 && (a.lane_stride & 15) == 0 && (((uintptr_t)a.base) & 15) == 0; }
__global__ __launch_bounds__(256) void k_save_integr_kf(ImgB iD_cur, ImgB rgb_cur, ImgB iD_kf, ImgB iD_raw, ImgB colors, ImgB weight, ImgB mask, const int* sw_int, const int* first) {
  const int lane = blockIdx.y;
  const bool sw = sw_int[lane] == 1, fr = first[lane] == 1;
  if (!sw && !fr) return;
  typedef float f4v __attribute__((ext_vector_type(4)));
  const int rows = iD_cur.rows, cols = iD_cur.cols;
  const bool v4 = rows16(iD_cur, 4 * cols) && rows16(iD_kf, 4 * cols) && rows16(iD_raw, 4 * cols) && rows16(weight, 4 * cols);
  const bool v3 = rows16(rgb_cur, 3 * cols) && rows16(colors, 3 * cols);
  const bool v1 = rows16(mask, cols);
  for (int y = blockIdx.x; y < rows; y += gridDim.x) {
    if (sw) {
      const float* s = row_ptr<float>(iD_cur, lane, y);
      float *d0 = row_ptr<float>(iD_kf, lane, y), *d1 = row_ptr<float>(iD_raw, lane, y), *w = row_ptr<float>(weight, lane, y);
      if (v4) {
        for (int i = threadIdx.x; i < cols / 4; i += 256) {
          const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(s) + i);
          __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(d0) + i);
          __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(d1) + i);
          __builtin_nontemporal_store(f4v{1.f, 1.f, 1.f, 1.f}, reinterpret_cast<f4v*>(w) + i);
        }
      } else {
        for (int i = threadIdx.x; i < cols; i += 256) { const float v = s[i]; d0[i] = v; d1[i] = v; w[i] = 1.f; }
      }
      const uint8_t* cs = row_ptr<uint8_t>(rgb_cur, lane, y);
      uint8_t* cd = row_ptr<uint8_t>(colors, lane, y);
      if (v3) {
        for (int i = threadIdx.x; i < 3 * cols / 16; i += 256) reinterpret_cast<uint4*>(cd)[i] = reinterpret_cast<const uint4*>(cs)[i];
      } else {
        for (int i = threadIdx.x; i < 3 * cols; i += 256) cd[i] = cs[i];
      }
    }
    if (fr) {
      uint8_t* mk = row_ptr<uint8_t>(mask, lane, y);
      if (v1) { for (int i = threadIdx.x; i < cols / 16; i += 256) reinterpret_cast<uint4*>(mk)[i] = make_uint4(0, 0, 0, 0); }
      else { for (int i = threadIdx.x; i < cols; i += 256) mk[i] = 0; }
    }
  }
}

// the four downloads of resetIntegrationKeyframe (:1638-1641) as one device-side packed copy into the lane's ring slot:
// section 0 overlap mask, 1 colours, 2 inverse depth, 3 normals (3*rows rows).  blockIdx = (row chunk, section, lane).
struct KfSrc { ImgB im[4]; int row_bytes[4]; size_t off[4]; };
__global__ __launch_bounds__(256) void k_export_keyframe(KfSrc src, char* blocks, size_t block_bytes, int cap, const int* kf_slot) {
  int lane = blockIdx.z, sec = blockIdx.y;
  int slot = kf_slot[lane];
  if (slot < 0) return;
  const ImgB& im = src.im[sec];
  const int rb = src.row_bytes[sec];
  char* dst = blocks + ((size_t)lane * cap + slot) * block_bytes + src.off[sec];
  for (int y = blockIdx.x; y < im.rows; y += gridDim.x) {
    const char* sp = row_ptr<char>(im, lane, y);
    char* dp = dst + (size_t)y * rb;
    if (((rb & 15) == 0) && ((((uintptr_t)sp | (uintptr_t)dp) & 15) == 0)) {
      for (int i = threadIdx.x; i < (rb >> 4); i += blockDim.x) reinterpret_cast<float4*>(dp)[i] = reinterpret_cast<const float4*>(sp)[i];
    } else {
      for (int i = threadIdx.x; i < rb; i += blockDim.x) dp[i] = sp[i];
    }
  }
}

// getImage (visodo.cpp:559-580): the Phong light sits at the integration keyframe's global position
__global__ void k_set_light(const LaneState* st, LightP* light, int B) {
  int lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= B) return;
  light[lane] = LightP{(float)st[lane].integrKF_t[0], (float)st[lane].integrKF_t[1], (float)st[lane].integrKF_t[2]};
}

// pose-record ring -> the 392-byte records of SURVEY 8e, lane-major; one thread per (lane, step)
__global__ void k_pack_gather(const rgbid_pose_record* ring, int capacity, int B, int first_step, int n_steps, rgbid_gather_record* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n_steps) return;
  int lane = i / n_steps, k = i - lane * n_steps;
  const rgbid_pose_record& r = ring[(size_t)((first_step + k) % capacity) * B + lane];
  rgbid_gather_record& g = out[i];
  g.frame_id = r.frame; g.status = r.status;
  for (int j = 0; j < 9; ++j) g.R[j] = r.odo_R[j];
  for (int j = 0; j < 3; ++j) g.t[j] = r.odo_t[j];
  for (int j = 0; j < 36; ++j) g.cov[j] = r.odo_cov[j];
}

__global__ void k_step_end(LaneState* st, rgbid_pose_record* rec, const int* kf_slot, int B) {
  int lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= B) return;
  LaneState& s = st[lane];
  rgbid_pose_record& R = rec[lane];
  if (kf_slot[lane] >= 0) s.status |= RGBID_ST_KF_EXPORTED;
  R.status = s.status;
  R.vis_odo = s.vis_odo; R.vis_integr = s.vis_int;
  if (s.status & RGBID_ST_FIRST) {
    R.frame = 0;
    se3::m3_id(R.R); se3::m3_id(R.odo_R); se3::m3_id(R.kf_R);
    for (int i = 0; i < 3; ++i) R.t[i] = R.odo_t[i] = R.kf_t[i] = 0.0;
    for (int i = 0; i < 36; ++i) R.odo_cov[i] = R.kf_cov[i] = 0.0;
    R.sigma_int = R.sigma_depthinv = R.nu_int = R.nu_depthinv = 0.f;
  }
}

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------
struct rgbid_engine {
  rgbid_ctx* ctx = nullptr;
  rgbid_engine_config cfg{};
  int B = 0, L = 0;
  std::vector<void*> allocs;
  size_t bytes = 0;
  // images
  ImgB in_depth, in_rgb;      // pitched staging copies of the inputs (graph replay needs fixed addresses)
  ImgB cur_depth, cur_rgb;    // what this step reads: the staging copies, or dense views of the caller's buffers (eager mode)
  ImgB iD_curr[MAXL], I_curr[MAXL], iD_kf[MAXL], I_kf[MAXL], iD_kf_f[MAXL], I_kf_f[MAXL];
  ImgB gxI[MAXL], gyI[MAXL], gxD[MAXL], gyD[MAXL], gxI_c[MAXL], gyI_c[MAXL], gxD_c[MAXL], gyD_c[MAXL];
  ImgB wiD[MAXL], wI[MAXL];
  ImgB r_curr, g_curr, b_curr;
  ImgB I_dist, iD_dist, iD_corr, iD_prereg, reg_f, reg_i;   // custom calibration (cfg.custom_registration): distorted maps, corrected map, pre-registration map, 3 x 3 enlarged splat canvases
  ImgB iD_integr, iD_integr_raw, w_integr, warped_iD_integr, warped_w, vmap, nmap, gxD_integr, gyD_integr, colors_integr, overlap_mask, preview;
  float *res_I = nullptr, *res_D = nullptr;
  float* lat_res = nullptr;        // fused path: residual lattice of both channels, [B][2 * lat_cap]
  size_t lat_cap = 0;
  float* lat_kf[MAXL] = {};        // fused path: the odometry keyframe's side of each level's lattice (W0 | I0), [B][2 * lat_cap], packed at keyframe switches
  float* chi_out = nullptr;
  double* partials = nullptr;
  int nblk_cap = 0;
  LaneState* state = nullptr;
  Flags flags{};
  WarpParams *wp = nullptr, *vis_ab = nullptr, *vis_ba = nullptr, *ivis_ab = nullptr, *ivis_ba = nullptr, *fuse_wp = nullptr;
  SysParams* sp = nullptr;
  LightP* light = nullptr;
  unsigned int* counts = nullptr;
  float* delta_t_dev = nullptr;   // StepCfg::delta_t
  rgbid_pose_record* records = nullptr;  // [capacity][B]
  rgbid_pose_record* rec_cur = nullptr;  // [B] staging written by the step, copied into the ring
  int* active_dev = nullptr;                // [B] lanes fed by the next steps (rgbid_engine_set_active)
  // keyframe export ring (cfg.keyframe_capacity > 0)
  rgbid_keyframe_header* kf_hdr = nullptr;  // [B][cap]
  char* kf_blocks = nullptr;                // [B][cap][kf_block_bytes]
  size_t kf_block_bytes = 0;
  char* kf_staging = nullptr;               // pinned host: header + block
  int* kf_counts_host = nullptr;            // pinned host [B]
  int steps = 0;
  int launches = 0;
  // algorithmic HBM bytes of the launch list, per lane (rgbid_engine_step_bytes): [0] every tracked frame, [1] extra per odometry-keyframe
  // switch, [2] extra per integration-keyframe switch, [3] extra per frame fused into the integration keyframe
  double step_bytes[4] = {0, 0, 0, 0};
  hipGraphExec_t graph_first = nullptr, graph_next = nullptr;
  bool graph_ready_first = false, graph_ready_next = false;
  // event timing of the dominant kernel (level-0 normal equations), see rgbid_engine_profile_begin
  std::vector<hipEvent_t> prof_ev;
  int prof_used = 0;
  bool prof_on = false;
  size_t lane_pad = 0, map_skew = 0;   // placement of the image maps (alloc_img)
  int n_maps = 0;
};

namespace {

int alloc_dev(rgbid_engine* e, void** p, size_t bytes, bool zero = true) {
  hipError_t err = hipMalloc(p, bytes);
  if (err != hipSuccess) return err == hipErrorOutOfMemory ? RGBID_E_NOMEM : (int)err;
  e->allocs.push_back(*p);
  e->bytes += bytes;
  if (zero) {
    err = hipMemsetAsync(*p, 0, bytes, e->ctx->stream);
    if (err != hipSuccess) return (int)err;
  }
  return RGBID_OK;
}

// Placement: a map is [lanes][rows][pitch] with the lanes `lane_stride` apart.  With lane_stride = rows * pitch the lanes of a 1280x960 map sit
// 75 x 64 KiB apart and every map of a level starts on the same allocation granule: the same tile of every lane, and of all eight maps the dominant kernel
// streams, then falls on the same HBM channel group -- whether the channels are loaded evenly depends on where the allocator happened to put the maps (round 5:
// 0.47 - 0.66 of the peak for the same binary).  lane_pad (a multiple of 256 B added to every lane) and map_skew (the k-th map of the engine starts k * map_skew
// bytes into its allocation) take both regularities out.  Defaults (rgbid_engine_create): map_skew = 4 KiB + 256 B (+0.5 - 1 % on the 640x480 level-0 kernel, the
// best of the sweep at 1280x960), lane_pad = 0 (no gain measured); RGBID_ENGINE_LANE_PAD / RGBID_ENGINE_MAP_SKEW (bytes) override them for experiments
// (profiles/r06_experiments/placement.md: the 0.47 case itself did not reproduce in 44 runs).
int alloc_img(rgbid_engine* e, ImgB* im, int rows, int cols, int elem) {
  size_t pitch = ((size_t)cols * elem + 255) & ~(size_t)255;
  size_t lane_stride = pitch * rows + e->lane_pad;
  const size_t skew = e->map_skew * (size_t)(e->n_maps++ % 16);
  void* p = nullptr;
  int r = alloc_dev(e, &p, lane_stride * e->B + skew);
  if (r) return r;
  *im = ImgB{static_cast<char*>(p) + skew, pitch, lane_stride, rows, cols};
  return RGBID_OK;
}

inline LaneMask M(const int* flag) { return LaneMask{flag, 1}; }
const LaneMask ALL{nullptr, 0};

StepCfg step_cfg(const rgbid_engine_config& c) {
  StepCfg s;
  s.fx = c.fx; s.fy = c.fy; s.cx = c.cx; s.cy = c.cy;
  s.levels = c.levels; s.finest_level = c.finest_level; s.motion_model = c.motion_model;
  s.max_odoKF_count = c.max_odoKF_count; s.max_integrKF_count = c.max_integrKF_count;
  s.visratio_odo = c.visratio_odo; s.visratio_integr = c.visratio_integr; s.delta_t = nullptr;   // set by enqueue_step (engine's device copy)
  s.mestimator = c.mestimator; s.weighting = c.weighting;
  // first level (coarse to fine) that runs at least one iteration; the covariance pass warps at the finest level
  s.start_warp_level = c.finest_level;
  bool any_gn = false;
  for (int l = c.levels - 1; l >= c.finest_level; --l) if (c.iters[l] > 0) { s.start_warp_level = l; any_gn = true; break; }
  if (c.warping == RGBID_WARP_FIRST && any_gn) s.start_warp_level = 0;  // warp-first warps the level-0 frame in every GN iteration
  return s;
}

// saveCurrentImagesAsOdoKeyframes (visodo.cpp:826-878) for the lanes flagged sw_odo
inline double npx(const ImgB& im) { return (double)im.rows * im.cols; }

void enqueue_save_odo_kf(rgbid_engine* e, hipStream_t s) {
  const int B = e->B, L = e->L;
  LaneMask m = M(e->flags.sw_odo);
  {
    // algorithmic bytes of a keyframe switch (bucket 1): 2 copies per level (4 B read + 4 B written per pixel; folded into the Sobel pass where possible), the lattice pack, two
    // bilateral filters (8 B/px), the pyramid of the filtered maps, Sobel (12 B/px) of the filtered and of the unfiltered maps
    double b = 0.0;
    for (int i = 0; i < L; ++i) {
      const double n = npx(e->iD_kf[i]);
      const bool keep = e->cfg.image_filtering != RGBID_FILTER_GRADS && e->iD_kf[i].cols % 4 == 0;   // Sobel pair + copy in one pass (16 B/px)
      if (!keep) b += 2 * 8 * n;
      if (e->lat_res) b += 16.0 * lattice_samples(e->iD_kf[i].rows, e->iD_kf[i].cols, e->cfg.nsamples);
      b += 2 * 12 * n;                                                            // gradients of the filtered maps
      if (i) b += 2 * (4 * npx(e->iD_kf_f[i - 1]) + 4 * n);                       // pyrDown of the filtered maps
      b += (e->cfg.image_filtering == RGBID_FILTER_GRADS) ? 4 * 8 * n : keep ? 2 * 16 * n : 2 * 12 * n;
    }
    b += 2 * 8 * npx(e->iD_kf[0]);
    e->step_bytes[1] = b;
  }
  // the current frame's pyramids become the keyframe's.  Where the Sobel pair of the unfiltered maps is taken anyway it writes the copy as well (one
  // read of the map instead of two, one launch instead of two per map and level); the gradients of those levels are done then, not at the end
  bool kept[MAXL] = {};
  for (int i = 0; i < L; ++i) {
    if (e->cfg.image_filtering != RGBID_FILTER_GRADS)
      kept[i] = launch_gradient_keep2(s, B, e->I_curr[i], e->gxI[i], e->gyI[i], e->I_kf[i], e->iD_curr[i], e->gxD[i], e->gyD[i], e->iD_kf[i], m);
    if (kept[i]) e->launches += 1;
    else {
      launch_copy_bytes(s, B, e->iD_curr[i], e->iD_kf[i], 4, m);
      launch_copy_bytes(s, B, e->I_curr[i], e->I_kf[i], 4, m);
      e->launches += 2;
    }
  }
  if (e->lat_res) {
    launch_lattice_pack_levels(s, B, L, e->iD_kf, e->I_kf, e->cfg.nsamples, e->lat_kf, 2 * e->lat_cap, m);
    e->launches += (L + 3) / 4;
  }
  launch_bilateral2(s, B, e->iD_kf[0], e->iD_kf_f[0], 2.f * 0.0025f, e->I_kf[0], e->I_kf_f[0], 3.f, m, e->cfg.fast_numerics != 0);
  launch_gradient2(s, B, e->I_kf_f[0], e->gxI_c[0], e->gyI_c[0], e->iD_kf_f[0], e->gxD_c[0], e->gyD_c[0], m);
  e->launches += 2;
  for (int i = 1; i < L; ++i) {
    launch_pyr_down2(s, B, e->iD_kf_f[i - 1], e->iD_kf_f[i], e->I_kf_f[i - 1], e->I_kf_f[i], m);
    launch_gradient2(s, B, e->I_kf_f[i], e->gxI_c[i], e->gyI_c[i], e->iD_kf_f[i], e->gxD_c[i], e->gyD_c[i], m);
    e->launches += 2;
  }
  for (int i = 0; i < L; ++i) {
    if (e->cfg.image_filtering == RGBID_FILTER_GRADS) {
      launch_copy_bytes(s, B, e->gxI_c[i], e->gxI[i], 4, m); launch_copy_bytes(s, B, e->gyI_c[i], e->gyI[i], 4, m);
      launch_copy_bytes(s, B, e->gxD_c[i], e->gxD[i], 4, m); launch_copy_bytes(s, B, e->gyD_c[i], e->gyD[i], 4, m);
      e->launches += 4;
    } else if (!kept[i]) {
      launch_gradient2(s, B, e->I_kf[i], e->gxI[i], e->gyI[i], e->iD_kf[i], e->gxD[i], e->gyD[i], m);
      e->launches += 1;
    }
  }
}

// the whole step as a launch sequence on stream s
int enqueue_step(rgbid_engine* e, hipStream_t s, bool first) {
  const rgbid_engine_config& c = e->cfg;
  const int B = e->B, L = e->L;
  StepCfg sc_ = step_cfg(c);
  sc_.kf_hdr = e->kf_hdr; sc_.kf_cap = c.keyframe_capacity;
  sc_.delta_t = e->delta_t_dev;
  sc_.active = e->active_dev;   // always passed (all ones by default): the captured graphs stay valid when the mask changes
  const StepCfg sc = sc_;
  const IntrP K0{c.fx, c.fy, c.cx, c.cy};
  const int tb = 64, gb = div_up(B, tb);
  // fast numerics per pyramid level, decided ONCE here for every gather kernel of the level (lattice pre-pass, warp pair, fused normal
  // equations; kernels.h gn_fast_supported): levels whose geometry does not allow the 16-byte / paired 8-byte accesses run the exact
  // kernels in BOTH the fused and the unfused path, so the two stay bit-identical to each other
  auto fast_at = [&](int level) {
    return c.fast_numerics != 0 && gn_fast_supported(e->iD_kf[level], e->I_kf[level], e->gxD[level], e->gyD[level], e->gxI[level], e->gyI[level], e->I_curr[level]);
  };
  e->launches = 0;
  const bool defer_maps = c.defer_keyframe_maps != 0 && !c.preview;   // the preview shades the maps every frame
  double* const sb = e->step_bytes;
  sb[0] = sb[2] = sb[3] = 0.0;
  const double N0 = (double)c.rows * c.cols;
  Flags& f = e->flags;
  // ---- prepareImages (visodo.cpp:760-773)
  const LaneMask fed = M(e->active_dev);   // lanes without a frame this step keep their current-frame pyramids untouched
  if (c.custom_registration) {
    // prepareImagesCustomCalibration (visodo.cpp:775-824): the converters write the DISTORTED maps; undistort the intensity (bilinear), correct the depth
    // sensor's distortion and undistort the inverse depth (point sample), register it onto the colour camera (z-buffer splat + homography)
    launch_prep_frame(s, B, e->cur_depth, e->cur_rgb, e->iD_dist, e->I_dist, e->r_curr, e->g_curr, e->b_curr, c.factor_depth, fed);
    const IntrK kc{c.fx, c.fy, c.cx, c.cy, c.rgb_dist[0], c.rgb_dist[1], c.rgb_dist[2], c.rgb_dist[3], c.rgb_dist[4]};
    const IntrK kd{c.depth_intr.fx, c.depth_intr.fy, c.depth_intr.cx, c.depth_intr.cy, c.depth_intr.k1, c.depth_intr.k2, c.depth_intr.k3, c.depth_intr.k4, c.depth_intr.k5};
    DepthDistP dp;
    dp.c1 = c.depth_dist.c1; dp.c0 = c.depth_dist.c0; dp.xshift = c.depth_dist.xshift; dp.yshift = c.depth_dist.yshift;
    for (int i = 0; i < 9; ++i) { dp.q0[i] = c.depth_dist.q0[i]; dp.q1[i] = c.depth_dist.q1[i]; }
    launch_undistort(s, B, e->I_dist, e->I_curr[0], kc, true, c.interp_mode, fed);
    launch_depthinv_correction(s, B, e->iD_dist, e->iD_corr, kd, dp, fed);
    launch_undistort(s, B, e->iD_corr, e->iD_prereg, kd, false, 0, fed);
    launch_register_depthinv(s, B, e->iD_prereg, e->reg_f, e->reg_i, e->iD_curr[0], c.dRc_proj, c.t_dc_proj, c.cRd_proj, fed);
    e->launches += 8;
    sb[0] += (25 + 12 + 8 + 8 + 9 * 4 + 8 + 9 * 8 + 8) * N0;                        // converters, three per-pixel resamplings, canvas clear / splat / conversion (9 N0 texels), homography
  } else {
  launch_prep_frame(s, B, e->cur_depth, e->cur_rgb, e->iD_curr[0], e->I_curr[0], e->r_curr, e->g_curr, e->b_curr, c.factor_depth, fed);
  e->launches += 1;
  sb[0] += 25 * N0;                                                               // 2 + 3 B/px read, five fp32 planes written
  }
  for (int i = 1; i < L; ++i) {
    launch_pyr_down2(s, B, e->I_curr[i - 1], e->I_curr[i], e->iD_curr[i - 1], e->iD_curr[i], fed);
    e->launches += 1;
    sb[0] += 2 * (4 * npx(e->I_curr[i - 1]) + 4 * npx(e->I_curr[i]));
  }
  // the Gauss-Newton stages of the step: the levels that iterate, coarse to fine, then the covariance pass
  int stage_level[MAXL + 1], n_stages = 0;
  for (int level = L - 1; level >= c.finest_level; --level) if (c.iters[level] > 0) stage_level[n_stages++] = level;
  const int n_gn_stages = n_stages;
  stage_level[n_stages++] = c.finest_level;   // covariance pass
  hipLaunchKernelGGL(k_step_begin, dim3(gb), dim3(tb), 0, s, e->state, f, e->wp, e->sp, sc, B, e->counts, stage_level[0], n_gn_stages == 0 ? 1 : 0);
  e->launches++;
  int stage = 0;
  // The first step after reset() is host-known to be every lane's first frame (visodo.cpp:1994-2045): only the
  // keyframe-creation part of the sequence below is enqueued (k_step_begin has set first / sw_odo / sw_int / maps).
  // ---- estimateVisualOdometry (visodo.cpp:1041-1281), PYR_FIRST
  const bool chi_stop = c.termination == RGBID_CHI_SQUARED;
  const LaneMask LV = M(f.lvl);   // the lanes iterating the current level (== f.gn unless chi_stop)
  // few-lane plan: every update but the frame's last rides in the next iteration's lattice pre-pass (k_lattice_after_update); needs that pre-pass (fused path with
  // estimated scales; CHI_SQUARED termination runs unfused)
  const bool prologue_plan = c.fused_gn && c.sigma_estimator == RGBID_SIGMA_PDF && !chi_stop && e->lat_res && B <= update_prologue_max_lanes();
  struct { bool on; int nblk, sys_level, sys_cov; } pend = {false, 0, -1, 0};
  int pose_in_alt = 0;   // which of the lane's two pose buffers holds the working pose (0: cur_*)
  for (int level = L - 1; !first && level >= c.finest_level; --level) {
    int iters = c.iters[level];
    if (iters > 0) ++stage;     // stage_level[stage]: what follows this level
    if (chi_stop && iters > 0) { hipLaunchKernelGGL(k_level_begin, dim3(gb), dim3(tb), 0, s, f, B); e->launches++; }
    for (int it = 0; it < iters; ++it) {
      bool last_of_level = (it == iters - 1);
      // level whose intrinsics project the NEXT warp: same level, the next lower level that iterates, or (after the very last
      // iteration) the finest level for the covariance pass; warp-first always warps at level 0 inside the GN loop
      int next_level = level;
      if (last_of_level) {
        next_level = c.finest_level;
        for (int l = level - 1; l >= c.finest_level; --l) if (c.iters[l] > 0) { next_level = l; break; }
      }
      bool more_gn = !last_of_level;
      for (int l = level - 1; l >= c.finest_level && !more_gn; --l) more_gn = c.iters[l] > 0;
      if (c.warping == RGBID_WARP_FIRST) next_level = more_gn ? 0 : c.finest_level;
      bool prof = e->prof_on && level == 0 && e->prof_used + 2 <= (int)e->prof_ev.size();
      int nblk;
      {
        // one Gauss-Newton iteration: the 8 maps of unit U1 (32 B/px); unfused additionally the warp pair (12 B/px read, 8 written); the
        // residual lattice: 36 B/sample fused (packed keyframe side 8, gathers 20, residuals written 8) + 8 read by the sigma / nu kernel,
        // 16 B/sample read from the stored maps otherwise
        const double nl = npx(e->iD_kf[level]);
        const double ns = c.sigma_estimator == RGBID_SIGMA_PDF ? (double)lattice_samples(e->iD_kf[level].rows, e->iD_kf[level].cols, c.nsamples) : 0.0;
        if (c.fused_gn) sb[0] += 32 * nl + 44 * ns;
        else sb[0] += (c.warping == RGBID_WARP_FIRST ? 20 * N0 : 20 * nl) + 32 * nl + 16 * ns;
      }
      if (c.fused_gn) {
        if (c.sigma_estimator == RGBID_SIGMA_PDF) {
          if (pend.on) {
            // few-lane plan: the previous iteration's update rides in this launch (k_lattice_after_update)
            int n_, lr_, lc_, st_;
            lattice_geometry(e->iD_kf[level].rows, e->iD_kf[level].cols, c.nsamples, &n_, &lr_, &lc_, &st_);
            const size_t kls = 2 * e->lat_cap;
            hipLaunchKernelGGL(k_lattice_after_update, dim3(div_up(n_, 256), B), dim3(256), 0, s, e->iD_curr[level], e->iD_kf[level], e->I_curr[level], e->I_kf[level], c.interp_mode,
                               n_, lc_, st_, e->lat_res, 2 * e->lat_cap, kls >= 2 * (size_t)n_ ? e->lat_kf[level] : nullptr, kls, fast_at(level) ? 1 : 0,
                               e->partials, pend.nblk, e->state, f, e->wp, sc, level, e->sp, pend.sys_level, pend.sys_cov, pose_in_alt);
            launch_sigma_pair_arrays(s, B, e->lat_res, 2 * e->lat_cap, n_, e->sp, c.mestimator, LV);
            pose_in_alt ^= 1; pend.on = false;
          } else {
            launch_sigma_pair_fused(s, B, e->iD_curr[level], e->iD_kf[level], e->I_curr[level], e->I_kf[level], e->wp, c.interp_mode, c.nsamples,
                                    e->sp, c.mestimator, LV, fast_at(level), e->lat_res, 2 * e->lat_cap, e->lat_kf[level], 2 * e->lat_cap);
          }
          e->launches += 2;
        }
        if (prof) { set_system_kernel_events(e->prof_ev[e->prof_used], e->prof_ev[e->prof_used + 1]); e->prof_used += 2; }
        nblk = launch_gn_fused(s, B, e->iD_kf[level], e->I_kf[level], e->gxD[level], e->gyD[level], e->gxI[level], e->gyI[level],
                               e->iD_curr[level], e->I_curr[level], e->wp, c.interp_mode, e->sp, e->partials, LV, level < 2 ? level : 2, fast_at(level),
                               c.weighting == RGBID_MIN_WEIGHT ? 0 : 1);   // k_set_sys: the Gauss-Newton iterations always estimate nu
        if (nblk < 0) return RGBID_E_INVALID;
      } else {
        if (c.warping == RGBID_WARP_FIRST) {
          // :1078-1105: warp the full-resolution frame, then reduce the WARPED maps down to the working level
          // (k_step_begin / k_solve_update project the pose with the level-0 intrinsics in this mode)
          if (!(fast_at(0) && launch_warp_pair_fast(s, B, e->iD_curr[0], e->I_curr[0], e->iD_kf[0], e->wiD[0], e->wI[0], nullptr, e->wp, c.interp_mode, LV)))
            launch_warp_pair(s, B, e->iD_curr[0], e->I_curr[0], e->iD_kf[0], e->wiD[0], e->wI[0], e->wp, c.interp_mode, LV);
          e->launches += 1;
          for (int i = 1; i <= level; ++i) {
            launch_pyr_down(s, B, e->wI[i - 1], e->wI[i], LV);
            launch_pyr_down(s, B, e->wiD[i - 1], e->wiD[i], LV);
            e->launches += 2;
          }
        } else {
          if (!(fast_at(level) && launch_warp_pair_fast(s, B, e->iD_curr[level], e->I_curr[level], e->iD_kf[level], e->wiD[level], e->wI[level], nullptr, e->wp, c.interp_mode, LV)))
            launch_warp_pair(s, B, e->iD_curr[level], e->I_curr[level], e->iD_kf[level], e->wiD[level], e->wI[level], e->wp, c.interp_mode, LV);
          e->launches += 1;
        }
        if (chi_stop && it != 0) {
          // :1134-1164 -- the full-lattice residuals of the LEVEL-0 warped maps (fresh with WARP_FIRST; with PYR_FIRST whatever the last level-0 warp left
          // there, as in the reference), their chi-square, and the per-lane decision
          int n, lr, lc, st_;
          lattice_geometry(e->wI[0].rows, e->wI[0].cols, 9999999, &n, &lr, &lc, &st_);
          launch_error_lattice(s, B, e->wI[0], e->I_kf[0], e->res_I, (size_t)c.rows * c.cols, lr, lc, st_, LV);
          launch_error_lattice(s, B, e->wiD[0], e->iD_kf[0], e->res_D, (size_t)c.rows * c.cols, lr, lc, st_, LV);
          launch_chi_square(s, B, e->res_I, e->res_D, (size_t)c.rows * c.cols, n, 5.f, 0.0025f, c.mestimator, e->chi_out, LV);
          int after_level = c.finest_level; bool more_below = false;
          for (int l = level - 1; l >= c.finest_level; --l) if (c.iters[l] > 0) { after_level = l; more_below = true; break; }
          if (c.warping == RGBID_WARP_FIRST) after_level = more_below ? 0 : c.finest_level;
          hipLaunchKernelGGL(k_chi_decide, dim3(gb), dim3(tb), 0, s, e->state, f, e->chi_out, e->wp, sc, it, after_level, B);
          e->launches += 4;
          sb[0] += 2 * 12 * N0 + 8 * N0;
        }
        if (c.sigma_estimator == RGBID_SIGMA_PDF) {
          launch_sigma_pair(s, B, e->wiD[level], e->iD_kf[level], e->wI[level], e->I_kf[level], c.nsamples, e->sp, c.mestimator, LV);
          e->launches++;
        }
        if (prof) { set_system_kernel_events(e->prof_ev[e->prof_used], e->prof_ev[e->prof_used + 1]); e->prof_used += 2; }
        nblk = launch_build_system(s, B, e->iD_kf[level], e->I_kf[level], e->gxD[level], e->gyD[level], e->gxI[level], e->gyI[level],
                                   e->wiD[level], e->wI[level], nullptr, e->sp, e->partials, LV, level < 2 ? level : 2);
      }
      const int sys_level = last_of_level ? stage_level[stage] : -1, sys_cov = (last_of_level && stage == n_gn_stages) ? 1 : 0;
      if (prologue_plan && more_gn) {
        // the update is deferred into the next iteration's lattice launch (same lanes: LV == f.gn here; next_level is that launch's level)
        pend.on = true; pend.nblk = nblk; pend.sys_level = sys_level; pend.sys_cov = sys_cov;
        e->launches += 1;
        continue;
      }
      if (scalar_block_threads(B) == 64)
        hipLaunchKernelGGL(k_solve_update<64>, dim3(B), dim3(64), 0, s, e->partials, nblk, e->state, f, e->wp, sc, next_level, e->sp, sys_level, sys_cov, pose_in_alt);
      else
        hipLaunchKernelGGL(k_solve_update<256>, dim3(B), dim3(256), 0, s, e->partials, nblk, e->state, f, e->wp, sc, next_level, e->sp, sys_level, sys_cov, pose_in_alt);
      pose_in_alt = 0;
      e->launches += 2;
    }
  }
  // ---- covariance pass (visodo.cpp:1283-1409)
  if (!first) {
    int fl = c.finest_level;
    bool prof = e->prof_on && fl == 0 && e->prof_used + 2 <= (int)e->prof_ev.size();
    bool fuse_cov = c.fused_gn && !c.chi_square_stats;  // the chi-square statistics need W1 / I1 in memory
    sb[0] += (fuse_cov ? 32 : 52) * npx(e->iD_kf[fl]);
    int nblk;
    if (fuse_cov) {
      if (prof) { set_system_kernel_events(e->prof_ev[e->prof_used], e->prof_ev[e->prof_used + 1]); e->prof_used += 2; }
      nblk = launch_gn_fused(s, B, e->iD_kf[fl], e->I_kf[fl], e->gxD_c[fl], e->gyD_c[fl], e->gxI_c[fl], e->gyI_c[fl],
                             e->iD_curr[fl], e->I_curr[fl], e->wp, c.interp_mode, e->sp, e->partials, M(f.gn), fl < 2 ? fl : 2, fast_at(fl),
                             c.weighting == RGBID_MIN_WEIGHT ? 0 : 2);   // k_set_sys: the covariance pass is fixed-nu STUDENT
      if (nblk < 0) return RGBID_E_INVALID;
      e->launches += 1;
    } else {
      if (!(fast_at(fl) && launch_warp_pair_fast(s, B, e->iD_curr[fl], e->I_curr[fl], e->iD_kf[fl], e->wiD[fl], e->wI[fl], nullptr, e->wp, c.interp_mode, M(f.gn))))
        launch_warp_pair(s, B, e->iD_curr[fl], e->I_curr[fl], e->iD_kf[fl], e->wiD[fl], e->wI[fl], e->wp, c.interp_mode, M(f.gn));
      if (prof) { set_system_kernel_events(e->prof_ev[e->prof_used], e->prof_ev[e->prof_used + 1]); e->prof_used += 2; }
      nblk = launch_build_system(s, B, e->iD_kf[fl], e->I_kf[fl], e->gxD_c[fl], e->gyD_c[fl], e->gxI_c[fl], e->gyI_c[fl],
                                 e->wiD[fl], e->wI[fl], nullptr, e->sp, e->partials, M(f.gn), fl < 2 ? fl : 2);
      e->launches += 2;
    }
    if (c.chi_square_stats) {  // :1411-1415 (results unused by the reference)
      int n, lr, lc, st;
      lattice_geometry(e->wI[fl].rows, e->wI[fl].cols, 9999999, &n, &lr, &lc, &st);
      launch_error_lattice(s, B, e->wI[fl], e->I_kf[fl], e->res_I, (size_t)c.rows * c.cols, lr, lc, st, M(f.gn));
      launch_error_lattice(s, B, e->wiD[fl], e->iD_kf[fl], e->res_D, (size_t)c.rows * c.cols, lr, lc, st, M(f.gn));
      launch_chi_square(s, B, e->res_I, e->res_D, (size_t)c.rows * c.cols, n, 5.f, 0.0025f, c.mestimator, e->chi_out, M(f.gn));
      e->launches += 3;
    }
    if (scalar_block_threads(B) == 64)
      hipLaunchKernelGGL(k_frame_finish<64>, dim3(B), dim3(64), 0, s, e->partials, nblk, e->state, f, e->sp, e->vis_ab, e->vis_ba,
                         e->ivis_ab, e->ivis_ba, e->rec_cur, sc);
    else
      hipLaunchKernelGGL(k_frame_finish<256>, dim3(B), dim3(256), 0, s, e->partials, nblk, e->state, f, e->sp, e->vis_ab, e->vis_ba,
                         e->ivis_ab, e->ivis_ba, e->rec_cur, sc);
    e->launches++;
  }
  // ---- covisibility with both keyframes (visodo.cpp:2172-2188), 4 ratio evaluations
  if (!first) {
    launch_visibility_pair2(s, B, e->iD_curr[0], e->iD_kf[0], e->vis_ab, e->vis_ba, e->counts + 0 * 2 * B, e->counts + 1 * 2 * B,
                            e->iD_integr_raw, e->ivis_ab, e->ivis_ba, e->counts + 2 * 2 * B, e->counts + 3 * 2 * B, M(f.vis), c.fast_numerics != 0);
    hipLaunchKernelGGL(k_decide, dim3(gb), dim3(tb), 0, s, e->state, f, e->counts, e->fuse_wp, sc, B);
    e->launches += 2;
    sb[0] += 2 * 16 * N0;                                                         // two covisibility pairs: both maps read + both gathered
  }
  // ---- odometry keyframe switch
  enqueue_save_odo_kf(e, s);
  if (!first && e->kf_hdr) {
    // the outgoing keyframe leaves for the back-end (:1631-1652) BEFORE computeOverlapping rewrites the mask and the incoming frame
    // overwrites the maps (:2197-2202): the exported mask is the keyframe's overlap with its predecessor
    const size_t N = (size_t)c.rows * c.cols;
    if (defer_maps) {   // the exported normals: from the fused map as the previous step left it -- what the per-frame schedule computed at the end of that step
      if (!launch_kf_maps(s, B, e->iD_integr, e->vmap, e->nmap, K0, M(f.sw_int))) {
        launch_vmap(s, B, e->iD_integr, e->vmap, K0, M(f.sw_int));
        launch_gradient(s, B, e->iD_integr, e->gxD_integr, e->gyD_integr, M(f.sw_int));
        launch_nmap_gradients(s, B, e->iD_integr, e->gxD_integr, e->gyD_integr, e->nmap, K0, M(f.sw_int));
        e->launches += 2;
      }
      e->launches++;
      sb[2] += 28 * N0;
    }
    KfSrc ks;
    ks.im[0] = e->overlap_mask; ks.im[1] = e->colors_integr; ks.im[2] = e->iD_integr; ks.im[3] = e->nmap;
    ks.row_bytes[0] = c.cols; ks.row_bytes[1] = 3 * c.cols; ks.row_bytes[2] = 4 * c.cols; ks.row_bytes[3] = 4 * c.cols;
    ks.off[0] = 0; ks.off[1] = N; ks.off[2] = 4 * N; ks.off[3] = 8 * N;
    hipLaunchKernelGGL(k_export_keyframe, dim3(min(c.rows, 8), 4, B), dim3(256), 0, s, ks, e->kf_blocks, e->kf_block_bytes, c.keyframe_capacity,
                       f.kf_slot);
    e->launches++;
    sb[2] += 40 * N0;                                                             // 20 B/px of keyframe images read and written
  }
  // ---- integration keyframe: computeOverlapping (:1517-1539) + saveCurrentImagesAsIntegrationKeyframes (:880-893) ...
  if (!first) {
    launch_visibility(s, B, e->iD_curr[0], e->iD_integr_raw, e->overlap_mask, nullptr, e->ivis_ab, e->counts, M(f.overlap));
    e->launches++;
  }
  hipLaunchKernelGGL(k_save_integr_kf, dim3(std::min(c.rows, 32), B), dim3(256), 0, s, e->iD_curr[0], e->cur_rgb, e->iD_integr, e->iD_integr_raw, e->colors_integr, e->w_integr,
                     e->overlap_mask, f.sw_int, f.first);   // three copies + weight fill; initialiseDeviceMemory2D(overlap_mask, 0) :2021
  e->launches += 1;
  sb[2] += (9 + 12 + 6 + 4) * N0;                                                 // overlap mask pass; inverse depth read once and written twice, colours, weight fill
  // ... or integrateImagesIntoKeyframes (:1674-1764)
  if (!first) {
    if (launch_fuse_frame(s, B, e->iD_curr[0], e->iD_integr, e->w_integr, e->warped_w, e->fuse_wp, M(f.fuse), c.fast_numerics != 0)) {
      e->launches -= 1;
      sb[3] += (c.fast_numerics ? 20 : 24) * N0;                                  // keyframe map + weight read and written, gather (+ the warped-weight buffer in the exact class)
    } else {
      sb[3] += 40 * N0;
      launch_warp_invdepth_weighted(s, B, e->iD_curr[0], e->iD_integr, e->warped_iD_integr, e->warped_w, nullptr, e->fuse_wp, M(f.fuse));
      launch_integrate_warped(s, B, e->warped_iD_integr, e->warped_w, e->iD_integr, e->w_integr, M(f.fuse));
    }
  }
  if (defer_maps) {
    e->launches += 2;                                                             // (the two launches counted with the maps: fusion, integration keyframe)
  } else if (launch_kf_maps(s, B, e->iD_integr, e->vmap, e->nmap, K0, M(f.maps))) {
    e->launches += 3;
    sb[0] += 28 * N0;
  } else {
    sb[0] += (16 + 12 + 24) * N0;
    launch_vmap(s, B, e->iD_integr, e->vmap, K0, M(f.maps));
    launch_gradient(s, B, e->iD_integr, e->gxD_integr, e->gyD_integr, M(f.maps));
    launch_nmap_gradients(s, B, e->iD_integr, e->gxD_integr, e->gyD_integr, e->nmap, K0, M(f.maps));
    e->launches += 5;
  }
  if (c.preview) {  // getImage :559-580
    hipLaunchKernelGGL(k_set_light, dim3(gb), dim3(tb), 0, s, e->state, e->light, B);
    e->launches++;
    launch_generate_image(s, B, e->vmap, e->nmap, e->colors_integr, e->preview, nullptr, e->light, ALL);
    e->launches++;
  }
  hipLaunchKernelGGL(k_step_end, dim3(gb), dim3(tb), 0, s, e->state, e->rec_cur, f.kf_slot, B);
  e->launches++;
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? RGBID_OK : (int)err;
}

}  // namespace

extern "C" {

void rgbid_engine_default_config(rgbid_engine_config* c) {
  if (!c) return;
  memset(c, 0, sizeof(*c));
  c->rows = 480; c->cols = 640; c->levels = 3; c->lanes = 1;
  c->iters[0] = 10; c->iters[1] = 5; c->iters[2] = 3;                     // visodo.cpp:65
  c->mestimator = RGBID_STUDENT; c->motion_model = RGBID_CONSTANT_VELOCITY; c->sigma_estimator = RGBID_SIGMA_PDF;
  c->weighting = RGBID_INDEPENDENT;
  c->max_odoKF_count = 9999999; c->finest_level = 0; c->image_filtering = RGBID_NO_FILTERS;
  c->visratio_odo = 0.9f; c->visratio_integr = 0.7f; c->max_integrKF_count = 9999999; c->nsamples = 10000;
  c->fx = 525.f; c->fy = 525.f; c->cx = 319.5f; c->cy = 239.5f; c->factor_depth = 1.f;   // calibration_factory.ini
  c->interp_mode = RGBID_INTERP_TEX8;
  c->delta_t = 0.03333f;
  c->use_graph = 1; c->fused_gn = 1; c->chi_square_stats = 0; c->preview = 0;
  c->record_capacity = 64;
  c->warping = RGBID_PYR_FIRST;
  c->keyframe_capacity = 0;
  c->fast_numerics = 1;
  c->termination = RGBID_ALL_ITERS;
  c->custom_registration = 0;
}

// K_d dRc K_c^-1, K_d t_dc and the inverse of the first, in float with Eigen's cofactor inverse: the constants prepareImagesCustomCalibration forms per frame
// (src/visodo.cpp:792-801), from the config's two intrinsics and the depth -> colour extrinsics
int rgbid_engine_config_set_stereo(rgbid_engine_config* c, const float dRc[9], const float t_dc[3]) {
  if (!c || !dRc || !t_dc) return RGBID_E_INVALID;
  auto mul = [](const float* A, const float* B, float* C) {
    float T[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    for (int i = 0; i < 9; ++i) C[i] = T[i];
  };
  auto inv = [](const float* A, float* I) {
    float c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    float det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    float id = 1.f / det;
    float T[9] = {c00 * id, (A[2] * A[7] - A[1] * A[8]) * id, (A[1] * A[5] - A[2] * A[4]) * id,
                  c01 * id, (A[0] * A[8] - A[2] * A[6]) * id, (A[2] * A[3] - A[0] * A[5]) * id,
                  c02 * id, (A[1] * A[6] - A[0] * A[7]) * id, (A[0] * A[4] - A[1] * A[3]) * id};
    for (int i = 0; i < 9; ++i) I[i] = T[i];
  };
  const float Kc[9] = {c->fx, 0.f, c->cx, 0.f, c->fy, c->cy, 0.f, 0.f, 1.f};
  const float Kd[9] = {c->depth_intr.fx, 0.f, c->depth_intr.cx, 0.f, c->depth_intr.fy, c->depth_intr.cy, 0.f, 0.f, 1.f};
  float Kci[9], T[9];
  inv(Kc, Kci);
  mul(Kd, dRc, T); mul(T, Kci, c->dRc_proj);
  for (int i = 0; i < 3; ++i) c->t_dc_proj[i] = Kd[i * 3] * t_dc[0] + Kd[i * 3 + 1] * t_dc[1] + Kd[i * 3 + 2] * t_dc[2];
  inv(c->dRc_proj, c->cRd_proj);
  return RGBID_OK;
}
size_t rgbid_engine_config_size(void) { return sizeof(rgbid_engine_config); }

int rgbid_engine_create(rgbid_engine** out, rgbid_ctx* ctx, const rgbid_engine_config* cfg) {
  if (!out || !ctx || !cfg) return RGBID_E_INVALID;
  *out = nullptr;
  if (cfg->rows <= 0 || cfg->cols <= 0 || cfg->levels < 1 || cfg->levels > MAXL || cfg->lanes < 1 || cfg->finest_level < 0 ||
      cfg->finest_level >= cfg->levels || (cfg->rows >> (cfg->levels - 1)) < 4 || (cfg->cols >> (cfg->levels - 1)) < 4 ||
      cfg->record_capacity < 1 || (cfg->warping != RGBID_PYR_FIRST && cfg->warping != RGBID_WARP_FIRST) ||
      cfg->keyframe_capacity < 0 || (cfg->termination != RGBID_ALL_ITERS && cfg->termination != RGBID_CHI_SQUARED) || (cfg->custom_registration != 0 && cfg->custom_registration != 1) ||
      !(cfg->delta_t > 0.f) || !(cfg->delta_t < INFINITY) ||
      cfg->cols > (1 << 20) || (unsigned long long)cfg->rows * 3ull * (((unsigned long long)cfg->cols * 4 + 255) & ~255ull) >= (1ull << 32) ||  // 24-bit row-offset arithmetic (common.h row_ptr)
      (cfg->custom_registration && (unsigned long long)cfg->rows * 3ull * (((unsigned long long)cfg->cols * 12 + 255) & ~255ull) >= (1ull << 32)))   // ... of the 3 x 3 registration canvases
    return RGBID_E_INVALID;
  rgbid_engine* e = new (std::nothrow) rgbid_engine();
  if (!e) return RGBID_E_NOMEM;
  e->ctx = ctx; e->cfg = *cfg; e->B = cfg->lanes; e->L = cfg->levels;
  if (cfg->warping == RGBID_WARP_FIRST) e->cfg.fused_gn = 0;   // warp-first pyramids the WARPED maps: they must exist in memory
  if (cfg->termination == RGBID_CHI_SQUARED) e->cfg.fused_gn = 0;   // the chi-square test reads the stored warped maps
  hipSetDevice(ctx->device);
  const int B = e->B, rows = cfg->rows, cols = cfg->cols;
  e->lane_pad = 0; e->map_skew = 0x1100;   // 4 KiB + 256 B per map (alloc_img; measured: profiles/r06_experiments/placement.md)
  if (const char* v = getenv("RGBID_ENGINE_LANE_PAD")) e->lane_pad = (size_t)strtoull(v, nullptr, 0) & ~(size_t)255;
  if (const char* v = getenv("RGBID_ENGINE_MAP_SKEW")) e->map_skew = (size_t)strtoull(v, nullptr, 0) & ~(size_t)255;
  int r = RGBID_OK;
#define A_IMG(im, rr, cc, el) if (!r) r = alloc_img(e, &(im), rr, cc, el)
  A_IMG(e->in_depth, rows, cols, 2); A_IMG(e->in_rgb, rows, cols, 3);
  for (int l = 0; l < e->L; ++l) {
    int pr = rows >> l, pc = cols >> l;
    ImgB* per_level[] = {&e->iD_curr[l], &e->I_curr[l], &e->iD_kf[l], &e->I_kf[l], &e->iD_kf_f[l], &e->I_kf_f[l], &e->gxI[l], &e->gyI[l],
                         &e->gxD[l], &e->gyD[l], &e->gxI_c[l], &e->gyI_c[l], &e->gxD_c[l], &e->gyD_c[l], &e->wiD[l], &e->wI[l]};
    for (ImgB* im : per_level) A_IMG(*im, pr, pc, 4);
  }
  A_IMG(e->r_curr, rows, cols, 4); A_IMG(e->g_curr, rows, cols, 4); A_IMG(e->b_curr, rows, cols, 4);
  A_IMG(e->iD_integr, rows, cols, 4); A_IMG(e->iD_integr_raw, rows, cols, 4); A_IMG(e->w_integr, rows, cols, 4);
  A_IMG(e->warped_iD_integr, rows, cols, 4); A_IMG(e->warped_w, rows, cols, 4);
  A_IMG(e->vmap, 3 * rows, cols, 4); A_IMG(e->nmap, 3 * rows, cols, 4);
  A_IMG(e->gxD_integr, rows, cols, 4); A_IMG(e->gyD_integr, rows, cols, 4);
  A_IMG(e->colors_integr, rows, cols, 3); A_IMG(e->overlap_mask, rows, cols, 1);
  if (cfg->preview) A_IMG(e->preview, rows, cols, 3);
  if (cfg->custom_registration) {
    A_IMG(e->I_dist, rows, cols, 4); A_IMG(e->iD_dist, rows, cols, 4); A_IMG(e->iD_corr, rows, cols, 4); A_IMG(e->iD_prereg, rows, cols, 4);
    A_IMG(e->reg_f, 3 * rows, 3 * cols, 4); A_IMG(e->reg_i, 3 * rows, 3 * cols, 4);   // depthinv_register_trans_ / _as_int_ (visodo.cpp:623-624)
  }
#undef A_IMG
  if (cfg->chi_square_stats || cfg->termination == RGBID_CHI_SQUARED) {
    if (!r) r = alloc_dev(e, (void**)&e->res_I, sizeof(float) * (size_t)rows * cols * B);
    if (!r) r = alloc_dev(e, (void**)&e->res_D, sizeof(float) * (size_t)rows * cols * B);
    if (!r) r = alloc_dev(e, (void**)&e->chi_out, sizeof(float) * 3 * B);
  }
  if (e->cfg.fused_gn && cfg->sigma_estimator == RGBID_SIGMA_PDF) {
    for (int l = 0; l < e->L; ++l) { size_t n = (size_t)lattice_samples(rows >> l, cols >> l, cfg->nsamples); if (n > e->lat_cap) e->lat_cap = n; }
    if (!r) r = alloc_dev(e, (void**)&e->lat_res, sizeof(float) * 2 * e->lat_cap * B);
    for (int l = 0; l < e->L; ++l) if (!r) r = alloc_dev(e, (void**)&e->lat_kf[l], sizeof(float) * 2 * e->lat_cap * B);
  }
  e->nblk_cap = system_blocks_per_lane(rows, cols, B);
  for (int l = 1; l < e->L; ++l) { int nb = system_blocks_per_lane(rows >> l, cols >> l, B); if (nb > e->nblk_cap) e->nblk_cap = nb; }
  if (!r) r = alloc_dev(e, (void**)&e->partials, sizeof(double) * SYS_TERMS * (size_t)e->nblk_cap * B);
  if (!r) r = alloc_dev(e, (void**)&e->state, sizeof(LaneState) * B);
  int** fl[] = {&e->flags.track, &e->flags.first, &e->flags.gn, &e->flags.vis, &e->flags.sw_odo, &e->flags.sw_int, &e->flags.overlap, &e->flags.fuse, &e->flags.maps};
  for (int** p : fl) if (!r) r = alloc_dev(e, (void**)p, sizeof(int) * B);
  if (cfg->termination == RGBID_CHI_SQUARED) { if (!r) r = alloc_dev(e, (void**)&e->flags.lvl, sizeof(int) * B); }
  else e->flags.lvl = e->flags.gn;   // one array: a level ends for every lane together
  WarpParams** wps[] = {&e->wp, &e->vis_ab, &e->vis_ba, &e->ivis_ab, &e->ivis_ba, &e->fuse_wp};
  for (WarpParams** p : wps) if (!r) r = alloc_dev(e, (void**)p, sizeof(WarpParams) * B);
  if (!r) r = alloc_dev(e, (void**)&e->sp, sizeof(SysParams) * B);
  if (!r) r = alloc_dev(e, (void**)&e->light, sizeof(LightP) * B);
  if (!r) r = alloc_dev(e, (void**)&e->counts, sizeof(unsigned int) * 8 * B);
  if (!r) r = alloc_dev(e, (void**)&e->records, sizeof(rgbid_pose_record) * (size_t)cfg->record_capacity * B);
  if (!r) r = alloc_dev(e, (void**)&e->rec_cur, sizeof(rgbid_pose_record) * B);
  if (!r) r = alloc_dev(e, (void**)&e->flags.kf_slot, sizeof(int) * B);
  if (!r) r = alloc_dev(e, (void**)&e->active_dev, sizeof(int) * B);
  if (!r) r = alloc_dev(e, (void**)&e->delta_t_dev, sizeof(float));
  if (!r && hipMemsetD32Async((hipDeviceptr_t)e->delta_t_dev, __builtin_bit_cast(int, cfg->delta_t), 1, ctx->stream) != hipSuccess) r = RGBID_E_NOMEM;
  if (!r) { std::vector<int> ones(B, 1); if (hipMemcpyAsync(e->active_dev, ones.data(), sizeof(int) * B, hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) r = RGBID_E_NOMEM; }
  if (!r && cfg->keyframe_capacity > 0) {
    e->kf_block_bytes = 20 * (size_t)rows * cols;   // u8 mask + 3 u8 colours + f32 inverse depth + 3 f32 normals per pixel
    r = alloc_dev(e, (void**)&e->kf_hdr, sizeof(rgbid_keyframe_header) * (size_t)cfg->keyframe_capacity * B);
    if (!r) r = alloc_dev(e, (void**)&e->kf_blocks, e->kf_block_bytes * cfg->keyframe_capacity * B, /*zero=*/false);
    if (!r && hipHostMalloc((void**)&e->kf_staging, sizeof(rgbid_keyframe_header) + e->kf_block_bytes, hipHostMallocDefault) != hipSuccess) r = RGBID_E_NOMEM;
    if (!r && hipHostMalloc((void**)&e->kf_counts_host, sizeof(int) * B, hipHostMallocDefault) != hipSuccess) r = RGBID_E_NOMEM;
  }
  if (!r) { hipError_t he = hipStreamSynchronize(ctx->stream); if (he != hipSuccess) r = (int)he; }
  if (r) { rgbid_engine_destroy(e); return r; }
  if (getenv("RGBID_ENGINE_DEBUG_ALLOC")) {   // diagnostics: where the level-0 maps of the dominant kernel landed (placement sensitivity of the 1280x960 case)
    const ImgB* m[] = {&e->iD_kf[0], &e->I_kf[0], &e->gxD[0], &e->gyD[0], &e->gxI[0], &e->gyI[0], &e->iD_curr[0], &e->I_curr[0]};
    fprintf(stderr, "rgbid_engine %dx%d x %d lanes: level-0 maps at", cfg->cols, cfg->rows, B);
    for (const ImgB* im : m) fprintf(stderr, " %p", im->base);
    fprintf(stderr, " (lane stride %zu)\n", e->iD_kf[0].lane_stride);
  }
  *out = e;
  return RGBID_OK;
}

int rgbid_engine_destroy(rgbid_engine* e) {
  if (!e) return RGBID_OK;
  hipSetDevice(e->ctx->device);
  hipStreamSynchronize(e->ctx->stream);
  if (e->graph_first) hipGraphExecDestroy(e->graph_first);
  if (e->graph_next) hipGraphExecDestroy(e->graph_next);
  for (void* p : e->allocs) hipFree(p);
  if (e->kf_staging) hipHostFree(e->kf_staging);
  if (e->kf_counts_host) hipHostFree(e->kf_counts_host);
  for (hipEvent_t ev : e->prof_ev) hipEventDestroy(ev);
  delete e;
  return RGBID_OK;
}

int rgbid_engine_reset(rgbid_engine* e) {
  if (!e) return RGBID_E_INVALID;
  hipSetDevice(e->ctx->device);
  hipError_t he = hipMemsetAsync(e->state, 0, sizeof(LaneState) * e->B, e->ctx->stream);  // global_time = 0 -> first-frame path
  if (he != hipSuccess) return (int)he;
  // warped_weight_curr_ is never initialised by the reference (uninitialised device memory); the engine defines it as 0
  he = hipMemsetAsync(e->warped_w.base, 0, e->warped_w.lane_stride * e->B, e->ctx->stream);
  if (he != hipSuccess) return (int)he;
  e->steps = 0;
  return RGBID_OK;
}

int rgbid_engine_set_active(rgbid_engine* e, const int* active_host) {
  if (!e) return RGBID_E_INVALID;
  hipSetDevice(e->ctx->device);
  std::vector<int> m(e->B, 1);
  if (active_host) for (int i = 0; i < e->B; ++i) m[i] = active_host[i] ? 1 : 0;
  hipError_t he = hipMemcpyAsync(e->active_dev, m.data(), sizeof(int) * e->B, hipMemcpyHostToDevice, e->ctx->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(e->ctx->stream);   // m is a temporary
  return he == hipSuccess ? RGBID_OK : (int)he;
}

int rgbid_engine_reset_lane(rgbid_engine* e, int lane) {
  if (!e || lane < 0 || lane >= e->B) return RGBID_E_INVALID;
  if (e->steps == 0) return RGBID_OK;   // nothing tracked yet: every lane starts fresh anyway
  hipSetDevice(e->ctx->device);
  // the lane's next frame takes the first-frame path of k_step_begin (global_time == 0) inside the ordinary step: its Gauss-Newton, covisibility
  // and fusion kernels are predicated off by the lane flags and the keyframe-creation kernels on, exactly as in the first step after reset()
  hipError_t he = hipMemsetAsync(&e->state[lane], 0, sizeof(LaneState), e->ctx->stream);
  if (he != hipSuccess) return (int)he;
  he = hipMemsetAsync((char*)e->warped_w.base + (size_t)lane * e->warped_w.lane_stride, 0, e->warped_w.lane_stride, e->ctx->stream);
  if (he != hipSuccess) return (int)he;
  // the lane's staged record goes with its state: while the lane sits steps out before its new first frame, its records read all-zero
  he = hipMemsetAsync(&e->rec_cur[lane], 0, sizeof(rgbid_pose_record), e->ctx->stream);
  return he == hipSuccess ? RGBID_OK : (int)he;
}

int rgbid_engine_step(rgbid_engine* e, const void* depth_dev, const void* rgb_dev) {
  if (!e) return RGBID_E_INVALID;
  const rgbid_engine_config& c = e->cfg;
  return rgbid_engine_step_strided(e, depth_dev, (size_t)c.cols * 2, (size_t)c.rows * c.cols * 2, rgb_dev, (size_t)c.cols * 3, (size_t)c.rows * c.cols * 3);
}

int rgbid_engine_set_delta_t(rgbid_engine* e, float delta_t) {
  if (!e || !(delta_t > 0.f) || !(delta_t < INFINITY)) return RGBID_E_INVALID;   // 1 / delta_t scales the velocity of the constant-velocity prediction: 0, negative, inf or NaN would lose the lane
  hipSetDevice(e->ctx->device);
  // the kernels read the value through a device pointer (StepCfg::delta_t): captured graphs stay valid; ordered on the context's stream
  if (hipError_t he = hipMemsetD32Async((hipDeviceptr_t)e->delta_t_dev, __builtin_bit_cast(int, delta_t), 1, e->ctx->stream); he != hipSuccess) return (int)he;
  e->cfg.delta_t = delta_t;
  return RGBID_OK;
}

int rgbid_engine_step_strided(rgbid_engine* e, const void* depth_dev, size_t depth_step, size_t depth_lane_stride, const void* rgb_dev, size_t rgb_step,
                              size_t rgb_lane_stride) {
  if (!e || !depth_dev || !rgb_dev) return RGBID_E_INVALID;
  hipStream_t s = e->ctx->stream;
  const rgbid_engine_config& c = e->cfg;
  if (depth_step < (size_t)c.cols * 2 || rgb_step < (size_t)c.cols * 3 || (depth_step & 1) || depth_step >= ((size_t)1 << 24) || rgb_step >= ((size_t)1 << 24) ||
      (e->B > 1 && (depth_lane_stride < depth_step * c.rows || rgb_lane_stride < rgb_step * c.rows || (depth_lane_stride & 1))))
    return RGBID_E_INVALID;
  hipSetDevice(e->ctx->device);
  // A captured graph bakes its kernel arguments in, so graph replay reads fixed staging buffers (-> pitched lanes); eager steps read the
  // caller's buffers in place (they must stay valid until the step has run, as documented).
  hipError_t he = hipSuccess;
  if (c.use_graph) {   // also while profiling (which only skips the replay): the buffer-lifetime rule of graph mode does not change
    // the staging lanes are contiguous (lane stride = pitch x rows): a source with contiguous lanes is ONE 2-D copy per map, otherwise one per lane
    const bool packed = e->B == 1 || (depth_lane_stride == depth_step * c.rows && rgb_lane_stride == rgb_step * c.rows);
    const int copies = packed ? 1 : e->B;
    const size_t nrows = packed ? (size_t)c.rows * e->B : (size_t)c.rows;
    for (int l = 0; l < copies && he == hipSuccess; ++l) {
      he = hipMemcpy2DAsync((char*)e->in_depth.base + (size_t)l * e->in_depth.lane_stride, e->in_depth.pitch, (const char*)depth_dev + (size_t)l * depth_lane_stride, depth_step,
                            (size_t)c.cols * 2, nrows, hipMemcpyDeviceToDevice, s);
      if (he == hipSuccess)
        he = hipMemcpy2DAsync((char*)e->in_rgb.base + (size_t)l * e->in_rgb.lane_stride, e->in_rgb.pitch, (const char*)rgb_dev + (size_t)l * rgb_lane_stride, rgb_step,
                              (size_t)c.cols * 3, nrows, hipMemcpyDeviceToDevice, s);
    }
    if (he != hipSuccess) return (int)he;
    e->cur_depth = e->in_depth; e->cur_rgb = e->in_rgb;
  } else {
    e->cur_depth = ImgB{const_cast<void*>(depth_dev), depth_step, depth_lane_stride, c.rows, c.cols};
    e->cur_rgb = ImgB{const_cast<void*>(rgb_dev), rgb_step, rgb_lane_stride, c.rows, c.cols};
  }
  int r = RGBID_OK;
  const bool first = (e->steps == 0);
  if (c.use_graph && !e->prof_on) {
    // the launch sequence is identical every step (flags live in device memory), so one captured graph is replayed
    // two graphs: the keyframe-creation sequence of the first frame, and every later frame
    hipGraphExec_t& gx = first ? e->graph_first : e->graph_next;
    bool& ready = first ? e->graph_ready_first : e->graph_ready_next;
    if (!ready) {
      hipGraph_t g = nullptr;
      he = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
      if (he != hipSuccess) return (int)he;
      r = enqueue_step(e, s, first);
      he = hipStreamEndCapture(s, &g);
      if (r || he != hipSuccess) {
        if (g) hipGraphDestroy(g);
        return r ? r : (int)he;
      }
      he = hipGraphInstantiate(&gx, g, nullptr, nullptr, 0);
      hipGraphDestroy(g);
      if (he != hipSuccess) return (int)he;
      ready = true;
    }
    he = hipGraphLaunch(gx, s);
    if (he != hipSuccess) return (int)he;
  } else {
    r = enqueue_step(e, s, first);
    if (r) return r;
  }
  int slot = e->steps % c.record_capacity;
  he = hipMemcpyAsync(e->records + (size_t)slot * e->B, e->rec_cur, sizeof(rgbid_pose_record) * e->B, hipMemcpyDeviceToDevice, s);
  if (he != hipSuccess) return (int)he;
  e->steps++;
  if (!e->ctx->async) { he = hipStreamSynchronize(s); if (he != hipSuccess) return (int)he; }
  return RGBID_OK;
}

int rgbid_engine_steps(const rgbid_engine* e) { return e ? e->steps : 0; }

int rgbid_engine_read_records(rgbid_engine* e, int first_step, int n_steps, rgbid_pose_record* out) {
  if (!e || !out || n_steps < 0 || first_step < 0 || first_step + n_steps > e->steps || e->steps - first_step > e->cfg.record_capacity) return RGBID_E_INVALID;
  for (int k = 0; k < n_steps; ++k) {
    int slot = (first_step + k) % e->cfg.record_capacity;
    hipError_t he = hipMemcpyAsync(out + (size_t)k * e->B, e->records + (size_t)slot * e->B, sizeof(rgbid_pose_record) * e->B, hipMemcpyDeviceToHost, e->ctx->stream);
    if (he != hipSuccess) return (int)he;
  }
  hipError_t he = hipStreamSynchronize(e->ctx->stream);
  return he == hipSuccess ? RGBID_OK : (int)he;
}

int rgbid_engine_pack_gather_records(rgbid_engine* e, int first_step, int n_steps, rgbid_gather_record* out_dev) {
  static_assert(sizeof(rgbid_gather_record) == 392, "SURVEY 8e record");
  if (!e || !out_dev || n_steps < 0 || first_step < 0 || first_step + n_steps > e->steps || e->steps - first_step > e->cfg.record_capacity) return RGBID_E_INVALID;
  if (n_steps == 0) return RGBID_OK;
  hipSetDevice(e->ctx->device);
  const int n = e->B * n_steps;
  hipLaunchKernelGGL(k_pack_gather, dim3(div_up(n, 128)), dim3(128), 0, e->ctx->stream, e->records, e->cfg.record_capacity, e->B, first_step, n_steps, out_dev);
  hipError_t he = hipGetLastError();
  if (he == hipSuccess && !e->ctx->async) he = hipStreamSynchronize(e->ctx->stream);
  return he == hipSuccess ? RGBID_OK : (int)he;
}

int rgbid_engine_records_dev(rgbid_engine* e, void** ptr, int* capacity) {
  if (!e || !ptr) return RGBID_E_INVALID;
  *ptr = e->records;
  if (capacity) *capacity = e->cfg.record_capacity;
  return RGBID_OK;
}

int rgbid_engine_keyframe_counts(rgbid_engine* e, int* counts) {
  if (!e || !counts) return RGBID_E_INVALID;
  if (!e->kf_hdr) { for (int i = 0; i < e->B; ++i) counts[i] = 0; return RGBID_OK; }
  hipSetDevice(e->ctx->device);
  hipError_t he = hipMemcpy2DAsync(e->kf_counts_host, sizeof(int), &e->state[0].kf_exported, sizeof(LaneState), sizeof(int), e->B,
                                   hipMemcpyDeviceToHost, e->ctx->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(e->ctx->stream);
  if (he != hipSuccess) return (int)he;
  memcpy(counts, e->kf_counts_host, sizeof(int) * e->B);
  return RGBID_OK;
}

int rgbid_engine_read_keyframe(rgbid_engine* e, int lane, int seq, rgbid_keyframe_header* header, unsigned char* overlap_mask,
                               unsigned char* colors, float* depthinv, float* normals) {
  if (!e || !e->kf_hdr || lane < 0 || lane >= e->B || seq < 0) return RGBID_E_INVALID;
  hipSetDevice(e->ctx->device);
  const int cap = e->cfg.keyframe_capacity, slot = seq % cap;
  hipStream_t s = e->ctx->stream;
  rgbid_keyframe_header* h = reinterpret_cast<rgbid_keyframe_header*>(e->kf_staging);
  const bool images = overlap_mask || colors || depthinv || normals;
  hipError_t he = hipMemcpyAsync(h, e->kf_hdr + (size_t)lane * cap + slot, sizeof(*h), hipMemcpyDeviceToHost, s);
  if (he == hipSuccess && images)
    he = hipMemcpyAsync(e->kf_staging + sizeof(*h), e->kf_blocks + ((size_t)lane * cap + slot) * e->kf_block_bytes, e->kf_block_bytes,
                        hipMemcpyDeviceToHost, s);
  if (he == hipSuccess) he = hipStreamSynchronize(s);
  if (he != hipSuccess) return (int)he;
  int count = 0;
  {
    he = hipMemcpyAsync(e->kf_counts_host, &e->state[lane].kf_exported, sizeof(int), hipMemcpyDeviceToHost, s);
    if (he == hipSuccess) he = hipStreamSynchronize(s);
    if (he != hipSuccess) return (int)he;
    count = e->kf_counts_host[0];
  }
  if (seq >= count || seq < count - cap || h->seq != seq) return RGBID_E_INVALID;  // not exported yet / already overwritten
  if (header) *header = *h;
  const size_t N = (size_t)e->cfg.rows * e->cfg.cols;
  const char* blk = e->kf_staging + sizeof(*h);
  if (overlap_mask) memcpy(overlap_mask, blk, N);
  if (colors) memcpy(colors, blk + N, 3 * N);
  if (depthinv) memcpy(depthinv, blk + 4 * N, 4 * N);
  if (normals) memcpy(normals, blk + 8 * N, 12 * N);
  return RGBID_OK;
}

int rgbid_engine_keyframes_dev(rgbid_engine* e, void** headers, void** blocks, size_t* block_bytes) {
  if (!e || !e->kf_hdr) return RGBID_E_INVALID;
  if (headers) *headers = e->kf_hdr;
  if (blocks) *blocks = e->kf_blocks;
  if (block_bytes) *block_bytes = e->kf_block_bytes;
  return RGBID_OK;
}

static rgbid_img lane_img(const ImgB& im, int lane) {
  rgbid_img r;
  r.data = (char*)im.base + (size_t)lane * im.lane_stride; r.step = im.pitch; r.rows = im.rows; r.cols = im.cols;
  return r;
}

int rgbid_engine_keyframe_maps(rgbid_engine* e, int lane, rgbid_img* depthinv, rgbid_img* weight, rgbid_img* vmap, rgbid_img* nmap, rgbid_img* overlap_mask) {
  if (!e || lane < 0 || lane >= e->B) return RGBID_E_INVALID;
  if ((vmap || nmap) && e->cfg.defer_keyframe_maps && !e->cfg.preview) {   // deferred schedule: the maps of the fused keyframe as it stands, now
    hipSetDevice(e->ctx->device);
    const IntrP K0{e->cfg.fx, e->cfg.fy, e->cfg.cx, e->cfg.cy};
    hipStream_t s = e->ctx->stream;
    if (!launch_kf_maps(s, e->B, e->iD_integr, e->vmap, e->nmap, K0, ALL)) {
      launch_vmap(s, e->B, e->iD_integr, e->vmap, K0, ALL);
      launch_gradient(s, e->B, e->iD_integr, e->gxD_integr, e->gyD_integr, ALL);
      launch_nmap_gradients(s, e->B, e->iD_integr, e->gxD_integr, e->gyD_integr, e->nmap, K0, ALL);
    }
  }
  if (depthinv) *depthinv = lane_img(e->iD_integr, lane);
  if (weight) *weight = lane_img(e->w_integr, lane);
  if (vmap) *vmap = lane_img(e->vmap, lane);
  if (nmap) *nmap = lane_img(e->nmap, lane);
  if (overlap_mask) *overlap_mask = lane_img(e->overlap_mask, lane);
  return RGBID_OK;
}

int rgbid_engine_current_maps(rgbid_engine* e, int lane, rgbid_img* depthinv, rgbid_img* intensity) {
  if (!e || lane < 0 || lane >= e->B) return RGBID_E_INVALID;
  if (depthinv) *depthinv = lane_img(e->iD_curr[0], lane);
  if (intensity) *intensity = lane_img(e->I_curr[0], lane);
  return RGBID_OK;
}

int rgbid_engine_preview(rgbid_engine* e, int lane, rgbid_img* preview, rgbid_img* colors) {
  if (!e || lane < 0 || lane >= e->B || !e->cfg.preview) return RGBID_E_INVALID;
  if (preview) *preview = lane_img(e->preview, lane);
  if (colors) *colors = lane_img(e->colors_integr, lane);
  return RGBID_OK;
}

int rgbid_engine_profile_begin(rgbid_engine* e, int max_launches) {
  if (!e || max_launches < 1) return RGBID_E_INVALID;
  while ((int)e->prof_ev.size() < 2 * max_launches) {
    hipEvent_t ev;
    // no system-scope fence around the timed kernel: the events only order against work on this stream
    hipError_t he = hipEventCreate(&ev);
    if (he != hipSuccess) return (int)he;
    e->prof_ev.push_back(ev);
  }
  e->prof_used = 0;
  e->prof_on = true;
  return RGBID_OK;
}

int rgbid_engine_profile_end(rgbid_engine* e, double* total_ms, int* n_launches, double* bytes_per_launch) {
  if (!e || !total_ms || !n_launches) return RGBID_E_INVALID;
  hipError_t he = hipStreamSynchronize(e->ctx->stream);
  if (he != hipSuccess) return (int)he;
  double tot = 0.0;
  for (int i = 0; i + 1 < e->prof_used; i += 2) {
    float ms = 0.f;
    he = hipEventElapsedTime(&ms, e->prof_ev[i], e->prof_ev[i + 1]);
    if (he != hipSuccess) return (int)he;
    tot += ms;
  }
  *total_ms = tot;
  *n_launches = e->prof_used / 2;
  // unit U1: 8 fp32 maps read per pixel of the level-0 frame, for every lane of the launch
  if (bytes_per_launch) *bytes_per_launch = 32.0 * (double)e->cfg.rows * (double)e->cfg.cols * (double)e->B;
  e->prof_on = false;
  return RGBID_OK;
}

int rgbid_engine_bytes(const rgbid_engine* e, size_t* bytes) { if (!e || !bytes) return RGBID_E_INVALID; *bytes = e->bytes; return RGBID_OK; }
int rgbid_engine_launches_per_step(const rgbid_engine* e) { return e ? e->launches : 0; }
int rgbid_engine_step_bytes(const rgbid_engine* e, double out[4]) {
  if (!e || !out) return RGBID_E_INVALID;
  for (int i = 0; i < 4; ++i) out[i] = e->step_bytes[i];
  return RGBID_OK;
}

}  // extern "C"
