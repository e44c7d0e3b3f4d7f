// engine_device.h -- device-side pieces of the batched engine (engine.hip) that the per-lane persistent Gauss-Newton kernel (tools/experiments/persistent_gn_level/kernels_gnlevel.hip: a measured experiment, profiles/r04_experiments/persistent_gn_level.md) shares:
// the per-lane tracker state, the per-step launch predicates, the by-value step configuration, the pose -> warp projection, and the bodies of
// k_set_sys / k_solve_update as functions of (lane, thread).  Moved here from engine.hip (round 4).
#pragma once
#include "../../include/rgbid_engine.h"
#include "kernels.h"
#include "../../include/rgbid/se3.h"

// as se3.h: the per-lane double-precision logic is evaluated operation by operation (RGBID_FP_STRICT inside each body) in every kernel that inlines it

namespace rgbid {
namespace eng {

// ---- per-lane tracker state (the scalar members of VisodoTracker, include/visodo.h:280-390) -----------------
struct LaneState {
  double delta_R[9], delta_t[3], delta_cov[36];          // delta_rotation_/translation_/covariance_
  double dprev_R[9], dprev_t[3], dprev_cov[36];          // their values at the start of trackNewFrame
  double prev_R[9], prev_t[3];                           // previous_rotation/translation of estimateVisualOdometry
  double cur_R[9], cur_t[3];                             // current_rotation/translation (GN working pose)
  double odoKF_R[9], odoKF_t[3];                         // last_odoKF_global_*
  double integrKF_R[9], integrKF_t[3];                   // last_integrKF_global_*
  double last_est_R[9], last_est_t[3];                   // last_estimated_*
  double o2i_last_R[9], o2i_last_t[3], o2i_last_cov[36]; // delta_*_odo2integr_last_
  double o2i_next_R[9], o2i_next_t[3], o2i_next_cov[36]; // delta_*_odo2integr_next_
  double dI_R[9], dI_t[3];                               // delta_integr_rotation/translation of this frame
  double velocity[3], omega[3];
  int global_time, lost, odoKF_count, integrKF_count, last_odoKF_index, last_integrKF_index;
  int gn_failed;
  int kf_exported;  // keyframes this lane has handed to the export ring
  int status;       // RGBID_ST_* bits of the current step
  float vis_odo, vis_int;
  float rec_sigma_i, rec_sigma_d, rec_nu_i, rec_nu_d;  // scale estimates of the last GN iteration (diagnostics)
  // CHI_SQUARED termination (visodo.cpp:1134-1164): the last increment (to undo it) and the previous RMSE of estimateVisualOdometry
  double inc_inv_R[9], inc_t[3];
  float rmse_prev;
  // round 6, few-lane plan: the Gauss-Newton update of an iteration runs as the prologue of the NEXT iteration's first launch, in every workgroup of it; the
  // working pose then alternates between cur_* and this second buffer (a workgroup that starts late must still read the pose its launch began with)
  double alt_R[9], alt_t[3];
};

struct Flags {  // int[B] each; consumed through LaneMask
  int *track, *first, *gn, *vis, *sw_odo, *sw_int, *overlap, *fuse, *maps;
  int* lvl;      // lanes still iterating the CURRENT pyramid level: == gn (the same array) unless termination is CHI_SQUARED, which ends levels per lane
  int* kf_slot;  // ring slot the lane exports its outgoing integration keyframe into this step, or -1 (not a LaneMask flag)
};

struct StepCfg {  // by-value kernel argument with what the scalar kernels need
  float fx, fy, cx, cy;
  int levels, finest_level, motion_model, max_odoKF_count, max_integrKF_count;
  float visratio_odo, visratio_integr;
  const float* delta_t;  // device: inter-frame time of the constant-velocity model (a pointer, so that captured graphs follow rgbid_engine_set_delta_t)
  int mestimator, weighting;
  int start_warp_level;  // pyramid level whose intrinsics project the first warp of a frame
  rgbid_keyframe_header* kf_hdr;  // export ring headers [B][kf_cap] (nullptr: no export)
  int kf_cap;
  const int* active;     // per-lane 0/1: lanes without a new frame this step sit it out (nullptr: every lane is fed)
};

__device__ inline void set_warp_from_pose(const StepCfg& c, int level, const double* R, const double* t, WarpParams& wp) { RGBID_FP_STRICT
  // inverse pose, projected with the level's K (visodo.cpp:1066-1067,1108-1114)
  double Ri[9], ti[3];
  se3::m3_inv(R, Ri);
  se3::m3_mulv(Ri, t, ti);
  ti[0] = -ti[0]; ti[1] = -ti[1]; ti[2] = -ti[2];
  int div = 1 << level;
  se3::project_trafo(c.fx / div, c.fy / div, c.cx / div, c.cy / div, Ri, ti, wp.R, wp.t);
}


// the per-level constants of the lane's SysParams before a level's iterations / the covariance pass (k_set_sys)
__device__ inline void set_sys_lane(SysParams* sp, LaneState* st, const int* track, const StepCfg& c, int level, int cov_pass, int lane) { RGBID_FP_STRICT
  int div = 1 << level;
  SysParams& p = sp[lane];
  if (cov_pass && track[lane]) {
    st[lane].rec_sigma_i = p.sigma_i; st[lane].rec_sigma_d = p.sigma_d;
    st[lane].rec_nu_i = fmaxf(p.nu_i, p.nu_d); st[lane].rec_nu_d = p.nu_d;  // nu_int = max(nu_int, nu_depthinv), visodo.cpp:1186
  }
  p.fx = c.fx / div; p.fy = c.fy / div; p.cx = c.cx / div; p.cy = c.cy / div;
  p.weighting = c.weighting;
  if (cov_pass) {
    // visodo.cpp:1349-1365: STUDENT with the fixed reference sigmas, zero bias
    p.mestimator = RGBID_STUDENT; p.student_nu = 0; p.nu_i_max = 0;
    p.sigma_i = (float)exp(log((double)5.f)); p.sigma_d = (float)exp(log((double)0.0025f));
    p.bias_i = 0.f; p.bias_d = 0.f; p.nu_i = 5.f; p.nu_d = 5.f;
  } else {
    p.mestimator = c.mestimator; p.student_nu = 1; p.nu_i_max = 1;
    p.sigma_i = 5.f; p.sigma_d = 0.0025f; p.bias_i = 0.f; p.bias_d = 0.f; p.nu_i = 5.f; p.nu_d = 5.f;  // visodo.cpp:1168-1173
  }
}

// Fixed-order reduction of a lane's partial rows by the NT (256 or 64) threads of a workgroup (every thread of the workgroup must call it: two barriers): slice sl of 8
// sums the rows sl, sl + 8, ... of one term, then the 8 slice sums are added in order -- the same doubles whatever NT is.  NT = 256: one slice per thread, the
// shortest path for a few lanes; NT = 64: four slices per thread, ONE wave per lane -- the kernels that follow keep the whole register file per wave
// (RGBID_SCALAR_KERNEL), so a 256-thread workgroup occupies a compute unit alone while three of its waves idle: with thousands of lanes the 64-thread form runs four
// lanes per compute unit at a time.  sm: [8][32] doubles, sums: [SYS_TERMS] doubles of LDS.
template <int NT>
__device__ inline void reduce_partials(const double* partials, int nblk, int lane, int tid, double (*sm)[32], double* sums) { RGBID_FP_STRICT
  static_assert(NT == 256 || NT == 64, "8 slices of 32 threads, or 2 x 4");
  const int k = tid & 31;
  if (tid < NT) {
    for (int sl = tid >> 5; sl < 8; sl += NT / 32) {
      double t = 0.0;
      if (k < SYS_TERMS) {
        const double* p = partials + (size_t)lane * nblk * SYS_TERMS + k;
        for (int b = sl; b < nblk; b += 8) t += p[(size_t)b * SYS_TERMS];
      }
      sm[sl][k] = t;
    }
  }
  __syncthreads();
  if (tid < SYS_TERMS) {
    double r = 0.0;
    for (int i = 0; i < 8; ++i) r += sm[i][tid];
    sums[tid] = r;
  }
  __syncthreads();
}

// the arithmetic of one GN update (visodo.cpp:1242-1274) from the 27 reduced sums and the pose (Rin, tin): the new pose, the increment a CHI_SQUARED stop
// undoes, and whether the update failed (NaN, :1265-1274).  Pure: callers decide who stores what.
struct GnUpdate { double R[9], t[3], inc_inv[9], tinc[3]; bool failed; };
__device__ inline void gn_update(const double* sums, const double* Rin, const double* tin, GnUpdate& u) { RGBID_FP_STRICT
  double A[36], b[6], x[6];
  int shift = 0;  // estimate_VO.cu:774-786
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) {
      double v = sums[shift++];
      if (j == 6) b[i] = v; else A[j * 6 + i] = A[i * 6 + j] = v;
    }
  se3::llt_solve6(A, b, x);
  double inc[9], tmp[3];
  se3::expmap_rot(x + 3, u.inc_inv);
  se3::m3_inv(u.inc_inv, inc);
  se3::m3_mulv(inc, x, u.tinc);
  u.tinc[0] = -u.tinc[0]; u.tinc[1] = -u.tinc[1]; u.tinc[2] = -u.tinc[2];
  se3::m3_mulv(inc, tin, tmp);
  for (int i = 0; i < 3; ++i) u.t[i] = tmp[i] + u.tinc[i];
  se3::m3_mul(inc, Rin, u.R);
  u.failed = se3::has_nan(u.R, u.t);
}
// what an update leaves in the lane's state: the pose into (Rout, tout), the increment, the failure flags, the next warp
__device__ inline void gn_commit(const GnUpdate& u, LaneState& s, double* Rout, double* tout, const Flags& f, WarpParams* wp, const WarpParams* next_warp, const StepCfg& c,
                                 int next_level, int lane) { RGBID_FP_STRICT
  se3::m3_copy(u.R, Rout);
  for (int i = 0; i < 3; ++i) tout[i] = u.t[i];
  se3::m3_copy(u.inc_inv, s.inc_inv_R);   // cam_rot_incremental_inv / cam_trans_incremental: what a CHI_SQUARED stop undoes
  for (int i = 0; i < 3; ++i) s.inc_t[i] = u.tinc[i];
  if (u.failed) {  // :1265-1274
    s.gn_failed = 1;
    f.gn[lane] = 0; f.lvl[lane] = 0;
    return;
  }
  if (next_warp) wp[lane] = *next_warp;
  else set_warp_from_pose(c, next_level, Rout, tout, wp[lane]);
}

// one GN update of one lane by the first NT threads of a workgroup (every thread of the workgroup must call it: two barriers): fixed-order reduction
// of the lane's partial sums, LLT solve, exp-map, pose update, next warp (visodo.cpp:1242-1274).  sm: [8][32] doubles, sums: [SYS_TERMS] doubles of LDS.
// pin: the working pose is in the lane's second buffer (the few-lane plan's prologue updates left it there); the result always goes to cur_*.
template <int NT = 256>
__device__ inline void solve_update_block(const double* partials, int nblk, LaneState* st, const Flags& f, WarpParams* wp, const StepCfg& c, int next_level, int lane,
                                          int tid, double (*sm)[32], double* sums, int pin = 0) { RGBID_FP_STRICT
  reduce_partials<NT>(partials, nblk, lane, tid, sm, sums);
  if (tid != 0) return;
  LaneState& s = st[lane];
  GnUpdate u;
  gn_update(sums, pin ? s.alt_R : s.cur_R, pin ? s.alt_t : s.cur_t, u);
  gn_commit(u, s, s.cur_R, s.cur_t, f, wp, nullptr, c, next_level, lane);
}

}  // namespace eng
}  // namespace rgbid

