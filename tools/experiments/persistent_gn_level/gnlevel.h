// gnlevel.h -- launcher of the per-lane persistent Gauss-Newton level kernel (kernels_gnlevel.hip) for the engine (engine.hip)
#pragma once
#include "kernels.h"
#include "system_device.h"
#include "engine_device.h"
#include "sigma_device.h"

namespace rgbid {

enum { GN_BARRIER_STRIDE = 64 };   // unsigned words between the per-lane barrier counters (256 bytes)

struct GnLevelArgs {
  ImgB W0, I0, gWx, gWy, gIx, gIy, Wcur, Icur;      // the level's keyframe-side maps and the current frame's maps
  WarpParams* wp; SysParams* sp;                    // per lane
  double* partials; int nblk, upt; SysTiles tp;     // the launch plan of the normal equations for this geometry and lane count (system_plan_vec)
  float* lat_res; size_t lat_res_lane_stride;       // residual lattice scratch [lane][2][n_lat]
  const float* kf_lat; size_t kf_lat_lane_stride;   // packed keyframe side of the lattice (nullable)
  int n_lat, lat_cols, lat_stride_px;
  NuTable T; int mestimator, interp_mode;
  eng::LaneState* st; eng::Flags f; eng::StepCfg c;
  int level, iters;
  int level_for_warp;        // level whose intrinsics project the warp of the NEXT iteration of this level (this level; 0 with warp-first)
  int next_level_after;      // ... and the warp after the level's last iteration (the next level that iterates, or the finest level for the covariance pass)
  unsigned* barrier; unsigned barrier_base; int wpl;   // per-lane counters (zeroed every step), this launch's first target, workgroups per lane
};

int gn_level_barriers(int iters);                                       // barriers one launch passes per workgroup (the next launch's base advances by wpl times this)
int gn_level_workgroups_per_lane(int lanes, bool fast, int weight_mode); // 0: the grid would not be resident (use the launch-per-phase path)
bool gn_level_supported(const GnLevelArgs& a);
// a: the level's argument block, a_dev: its copy in device memory (static per engine and level: uploaded once).  RGBID_GN_DEBUG_SKIP (timing
// experiments): bit 0 lattice, 1 sigma / nu, 2 normal equations, 3 solve switched off, 4 no fences.
int launch_gn_level(hipStream_t s, int lanes, const GnLevelArgs& a, const GnLevelArgs* a_dev, bool fast, int weight_mode);

}  // namespace rgbid
