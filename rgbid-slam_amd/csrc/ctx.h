// ctx.h -- the opaque rgbid_ctx: device, stream, timing events and a small scratch arena.  The reference
// allocates and frees its scratch on every sigma / visibility call (sigmaFuncs.cu:872-898, warping_
// registration.cu:838-858); here it lives in the context.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct rgbid_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  int async = 0;
  int interp_mode = 1;
  int numerics = 0;            // RGBID_NUMERICS_EXACT / _FAST for the bridge calls that have a fast variant
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  void* small_dev = nullptr;   // ctx_small_bytes of device scratch
  void* small_host = nullptr;  // pinned mirror
  double* partials = nullptr;  // [nblk][27] workgroup partial sums of the normal equations
  size_t partials_cap = 0;     // in doubles
  void* lane_dev = nullptr;    // per-lane parameter / result scratch of the batched C-ABI (rgbid_batched.h), grown on demand
  void* lane_host = nullptr;   // pinned mirror
  size_t lane_cap = 0;         // bytes
  hipEvent_t lane_ev = nullptr; // recorded behind the last H2D out of lane_host: the next call waits for it before it rewrites the staging area
  bool lane_ev_pending = false;
};

namespace rgbid {
constexpr size_t ctx_off_sums = 0;      // 27 doubles
constexpr size_t ctx_off_counts = 256;  // 2 uint
constexpr size_t ctx_off_sigma = 320;   // SigmaIO
constexpr size_t ctx_off_chi = 384;     // 3 floats
constexpr size_t ctx_small_bytes = 1024;
int ctx_reserve_partials(rgbid_ctx* c, size_t n_doubles);
int ctx_reserve_lane(rgbid_ctx* c, size_t bytes);
}  // namespace rgbid
