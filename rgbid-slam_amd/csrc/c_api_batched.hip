// c_api_batched.hip -- the batched C-ABI of include/rgbid_batched.h: every kernel the engine (engine.hip) launches on its hot path as a
// single call over `lanes` images.  Argument validation, per-lane parameters staged through the context's pinned scratch (an event behind each
// staging copy lets an asynchronous context issue the next call without overwriting parameters still in flight), the SAME launchers
// (kernels.h) the engine calls -- nothing here has an implementation of its own, and nothing falls back to another numerics class or to a CPU.
#include "../../include/rgbid_batched.h"
#include "ctx.h"
#include "kernels.h"

#include <cstring>
#include <vector>

using namespace rgbid;

namespace {

#define RGBID_HIPB(expr)                                               \
  do {                                                                 \
    hipError_t e_ = (expr);                                            \
    if (e_ != hipSuccess) { (void)hipGetLastError(); return (int)e_; } \
  } while (0)

const LaneMask ALL{nullptr, 0};

// rows and the row pitch enter the kernels' 24-bit row-offset multiply (common.h row_ptr): both stay below 2^24 and a lane's image below 4 GB.
// elem: bytes per pixel of the map (4: the fp32 maps; 2 / 3: the 16-bit depth and packed rgb inputs of frame preparation) -- a row must hold its
// `cols` pixels (a caller-described step < cols * elem would make kernels write past rows and lanes) and rows / lanes must start on element
// boundaries; lanes is a grid dimension of the kernels (<= 65535).
inline bool ok_b(const rgbid_imgb* i, int lanes, size_t elem = 4) {
  const size_t align = elem == 3 ? 1 : elem;
  return i && i->data && lanes >= 1 && lanes <= 65535 && i->rows > 0 && i->cols > 0 && i->step >= (size_t)i->cols * elem && i->step % align == 0 &&
         ((uintptr_t)i->data) % align == 0 && i->rows < (1 << 24) && i->step < ((size_t)1 << 24) && (unsigned long long)i->rows * i->step < (1ull << 32) &&
         (lanes == 1 || (i->lane_stride >= (size_t)i->rows * i->step && i->lane_stride % align == 0));
}
inline bool same_b(const rgbid_imgb* a, const rgbid_imgb* b) { return a->rows == b->rows && a->cols == b->cols; }
inline ImgB BB(const rgbid_imgb* i, int lanes) { return ImgB{i->data, i->step, lanes == 1 ? 0 : i->lane_stride, i->rows, i->cols}; }
inline bool numerics_ok(int n) { return n == RGBID_NUMERICS_EXACT || n == RGBID_NUMERICS_FAST; }
inline bool al16(const rgbid_imgb* a, int lanes) { return (a->step & 15) == 0 && (lanes == 1 || (a->lane_stride & 15) == 0) && (((uintptr_t)a->data) & 15) == 0; }

// one call's lifetime: optional event timing of its kernels, the synchronous-on-return contract, read-back of host results
struct Call {
  rgbid_ctx* c;
  float* ms;
  Call(rgbid_ctx* c_, float* ms_) : c(c_), ms(ms_) { if (ms) hipEventRecord(c->ev0, c->stream); }
  // results: bytes of the lane scratch (from offset 0 of `from`) the caller reads on the host afterwards
  int finish(const void* from = nullptr, size_t result_bytes = 0, size_t host_off = 0) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (ms) hipEventRecord(c->ev1, c->stream);
    if (result_bytes) {
      e = hipMemcpyAsync((char*)c->lane_host + host_off, from, result_bytes, hipMemcpyDeviceToHost, c->stream);
      if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    }
    if (result_bytes || ms || !c->async) {
      e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    }
    if (ms) hipEventElapsedTime(ms, c->ev0, c->ev1);
    return RGBID_OK;
  }
};

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// the pinned staging area may be rewritten once the previous call's H2D out of it has run (the kernels behind it need not have)
int staging_free(rgbid_ctx* c) {
  if (c->lane_ev_pending) { RGBID_HIPB(hipEventSynchronize(c->lane_ev)); c->lane_ev_pending = false; }
  return RGBID_OK;
}
int staging_sent(rgbid_ctx* c) {
  RGBID_HIPB(hipEventRecord(c->lane_ev, c->stream));
  c->lane_ev_pending = true;
  return RGBID_OK;
}
// per-lane warps host -> pinned staging -> device at `off` of the lane scratch (the scratch must have been reserved; call staging_free() first
// and staging_sent() after the last stage_* of the call)
int stage_warps(rgbid_ctx* c, int lanes, const float* R, const float* t, size_t off) {
  WarpParams* h = reinterpret_cast<WarpParams*>((char*)c->lane_host + off);
  for (int l = 0; l < lanes; ++l) {
    for (int i = 0; i < 9; ++i) h[l].R[i] = R[(size_t)l * 9 + i];
    for (int i = 0; i < 3; ++i) h[l].t[i] = t[(size_t)l * 3 + i];
  }
  RGBID_HIPB(hipMemcpyAsync((char*)c->lane_dev + off, h, sizeof(WarpParams) * lanes, hipMemcpyHostToDevice, c->stream));
  return RGBID_OK;
}

int stage_sys(rgbid_ctx* c, int lanes, rgbid_intr k, const rgbid_sys_params* p, size_t off) {
  SysParams* h = reinterpret_cast<SysParams*>((char*)c->lane_host + off);
  for (int l = 0; l < lanes; ++l)
    h[l] = SysParams{k.fx, k.fy, k.cx, k.cy, p[l].sigma_depthinv, p[l].sigma_int, p[l].bias_depthinv, p[l].bias_int, p[l].nu_depthinv, p[l].nu_int,
                     p[l].mestimator, p[l].weighting, p[l].student_nu ? 1 : 0, p[l].nu_int_from_max ? 1 : 0};
  RGBID_HIPB(hipMemcpyAsync((char*)c->lane_dev + off, h, sizeof(SysParams) * lanes, hipMemcpyHostToDevice, c->stream));
  return RGBID_OK;
}

bool sys_params_ok(int lanes, const rgbid_sys_params* p) {
  for (int l = 0; l < lanes; ++l)
    if (p[l].mestimator < RGBID_LSQ || p[l].mestimator > RGBID_STUDENT || p[l].weighting < RGBID_INDEPENDENT || p[l].weighting > RGBID_PHOT_ONLY) return false;
  return true;
}

void unpack_system(const double* sums, int lanes, double* A, double* b) {
  for (int l = 0; l < lanes; ++l) {
    const double* h = sums + (size_t)l * SYS_TERMS;
    double* Al = A + (size_t)l * 36;
    double* bl = b + (size_t)l * 6;
    int shift = 0;  // estimate_VO.cu:774-786
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 7; ++j) {
        const double v = h[shift++];
        if (j == 6) bl[i] = v; else Al[j * 6 + i] = Al[i * 6 + j] = v;
      }
  }
}

}  // namespace

extern "C" {

int rgbid_gn_fused_batched(rgbid_ctx* c, int lanes, const rgbid_imgb* W0, const rgbid_imgb* I0, const rgbid_imgb* gWx, const rgbid_imgb* gWy,
                           const rgbid_imgb* gIx, const rgbid_imgb* gIy, const rgbid_imgb* Wcur, const rgbid_imgb* Icur, const float* R, const float* t,
                           rgbid_intr intr, const rgbid_sys_params* params, int numerics, int weight_mode, double* A, double* b, float* ms) {
  const rgbid_imgb* all[8] = {W0, I0, gWx, gWy, gIx, gIy, Wcur, Icur};
  if (!c || lanes < 1 || !R || !t || !params || !A || !b || !numerics_ok(numerics) || weight_mode < RGBID_WM_AUTO || weight_mode > RGBID_WM_STUDENT_FIXED) return RGBID_E_INVALID;
  for (int i = 0; i < 8; ++i) if (!ok_b(all[i], lanes) || !same_b(all[i], W0)) return RGBID_E_INVALID;
  if (!sys_params_ok(lanes, params)) return RGBID_E_INVALID;
  bool all_nu = true, all_fixed = true;
  for (int l = 0; l < lanes; ++l) {
    const bool notmin = params[l].weighting != RGBID_MIN_WEIGHT;
    all_nu = all_nu && params[l].student_nu && notmin;
    all_fixed = all_fixed && !params[l].student_nu && params[l].mestimator == RGBID_STUDENT && notmin;
  }
  int wm = weight_mode;
  if (wm == RGBID_WM_AUTO) wm = all_nu ? RGBID_WM_STUDENT_NU : all_fixed ? RGBID_WM_STUDENT_FIXED : RGBID_WM_GENERIC;
  if ((wm == RGBID_WM_STUDENT_NU && !all_nu) || (wm == RGBID_WM_STUDENT_FIXED && !all_fixed)) return RGBID_E_INVALID;
  const bool fast = numerics == RGBID_NUMERICS_FAST;
  const ImgB bW0 = BB(W0, lanes), bI0 = BB(I0, lanes), bgWx = BB(gWx, lanes), bgWy = BB(gWy, lanes), bgIx = BB(gIx, lanes), bgIy = BB(gIy, lanes),
             bWc = BB(Wcur, lanes), bIc = BB(Icur, lanes);
  if (fast && !gn_fast_supported(bW0, bI0, bgWx, bgWy, bgIx, bgIy, bIc)) return RGBID_E_INVALID;
  hipSetDevice(c->device);
  const size_t off_wp = 0, off_sp = up256(sizeof(WarpParams) * lanes), off_sums = off_sp + up256(sizeof(SysParams) * lanes);
  int e = ctx_reserve_lane(c, off_sums + sizeof(double) * SYS_TERMS * lanes);
  if (e) return e;
  const int nb = system_blocks_per_lane(W0->rows, W0->cols, lanes);
  e = ctx_reserve_partials(c, (size_t)nb * SYS_TERMS * lanes);
  if (e) return e;
  if ((e = staging_free(c))) return e;
  if ((e = stage_warps(c, lanes, R, t, off_wp))) return e;
  if ((e = stage_sys(c, lanes, intr, params, off_sp))) return e;
  if ((e = staging_sent(c))) return e;
  double* sums_d = reinterpret_cast<double*>((char*)c->lane_dev + off_sums);
  Call call(c, ms);
  const int nblk = launch_gn_fused(c->stream, lanes, bW0, bI0, bgWx, bgWy, bgIx, bgIy, bWc, bIc, reinterpret_cast<const WarpParams*>((char*)c->lane_dev + off_wp),
                                   c->interp_mode, reinterpret_cast<const SysParams*>((char*)c->lane_dev + off_sp), c->partials, ALL, 0, fast, wm);
  if (nblk < 0) return RGBID_E_INVALID;
  launch_reduce_system(c->stream, lanes, c->partials, nblk, sums_d, ALL);
  e = call.finish(sums_d, sizeof(double) * SYS_TERMS * lanes, off_sums);
  if (e) return e;
  unpack_system(reinterpret_cast<const double*>((char*)c->lane_host + off_sums), lanes, A, b);
  return RGBID_OK;
}

int rgbid_build_system_batched(rgbid_ctx* c, int lanes, const rgbid_imgb* W0, const rgbid_imgb* I0, const rgbid_imgb* gWx, const rgbid_imgb* gWy,
                               const rgbid_imgb* gIx, const rgbid_imgb* gIy, const rgbid_imgb* W1, const rgbid_imgb* I1, rgbid_intr intr,
                               const rgbid_sys_params* params, double* A, double* b, float* ms) {
  const rgbid_imgb* all[8] = {W0, I0, gWx, gWy, gIx, gIy, W1, I1};
  if (!c || lanes < 1 || !params || !A || !b) return RGBID_E_INVALID;
  for (int i = 0; i < 8; ++i) if (!ok_b(all[i], lanes) || !same_b(all[i], W0)) return RGBID_E_INVALID;
  if (!sys_params_ok(lanes, params)) return RGBID_E_INVALID;
  hipSetDevice(c->device);
  const size_t off_sp = 0, off_sums = up256(sizeof(SysParams) * lanes);
  int e = ctx_reserve_lane(c, off_sums + sizeof(double) * SYS_TERMS * lanes);
  if (e) return e;
  const int nb = system_blocks_per_lane(W0->rows, W0->cols, lanes);
  e = ctx_reserve_partials(c, (size_t)nb * SYS_TERMS * lanes);
  if (e) return e;
  if ((e = staging_free(c))) return e;
  if ((e = stage_sys(c, lanes, intr, params, off_sp))) return e;
  if ((e = staging_sent(c))) return e;
  double* sums_d = reinterpret_cast<double*>((char*)c->lane_dev + off_sums);
  Call call(c, ms);
  const int nblk = launch_build_system(c->stream, lanes, BB(W0, lanes), BB(I0, lanes), BB(gWx, lanes), BB(gWy, lanes), BB(gIx, lanes), BB(gIy, lanes), BB(W1, lanes),
                                       BB(I1, lanes), nullptr, reinterpret_cast<const SysParams*>((char*)c->lane_dev + off_sp), c->partials, ALL, 0);
  launch_reduce_system(c->stream, lanes, c->partials, nblk, sums_d, ALL);
  e = call.finish(sums_d, sizeof(double) * SYS_TERMS * lanes, off_sums);
  if (e) return e;
  unpack_system(reinterpret_cast<const double*>((char*)c->lane_host + off_sums), lanes, A, b);
  return RGBID_OK;
}

int rgbid_warp_pair_batched(rgbid_ctx* c, int lanes, const rgbid_imgb* src_iD, const rgbid_imgb* src_I, const rgbid_imgb* grid, const rgbid_imgb* dst_iD,
                            const rgbid_imgb* dst_I, const float* R, const float* t, int numerics, float* ms) {
  const rgbid_imgb* all[5] = {src_iD, src_I, grid, dst_iD, dst_I};
  if (!c || lanes < 1 || !R || !t || !numerics_ok(numerics)) return RGBID_E_INVALID;
  for (int i = 0; i < 5; ++i) if (!ok_b(all[i], lanes) || !same_b(all[i], src_iD)) return RGBID_E_INVALID;
  const bool fast = numerics == RGBID_NUMERICS_FAST;
  if (fast && (src_I->cols < 2 || src_I->rows < 2)) return RGBID_E_INVALID;
  hipSetDevice(c->device);
  int e = ctx_reserve_lane(c, sizeof(WarpParams) * lanes);
  if (e) return e;
  if ((e = staging_free(c))) return e;
  if ((e = stage_warps(c, lanes, R, t, 0))) return e;
  if ((e = staging_sent(c))) return e;
  const WarpParams* wp = reinterpret_cast<const WarpParams*>(c->lane_dev);
  Call call(c, ms);
  if (fast) {
    if (!launch_warp_pair_fast(c->stream, lanes, BB(src_iD, lanes), BB(src_I, lanes), BB(grid, lanes), BB(dst_iD, lanes), BB(dst_I, lanes), nullptr, wp, c->interp_mode, ALL))
      return RGBID_E_INVALID;
  } else {
    launch_warp_pair(c->stream, lanes, BB(src_iD, lanes), BB(src_I, lanes), BB(grid, lanes), BB(dst_iD, lanes), BB(dst_I, lanes), wp, c->interp_mode, ALL);
  }
  return call.finish();
}

int rgbid_lattice_pack_batched(rgbid_ctx* c, int lanes, const rgbid_imgb* W0, const rgbid_imgb* I0, int min_nsamples, float* out_dev, size_t out_lane_stride, float* ms) {
  if (!c || lanes < 1 || !ok_b(W0, lanes) || !ok_b(I0, lanes) || !same_b(W0, I0) || !out_dev) return RGBID_E_INVALID;
  const int n = lattice_samples(W0->rows, W0->cols, min_nsamples);
  if (out_lane_stride < 2 * (size_t)n) return RGBID_E_INVALID;
  hipSetDevice(c->device);
  Call call(c, ms);
  launch_lattice_pack(c->stream, lanes, BB(W0, lanes), BB(I0, lanes), min_nsamples, out_dev, out_lane_stride, ALL);
  return call.finish();
}

int rgbid_lattice_residuals_batched(rgbid_ctx* c, int lanes, const rgbid_imgb* Wcur, const rgbid_imgb* W0, const rgbid_imgb* Icur, const rgbid_imgb* I0,
                                    const float* R, const float* t, int min_nsamples, int numerics, const float* kf_lat_dev, size_t kf_lat_lane_stride,
                                    float* res_dev, size_t res_lane_stride, float* ms) {
  const rgbid_imgb* all[4] = {Wcur, W0, Icur, I0};
  if (!c || lanes < 1 || !R || !t || !res_dev || !numerics_ok(numerics)) return RGBID_E_INVALID;
  for (int i = 0; i < 4; ++i) if (!ok_b(all[i], lanes) || !same_b(all[i], W0)) return RGBID_E_INVALID;
  const int n = lattice_samples(W0->rows, W0->cols, min_nsamples);
  if (res_lane_stride < 2 * (size_t)n || (kf_lat_dev && kf_lat_lane_stride < 2 * (size_t)n)) return RGBID_E_INVALID;
  const bool fast = numerics == RGBID_NUMERICS_FAST;
  // FAST: the part of gn_fast_supported() that can be checked on the maps at hand -- the lattice must run in the class of the normal equations that follow
  if (fast && !(W0->cols % 4 == 0 && W0->cols >= 4 && W0->rows >= 2 && al16(W0, lanes) && al16(I0, lanes) && W0->step == I0->step)) return RGBID_E_INVALID;
  hipSetDevice(c->device);
  int e = ctx_reserve_lane(c, sizeof(WarpParams) * lanes);
  if (e) return e;
  if ((e = staging_free(c))) return e;
  if ((e = stage_warps(c, lanes, R, t, 0))) return e;
  if ((e = staging_sent(c))) return e;
  Call call(c, ms);
  launch_lattice_residuals_fused(c->stream, lanes, BB(Wcur, lanes), BB(W0, lanes), BB(Icur, lanes), BB(I0, lanes), reinterpret_cast<const WarpParams*>(c->lane_dev),
                                 c->interp_mode, min_nsamples, ALL, fast, res_dev, res_lane_stride, kf_lat_dev, kf_lat_dev ? kf_lat_lane_stride : 0);
  return call.finish();
}

int rgbid_sigma_pair_batched(rgbid_ctx* c, int lanes, const float* res_dev, size_t res_lane_stride, int n, int mestimator, rgbid_scale_pair* out, float* ms) {
  if (!c || lanes < 1 || !res_dev || n < 1 || res_lane_stride < 2 * (size_t)n || !out || mestimator < RGBID_LSQ || mestimator > RGBID_STUDENT) return RGBID_E_INVALID;
  hipSetDevice(c->device);
  int e = ctx_reserve_lane(c, sizeof(SysParams) * lanes);
  if (e) return e;
  SysParams* sp = reinterpret_cast<SysParams*>(c->lane_dev);
  RGBID_HIPB(hipMemsetAsync(sp, 0, sizeof(SysParams) * lanes, c->stream));
  Call call(c, ms);
  launch_sigma_pair_arrays(c->stream, lanes, res_dev, res_lane_stride, n, sp, mestimator, ALL);
  e = call.finish(sp, sizeof(SysParams) * lanes, 0);
  if (e) return e;
  const SysParams* h = reinterpret_cast<const SysParams*>(c->lane_host);
  for (int l = 0; l < lanes; ++l) out[l] = rgbid_scale_pair{h[l].bias_d, h[l].sigma_d, h[l].nu_d, h[l].bias_i, h[l].sigma_i, h[l].nu_i};
  return RGBID_OK;
}

int rgbid_fuse_frame_batched(rgbid_ctx* c, int lanes, const rgbid_imgb* cur, const rgbid_imgb* kf, const rgbid_imgb* kfw, const rgbid_imgb* wweight,
                             const float* R, const float* t, int numerics, float* ms) {
  const rgbid_imgb* all[4] = {cur, kf, kfw, wweight};
  if (!c || lanes < 1 || !R || !t || !numerics_ok(numerics)) return RGBID_E_INVALID;
  for (int i = 0; i < 4; ++i) if (!ok_b(all[i], lanes) || !same_b(all[i], kf)) return RGBID_E_INVALID;
  if (kf->cols % 4 != 0 || !al16(kf, lanes) || !al16(kfw, lanes) || !al16(wweight, lanes)) return RGBID_E_INVALID;   // 16-byte geometry
  hipSetDevice(c->device);
  int e = ctx_reserve_lane(c, sizeof(WarpParams) * lanes);
  if (e) return e;
  if ((e = staging_free(c))) return e;
  if ((e = stage_warps(c, lanes, R, t, 0))) return e;
  if ((e = staging_sent(c))) return e;
  Call call(c, ms);
  if (!launch_fuse_frame(c->stream, lanes, BB(cur, lanes), BB(kf, lanes), BB(kfw, lanes), BB(wweight, lanes), reinterpret_cast<const WarpParams*>(c->lane_dev), ALL,
                         numerics == RGBID_NUMERICS_FAST))
    return RGBID_E_INVALID;
  return call.finish();
}

int rgbid_kf_maps_batched(rgbid_ctx* c, int lanes, rgbid_intr k, const rgbid_imgb* depthinv, const rgbid_imgb* vmap, const rgbid_imgb* nmap, float* ms) {
  if (!c || lanes < 1 || !ok_b(depthinv, lanes) || !ok_b(vmap, lanes) || !ok_b(nmap, lanes) || vmap->rows != 3 * depthinv->rows || nmap->rows != 3 * depthinv->rows ||
      vmap->cols != depthinv->cols || nmap->cols != depthinv->cols)
    return RGBID_E_INVALID;
  if (depthinv->cols % 4 != 0 || !al16(depthinv, lanes) || !al16(vmap, lanes) || !al16(nmap, lanes)) return RGBID_E_INVALID;   // 16-byte geometry
  hipSetDevice(c->device);
  Call call(c, ms);
  if (!launch_kf_maps(c->stream, lanes, BB(depthinv, lanes), BB(vmap, lanes), BB(nmap, lanes), IntrP{k.fx, k.fy, k.cx, k.cy}, ALL)) return RGBID_E_INVALID;
  return call.finish();
}

int rgbid_visibility_pair_batched(rgbid_ctx* c, int lanes, const rgbid_imgb* a, const rgbid_imgb* b, const float* R_ab, const float* t_ab, const float* R_ba,
                                  const float* t_ba, int numerics, unsigned int* counts, float* ms) {
  if (!c || lanes < 1 || !ok_b(a, lanes) || !ok_b(b, lanes) || !same_b(a, b) || !R_ab || !t_ab || !R_ba || !t_ba || !counts || !numerics_ok(numerics)) return RGBID_E_INVALID;
  hipSetDevice(c->device);
  const size_t off_ab = 0, off_ba = up256(sizeof(WarpParams) * lanes), off_cnt = off_ba + up256(sizeof(WarpParams) * lanes);
  int e = ctx_reserve_lane(c, off_cnt + sizeof(unsigned int) * 4 * lanes);
  if (e) return e;
  if ((e = staging_free(c))) return e;
  if ((e = stage_warps(c, lanes, R_ab, t_ab, off_ab))) return e;
  if ((e = stage_warps(c, lanes, R_ba, t_ba, off_ba))) return e;
  if ((e = staging_sent(c))) return e;
  unsigned int* cnt = reinterpret_cast<unsigned int*>((char*)c->lane_dev + off_cnt);   // [2][lanes][2]: a->b block, then b->a block
  RGBID_HIPB(hipMemsetAsync(cnt, 0, sizeof(unsigned int) * 4 * lanes, c->stream));
  Call call(c, ms);
  launch_visibility_pair(c->stream, lanes, BB(a, lanes), BB(b, lanes), reinterpret_cast<const WarpParams*>((char*)c->lane_dev + off_ab),
                         reinterpret_cast<const WarpParams*>((char*)c->lane_dev + off_ba), cnt, cnt + 2 * lanes, ALL, numerics == RGBID_NUMERICS_FAST);
  e = call.finish(cnt, sizeof(unsigned int) * 4 * lanes, off_cnt);
  if (e) return e;
  const unsigned int* h = reinterpret_cast<const unsigned int*>((char*)c->lane_host + off_cnt);
  for (int l = 0; l < lanes; ++l) {
    counts[4 * l + 0] = h[2 * l + 0]; counts[4 * l + 1] = h[2 * l + 1];
    counts[4 * l + 2] = h[2 * lanes + 2 * l + 0]; counts[4 * l + 3] = h[2 * lanes + 2 * l + 1];
  }
  return RGBID_OK;
}

int rgbid_prep_frame_batched(rgbid_ctx* c, int lanes, const rgbid_imgb* depth, const rgbid_imgb* rgb, const rgbid_imgb* iD, const rgbid_imgb* I, const rgbid_imgb* r,
                             const rgbid_imgb* g, const rgbid_imgb* b, float factor_depth, float* ms) {
  const rgbid_imgb* all[7] = {depth, rgb, iD, I, r, g, b};
  if (!c || lanes < 1 || !(factor_depth == factor_depth) || factor_depth == 0.f) return RGBID_E_INVALID;
  for (int i = 0; i < 7; ++i) if (!ok_b(all[i], lanes, i == 0 ? 2 : i == 1 ? 3 : 4) || !same_b(all[i], iD)) return RGBID_E_INVALID;
  hipSetDevice(c->device);
  Call call(c, ms);
  launch_prep_frame(c->stream, lanes, BB(depth, lanes), BB(rgb, lanes), BB(iD, lanes), BB(I, lanes), BB(r, lanes), BB(g, lanes), BB(b, lanes), factor_depth, ALL);
  return call.finish();
}

int rgbid_pyr_down_batched(rgbid_ctx* c, int lanes, const rgbid_imgb* src, const rgbid_imgb* dst, float* ms) {
  if (!c || lanes < 1 || !ok_b(src, lanes) || !ok_b(dst, lanes) || dst->rows != src->rows / 2 || dst->cols != src->cols / 2) return RGBID_E_INVALID;
  hipSetDevice(c->device);
  Call call(c, ms);
  launch_pyr_down(c->stream, lanes, BB(src, lanes), BB(dst, lanes), ALL);
  return call.finish();
}

int rgbid_compute_gradient_batched(rgbid_ctx* c, int lanes, const rgbid_imgb* src, const rgbid_imgb* gx, const rgbid_imgb* gy, float* ms) {
  if (!c || lanes < 1 || !ok_b(src, lanes) || !ok_b(gx, lanes) || !ok_b(gy, lanes) || !same_b(src, gx) || !same_b(src, gy)) return RGBID_E_INVALID;
  hipSetDevice(c->device);
  Call call(c, ms);
  launch_gradient(c->stream, lanes, BB(src, lanes), BB(gx, lanes), BB(gy, lanes), ALL);
  return call.finish();
}

int rgbid_gradient_keep_batched(rgbid_ctx* c, int lanes, const rgbid_imgb* src, const rgbid_imgb* gx, const rgbid_imgb* gy, const rgbid_imgb* keep, float* ms) {
  if (!c || lanes < 1 || !ok_b(src, lanes) || !ok_b(gx, lanes) || !ok_b(gy, lanes) || !ok_b(keep, lanes) || !same_b(src, gx) || !same_b(src, gy) || !same_b(src, keep) ||
      src->data == keep->data)
    return RGBID_E_INVALID;
  hipSetDevice(c->device);
  Call call(c, ms);
  const bool launched = launch_gradient_keep(c->stream, lanes, BB(src, lanes), BB(gx, lanes), BB(gy, lanes), BB(keep, lanes), ALL);
  const int r = call.finish();
  return launched ? r : RGBID_E_INVALID;
}

int rgbid_bilateral_filter_batched(rgbid_ctx* c, int lanes, const rgbid_imgb* src, const rgbid_imgb* dst, float sigma_floatmap, int numerics, float* ms) {
  if (!c || lanes < 1 || !ok_b(src, lanes) || !ok_b(dst, lanes) || !same_b(src, dst) || src->data == dst->data || !numerics_ok(numerics) ||
      !(sigma_floatmap > 0.f))
    return RGBID_E_INVALID;
  hipSetDevice(c->device);
  Call call(c, ms);
  launch_bilateral(c->stream, lanes, BB(src, lanes), BB(dst, lanes), sigma_floatmap, ALL, numerics == RGBID_NUMERICS_FAST);
  return call.finish();
}

}  // extern "C"
