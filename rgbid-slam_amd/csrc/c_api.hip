// c_api.hip -- the C-ABI of include/rgbid.h: context, device memory and one wrapper per reference
// bridge function.  Every wrapper validates its arguments, launches the batched kernels of kernels.h
// with B = 1 on the context's stream and (unless the context is asynchronous) synchronises before
// returning, which is the contract of the reference's bridge (cudaStreamSynchronize after every launch).
// There is NO CPU fallback: without a HIP device every entry point returns an error.
#include "../../include/rgbid.h"
#include "ctx.h"
#include "kernels.h"
#include "guard_band.h"

#include <cstdio>
#include <cstring>
#include <new>

using namespace rgbid;

// a failing HIP call is reported through the return value; the runtime's sticky "last error" is cleared so that the caller's
// other HIP users (e.g. a framework sharing the process) do not trip over it later
#define RGBID_HIP(expr)                                             \
  do {                                                              \
    hipError_t e_ = (expr);                                         \
    if (e_ != hipSuccess) { (void)hipGetLastError(); return (int)e_; } \
  } while (0)

extern "C" {

const char* rgbid_version(void) { return "rgbid-mi355x 0.1 (gfx950)"; }

const char* rgbid_error_string(int err) {
  if (err == RGBID_OK) return "ok";
  if (err == RGBID_E_INVALID) return "rgbid: invalid argument";
  if (err == RGBID_E_NOMEM) return "rgbid: out of memory";
  if (err == RGBID_E_NODEV) return "rgbid: no usable HIP device";
  if (err > 0) return hipGetErrorString((hipError_t)err);
  return "rgbid: unknown error";
}

int rgbid_device_count(int* n) {
  if (!n) return RGBID_E_INVALID;
  *n = 0;
  hipError_t e = hipGetDeviceCount(n);
  if (e != hipSuccess) { *n = 0; return RGBID_E_NODEV; }
  return RGBID_OK;
}

int rgbid_get_device_prop(int device, rgbid_device_prop* prop) {
  if (!prop) return RGBID_E_INVALID;
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) return RGBID_E_NODEV;
  memset(prop, 0, sizeof(*prop));
  strncpy(prop->name, p.name, sizeof(prop->name) - 1);
  prop->multiProcessorCount = p.multiProcessorCount;
  prop->maxThreadsPerMultiProcessor = p.maxThreadsPerMultiProcessor;
  prop->warpSize = p.warpSize;
  prop->clockRateKHz = p.clockRate;
  prop->totalGlobalMem = p.totalGlobalMem;
  prop->sharedMemPerBlock = p.sharedMemPerBlock;
  strncpy(prop->gcnArchName, p.gcnArchName, sizeof(prop->gcnArchName) - 1);
  return RGBID_OK;
}

int rgbid_set_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); return RGBID_E_NODEV; }  // no sticky error left behind
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { (void)hipGetLastError(); return RGBID_E_NODEV; }
  return RGBID_OK;
}

int rgbid_ctx_create(rgbid_ctx** out, int device, void* stream) {
  if (!out) return RGBID_E_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return RGBID_E_NODEV;
  RGBID_HIP(hipSetDevice(device));
  rgbid_ctx* c = new (std::nothrow) rgbid_ctx();
  if (!c) return RGBID_E_NOMEM;
  c->device = device;
  c->async = 0;
  c->interp_mode = RGBID_INTERP_TEX8;
  c->owns_stream = (stream == nullptr);
  if (stream) c->stream = (hipStream_t)stream;
  else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return (int)e; }
  }
  hipError_t e = hipEventCreate(&c->ev0);
  if (e == hipSuccess) e = hipEventCreate(&c->ev1);
  if (e == hipSuccess) e = hipMalloc(&c->small_dev, ctx_small_bytes);
  if (e == hipSuccess) e = hipHostMalloc(&c->small_host, ctx_small_bytes, hipHostMallocDefault);
  if (e != hipSuccess) { rgbid_ctx_destroy(c); return (int)e; }
  *out = c;
  return RGBID_OK;
}

int rgbid_ctx_destroy(rgbid_ctx* c) {
  if (!c) return RGBID_OK;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->partials) hipFree(c->partials);
  if (c->lane_dev) hipFree(c->lane_dev);
  if (c->lane_host) hipHostFree(c->lane_host);
  if (c->lane_ev) hipEventDestroy(c->lane_ev);
  if (c->small_dev) hipFree(c->small_dev);
  if (c->small_host) hipHostFree(c->small_host);
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  if (c->owns_stream && c->stream) hipStreamDestroy(c->stream);
  delete c;
  return RGBID_OK;
}

int rgbid_ctx_set_stream(rgbid_ctx* c, void* stream) {
  if (!c) return RGBID_E_INVALID;
  RGBID_HIP(hipStreamSynchronize(c->stream));
  if (c->owns_stream) { hipStreamDestroy(c->stream); c->owns_stream = false; }
  if (stream) c->stream = (hipStream_t)stream;
  else { RGBID_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->owns_stream = true; }
  return RGBID_OK;
}
int rgbid_ctx_set_async(rgbid_ctx* c, int on) { if (!c) return RGBID_E_INVALID; c->async = on ? 1 : 0; return RGBID_OK; }
int rgbid_ctx_get_async(rgbid_ctx* c, int* on) { if (!c || !on) return RGBID_E_INVALID; *on = c->async; return RGBID_OK; }
int rgbid_ctx_set_interp_mode(rgbid_ctx* c, int mode) {
  if (!c || (mode != RGBID_INTERP_EXACT && mode != RGBID_INTERP_TEX8)) return RGBID_E_INVALID;
  c->interp_mode = mode;
  return RGBID_OK;
}
int rgbid_ctx_get_interp_mode(rgbid_ctx* c, int* mode) { if (!c || !mode) return RGBID_E_INVALID; *mode = c->interp_mode; return RGBID_OK; }
int rgbid_ctx_set_numerics(rgbid_ctx* c, int numerics) {
  if (!c || (numerics != RGBID_NUMERICS_EXACT && numerics != RGBID_NUMERICS_FAST)) return RGBID_E_INVALID;
  c->numerics = numerics;
  return RGBID_OK;
}
int rgbid_ctx_sync(rgbid_ctx* c) { if (!c) return RGBID_E_INVALID; RGBID_HIP(hipStreamSynchronize(c->stream)); return RGBID_OK; }
namespace {
__global__ __launch_bounds__(256) void k_selftest_rcp(unsigned long long* mismatches) {
  const uint32_t hi = blockIdx.x;  // 2^16 workgroups x 2^16 bit patterns
  unsigned int bad = 0;
  for (uint32_t lo = threadIdx.x; lo < 65536u; lo += blockDim.x) {
    const float x = __uint_as_float((hi << 16) | lo);
    const float a = rgbid::rcp_exact(x), b = 1.0f / x;
    const bool same = (a != a && b != b) || (__float_as_uint(a) == __float_as_uint(b));
    bad += same ? 0u : 1u;
  }
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}
}  // namespace
int rgbid_selftest_rcp(rgbid_ctx* c, unsigned long long* mismatches) {
  if (!c || !mismatches) return RGBID_E_INVALID;
  unsigned long long* d = static_cast<unsigned long long*>(c->small_dev);
  RGBID_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(k_selftest_rcp, dim3(65536), dim3(256), 0, c->stream, d);
  RGBID_HIP(hipGetLastError());
  RGBID_HIP(hipMemcpyAsync(mismatches, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  RGBID_HIP(hipStreamSynchronize(c->stream));
  return RGBID_OK;
}
int rgbid_selftest_div_const(rgbid_ctx* c, float divisor, unsigned long long* mismatches, int* used_by_filter) {
  if (!c || !mismatches || !(divisor == divisor) || divisor == 0.f) return RGBID_E_INVALID;
  unsigned long long* d = static_cast<unsigned long long*>(c->small_dev);
  RGBID_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), c->stream));
  rgbid::launch_selftest_div_const(c->stream, divisor, d);
  RGBID_HIP(hipGetLastError());
  RGBID_HIP(hipMemcpyAsync(mismatches, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  RGBID_HIP(hipStreamSynchronize(c->stream));
  if (used_by_filter) *used_by_filter = rgbid::div_const_verified(divisor) ? 1 : 0;
  return RGBID_OK;
}
int rgbid_selftest_cvt_flr(rgbid_ctx* c, unsigned stride, unsigned long long* mismatches) {
  if (!c || !mismatches || stride < 1) return RGBID_E_INVALID;
  unsigned long long* d = static_cast<unsigned long long*>(c->small_dev);
  RGBID_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), c->stream));
  rgbid::launch_selftest_cvt_flr(c->stream, d, stride);
  RGBID_HIP(hipGetLastError());
  RGBID_HIP(hipMemcpyAsync(mismatches, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  RGBID_HIP(hipStreamSynchronize(c->stream));
  return RGBID_OK;
}
namespace {
// v_rcp_f32 within 1 ulp; v_med3_f32 of a NaN first operand returns the smaller bound; v_fract_f32(x) == x - floor(x) in [0, 1)
__global__ __launch_bounds__(256) void k_selftest_fast_primitives(unsigned long long* mismatches) {
  const uint32_t hi = blockIdx.x;
  unsigned int bad = 0;
  for (uint32_t lo = threadIdx.x; lo < 65536u; lo += blockDim.x) {
    const float x = __uint_as_float((hi << 16) | lo);
    const float r = __builtin_amdgcn_rcpf(x);
    const double e = 1.0 / (double)x;
    if (__builtin_amdgcn_classf(x, 0x108) && __builtin_amdgcn_classf((float)e, 0x108)) {   // x and its reciprocal normal numbers (a denormal x reads as 0: inf, which the guard treats as "recompute")
      const float rn = (float)e;                                       // correctly rounded
      const int d = (int)__float_as_uint(r) - (int)__float_as_uint(rn);
      bad += (d < -1 || d > 1) ? 1u : 0u;
    }
    const float m = __builtin_amdgcn_fmed3f(x, rgbid::fastnum::W_LO, rgbid::fastnum::W_HI);
    const bool in = x >= rgbid::fastnum::W_LO && x <= rgbid::fastnum::W_HI;
    bad += ((m == x) != in) ? 1u : 0u;                                 // NaN, out-of-range values: med3 != x
    bad += !(m >= rgbid::fastnum::W_LO && m <= rgbid::fastnum::W_HI) ? 1u : 0u;   // and always a finite value of the domain
    if (fabsf(x) < 8388608.f) {
      const float f = __builtin_amdgcn_fractf(x);
      bad += !(f >= 0.f && f < 1.f && f == fminf(x - floorf(x), 0x1.fffffep-1f)) ? 1u : 0u;   // tiny negative x: x - floor(x) rounds to 1, v_fract stays below
    }
  }
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}
}  // namespace
int rgbid_selftest_fast_primitives(rgbid_ctx* c, unsigned long long* mismatches) {
  if (!c || !mismatches) return RGBID_E_INVALID;
  unsigned long long* d = static_cast<unsigned long long*>(c->small_dev);
  RGBID_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(k_selftest_fast_primitives, dim3(65536), dim3(256), 0, c->stream, d);
  RGBID_HIP(hipGetLastError());
  RGBID_HIP(hipMemcpyAsync(mismatches, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  RGBID_HIP(hipStreamSynchronize(c->stream));
  return RGBID_OK;
}
int rgbid_fast_guard(const float R_proj[9], const float t_proj[3], int cols, int rows, float out[10], int* zsafe) {
  if (!R_proj || !t_proj || !out || cols < 1 || rows < 1) return RGBID_E_INVALID;
  const rgbid::fastnum::Guard g = rgbid::fastnum::make_guard(R_proj, t_proj, cols, rows);
  const float v[10] = {g.d1, g.c2, g.d2, g.q0, g.q1, g.db, g.g0, g.g1, g.e0, g.e1};
  for (int i = 0; i < 10; ++i) out[i] = v[i];
  if (zsafe) *zsafe = g.zsafe;
  return RGBID_OK;
}
int rgbid_fast_guard_lane(const float R_proj[9], const float t_proj[3], int cols, int rows, float out[4]) {
  if (!R_proj || !t_proj || !out || cols < 1 || rows < 1) return RGBID_E_INVALID;
  const rgbid::fastnum::Guard g = rgbid::fastnum::make_guard(R_proj, t_proj, cols, rows);
  out[0] = g.bL; out[1] = g.cL; out[2] = g.kL; out[3] = g.wcore;
  return RGBID_OK;
}
int rgbid_ctx_wait_event(rgbid_ctx* c, void* ev) {
  if (!c || !ev) return RGBID_E_INVALID;
  RGBID_HIP(hipStreamWaitEvent(c->stream, (hipEvent_t)ev, 0));
  return RGBID_OK;
}
int rgbid_ctx_get_stream(rgbid_ctx* c, void** s) { if (!c || !s) return RGBID_E_INVALID; *s = (void*)c->stream; return RGBID_OK; }
int rgbid_mem_info(size_t* f, size_t* t) { if (!f || !t) return RGBID_E_INVALID; RGBID_HIP(hipMemGetInfo(f, t)); return RGBID_OK; }

// ---- memory -----------------------------------------------------------------------------------------
int rgbid_malloc(void** p, size_t bytes) {
  if (!p) return RGBID_E_INVALID;
  *p = nullptr;
  if (bytes == 0) return RGBID_OK;
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory) return RGBID_E_NOMEM;
  return (int)e;
}
int rgbid_malloc_pitch(void** p, size_t* step, size_t width_bytes, size_t rows) {
  if (!p || !step) return RGBID_E_INVALID;
  size_t st = (width_bytes + 255) & ~(size_t)255;  // 256-B rows: every row start is 16-B (float4) aligned
  *step = st;
  return rgbid_malloc(p, st * rows);
}
int rgbid_free(void* p) { if (p) RGBID_HIP(hipFree(p)); return RGBID_OK; }
int rgbid_malloc_host(void** p, size_t bytes) {
  if (!p) return RGBID_E_INVALID;
  *p = nullptr;
  if (bytes == 0) return RGBID_OK;
  hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
  if (e != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return e == hipErrorOutOfMemory ? RGBID_E_NOMEM : (int)e; }
  return RGBID_OK;
}
int rgbid_free_host(void* p) { if (p) RGBID_HIP(hipHostFree(p)); return RGBID_OK; }

static int copy1d(rgbid_ctx* c, void* d, const void* s, size_t n, hipMemcpyKind k) {
  if (!c || (n && (!d || !s))) return RGBID_E_INVALID;
  if (!n) return RGBID_OK;
  RGBID_HIP(hipMemcpyAsync(d, s, n, k, c->stream));
  RGBID_HIP(hipStreamSynchronize(c->stream));  // upload/download are synchronous in the reference containers
  return RGBID_OK;
}
static int copy2d(rgbid_ctx* c, void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k) {
  if (!c || (w && h && (!d || !s))) return RGBID_E_INVALID;
  if (!w || !h) return RGBID_OK;
  RGBID_HIP(hipMemcpy2DAsync(d, dp, s, sp, w, h, k, c->stream));
  RGBID_HIP(hipStreamSynchronize(c->stream));
  return RGBID_OK;
}
int rgbid_memcpy_h2d(rgbid_ctx* c, void* d, const void* s, size_t n) { return copy1d(c, d, s, n, hipMemcpyHostToDevice); }
int rgbid_memcpy_d2h(rgbid_ctx* c, void* d, const void* s, size_t n) { return copy1d(c, d, s, n, hipMemcpyDeviceToHost); }
int rgbid_memcpy_d2d(rgbid_ctx* c, void* d, const void* s, size_t n) { return copy1d(c, d, s, n, hipMemcpyDeviceToDevice); }
int rgbid_memcpy2d_h2d(rgbid_ctx* c, void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h) { return copy2d(c, d, dp, s, sp, w, h, hipMemcpyHostToDevice); }
int rgbid_memcpy2d_d2h(rgbid_ctx* c, void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h) { return copy2d(c, d, dp, s, sp, w, h, hipMemcpyDeviceToHost); }
int rgbid_memcpy2d_d2d(rgbid_ctx* c, void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h) { return copy2d(c, d, dp, s, sp, w, h, hipMemcpyDeviceToDevice); }

}  // extern "C"

// ---- helpers ------------------------------------------------------------------------------------------
namespace {

// rows and the row pitch enter the kernels' 24-bit row-offset multiply (common.h row_ptr): both must stay below 2^24 and the image below 4 GB
inline bool ok_img(const rgbid_img* i) {
  return i && i->data && i->rows > 0 && i->cols > 0 && i->step > 0 && i->rows < (1 << 24) && i->step < ((size_t)1 << 24) &&
         (unsigned long long)i->rows * i->step < (1ull << 32);
}
inline bool same_size(const rgbid_img* a, const rgbid_img* b) { return a->rows == b->rows && a->cols == b->cols; }
inline ImgB B1(const rgbid_img* i) { return ImgB{i->data, i->step, 0, i->rows, i->cols}; }
inline ImgB Bnull() { return ImgB{nullptr, 0, 0, 0, 0}; }
const LaneMask ALL{nullptr, 0};

struct Timed {  // cudaTimer (device.hpp:83-106) with hipEvents on the context's stream
  rgbid_ctx* c;
  float* ms;
  Timed(rgbid_ctx* c_, float* ms_) : c(c_), ms(ms_) { if (ms) hipEventRecord(c->ev0, c->stream); }
  int finish() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (ms) {
      hipEventRecord(c->ev1, c->stream);
      e = hipEventSynchronize(c->ev1);
      if (e != hipSuccess) return (int)e;
      hipEventElapsedTime(ms, c->ev0, c->ev1);
    } else if (!c->async) {
      e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) return (int)e;
    }
    return RGBID_OK;
  }
};

inline WarpParams make_wp(const float R[9], const float t[3]) {
  WarpParams p;
  for (int i = 0; i < 9; ++i) p.R[i] = R[i];
  for (int i = 0; i < 3; ++i) p.t[i] = t[i];
  return p;
}

// blocking read-back of a few bytes from the context's small device scratch through pinned memory
int fetch_small(rgbid_ctx* c, size_t dev_off, size_t bytes) {
  hipError_t e = hipMemcpyAsync((char*)c->small_host + dev_off, (char*)c->small_dev + dev_off, bytes, hipMemcpyDeviceToHost, c->stream);
  if (e != hipSuccess) return (int)e;
  e = hipStreamSynchronize(c->stream);
  return (int)e;
}

}  // namespace

int rgbid::ctx_reserve_partials(rgbid_ctx* c, size_t n_doubles) {
  if (c->partials_cap >= n_doubles) return RGBID_OK;
  hipStreamSynchronize(c->stream);
  if (c->partials) hipFree(c->partials);
  c->partials = nullptr; c->partials_cap = 0;
  hipError_t e = hipMalloc((void**)&c->partials, n_doubles * sizeof(double));
  if (e != hipSuccess) return e == hipErrorOutOfMemory ? RGBID_E_NOMEM : (int)e;
  c->partials_cap = n_doubles;
  return RGBID_OK;
}

int rgbid::ctx_reserve_lane(rgbid_ctx* c, size_t bytes) {
  if (c->lane_cap >= bytes) return RGBID_OK;
  hipStreamSynchronize(c->stream);
  if (c->lane_dev) hipFree(c->lane_dev);
  if (c->lane_host) hipHostFree(c->lane_host);
  c->lane_dev = c->lane_host = nullptr; c->lane_cap = 0;
  bytes = (bytes + 4095) & ~(size_t)4095;
  hipError_t e = hipMalloc(&c->lane_dev, bytes);
  if (e == hipSuccess) e = hipHostMalloc(&c->lane_host, bytes, hipHostMallocDefault);
  if (e == hipSuccess && !c->lane_ev) e = hipEventCreateWithFlags(&c->lane_ev, hipEventDisableTiming);
  c->lane_ev_pending = false;
  if (e != hipSuccess) {
    if (c->lane_dev) { hipFree(c->lane_dev); c->lane_dev = nullptr; }
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? RGBID_E_NOMEM : (int)e;
  }
  c->lane_cap = bytes;
  return RGBID_OK;
}

extern "C" {

// ---- frame preparation ------------------------------------------------------------------------------
int rgbid_depth_to_invdepth(rgbid_ctx* c, const rgbid_img* src, const rgbid_img* dst, float factor_depth) {
  if (!c || !ok_img(src) || !ok_img(dst) || !same_size(src, dst)) return RGBID_E_INVALID;
  Timed t(c, nullptr);
  launch_depth_to_invdepth(c->stream, 1, B1(src), B1(dst), factor_depth, ALL);
  return t.finish();
}
int rgbid_compute_intensity(rgbid_ctx* c, const rgbid_img* rgb, const rgbid_img* dst) {
  if (!c || !ok_img(rgb) || !ok_img(dst) || rgb->rows < dst->rows || rgb->cols < dst->cols) return RGBID_E_INVALID;
  Timed t(c, nullptr);
  launch_intensity(c->stream, 1, B1(rgb), B1(dst), ALL);
  return t.finish();
}
int rgbid_decompose_rgb(rgbid_ctx* c, const rgbid_img* rgb, const rgbid_img* r, const rgbid_img* g, const rgbid_img* b) {
  if (!c || !ok_img(rgb) || !ok_img(r) || !ok_img(g) || !ok_img(b) || !same_size(r, g) || !same_size(r, b) ||
      rgb->rows < r->rows || rgb->cols < r->cols) return RGBID_E_INVALID;
  Timed t(c, nullptr);
  launch_decompose_rgb(c->stream, 1, B1(rgb), B1(r), B1(g), B1(b), ALL);
  return t.finish();
}
int rgbid_compute_gradient(rgbid_ctx* c, const rgbid_img* src, const rgbid_img* gx, const rgbid_img* gy, float* ms) {
  if (!c || !ok_img(src) || !ok_img(gx) || !ok_img(gy) || !same_size(src, gx) || !same_size(src, gy)) return RGBID_E_INVALID;
  Timed t(c, ms);
  launch_gradient(c->stream, 1, B1(src), B1(gx), B1(gy), ALL);
  return t.finish();
}
int rgbid_copy_images(rgbid_ctx* c, const rgbid_img* sd, const rgbid_img* si, const rgbid_img* dd, const rgbid_img* di) {
  if (!c || !ok_img(sd) || !ok_img(si) || !ok_img(dd) || !ok_img(di) || !same_size(sd, dd) || !same_size(sd, si) || !same_size(sd, di)) return RGBID_E_INVALID;
  Timed t(c, nullptr);
  launch_copy_bytes(c->stream, 1, B1(sd), B1(dd), 4, ALL);
  launch_copy_bytes(c->stream, 1, B1(si), B1(di), 4, ALL);
  return t.finish();
}
int rgbid_copy_image(rgbid_ctx* c, const rgbid_img* s, const rgbid_img* d) {
  if (!c || !ok_img(s) || !ok_img(d) || !same_size(s, d)) return RGBID_E_INVALID;
  Timed t(c, nullptr);
  launch_copy_bytes(c->stream, 1, B1(s), B1(d), 4, ALL);
  return t.finish();
}
int rgbid_copy_image_rgb(rgbid_ctx* c, const rgbid_img* s, const rgbid_img* d) {
  if (!c || !ok_img(s) || !ok_img(d) || !same_size(s, d)) return RGBID_E_INVALID;
  Timed t(c, nullptr);
  launch_copy_bytes(c->stream, 1, B1(s), B1(d), 3, ALL);
  return t.finish();
}
int rgbid_init_weight_keyframe(rgbid_ctx* c, const rgbid_img* src_depth, const rgbid_img* dst_weight) {
  if (!c || !ok_img(src_depth) || !ok_img(dst_weight) || !same_size(src_depth, dst_weight)) return RGBID_E_INVALID;
  Timed t(c, nullptr);
  launch_fill(c->stream, 1, B1(dst_weight), 4, 0x3f800000u /* 1.0f: both branches of misc.cu:280-285 assign 1 */, ALL);
  return t.finish();
}
int rgbid_fill_2d(rgbid_ctx* c, const rgbid_img* img, int elem_size, uint32_t bits) {
  if (!c || !ok_img(img) || (elem_size != 1 && elem_size != 4)) return RGBID_E_INVALID;
  Timed t(c, nullptr);
  launch_fill(c->stream, 1, B1(img), elem_size, bits, ALL);
  return t.finish();
}

int rgbid_pyr_down(rgbid_ctx* c, const rgbid_img* src, const rgbid_img* dst, float* ms) {
  if (!c || !ok_img(src) || !ok_img(dst) || dst->rows != src->rows / 2 || dst->cols != src->cols / 2) return RGBID_E_INVALID;
  Timed t(c, ms);
  launch_pyr_down(c->stream, 1, B1(src), B1(dst), ALL);
  return t.finish();
}
int rgbid_bilateral_filter(rgbid_ctx* c, const rgbid_img* src, const rgbid_img* dst, float sigma_floatmap, float* ms) {
  if (!c || !ok_img(src) || !ok_img(dst) || !same_size(src, dst) || src->data == dst->data) return RGBID_E_INVALID;
  Timed t(c, ms);
  launch_bilateral(c->stream, 1, B1(src), B1(dst), sigma_floatmap, ALL, c->numerics == RGBID_NUMERICS_FAST);
  return t.finish();
}

// ---- custom-calibration front-end --------------------------------------------------------------------
static IntrK to_k(const rgbid_intr_k* k) { return IntrK{k->fx, k->fy, k->cx, k->cy, k->k1, k->k2, k->k3, k->k4, k->k5}; }
int rgbid_undistort_intensity(rgbid_ctx* c, const rgbid_img* src, const rgbid_img* dst, const rgbid_intr_k* intr, float* ms) {
  if (!c || !ok_img(src) || !ok_img(dst) || !intr || !same_size(src, dst) || src->data == dst->data) return RGBID_E_INVALID;
  Timed t(c, ms);
  launch_undistort(c->stream, 1, B1(src), B1(dst), to_k(intr), true, c->interp_mode, ALL);
  return t.finish();
}
int rgbid_undistort_depthinv(rgbid_ctx* c, const rgbid_img* src, const rgbid_img* src_corr, const rgbid_img* dst, const rgbid_intr_k* intr,
                             const rgbid_depth_dist* dp, float* ms) {
  if (!c || !ok_img(src) || !ok_img(src_corr) || !ok_img(dst) || !intr || !dp || !same_size(src, dst) || !same_size(src, src_corr) ||
      src->data == src_corr->data || src_corr->data == dst->data) return RGBID_E_INVALID;
  DepthDistP d;
  d.c1 = dp->c1; d.c0 = dp->c0; d.xshift = dp->xshift; d.yshift = dp->yshift;
  for (int i = 0; i < 9; ++i) { d.q0[i] = dp->q0[i]; d.q1[i] = dp->q1[i]; }
  Timed t(c, ms);
  launch_depthinv_correction(c->stream, 1, B1(src), B1(src_corr), to_k(intr), d, ALL);
  launch_undistort(c->stream, 1, B1(src_corr), B1(dst), to_k(intr), false, 0, ALL);
  return t.finish();
}
int rgbid_register_depthinv(rgbid_ctx* c, const rgbid_img* src, const rgbid_img* inter, const rgbid_img* inter_i, const rgbid_img* dst,
                            const float dRc_proj[9], const float t_dc_proj[3], const float cRd_proj[9], float* ms) {
  if (!c || !ok_img(src) || !ok_img(inter) || !ok_img(inter_i) || !ok_img(dst) || !dRc_proj || !t_dc_proj || !cRd_proj || !same_size(inter, inter_i) ||
      inter->rows < src->rows || inter->cols < src->cols || src->data == dst->data) return RGBID_E_INVALID;
  Timed t(c, ms);
  launch_register_depthinv(c->stream, 1, B1(src), B1(inter), B1(inter_i), B1(dst), dRc_proj, t_dc_proj, cRd_proj, ALL);
  return t.finish();
}

// ---- warps / fusion / visibility ----------------------------------------------------------------------
int rgbid_warp_invdepth(rgbid_ctx* c, const rgbid_img* src, const rgbid_img* dst, const rgbid_img* prev, const float R[9], const float tv[3], float* ms) {
  if (!c || !ok_img(src) || !ok_img(dst) || !ok_img(prev) || !R || !tv || !same_size(src, dst) || !same_size(src, prev)) return RGBID_E_INVALID;
  WarpParams p = make_wp(R, tv);
  Timed t(c, ms);
  launch_warp_invdepth(c->stream, 1, B1(src), B1(prev), B1(dst), &p, nullptr, ALL);
  return t.finish();
}
int rgbid_warp_intensity(rgbid_ctx* c, const rgbid_img* src, const rgbid_img* dst, const rgbid_img* prev, const float R[9], const float tv[3], float* ms) {
  if (!c || !ok_img(src) || !ok_img(dst) || !ok_img(prev) || !R || !tv || !same_size(src, dst) || !same_size(src, prev)) return RGBID_E_INVALID;
  WarpParams p = make_wp(R, tv);
  Timed t(c, ms);
  launch_warp_intensity(c->stream, 1, B1(src), B1(prev), B1(dst), &p, nullptr, c->interp_mode, ALL);
  return t.finish();
}
int rgbid_warp_pair(rgbid_ctx* c, const rgbid_img* src_iD, const rgbid_img* src_I, const rgbid_img* grid, const rgbid_img* dst_iD, const rgbid_img* dst_I,
                    const float R[9], const float tv[3], int numerics, float* ms) {
  if (!c || !ok_img(src_iD) || !ok_img(src_I) || !ok_img(grid) || !ok_img(dst_iD) || !ok_img(dst_I) || !R || !tv || !same_size(src_iD, src_I) ||
      !same_size(src_iD, grid) || !same_size(src_iD, dst_iD) || !same_size(src_iD, dst_I) || (numerics != RGBID_NUMERICS_EXACT && numerics != RGBID_NUMERICS_FAST))
    return RGBID_E_INVALID;
  WarpParams p = make_wp(R, tv);
  Timed t(c, ms);
  if (!(numerics == RGBID_NUMERICS_FAST && launch_warp_pair_fast(c->stream, 1, B1(src_iD), B1(src_I), B1(grid), B1(dst_iD), B1(dst_I), &p, nullptr, c->interp_mode, ALL)))
    launch_warp_pair(c->stream, 1, B1(src_iD), B1(src_I), B1(grid), B1(dst_iD), B1(dst_I), nullptr, c->interp_mode, ALL, &p);
  return t.finish();
}
int rgbid_warp_invdepth_weighted(rgbid_ctx* c, const rgbid_img* src, const rgbid_img* dst, const rgbid_img* prev, const rgbid_img* weight,
                                 const float R[9], const float tv[3], float* ms) {
  if (!c || !ok_img(src) || !ok_img(dst) || !ok_img(prev) || !ok_img(weight) || !R || !tv || !same_size(src, dst) ||
      !same_size(src, prev) || !same_size(src, weight)) return RGBID_E_INVALID;
  WarpParams p = make_wp(R, tv);
  Timed t(c, ms);
  launch_warp_invdepth_weighted(c->stream, 1, B1(src), B1(prev), B1(dst), B1(weight), &p, nullptr, ALL);
  return t.finish();
}
int rgbid_integrate_warped_frame(rgbid_ctx* c, const rgbid_img* wd, const rgbid_img* ww, const rgbid_img* dd, const rgbid_img* dw, float* ms) {
  if (!c || !ok_img(wd) || !ok_img(ww) || !ok_img(dd) || !ok_img(dw) || !same_size(wd, ww) || !same_size(wd, dd) || !same_size(wd, dw)) return RGBID_E_INVALID;
  Timed t(c, ms);
  launch_integrate_warped(c->stream, 1, B1(wd), B1(ww), B1(dd), B1(dw), ALL);
  return t.finish();
}
int rgbid_visibility_ratio(rgbid_ctx* c, const rgbid_img* src, const rgbid_img* dst, const float R[9], const float tv[3],
                           const rgbid_img* mask, float* ratio, float* ms) {
  if (!c || !ok_img(src) || !ok_img(dst) || !R || !tv || !ratio || !same_size(src, dst) || (mask && (!ok_img(mask) || !same_size(src, mask)))) return RGBID_E_INVALID;
  WarpParams p = make_wp(R, tv);
  unsigned int* counts = (unsigned int*)((char*)c->small_dev + ctx_off_counts);
  if (ms) hipEventRecord(c->ev0, c->stream);
  RGBID_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(unsigned int), c->stream));
  launch_visibility(c->stream, 1, B1(src), B1(dst), mask ? B1(mask) : Bnull(), &p, nullptr, counts, ALL);
  RGBID_HIP(hipGetLastError());
  if (ms) hipEventRecord(c->ev1, c->stream);
  int e = fetch_small(c, ctx_off_counts, 2 * sizeof(unsigned int));
  if (e) return e;
  if (ms) hipEventElapsedTime(ms, c->ev0, c->ev1);
  const unsigned int* h = (const unsigned int*)((char*)c->small_host + ctx_off_counts);
  float visible = (float)h[0], valid = (float)h[1];
  *ratio = (valid < 1.f) ? 0.f : visible / valid;  // warping_registration.cu:862-865
  return RGBID_OK;
}

// ---- maps ---------------------------------------------------------------------------------------------
int rgbid_create_vmap(rgbid_ctx* c, rgbid_intr k, const rgbid_img* depthinv, const rgbid_img* vmap) {
  if (!c || !ok_img(depthinv) || !ok_img(vmap) || vmap->rows != 3 * depthinv->rows || vmap->cols != depthinv->cols) return RGBID_E_INVALID;
  Timed t(c, nullptr);
  launch_vmap(c->stream, 1, B1(depthinv), B1(vmap), IntrP{k.fx, k.fy, k.cx, k.cy}, ALL);
  return t.finish();
}
int rgbid_depth_to_float(rgbid_ctx* c, const rgbid_img* src, const rgbid_img* dst) {
  if (!c || !ok_img(src) || !ok_img(dst) || !same_size(src, dst)) return RGBID_E_INVALID;
  Timed t(c, nullptr);
  launch_depth_to_float(c->stream, 1, B1(src), B1(dst), ALL);
  return t.finish();
}
int rgbid_float_to_rgb(rgbid_ctx* c, const rgbid_img* src, const rgbid_img* dst) {
  if (!c || !ok_img(src) || !ok_img(dst) || !same_size(src, dst)) return RGBID_E_INVALID;
  Timed t(c, nullptr);
  launch_float_to_rgb(c->stream, 1, B1(src), B1(dst), ALL);
  return t.finish();
}
int rgbid_create_nmap(rgbid_ctx* c, const rgbid_img* vmap, const rgbid_img* nmap) {
  if (!c || !ok_img(vmap) || !ok_img(nmap) || !same_size(vmap, nmap) || vmap->rows % 3 != 0 || vmap->data == nmap->data) return RGBID_E_INVALID;
  Timed t(c, nullptr);
  launch_nmap_cross(c->stream, 1, B1(vmap), B1(nmap), ALL);
  return t.finish();
}
int rgbid_integrate_warped_rgb(rgbid_ctx* c, const rgbid_img* warped, const rgbid_img* r, const rgbid_img* g, const rgbid_img* b, const rgbid_img* wweight,
                               const rgbid_img* kf, const rgbid_img* colors, const rgbid_img* kfw, float* ms) {
  if (!c || !ok_img(warped) || !ok_img(r) || !ok_img(g) || !ok_img(b) || !ok_img(wweight) || !ok_img(kf) || !ok_img(colors) || !ok_img(kfw) ||
      !same_size(warped, r) || !same_size(warped, g) || !same_size(warped, b) || !same_size(warped, wweight) || !same_size(warped, kf) ||
      !same_size(warped, colors) || !same_size(warped, kfw)) return RGBID_E_INVALID;
  Timed t(c, ms);
  launch_integrate_warped_rgb(c->stream, 1, B1(warped), B1(r), B1(g), B1(b), B1(wweight), B1(kf), B1(colors), B1(kfw), ALL);
  return t.finish();
}
int rgbid_create_nmap_gradients(rgbid_ctx* c, rgbid_intr k, const rgbid_img* depthinv, const rgbid_img* gx, const rgbid_img* gy, const rgbid_img* nmap) {
  if (!c || !ok_img(depthinv) || !ok_img(gx) || !ok_img(gy) || !ok_img(nmap) || !same_size(depthinv, gx) || !same_size(depthinv, gy) ||
      nmap->rows != 3 * depthinv->rows || nmap->cols != depthinv->cols) return RGBID_E_INVALID;
  Timed t(c, nullptr);
  launch_nmap_gradients(c->stream, 1, B1(depthinv), B1(gx), B1(gy), B1(nmap), IntrP{k.fx, k.fy, k.cx, k.cy}, ALL);
  return t.finish();
}
int rgbid_generate_image(rgbid_ctx* c, const rgbid_img* vmap, const rgbid_img* nmap, const rgbid_img* rgb, const float light[3], const rgbid_img* dst) {
  if (!c || !ok_img(vmap) || !ok_img(nmap) || !ok_img(dst) || !light || vmap->rows != 3 * dst->rows || nmap->rows != 3 * dst->rows ||
      (rgb && !ok_img(rgb))) return RGBID_E_INVALID;
  LightP L{light[0], light[1], light[2]};
  Timed t(c, nullptr);
  launch_generate_image(c->stream, 1, B1(vmap), B1(nmap), rgb ? B1(rgb) : Bnull(), B1(dst), &L, nullptr, ALL);
  return t.finish();
}

// ---- residual lattice + scale estimation -----------------------------------------------------------------
int rgbid_error_lattice_size(int rows, int cols, int min_nsamples, int* n, int* lr, int* lc, int* stride) {
  if (rows <= 0 || cols <= 0 || !n) return RGBID_E_INVALID;
  int a, b, d;
  lattice_geometry(rows, cols, min_nsamples, n, &a, &b, &d);
  if (lr) *lr = a;
  if (lc) *lc = b;
  if (stride) *stride = d;
  return RGBID_OK;
}
int rgbid_compute_error(rgbid_ctx* c, const rgbid_img* im1, const rgbid_img* im0, float* err, int min_nsamples, int* n_samples, float* ms) {
  if (!c || !ok_img(im1) || !ok_img(im0) || !err || !same_size(im1, im0)) return RGBID_E_INVALID;
  int n, lr, lc, st;
  lattice_geometry(im0->rows, im0->cols, min_nsamples, &n, &lr, &lc, &st);
  if (n_samples) *n_samples = n;
  Timed t(c, ms);
  launch_error_lattice(c->stream, 1, B1(im1), B1(im0), err, 0, lr, lc, st, ALL);
  return t.finish();
}

static int run_sigma(rgbid_ctx* c, int mode, const float* err, int n, float* bias, float* sigma, float* nu, int mest, float* ms) {
  if (!c || !err || n <= 0) return RGBID_E_INVALID;
  SigmaIO* io_d = (SigmaIO*)((char*)c->small_dev + ctx_off_sigma);
  SigmaIO* io_h = (SigmaIO*)((char*)c->small_host + ctx_off_sigma);
  io_h->bias = *bias; io_h->sigma = *sigma; io_h->nu = *nu;
  if (ms) hipEventRecord(c->ev0, c->stream);
  RGBID_HIP(hipMemcpyAsync(io_d, io_h, sizeof(SigmaIO), hipMemcpyHostToDevice, c->stream));
  launch_sigma(c->stream, 1, mode, err, 0, n, io_d, mest, ALL);
  RGBID_HIP(hipGetLastError());
  if (ms) hipEventRecord(c->ev1, c->stream);
  int e = fetch_small(c, ctx_off_sigma, sizeof(SigmaIO));
  if (e) return e;
  if (ms) hipEventElapsedTime(ms, c->ev0, c->ev1);
  *bias = io_h->bias; *sigma = io_h->sigma; *nu = io_h->nu;
  return RGBID_OK;
}
int rgbid_sigma_nu_student(rgbid_ctx* c, const float* err, int n, float* bias, float* sigma, float* nu, int mest, float* ms) {
  if (!bias || !sigma || !nu) return RGBID_E_INVALID;
  return run_sigma(c, 0, err, n, bias, sigma, nu, mest, ms);
}
int rgbid_nu_student(rgbid_ctx* c, const float* err, int n, float bias, float sigma, float* nu, float* ms) {
  if (!nu) return RGBID_E_INVALID;
  return run_sigma(c, 1, err, n, &bias, &sigma, nu, RGBID_STUDENT, ms);
}
int rgbid_sigma_pdf(rgbid_ctx* c, const float* err, int n, float* bias, float* sigma, int mest, float* ms) {
  if (!bias || !sigma) return RGBID_E_INVALID;
  float nu = 5.f;
  return run_sigma(c, 2, err, n, bias, sigma, &nu, mest, ms);
}
int rgbid_chi_square(rgbid_ctx* c, const float* ei, const float* ed, int n, float sigma_int, float sigma_depth, int mest,
                     float* chi_square, float* chi_test, float* ndof, float* ms) {
  if (!c || !ei || !ed || n <= 0 || !chi_square || !chi_test || !ndof) return RGBID_E_INVALID;
  float* out_d = (float*)((char*)c->small_dev + ctx_off_chi);
  if (ms) hipEventRecord(c->ev0, c->stream);
  launch_chi_square(c->stream, 1, ei, ed, 0, n, sigma_int, sigma_depth, mest, out_d, ALL);
  RGBID_HIP(hipGetLastError());
  if (ms) hipEventRecord(c->ev1, c->stream);
  int e = fetch_small(c, ctx_off_chi, 3 * sizeof(float));
  if (e) return e;
  if (ms) hipEventElapsedTime(ms, c->ev0, c->ev1);
  const float* h = (const float*)((char*)c->small_host + ctx_off_chi);
  *chi_square = h[0]; *chi_test = h[1]; *ndof = h[2];
  return RGBID_OK;
}

// ---- normal equations ---------------------------------------------------------------------------------------
static int run_system(rgbid_ctx* c, const rgbid_img* W0, const rgbid_img* I0, const rgbid_img* gWx, const rgbid_img* gWy,
                      const rgbid_img* gIx, const rgbid_img* gIy, const rgbid_img* W1, const rgbid_img* I1, const SysParams& P,
                      double A[36], double b[6], float* ms) {
  const rgbid_img* all[8] = {W0, I0, gWx, gWy, gIx, gIy, W1, I1};
  if (!c || !A || !b) return RGBID_E_INVALID;
  for (int i = 0; i < 8; ++i) if (!ok_img(all[i]) || !same_size(all[i], W0)) return RGBID_E_INVALID;
  int nb = system_blocks_per_lane(W0->rows, W0->cols, 1);
  int e = ctx_reserve_partials(c, (size_t)nb * SYS_TERMS);
  if (e) return e;
  double* sums_d = (double*)((char*)c->small_dev + ctx_off_sums);
  if (ms) hipEventRecord(c->ev0, c->stream);
  int nblk = launch_build_system(c->stream, 1, B1(W0), B1(I0), B1(gWx), B1(gWy), B1(gIx), B1(gIy), B1(W1), B1(I1), &P, nullptr, c->partials, ALL);
  launch_reduce_system(c->stream, 1, c->partials, nblk, sums_d, ALL);
  RGBID_HIP(hipGetLastError());
  if (ms) hipEventRecord(c->ev1, c->stream);
  e = fetch_small(c, ctx_off_sums, SYS_TERMS * sizeof(double));
  if (e) return e;
  if (ms) hipEventElapsedTime(ms, c->ev0, c->ev1);
  const double* host_data = (const double*)((char*)c->small_host + ctx_off_sums);
  int shift = 0;  // estimate_VO.cu:774-786
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) {
      double value = host_data[shift++];
      if (j == 6) b[i] = value;
      else A[j * 6 + i] = A[i * 6 + j] = value;
    }
  return RGBID_OK;
}

int rgbid_build_system(rgbid_ctx* c, const rgbid_img* W0, const rgbid_img* I0, const rgbid_img* gWx, const rgbid_img* gWy,
                       const rgbid_img* gIx, const rgbid_img* gIy, const rgbid_img* W1, const rgbid_img* I1, int mest, int weighting,
                       float sigma_d, float sigma_i, float bias_d, float bias_i, rgbid_intr k, double A[36], double b[6], float* ms) {
  SysParams P{k.fx, k.fy, k.cx, k.cy, sigma_d, sigma_i, bias_d, bias_i, 5.f, 5.f, mest, weighting, 0, 0};
  return run_system(c, W0, I0, gWx, gWy, gIx, gIy, W1, I1, P, A, b, ms);
}
int rgbid_build_system_student_nu(rgbid_ctx* c, const rgbid_img* W0, const rgbid_img* I0, const rgbid_img* gWx, const rgbid_img* gWy,
                                  const rgbid_img* gIx, const rgbid_img* gIy, const rgbid_img* W1, const rgbid_img* I1, int mest, int weighting,
                                  float sigma_d, float sigma_i, float bias_d, float bias_i, float nu_d, float nu_i, rgbid_intr k,
                                  double A[36], double b[6], float* ms) {
  SysParams P{k.fx, k.fy, k.cx, k.cy, sigma_d, sigma_i, bias_d, bias_i, nu_d, nu_i, mest, weighting, 1, 0};
  return run_system(c, W0, I0, gWx, gWy, gIx, gIy, W1, I1, P, A, b, ms);
}

}  // extern "C"
