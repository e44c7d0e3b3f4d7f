// which of the three primitive facts of rgbid_selftest_fast_primitives fails, and where
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
__global__ void k(unsigned long long* bad, float* ex) {
  const uint32_t hi = blockIdx.x;
  for (uint32_t lo = threadIdx.x; lo < 65536u; lo += blockDim.x) {
    const float x = __uint_as_float((hi << 16) | lo);
    const float r = __builtin_amdgcn_rcpf(x);
    const double e = 1.0 / (double)x;
    if (x == x && __builtin_amdgcn_classf((float)e, 0x108)) {
      const float rn = (float)e;
      const int d = (int)__float_as_uint(r) - (int)__float_as_uint(rn);
      if (d < -1 || d > 1) { if (atomicAdd(&bad[0], 1ull) == 0) { ex[0] = x; ex[1] = r; ex[2] = rn; } }
    }
    const float lo_ = 0x1p-14f, hi_ = 0x1p14f;
    const float m = __builtin_amdgcn_fmed3f(x, lo_, hi_);
    const bool in = x >= lo_ && x <= hi_;
    if ((m == x) != in) { if (atomicAdd(&bad[1], 1ull) == 0) { ex[3] = x; ex[4] = m; } }
    if (!(m >= lo_ && m <= hi_)) { if (atomicAdd(&bad[2], 1ull) == 0) { ex[5] = x; ex[6] = m; } }
    if (fabsf(x) < 8388608.f) {
      const float f = __builtin_amdgcn_fractf(x);
      if (!(f >= 0.f && f < 1.f && f == x - floorf(x))) { if (atomicAdd(&bad[3], 1ull) == 0) { ex[7] = x; ex[8] = f; ex[9] = x - floorf(x); } }
    }
  }
}
int main() {
  unsigned long long* b; float* ex; hipMalloc(&b, 32); hipMalloc(&ex, 64); hipMemset(b, 0, 32); hipMemset(ex, 0, 64);
  hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, b, ex);
  unsigned long long hb[4]; float he[16]; hipMemcpy(hb, b, 32, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 64, hipMemcpyDeviceToHost);
  printf("rcp>1ulp %llu (x %a r %a rn %a)\nmed3 eq-mismatch %llu (x %a m %a)\nmed3 range %llu (x %a m %a)\nfract %llu (x %a f %a x-floor %a)\n", hb[0], he[0], he[1], he[2], hb[1], he[3], he[4], hb[2], he[5], he[6], hb[3], he[7], he[8], he[9]);
}
