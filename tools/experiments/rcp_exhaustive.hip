// Exhaustive check over all 2^32 float bit patterns: which short reciprocal sequences equal the IEEE 1.0f/x the compiler emits
// (v_div_scale x2, v_rcp, 5 fma, v_div_fmas, v_div_fixup)?   hipcc --offload-arch=gfx950 -O3 rcp_exhaustive.hip -o rcp_exhaustive
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

#pragma clang fp contract(off)

__device__ __forceinline__ float cand_a(float x) {  // rcp + 1 Newton step
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float cand_b(float x) {  // rcp + 2 Newton steps
  float r = cand_a(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float cand_c(float x) { return __builtin_amdgcn_div_fixupf(cand_a(x), x, 1.0f); }
__device__ __forceinline__ float cand_d(float x) { return __builtin_amdgcn_div_fixupf(cand_b(x), x, 1.0f); }

__device__ __forceinline__ bool same(float a, float b) {
  if (a != a && b != b) return true;
  return __float_as_uint(a) == __float_as_uint(b);
}

// fail[c][e]: failures of candidate c for inputs with biased exponent e
__global__ void k_check(unsigned long long* fail, uint32_t* first) {
  uint32_t hi = blockIdx.x;  // 2^16 blocks x 2^16 patterns
  for (uint32_t lo = threadIdx.x; lo < 65536u; lo += blockDim.x) {
    uint32_t bits = (hi << 16) | lo;
    float x = __uint_as_float(bits);
    float ref = 1.0f / x;
    float c[4] = {cand_a(x), cand_b(x), cand_c(x), cand_d(x)};
    int ex = (bits >> 23) & 255;
    for (int k = 0; k < 4; ++k)
      if (!same(c[k], ref)) {
        unsigned long long old = atomicAdd(&fail[k * 256 + ex], 1ull);
        if (old == 0) first[k * 256 + ex] = bits;
      }
  }
}

int main() {
  unsigned long long* d_fail; uint32_t* d_first;
  hipMalloc(&d_fail, 4 * 256 * 8); hipMalloc(&d_first, 4 * 256 * 4);
  hipMemset(d_fail, 0, 4 * 256 * 8); hipMemset(d_first, 0, 4 * 256 * 4);
  hipLaunchKernelGGL(k_check, dim3(65536), dim3(256), 0, 0, d_fail, d_first);
  hipDeviceSynchronize();
  static unsigned long long fail[4 * 256]; static uint32_t first[4 * 256];
  hipMemcpy(fail, d_fail, sizeof(fail), hipMemcpyDeviceToHost); hipMemcpy(first, d_first, sizeof(first), hipMemcpyDeviceToHost);
  const char* names[4] = {"A rcp+1NR", "B rcp+2NR", "C rcp+1NR+fixup", "D rcp+2NR+fixup"};
  for (int k = 0; k < 4; ++k) {
    unsigned long long tot = 0, normal = 0;
    for (int e = 0; e < 256; ++e) { tot += fail[k * 256 + e]; if (e >= 2 && e <= 252) normal += fail[k * 256 + e]; }
    printf("%-18s mismatches: total %llu, biased exponent in [2,252]: %llu\n", names[k], tot, normal);
    int shown = 0;
    for (int e = 0; e < 256 && shown < 12; ++e)
      if (fail[k * 256 + e]) { float x; memcpy(&x, &first[k * 256 + e], 4); printf("    exp %3d: %llu failures, e.g. 0x%08x (%g)\n", e, fail[k * 256 + e], first[k * 256 + e], x); ++shown; }
  }
  return 0;
}
