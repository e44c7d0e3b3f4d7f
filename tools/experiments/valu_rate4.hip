// VALU issue-rate probe 4: selects.  (probe 3 with the select / mask forms)
// VALU issue-rate probe 3 (gfx950): the non-FMA instructions of the fused Gauss-Newton loop, one opcode at a time (inline asm, 16 independent
// chains per thread): which of them issue at the FMA rate and which at a quarter.   hipcc --offload-arch=gfx950 -O3 valu_rate3.hip -o valu_rate3
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP1(name, str)                                                                                           \
  struct name { static __device__ __forceinline__ void op(float& a, float b, float c) { asm volatile(str : "+v"(a) : "v"(b), "v"(c)); } \
                static const char* nm() { return #name; } };
OP1(fma_ref, "v_fma_f32 %0, %0, %1, %2")
OP1(mul, "v_mul_f32 %0, %0, %1")
OP1(add, "v_add_f32 %0, %0, %1")
OP1(max_, "v_max_f32 %0, %0, %1")
OP1(min_, "v_min_f32 %0, %0, %1")
OP1(med3, "v_med3_f32 %0, %0, %1, %2")
OP1(max3, "v_max3_f32 %0, %0, %1, %2")
OP1(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
OP1(cmp_lt, "v_cmp_lt_f32 vcc, %0, %1")
OP1(cmp_o, "v_cmp_o_f32 vcc, %0, %1")
OP1(cmp_class, "v_cmp_class_f32 vcc, %0, %1")
OP1(fract, "v_fract_f32 %0, %0")
OP1(rndne, "v_rndne_f32 %0, %0")
OP1(floor_, "v_floor_f32 %0, %0")
OP1(cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
OP1(cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
OP1(lshl, "v_lshlrev_b32 %0, 1, %0")
OP1(add_u32, "v_add_u32 %0, %0, %1")
OP1(and_b32, "v_and_b32 %0, %0, %1")
OP1(mov, "v_mov_b32 %0, %1")
OP1(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
OP1(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
OP1(add_lshl, "v_add_lshl_u32 %0, %0, %1, 2")
OP1(lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
OP1(bfe, "v_bfe_u32 %0, %0, 3, 5")
OP1(rsq, "v_rsq_f32 %0, %0")
OP1(rcp, "v_rcp_f32 %0, %0")
OP1(exp_, "v_exp_f32 %0, %0")
OP1(mul_e64_abs, "v_mul_f32 %0, |%0|, %1")
OP1(sub_neg, "v_fma_f32 %0, -%0, %1, %2")
OP1(fmac_dpp, "v_fmac_f32 %0, %1, %2")
OP1(bfi, "v_bfi_b32 %0, %1, %0, %2")
OP1(ashr, "v_ashrrev_i32 %0, 31, %0")
OP1(sub_u32, "v_sub_u32 %0, %1, %0")
OP1(and_or, "v_and_or_b32 %0, %0, %1, %2")
struct cnd_sgpr { static __device__ __forceinline__ void op(float& a, float b, float c) { unsigned long long m = __builtin_amdgcn_read_exec() >> 1; asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(m)); }
                  static const char* nm() { return "cndmask_e64_sgpr"; } };
struct cnd_vcc { static __device__ __forceinline__ void op(float& a, float b, float c) { asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc"); }
                 static const char* nm() { return "cmp+cndmask_vcc"; } };
struct cnd_pair { static __device__ __forceinline__ void op(float& a, float b, float c) { unsigned long long m; asm volatile("v_cmp_lt_f32 %1, %0, %2\n v_cndmask_b32_e64 %0, %0, %3, %1" : "+v"(a), "=&s"(m) : "v"(b), "v"(c)); }
                  static const char* nm() { return "cmp+cndmask_e64"; } };
struct cnd_c { static __device__ __forceinline__ void op(float& a, float b, float c) { a = a > b ? a : c; }
               static const char* nm() { return "C: a>b?a:c"; } };
struct cnd_zero { static __device__ __forceinline__ void op(float& a, float b, float c) { asm volatile("v_cndmask_b32_e64 %0, 0, %0, vcc" : "+v"(a)); }
               static const char* nm() { return "cndmask 0,v,vcc"; } };
template <class O>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  float b = seed * 1.0000001f, c = seed * 1e-7f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) O::op(a[i], b, c);
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class O> void run(double ref_ms[2]) {
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int j = 0;
  printf("%-14s", O::nm());
  for (int wg_per_cu : {2, 8}) {
    int blocks = 256 * wg_per_cu, iters = 20000;
    k<O><<<blocks, 256>>>(d, 100, 1.f);
    hipEventRecord(e0); k<O><<<blocks, 256>>>(d, iters, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ref_ms[j] == 0) ref_ms[j] = ms;
    printf("  waves/SIMD %d: %8.3f ms = %.2f x v_fma_f32", wg_per_cu, ms, ms / ref_ms[j]);
    ++j;
  }
  printf("\n");
  hipFree(d);
}
int main() {
  double ref[2] = {0, 0};
  run<fma_ref>(ref); run<cnd_sgpr>(ref); run<cnd_vcc>(ref); run<cnd_pair>(ref); run<cnd_c>(ref); run<cnd_zero>(ref); run<bfi>(ref); run<ashr>(ref); run<sub_u32>(ref); run<and_or>(ref); return 0;
  run<mul>(ref); run<add>(ref); run<fmac_dpp>(ref); run<sub_neg>(ref); run<mul_e64_abs>(ref);
  run<max_>(ref); run<min_>(ref); run<med3>(ref); run<max3>(ref);
  run<cndmask>(ref); run<cmp_lt>(ref); run<cmp_o>(ref); run<cmp_class>(ref);
  run<fract>(ref); run<rndne>(ref); run<floor_>(ref); run<cvt_f32_i32>(ref); run<cvt_i32_f32>(ref);
  run<lshl>(ref); run<add_u32>(ref); run<and_b32>(ref); run<mov>(ref); run<mad_u32_u24>(ref); run<mul_u32_u24>(ref); run<add_lshl>(ref); run<lshl_add>(ref); run<bfe>(ref);
  run<rsq>(ref); run<rcp>(ref); run<exp_>(ref);
  return 0;
}
