// VALU issue-rate probe 2 (gfx950): v_fma_f32 / v_pk_fma_f32 with THREE VGPR operands (the accumulate pattern acc = fma(a, b, acc)),
// v_mul_f32, v_add_f32 with two VGPR operands.  hipcc --offload-arch=gfx950 -O3 valu_rate2.hip -o valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* in) {
  float a[12], b[12], c[12];
  v2f pa[6], pb[6], pc[6];
#pragma unroll
  for (int i = 0; i < 12; ++i) { a[i] = in[threadIdx.x + i]; b[i] = in[threadIdx.x + 64 + i]; c[i] = in[threadIdx.x + 128 + i]; }
#pragma unroll
  for (int i = 0; i < 6; ++i) { pa[i] = v2f{a[2 * i], a[2 * i + 1]}; pb[i] = v2f{b[2 * i], b[2 * i + 1]}; pc[i] = v2f{c[2 * i], c[2 * i + 1]}; }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 12; ++i) a[i] = __builtin_fmaf(b[i], c[(i + 1) % 12], a[i]);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 6; ++i) pa[i] = __builtin_elementwise_fma(pb[i], pc[(i + 1) % 6], pa[i]);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 12; ++i) a[i] = a[i] * b[i];
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 12; ++i) a[i] = a[i] + b[i];
    } else if (MODE == 4) {   // broadcast operand: pk_fma with one scalar-in-VGPR operand splatted (op_sel) -- acc2 += J2 * w
#pragma unroll
      for (int i = 0; i < 6; ++i) pa[i] = __builtin_elementwise_fma(pb[i], v2f{c[i], c[i]}, pa[i]);
    }
  }
  float s = 0;
  if (MODE == 1 || MODE == 4) { for (int i = 0; i < 6; ++i) s += pa[i].x + pa[i].y; } else { for (int i = 0; i < 12; ++i) s += a[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int per_iter_instr, int flops_per_instr) {
  float *d, *in; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float)); (void)hipMalloc(&in, 1024 * sizeof(float)); (void)hipMemset(in, 0, 1024 * sizeof(float));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wg_per_cu : {2, 4, 8}) {
    int blocks = 256 * wg_per_cu, iters = 20000;
    k<MODE><<<blocks, 256>>>(d, 100, in);
    (void)hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters, in); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * iters * per_iter_instr;
    double per_simd_per_s = wave_instr / 1024 / (ms * 1e-3);
    printf("%-22s waves/SIMD %d: %.3f ms -> %.2f cycles/instr at 2.4 GHz, %.1f TFLOP/s\n", name, wg_per_cu, ms, 2.4e9 / per_simd_per_s,
           wave_instr * 64 * flops_per_instr / (ms * 1e-3) / 1e12);
  }
}
int main() {
  run<0>("v_fma_f32 3xVGPR", 12, 2);
  run<1>("v_pk_fma_f32 3xVGPR", 6, 4);
  run<4>("v_pk_fma_f32 splat", 6, 4);
  run<2>("v_mul_f32 2xVGPR", 12, 1);
  run<3>("v_add_f32 2xVGPR", 12, 1);
  return 0;
}
