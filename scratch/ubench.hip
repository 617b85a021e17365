// micro-benchmark: how do VALU ops co-issue with the fp32 MFMA on one SIMD? (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND, int NM>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc = {0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float x[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = i + a; u[i] = i * 77 + threadIdx.x; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m)
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[v & 7]) : "v"(b));
      if (KIND == 1) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[v & 7]) : "v"(a));
      if (KIND == 2) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u[v & 7]) : "v"(a), "v"(b));
      if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&x[(v & 3) * 2]) : "v"(*(double*)&x[0]));
      if (KIND == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x[v & 7]), "v"(b) : "vcc");
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int i = 0; i < 8; ++i) s += x[i] + u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int KIND, int NM>
void run(const char* name, int blocks_per_cu) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * blocks_per_cu), block(256);
  hipLaunchKernelGGL((k<NV, KIND, NM>), grid, block, 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, KIND, NM>), grid, block, 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-10s NM=%d NV=%2d waves/SIMD=%d: %.1f ns/iter  (%.0f cyc @2.4GHz)\n", name, NM, NV, blocks_per_cu,
         ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0, 0, 1>("mfma only", w);
    run<8, 0, 1>("add", w); run<12, 0, 1>("add", w); run<16, 0, 1>("add", w); run<24, 0, 1>("add", w);
    run<12, 1, 1>("alignbit", w); run<12, 2, 1>("min3_u32", w); run<6, 3, 1>("pk_add", w); run<12, 4, 1>("cmp", w);
    run<16, 0, 0>("add only", w); run<16, 1, 0>("alignb only", w); run<8, 3, 0>("pk only", w);
  }
  return 0;
}
