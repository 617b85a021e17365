// microbenchmark: do v_mfma_f32_32x32x16_f16 and plain VALU work overlap on one SIMD (gfx950)?
// times three loops per occupancy: MFMA only, VALU only, both interleaved (NV VALU ops per MFMA)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NV, int DEP>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  const int lane = threadIdx.x;
  h16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE & 1) {
        if (DEP) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
        else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
      }
      if (MODE & 2) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          float& x = v[q & 7];
          asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(v[(q + 1) & 7]), "v"(v[(q + 3) & 7]));
        }
      }
    }
  }
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV, int DEP>
float run(float* d, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NV, DEP>), dim3(blocks), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV, DEP>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

template <int NV, int DEP>
void suite(float* d, int iters) {
  for (int occ = 1; occ <= 2; ++occ) {
    const int blocks = 256 * occ;   // 256 CUs x 4 waves: occ waves per SIMD
    const float m = run<1, NV, DEP>(d, blocks, iters), v = run<2, NV, DEP>(d, blocks, iters), both = run<3, NV, DEP>(d, blocks, iters);
    // per wave per MFMA slot, in ns
    const double per = 1e6 / (4.0 * iters);
    printf("NV=%2d dep=%d waves/SIMD=%d: mfma %.2f ns  valu %.2f ns  both %.2f ns   (sum %.2f, max %.2f) per MFMA slot per wave\n", NV, DEP, occ,
           m * per, v * per, both * per, (m + v) * per, (m > v ? m : v) * per);
  }
}

int main() {
  float* d; hipMalloc(&d, sizeof(float) * 256 * 1024);
  const int iters = 20000;
  suite<4, 0>(d, iters);
  suite<8, 0>(d, iters);
  suite<12, 0>(d, iters);
  suite<8, 1>(d, iters);
  return 0;
}
