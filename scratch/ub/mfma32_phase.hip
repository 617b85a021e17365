// microbenchmark (round 6): does VALU work of ONE wave overlap with fp32-input MFMAs of ANOTHER wave of the same SIMD?
// 512-thread workgroups (two waves per SIMD: wave w and w + 4).  Every wave alternates a clump of 20 MFMAs (four chains
// of five dependent v_mfma_f32_32x32x2_f32) with a clump of NV VALU instructions per chain.
//   mode 0: both waves of a SIMD in phase (a barrier per clump keeps them there)
//   mode 1: anti-phase -- waves 4..7 run their VALU clump while waves 0..3 run their MFMA clump (barrier per clump)
//   mode 2: free-running, all waves start together        mode 3: free-running, waves 4..7 start with the VALU clump
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mfma_clump(const float (&a)[5], const float (&b)[4][5], const f32x16& c0, f32x16 (&acc)[4]) {
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[q][s], s == 0 ? c0 : acc[q], 0, 0, 0);
}
template <int MIX>
__device__ __forceinline__ void valu_clump(const f32x16 (&old)[4], f32x2 nlo, uint32_t& cnt, float& fm) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if constexpr (MIX == 0) {   // population epilogue: 8 v_pk_add_f32 + 16 v_alignbit + count
      f32x2 t[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) { const f32x2 v = {old[q][r], old[q][r + 1]}; asm("v_pk_add_f32 %0, %1, %2" : "=v"(t[r / 2]) : "v"(v), "v"(nlo)); }
      uint32_t bits = 0;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t[r / 2].x), 30);
        bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t[r / 2].y), 30);
      }
      cnt += __builtin_popcount(bits & 0xAAAAAAAAu);
    } else {                    // neighbour epilogue: 8 v_min3_f32
#pragma unroll
      for (int r = 0; r < 16; r += 2) fm = fminf(fm, fminf(old[q][r], old[q][r + 1]));
    }
  }
}

template <int MODE, int MIX>
__global__ __launch_bounds__(512, 1) void k(float* out, const float* in, int iters) {
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 8;   // grp 1: waves 4..7
  float a[5], b[4][5];
  for (int s = 0; s < 5; ++s) { a[s] = in[lane + 64 * s]; for (int q = 0; q < 4; ++q) b[q][s] = in[lane + 64 * (s + 5 + q)]; }
  f32x16 c0, acc[4], old[4];
  for (int i = 0; i < 16; ++i) { c0[i] = in[i]; for (int q = 0; q < 4; ++q) { acc[q][i] = 0.f; old[q][i] = in[16 + i + q]; } }
  uint32_t cnt = 0;
  float fm = 1e30f;
  const f32x2 nlo = {in[lane], in[lane]};
  constexpr bool kBarrier = MODE < 2, kAnti = (MODE == 1 || MODE == 3);
  if (kAnti && grp == 1) {   // start half a period late: the VALU clump of a pretend tile first
    valu_clump<MIX>(old, nlo, cnt, fm);
    if (kBarrier) __syncthreads();
  }
  for (int it = 0; it < iters; it += 2) {
    mfma_clump(a, b, c0, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (kBarrier) __syncthreads();
    valu_clump<MIX>(acc, nlo, cnt, fm);
    __builtin_amdgcn_sched_barrier(0);
    if (kBarrier) __syncthreads();
    asm volatile("" ::"v"(c0));
    a[0] += 1e-9f;
    mfma_clump(a, b, c0, old);
    __builtin_amdgcn_sched_barrier(0);
    if (kBarrier) __syncthreads();
    valu_clump<MIX>(old, nlo, cnt, fm);
    __builtin_amdgcn_sched_barrier(0);
    if (kBarrier) __syncthreads();
    asm volatile("" ::"v"(c0));
    a[0] += 1e-9f;
  }
  if (kAnti && grp == 0 && kBarrier) __syncthreads();   // (matches the extra barrier of the late group)
  float s = fm + cnt;
  for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i] + old[q][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int MIX>
void run(float* d, const float* in, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000, blocks = 256;
  hipLaunchKernelGGL((k<MODE, MIX>), dim3(blocks), dim3(512), 0, 0, d, in, 10);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, MIX>), dim3(blocks), dim3(512), 0, 0, d, in, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  // per SIMD: 2 waves x iters clumps x 4 chains
  printf("%-44s %.1f ns per chain and SIMD (five MFMAs alone: 137.4)\n", name, best * 1e6 / (2.0 * iters * 4));
}

int main() {
  float *d, *in; (void)hipMalloc(&d, sizeof(float) * 256 * 512); (void)hipMalloc(&in, 4096 * 4);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) >> 8) * 1e-7f;
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  run<0, 0>(d, in, "pop epilogue, in phase (barriers)");
  run<1, 0>(d, in, "pop epilogue, ANTI-phase (barriers)");
  run<2, 0>(d, in, "pop epilogue, free-running, same start");
  run<3, 0>(d, in, "pop epilogue, free-running, offset start");
  run<0, 1>(d, in, "nn epilogue, in phase (barriers)");
  run<1, 1>(d, in, "nn epilogue, ANTI-phase (barriers)");
  run<2, 1>(d, in, "nn epilogue, free-running, same start");
  run<3, 1>(d, in, "nn epilogue, free-running, offset start");
  return 0;
}
