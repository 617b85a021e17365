// microbenchmark (round 6): how should the epilogue VALU work of the fp32-input MFMA sweep be placed?
// One "chain" = 5 dependent v_mfma_f32_32x32x2_f32 (D = 10).  Per chain NVC VALU instructions of a given mix
// work on the accumulator of an OLDER chain.  Placements:
//   P=0  [MFMA ; NVC/5 VALU] x 5            (the round-1..5 kernel: epilogue dealt to the MFMA slots)
//   P=1  5 MFMA ; NVC VALU                   (one clump per chain)
//   P=2  10 MFMA (two chains interleaved) ; 2 NVC VALU
//   P=3  20 MFMA (four chains interleaved) ; 4 NVC VALU
// Mix: 0 = sub(e32) + alignbit + min_u32 per element (48/chain), 1 = pk_add per 2 + alignbit per element (24/chain),
//      2 = 8 v_min3_f32 + 4 misc (12/chain: the neighbour epilogue), 3 = no VALU at all
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MIX, int R0, int R1>
__device__ __forceinline__ void epi(const f32x16& acc, float lo, uint32_t& bits, uint32_t& tmin, float& fm) {
  if constexpr (MIX == 0) {
#pragma unroll
    for (int r = R0; r < R1; ++r) {
      const uint32_t tb = __float_as_uint(acc[r] - lo);
      bits = __builtin_amdgcn_alignbit(bits, tb, 31);
      tmin = min(tmin, tb);
    }
  } else if constexpr (MIX == 1) {
#pragma unroll
    for (int r = R0; r < R1; r += 2) {
      const f32x2 t = f32x2{acc[r], acc[r + 1]} - f32x2{lo, lo};
      bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t[0]), 30);
      bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t[1]), 30);
    }
  } else if constexpr (MIX == 4) {
    f32x2 t[(R1 - R0) / 2];
#pragma unroll
    for (int r = R0; r < R1; r += 2) t[(r - R0) / 2] = f32x2{acc[r], acc[r + 1]} - f32x2{lo, lo};
#pragma unroll
    for (int r = R0; r < R1; r += 2) asm volatile("" : "+v"(t[(r - R0) / 2]));
#pragma unroll
    for (int r = R0; r < R1; r += 2) {
      bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t[(r - R0) / 2][0]), 30);
      bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t[(r - R0) / 2][1]), 30);
    }
  } else if constexpr (MIX == 2) {
#pragma unroll
    for (int r = R0; r < R1; r += 2) fm = fminf(fm, fminf(acc[r], acc[r + 1]));
  }
}

template <int P, int MIX>
__device__ __forceinline__ void step(const float (&a)[5], const float (&b)[4][5], const f32x16& c0, f32x16 (&acc)[4],
                                     const f32x16 (&old)[4], float lo, uint32_t& bits, uint32_t& tmin, float& fm, uint32_t& cnt) {
  if constexpr (P == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[q][s], s == 0 ? c0 : acc[q], 0, 0, 0);
        if (s == 0) epi<MIX, 0, 4>(old[q], lo, bits, tmin, fm);
        if (s == 1) epi<MIX, 4, 6>(old[q], lo, bits, tmin, fm);
        if (s == 2) epi<MIX, 6, 10>(old[q], lo, bits, tmin, fm);
        if (s == 3) epi<MIX, 10, 12>(old[q], lo, bits, tmin, fm);
        if (s == 4) epi<MIX, 12, 16>(old[q], lo, bits, tmin, fm);
        __builtin_amdgcn_sched_barrier(0);
      }
      cnt += __builtin_popcount(bits) + (tmin < 77u ? 1u : 0u);
    }
  } else {
    constexpr int NC = (P == 1) ? 1 : (P == 2 ? 2 : 4);
#pragma unroll
    for (int q0 = 0; q0 < 4; q0 += NC) {
#pragma unroll
      for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int q = q0; q < q0 + NC; ++q)
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[q][s], s == 0 ? c0 : acc[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = q0; q < q0 + NC; ++q) {
        epi<MIX, 0, 16>(old[q], lo, bits, tmin, fm);
        cnt += __builtin_popcount(bits) + (tmin < 77u ? 1u : 0u);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int P, int MIX>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* in, int iters) {
  const int lane = threadIdx.x & 63;
  float a[5], b[4][5];
  for (int s = 0; s < 5; ++s) {
    a[s] = in[lane + 64 * s];
    for (int q = 0; q < 4; ++q) b[q][s] = in[lane + 64 * (s + 5 + q)];
  }
  f32x16 c0, acc[4], old[4];
  for (int i = 0; i < 16; ++i) { c0[i] = in[i]; for (int q = 0; q < 4; ++q) { acc[q][i] = 0.f; old[q][i] = in[16 + i + q]; } }
  uint32_t bits = 0, tmin = 0xFFFFFFFFu, cnt = 0;
  float fm = 1e30f, lo = in[lane];
  for (int it = 0; it < iters; it += 2) {
    step<P, MIX>(a, b, c0, acc, old, lo, bits, tmin, fm, cnt);
    asm volatile("" ::"v"(c0));
    a[0] += 1e-9f;
    step<P, MIX>(a, b, c0, old, acc, lo, bits, tmin, fm, cnt);
    asm volatile("" ::"v"(c0));
    a[0] += 1e-9f;
  }
  float s = fm + cnt;
  for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i] + old[q][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int P, int MIX>
void run(float* d, const float* in, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int occ = 1; occ <= 2; ++occ) {
    const int blocks = 256 * occ;
    hipLaunchKernelGGL((k<P, MIX>), dim3(blocks), dim3(256), 0, 0, d, in, 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<P, MIX>), dim3(blocks), dim3(256), 0, 0, d, in, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    // per SIMD: occ waves x iters x 4 chains
    const double ns_per_chain = best * 1e6 / (double)(occ * iters * 4);
    printf("%-28s waves/SIMD=%d  %.1f ns per chain and SIMD  (pure MFMA at 2.4 GHz = 133.3)\n", name, occ, ns_per_chain);
  }
}

int main() {
  float *d, *in; hipMalloc(&d, sizeof(float) * 256 * 1024); hipMalloc(&in, 4096 * 4);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) >> 8) * 1e-7f;
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  run<1, 3>(d, in, "MFMA only");
  run<0, 0>(d, in, "P0 dealt,   mix0 (48)");
  run<1, 0>(d, in, "P1 clump1,  mix0 (48)");
  run<2, 0>(d, in, "P2 clump2,  mix0 (48)");
  run<3, 0>(d, in, "P3 clump4,  mix0 (48)");
  run<0, 1>(d, in, "P0 dealt,   mix1 (24)");
  run<1, 1>(d, in, "P1 clump1,  mix1 (24)");
  run<2, 1>(d, in, "P2 clump2,  mix1 (24)");
  run<3, 1>(d, in, "P3 clump4,  mix1 (24)");
  run<0, 4>(d, in, "P0 dealt,   mix4 (24, 2-phase)");
  run<1, 4>(d, in, "P1 clump1,  mix4 (24, 2-phase)");
  run<2, 4>(d, in, "P2 clump2,  mix4 (24, 2-phase)");
  run<3, 4>(d, in, "P3 clump4,  mix4 (24, 2-phase)");
  run<0, 2>(d, in, "P0 dealt,   mix2 (nn)");
  run<1, 2>(d, in, "P1 clump1,  mix2 (nn)");
  run<2, 2>(d, in, "P2 clump2,  mix2 (nn)");
  run<3, 2>(d, in, "P3 clump4,  mix2 (nn)");
  return 0;
}
