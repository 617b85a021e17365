// microbenchmark for VERDICT r4 item 4: the per-radius threshold shift of the multi-radius population
// epilogue (pop_msym_kernel: t = acc - delta_r as 8 v_pk_add_f32 per radius and chain, then 16 v_alignbit) against the
// same shift done on the matrix pipe as an in-place cumulative rank-1 update acc += ones x (-step_r) (one
// v_mfma_f32_32x32x16_f16 per radius and chain, then 16 v_alignbit on the accumulator itself).
// Shape of the real loop: two chains per reference tile, 6 Gram MFMAs each; the epilogue of chain 0 sits between the
// MFMAs of chain 1, the epilogue of chain 1 stands alone.  Two waves per SIMD, 8 radii.
//   hipcc -O3 --offload-arch=gfx950 -o shift_mfma shift_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NM = 6, NR = 8;

template <int R0, int R1>
__device__ __forceinline__ void strings(const f32x16& t, uint32_t& bits) {
#pragma unroll
  for (int r = R0; r < R1; ++r) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t[r]), 30);
}

// MODE 0: VALU shift (the kernel's form).  MODE 1: in-place MFMA shift.  MODE 2: MFMA shift into a second tile
// (ping-pong: needs 16 more registers per chain in the real kernel)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float d0) {
  const int lane = threadIdx.x & 63;
  h16x8 a[NM], b[NM], one, step[NR];
  for (int m = 0; m < NM; ++m)
    for (int i = 0; i < 8; ++i) {
      a[m][i] = (_Float16)(0.001f * lane + 0.01f * i + m);
      b[m][i] = (_Float16)(1.0f + 0.01f * i - 0.1f * m);
    }
  for (int i = 0; i < 8; ++i) one[i] = (_Float16)((i == 0 && lane < 32) ? 1.0f : 0.0f);
  for (int r = 0; r < NR; ++r)
    for (int i = 0; i < 8; ++i) step[r][i] = (_Float16)((i == 0 && lane < 32) ? -(d0 + r) : 0.0f);
  float dl[NR];
  for (int r = 0; r < NR; ++r) dl[r] = d0 * r + 0.5f * r * r;
  f32x16 c0;
  for (int i = 0; i < 16; ++i) c0[i] = 1.0f + i;
  uint32_t bits[NR];
  for (int r = 0; r < NR; ++r) bits[r] = 0;
  uint32_t total = 0;
  for (int it = 0; it < iters; ++it) {
    // chain 0
    f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c0, 0, 0, 0);
#pragma unroll
    for (int m = 1; m < NM; ++m) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[m], acc0, 0, 0, 0);
    // chain 1 with the epilogue of chain 0 between its MFMAs, then its own epilogue
    f32x16 acc1;
#pragma unroll
    for (int phase = 0; phase < 2; ++phase) {
      f32x16& cur = phase == 0 ? acc0 : acc1;
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) {
        if (phase == 0 && rr < NM) {
          if (rr == 0) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], c0, 0, 0, 0);
          else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rr], b[(rr + 1) % NM], acc1, 0, 0, 0);
        }
        if (MODE == 0) {
          f32x16 t = cur;
          if (rr != 0) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              f32x2 v = {cur[r], cur[r + 1]};
              v = v - f32x2{dl[rr], dl[rr]};
              t[r] = v.x;
              t[r + 1] = v.y;
            }
          }
          strings<0, 16>(t, bits[rr]);
        } else if (MODE == 1) {
          if (rr != 0) cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(one, step[rr], cur, 0, 0, 0);
          strings<0, 16>(cur, bits[rr]);
        } else {
          // shift of radius rr + 1 issued into the other tile before the strings of radius rr are built
          static_assert(true, "");
          f32x16 nxt = cur;
          if (rr + 1 < NR) nxt = __builtin_amdgcn_mfma_f32_32x32x16_f16(one, step[rr + 1], cur, 0, 0, 0);
          strings<0, 16>(cur, bits[rr]);
          cur = nxt;
        }
      }
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) total += __builtin_popcount(bits[rr] & 0xAAAAAAAAu);
    }
    c0[it & 15] += 1.0f;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)total;
}

template <int MODE>
float run(float* d, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, 10, 3.0f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, iters, 3.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  float* d;
  hipMalloc(&d, sizeof(float) * 256 * 2048);
  const int iters = 4000;
  for (int occ = 1; occ <= 2; ++occ) {
    const int blocks = 256 * occ;
    const float t0 = run<0>(d, blocks, iters), t1 = run<1>(d, blocks, iters), t2 = run<2>(d, blocks, iters);
    const double per = 1e6 / (2.0 * iters);   // ns per chain and wave
    printf("waves/SIMD=%d: per chain (8 radii)  VALU shift %.1f ns   in-place MFMA shift %.1f ns   ping-pong MFMA shift %.1f ns\n", occ,
           t0 * per, t1 * per, t2 * per);
  }
  return 0;
}
