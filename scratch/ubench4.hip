// micro-benchmark 4: one 32x32 tile pair per "chain" as (A) 2 dependent v_mfma_f32_32x32x16_f16 or (B) 4 independent
// v_mfma_f32_16x16x32_f16 (K = 32 in one instruction), each with the population epilogue of the previous chain
// interleaved in program order (16 v_alignbit + 8 v_min3_u32 + popcount), operands resident, 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/ubench4 scratch/ubench4.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
constexpr int TQ = 6;

__device__ __forceinline__ void epi(const f32x16& acc, unsigned& cnt, unsigned& trig, unsigned wbits, int lo, int hi,
                                    unsigned& bits, unsigned& mn) {
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (r >= lo && r < hi) {
      const unsigned tb = __float_as_uint(acc[r]);
      bits = __builtin_amdgcn_alignbit(bits, tb, 31);
      mn = min(mn, tb);
    }
}

template <int MODE>   // 0: 2 x 32x32x16; 1: 4 x 16x16x32; 2/3: the same without epilogue
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int T, unsigned wbits) {
  const int lane = threadIdx.x & 63;
  h16x8 a[2], b[TQ][2];
  for (int m = 0; m < 2; ++m)
    for (int j = 0; j < 8; ++j) {
      a[m][j] = (_Float16)(0.01f * ((lane * 3 + m * 5 + j) & 31));
      for (int q = 0; q < TQ; ++q) b[q][m][j] = (_Float16)(0.02f * ((lane + q * 7 + m + j) & 15));
    }
  f32x16 c0;
  for (int r = 0; r < 16; ++r) c0[r] = 1.0f + 0.001f * (lane + r);
  unsigned cnt[TQ], trig = 0;
  for (int q = 0; q < TQ; ++q) cnt[q] = 0;
  f32x16 accA, accB;
  for (int r = 0; r < 16; ++r) accB[r] = 1.0f;
  for (int t = 0; t < T; ++t) {
    asm volatile("" : "+v"(c0));
#pragma unroll
    for (int q = 0; q < TQ; q += 2) {
      auto chain = [&](f32x16& nw, const f32x16& old, int qn, int qo) {
        unsigned bits = 0, mn = 0xffffffffu;
        if (MODE == 0 || MODE == 2) {
          nw = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[qn][0], c0, 0, 0, 0);
          if (MODE == 0) epi(old, cnt[qo], trig, wbits, 0, 8, bits, mn);
          nw = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[qn][1], nw, 0, 0, 0);
          if (MODE == 0) epi(old, cnt[qo], trig, wbits, 8, 16, bits, mn);
        } else {
          f32x4 p[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 ci = {c0[4 * i], c0[4 * i + 1], c0[4 * i + 2], c0[4 * i + 3]};
            p[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 1], b[qn][i >> 1], ci, 0, 0, 0);
            if (MODE == 1) epi(old, cnt[qo], trig, wbits, 4 * i, 4 * i + 4, bits, mn);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            nw[4 * i] = p[i][0]; nw[4 * i + 1] = p[i][1]; nw[4 * i + 2] = p[i][2]; nw[4 * i + 3] = p[i][3];
          }
        }
        if (MODE <= 1) {
          cnt[qo] += __builtin_popcount(bits);
          if (__builtin_amdgcn_ballot_w64(mn < wbits) != 0) trig++;
        } else {
          cnt[qo] += __float_as_uint(old[3]) >> 31;
        }
      };
      chain(accA, accB, q, (q + TQ - 1) % TQ);
      chain(accB, accA, q + 1, q);
    }
  }
  unsigned s = trig;
  for (int q = 0; q < TQ; ++q) s += cnt[q];
  out[(blockIdx.x * 256 + threadIdx.x)] = s + __float_as_uint(accB[1]);
}

template <int MODE>
void run(const char* name, unsigned* out, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int T = 4000;
  k<MODE><<<blocks, 256>>>(out, T, 0x00800000u);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, T, 0x00800000u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double chains = (double)blocks * 4 * T * TQ, per_simd = chains / 1024.0;
  printf("%-44s %8.3f ms  %6.1f ns per tile pair per SIMD\n", name, best, best * 1e6 / per_simd);
}
int main() {
  unsigned* out;
  hipMalloc(&out, 2048 * 256 * 4);
  for (int blocks : {256, 512}) {
    printf("blocks %d (waves per SIMD %.0f)\n", blocks, blocks * 4 / 1024.0);
    run<0>("2 x mfma 32x32x16 f16 + epilogue", out, blocks);
    run<1>("4 x mfma 16x16x32 f16 + epilogue", out, blocks);
    run<2>("2 x mfma 32x32x16 f16 only", out, blocks);
    run<3>("4 x mfma 16x16x32 f16 only", out, blocks);
  }
  return 0;
}
