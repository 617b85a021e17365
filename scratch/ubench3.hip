// micro-benchmark 3: per-chain cost of the population epilogue behind (a) the fp32 MFMA Gram chain
// and (b) a bf16x3 split Gram chain (4 x v_mfma_f32_32x32x16_bf16 for D=10), gfx950.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o scratch/ubench3 scratch/ubench3.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int TQ = 4;
#ifndef LOCAL
#define LOCAL 0
#endif
#ifndef NLDS
#define NLDS 0
#endif

// MODE 0: fp32 chain (S=5) + sub epilogue; 1: bf16 NM MFMAs + sub epilogue; 2: bf16, lo folded (no sub)
template <int MODE, int NM>
__global__ __launch_bounds__(256, 2) void chain_kernel(const uint4* __restrict__ img, const float* __restrict__ imgf,
                                                      const float4* __restrict__ norms, int T, int n_tiles_img,
                                                      unsigned* __restrict__ out, unsigned wbits) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  s16x8 b[TQ][NM];
  float bf[TQ][5];
  float lo[TQ];
  unsigned cnt[TQ];
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int j = 0; j < 8; ++j) b[qt][m][j] = (short)(0x3c00 + ((lane * 7 + qt * 13 + m * 3 + j) & 0xff));
#pragma unroll
    for (int s = 0; s < 5; ++s) bf[qt][s] = 0.01f * (float)((lane + qt + s) & 15);
    lo[qt] = 0.5f + 0.001f * lane;
    cnt[qt] = 0;
  }
  const int base = (LOCAL ? 0 : ((wave >> 2) * 16)) % n_tiles_img;
  unsigned trig = 0;
  __shared__ float nlds[4][2][32];
  float* mylds = &nlds[threadIdx.x >> 6][0][0];
  struct Ref { s16x8 a[NM]; float af[5]; f32x16 cinit; float nval; };
  auto load = [&](Ref& R, int i) {
    int t = base + i;
    if (t >= n_tiles_img) t -= n_tiles_img;
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 5; ++s) R.af[s] = imgf[((size_t)t * 5 + s) * 64 + lane];
    } else {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const uint4 v = img[((size_t)t * NM + m) * 64 + lane];
        R.a[m] = *reinterpret_cast<const s16x8*>(&v);
      }
    }
#if NLDS
    R.nval = reinterpret_cast<const float*>(norms)[(size_t)t * 32 + (lane & 31)];
#else
    const float4* np = norms + ((size_t)t * 2 + (lane >> 5)) * 4;
    const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    R.cinit = (f32x16){n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, n3.x, n3.y, n3.z, n3.w};
#endif
  };
  int flip = 0;
  auto process = [&](Ref& R) {
#if NLDS
    {
      float* buf = mylds + 32 * flip;
      flip ^= 1;
      if (lane < 32) buf[lane] = R.nval;
      const float4* np = reinterpret_cast<const float4*>(buf + 4 * (lane >> 5));
      const float4 n0 = np[0], n1 = np[2], n2 = np[4], n3 = np[6];
      R.cinit = (f32x16){n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, n3.x, n3.y, n3.z, n3.w};
    }
#endif
#pragma unroll
    for (int qt = 0; qt < TQ; ++qt) {
      f32x16 acc;
      if (MODE == 0) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(R.af[0], bf[qt][0], R.cinit, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 5; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(R.af[s], bf[qt][s], acc, 0, 0, 0);
      } else if (MODE == 4) {   // no MFMA: epilogue on a cheaply perturbed accumulator
        acc = R.cinit;
        acc[0] += lo[qt];
        asm volatile("" : "+v"(acc));
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R.a[0], b[qt][0], R.cinit, 0, 0, 0);
#pragma unroll
        for (int m = 1; m < NM; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R.a[m], b[qt][m], acc, 0, 0, 0);
      }
      if (MODE == 3 || MODE == 5) {   // MFMA only: keep the accumulator alive with one op
        cnt[qt] += __float_as_uint(acc[3]) >> 31;
        continue;
      }
      unsigned bits = 0, mn = 0xffffffffu;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float tv = (MODE == 2 || MODE == 4) ? acc[r] : acc[r] - lo[qt];
        const unsigned tb = __float_as_uint(tv);
        bits = __builtin_amdgcn_alignbit(bits, tb, 31);
        mn = min(mn, tb);
      }
      cnt[qt] += __builtin_popcount(bits & 0xffffu);
      if (__builtin_amdgcn_ballot_w64(mn < wbits) != 0) trig++;
    }
  };
  Ref R0, R1;
  load(R0, 0);
  if (MODE == 5) {   // no loads in the loop at all: pure MFMA chains on resident operands
    load(R1, 1);
    for (int i = 0; i < T; i += 2) {
      process(R0);
      process(R1);
      asm volatile("" : "+v"(R0.cinit), "+v"(R1.cinit));
    }
  } else
  for (int i = 0; i < T; i += 2) {
    load(R1, i + 1);
    process(R0);
    load(R0, i + 2);
    process(R1);
  }
  unsigned s = trig;
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) s += cnt[qt];
  out[wave * 64 + lane] = s;
}

template <int MODE, int NM>
static void run(const char* name, const uint4* img, const float* imgf, const float4* norms, int T, int n_img,
                unsigned* out, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const unsigned wbits = 0x00800000u;  // never triggers on these magnitudes
  chain_kernel<MODE, NM><<<blocks, 256>>>(img, imgf, norms, T, n_img, out, wbits);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    chain_kernel<MODE, NM><<<blocks, 256>>>(img, imgf, norms, T, n_img, out, wbits);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double chains = (double)blocks * 4 * T * TQ;
  const double per_simd = chains / 1024.0;
  printf("%-34s %8.3f ms  %7.1f ns/chain/SIMD  (%6.1f cycles @2.4GHz)  chains/s %.3e  -> D=10 'fp32 roof' %.1f%%\n", name,
         best, best * 1e6 / per_simd, best * 1e6 / per_simd * 2.4, chains / (best * 1e-3),
         chains * 1024 * 20 / (best * 1e-3) / 157.3e12 * 100);
}

int main() {
  const int n_img = 31250, NMMAX = 6;
  uint4* img;
  float* imgf;
  float4* norms;
  unsigned* out;
  const size_t img_bytes = (size_t)n_img * NMMAX * 64 * 16;
  hipMalloc(&img, img_bytes);
  hipMalloc(&imgf, (size_t)n_img * 5 * 64 * 4);
  hipMalloc(&norms, (size_t)n_img * 32 * 4);
  hipMalloc(&out, 4096 * 4 * 64 * 4);
  std::vector<unsigned short> h(img_bytes / 2);
  const bool zero = getenv("ZERO") != nullptr;
  for (size_t i = 0; i < h.size(); ++i) h[i] = zero ? 0 : (unsigned short)(0x3c00 + (rand() & 0x1ff));
  hipMemcpy(img, h.data(), img_bytes, hipMemcpyHostToDevice);
  std::vector<float> hf((size_t)n_img * 5 * 64);
  for (auto& v : hf) v = 0.001f * (rand() & 1023);
  hipMemcpy(imgf, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> hn((size_t)n_img * 32);
  for (auto& v : hn) v = 1.0f + 0.001f * (rand() & 1023);
  hipMemcpy(norms, hn.data(), hn.size() * 4, hipMemcpyHostToDevice);
  const int T = 1500;
  for (int blocks : {512, 1024}) {
    printf("blocks %d (waves/SIMD %.1f)\n", blocks, blocks * 4 / 1024.0);
    run<0, 1>("fp32 5xMFMA + sub epilogue", img, imgf, norms, T, n_img, out, blocks);
    run<1, 4>("bf16 4xMFMA + sub epilogue", img, imgf, norms, T, n_img, out, blocks);
    run<2, 4>("bf16 4xMFMA, lo folded", img, imgf, norms, T, n_img, out, blocks);
    run<3, 4>("bf16 4xMFMA only", img, imgf, norms, T, n_img, out, blocks);
    run<5, 4>("bf16 4xMFMA only, no loads", img, imgf, norms, T, n_img, out, blocks);
    run<4, 4>("epilogue only (loads as for 4xMFMA)", img, imgf, norms, T, n_img, out, blocks);
  }
  return 0;
}
