// Variants of the row-statistics pass (max |x - mu|^2 in double, extent of columns 0/1, non-finite flag) timed in isolation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cfloat>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint32_t fkey(float f) { const uint32_t u = __float_as_uint(f); return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ __forceinline__ void publish_max(uint32_t* addr, uint32_t v, uint32_t* wave_max) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off, 64));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    v = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    if (v > __atomic_load_n(addr, __ATOMIC_RELAXED)) atomicMax(addr, v);
  }
  __syncthreads();
}
struct Acc { uint32_t m_norm = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0; bool bad = false; };
__device__ __forceinline__ void take(Acc& a, double nrm, float c0, float c1) {
  const float nf = (float)nrm;
  const bool ok = nf <= 1.0e30f;
  a.bad |= !ok;
  a.m_norm = max(a.m_norm, ok ? __float_as_uint(nf) : 0u);
  const bool fin = (fabsf(c0) <= FLT_MAX) && (fabsf(c1) <= FLT_MAX);
  a.m0 = max(a.m0, fin ? ~fkey(c0) : 0u); a.m1 = max(a.m1, fin ? fkey(c0) : 0u);
  a.m2 = max(a.m2, fin ? ~fkey(c1) : 0u); a.m3 = max(a.m3, fin ? fkey(c1) : 0u);
}
__device__ __forceinline__ void finish(const Acc& a, uint32_t* hdr, uint32_t* wave_max) {
  if (a.bad) atomicOr(hdr + 1, 1u);
  publish_max(hdr, a.m_norm, wave_max); publish_max(hdr + 8, a.m0, wave_max); publish_max(hdr + 9, a.m1, wave_max);
  publish_max(hdr + 10, a.m2, wave_max); publish_max(hdr + 11, a.m3, wave_max);
}
// A: a row per lane, one float2 load per trip
__global__ void v_rowloop(const float* __restrict__ coords, uint32_t n, uint32_t D, const float* __restrict__ means, uint32_t* hdr) {
  __shared__ uint32_t wave_max[4]; __shared__ float mu[64];
  if (threadIdx.x < D) mu[threadIdx.x] = means[threadIdx.x];
  __syncthreads();
  Acc a;
  for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x) {
    const float2* x2 = reinterpret_cast<const float2*>(coords + (size_t)row * D);
    double nrm = 0.0; const float2 f = x2[0];
    for (uint32_t k = 0; k < D; k += 2) { const float2 v = x2[k >> 1]; const float p = v.x - mu[k], q = v.y - mu[k + 1]; nrm += (double)p * (double)p; nrm += (double)q * (double)q; }
    take(a, nrm, f.x, f.y);
  }
  finish(a, hdr, wave_max);
}
// B: 256 rows per block through LDS, float4 loads statically unrolled four deep
__global__ void v_lds4(const float* __restrict__ coords, uint32_t n, uint32_t D, const float* __restrict__ means, uint32_t* hdr) {
  extern __shared__ float tile[]; __shared__ uint32_t wave_max[4]; __shared__ float mu[64];
  if (threadIdx.x < D) mu[threadIdx.x] = means[threadIdx.x];
  const uint32_t Dp = D | 1u; const size_t total = (size_t)n * D; const uint32_t n_chunks = (n + 255) / 256;
  Acc a;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const size_t base = (size_t)chunk * 256 * D;
    __syncthreads();
    const uint32_t nv = 64 * D;   // float4 per chunk (256 * D / 4)
    const float4* src = reinterpret_cast<const float4*>(coords + base);
    for (uint32_t e0 = threadIdx.x; e0 < nv; e0 += 1024) {
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const uint32_t e = e0 + 256 * j; v[j] = (e < nv && base + 4 * (size_t)e + 3 < total) ? src[e] : make_float4(0, 0, 0, 0); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { const uint32_t e = e0 + 256 * j; if (e < nv) { const uint32_t f0 = 4 * e; const float w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) { const uint32_t f = f0 + t, r = f / D, k = f - r * D; tile[r * Dp + k] = w[t]; } } }
    }
    __syncthreads();
    const uint32_t row = chunk * 256 + threadIdx.x;
    if (row < n) {
      const float* x = tile + threadIdx.x * Dp; double nrm = 0.0;
      for (uint32_t k = 0; k < D; ++k) { const float p = x[k] - mu[k]; nrm += (double)p * (double)p; }
      take(a, nrm, x[0], D > 1 ? x[1] : 0.0f);
    }
  }
  finish(a, hdr, wave_max);
}
// C: a row per lane, D as a template parameter (fully unrolled loads)
template <int DD>
__global__ void v_rowT(const float* __restrict__ coords, uint32_t n, const float* __restrict__ means, uint32_t* hdr) {
  __shared__ uint32_t wave_max[4]; __shared__ float mu[64];
  if (threadIdx.x < DD) mu[threadIdx.x] = means[threadIdx.x];
  __syncthreads();
  Acc a;
  for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x) {
    const float2* x2 = reinterpret_cast<const float2*>(coords + (size_t)row * DD);
    float2 v[DD / 2];
#pragma unroll
    for (int k = 0; k < DD / 2; ++k) v[k] = x2[k];
    double nrm = 0.0;
#pragma unroll
    for (int k = 0; k < DD / 2; ++k) { const float p = v[k].x - mu[2 * k], q = v[k].y - mu[2 * k + 1]; nrm += (double)p * (double)p; nrm += (double)q * (double)q; }
    take(a, nrm, v[0].x, v[0].y);
  }
  finish(a, hdr, wave_max);
}
// D: as C but the squared norm in float pairs (what does the double arithmetic cost?)
template <int DD>
__global__ void v_rowT_f32(const float* __restrict__ coords, uint32_t n, const float* __restrict__ means, uint32_t* hdr) {
  __shared__ uint32_t wave_max[4]; __shared__ float mu[64];
  if (threadIdx.x < DD) mu[threadIdx.x] = means[threadIdx.x];
  __syncthreads();
  Acc a;
  for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x) {
    const float2* x2 = reinterpret_cast<const float2*>(coords + (size_t)row * DD);
    float2 v[DD / 2];
#pragma unroll
    for (int k = 0; k < DD / 2; ++k) v[k] = x2[k];
    float nrm = 0.0f;
#pragma unroll
    for (int k = 0; k < DD / 2; ++k) { const float p = v[k].x - mu[2 * k], q = v[k].y - mu[2 * k + 1]; nrm += p * p; nrm += q * q; }
    take(a, (double)nrm, v[0].x, v[0].y);
  }
  finish(a, hdr, wave_max);
}
int main() {
  const uint32_t n = 1000000, D = 10;
  std::vector<float> h((size_t)n * D); std::mt19937 g(1); std::normal_distribution<float> nd(0.f, 0.3f);
  for (auto& x : h) x = nd(g);
  float *d, *mu; uint32_t* hdr;
  CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&mu, 256)); CK(hipMalloc(&hdr, 1024));
  CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(mu, 0, 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
      (void)hipMemsetAsync(hdr, 0, 1024, 0);
      (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep > 1 && ms < best) best = ms;
    }
    uint32_t out[12]; (void)hipMemcpy(out, hdr, 48, hipMemcpyDeviceToHost);
    printf("%-34s %7.1f us   (max norm bits %08x, flag %u)\n", name, best * 1e3f, out[0], out[1]);
  };
  for (uint32_t blocks : {128u, 256u, 512u, 1024u}) {
    printf("grid %u\n", blocks);
    time("row per lane, run-time loop", [&] { hipLaunchKernelGGL(v_rowloop, dim3(blocks), dim3(256), 0, 0, d, n, D, mu, hdr); });
    time("LDS tile, float4 x 4 in flight", [&] { hipLaunchKernelGGL(v_lds4, dim3(std::min(blocks, 3907u)), dim3(256), 4 * 256 * (D | 1u), 0, d, n, D, mu, hdr); });
    time("row per lane, D = 10 template", [&] { hipLaunchKernelGGL(v_rowT<10>, dim3(blocks), dim3(256), 0, 0, d, n, mu, hdr); });
    time("same, float norm", [&] { hipLaunchKernelGGL(v_rowT_f32<10>, dim3(blocks), dim3(256), 0, 0, d, n, mu, hdr); });
  }
  return 0;
}
