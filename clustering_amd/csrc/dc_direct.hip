// dc_direct.hip -- exact-by-construction VALU kernels of the density hot path (gfx950).
//
// Replaces the reference's CUDA kernels population_count
// (density_clustering_cuda_kernels.cu:9-56) and nearest_neighbor_search (:58-130), but with the
// arithmetic and the decision rules of the reference's CPU path (density_clustering.cpp:126-288),
// which is the parity target: strict '<', duplicates are neighbours, lowest index wins ties,
// "none" = (n_rows+1, FLT_MAX), and d2 in the reference binary's float order (dist2_canon).
//
// Shape (one launch sweeps ALL reference frames; the reference needs ceil(N/512) launches):
//   - a workgroup of 256 lanes owns 256*Q query frames; each lane keeps its Q query rows in
//     VGPRs for the whole sweep (Q*D registers);
//   - reference frames stream through one LDS tile of 256 rows (row stride padded to a
//     multiple of 4 floats so that a row is read with ds_read_b128); every lane of a wave reads
//     the SAME row -> LDS broadcast, no bank conflicts;
//   - coordinates are read row-major and fully coalesced from HBM/L2 (a tile is one contiguous
//     256*D*4-byte span);
//   - counters / running minima are per-lane registers, so no cross-lane reduction is needed.
// VALU-bound by design: 3D-1 non-fusable float ops per pair (no FMA is allowed by the spec).
#include "dc_common.hpp"

#include <float.h>
#include <array>
#include <utility>

namespace dc {

namespace {

constexpr int kBlock = 256;   // lanes per workgroup (4 waves: one per SIMD)
constexpr int kTile = 256;    // reference frames per LDS tile

template <int D>
struct Cfg {
  static constexpr int S = (D + 3) & ~3;                       // LDS row stride (floats)
  static constexpr int Q = (D <= 12) ? 4 : (D <= 24 ? 2 : 1);  // query rows per lane
};

// contiguous span of nt rows -> padded LDS tile
template <int D>
__device__ __forceinline__ void stage_tile(const float* __restrict__ coords, uint32_t t0,
                                           uint32_t nt, float* tile) {
  constexpr int S = Cfg<D>::S;
  const float* src = coords + (size_t)t0 * D;
  const uint32_t ne = nt * D;
  for (uint32_t e = threadIdx.x; e < ne; e += kBlock) {
    const uint32_t row = e / D, col = e - row * D;
    tile[row * S + col] = src[e];
  }
}

template <int D>
__device__ __forceinline__ void load_ref(const float* tile, uint32_t r, float (&ref)[D]) {
  constexpr int S = Cfg<D>::S;
  const float4* r4 = reinterpret_cast<const float4*>(tile + r * S);
#pragma unroll
  for (int m = 0; m < S / 4; ++m) {
    const float4 v = r4[m];
    if (4 * m + 0 < D) ref[4 * m + 0] = v.x;
    if (4 * m + 1 < D) ref[4 * m + 1] = v.y;
    if (4 * m + 2 < D) ref[4 * m + 2] = v.z;
    if (4 * m + 3 < D) ref[4 * m + 3] = v.w;
  }
}

// -----------------------------------------------------------------------------------------
// population count: pops[r][i] = 1 + #{ j != i : d2(i,j) < rad2[r] }
// -----------------------------------------------------------------------------------------
template <int D, int NR>
__global__ __launch_bounds__(kBlock) void pop_direct_kernel(const float* __restrict__ coords,
                                                            uint32_t n_rows, uint32_t i_from,
                                                            uint32_t i_to, Rad2 rad2, int n_rad,
                                                            uint32_t* __restrict__ pops,
                                                            const uint32_t* __restrict__ gate) {
  constexpr int S = Cfg<D>::S, Q = Cfg<D>::Q;
  if (gate && gate[1] == 0) return;
  __shared__ __attribute__((aligned(16))) float tile[kTile * S];
  const uint32_t qbase = i_from + blockIdx.x * (kBlock * Q);

  float q[Q][D];
  uint32_t qi[Q];
#pragma unroll
  for (int a = 0; a < Q; ++a) {
    qi[a] = qbase + a * kBlock + threadIdx.x;
    const uint32_t row = qi[a] < i_to ? qi[a] : i_to - 1;   // clamp: result discarded
#pragma unroll
    for (int k = 0; k < D; ++k) q[a][k] = coords[(size_t)row * D + k];
  }
  uint32_t cnt[Q][NR];
#pragma unroll
  for (int a = 0; a < Q; ++a)
#pragma unroll
    for (int r = 0; r < NR; ++r) cnt[a][r] = 0;

  for (uint32_t t0 = 0; t0 < n_rows; t0 += kTile) {
    const uint32_t nt = min((uint32_t)kTile, n_rows - t0);
    __syncthreads();
    stage_tile<D>(coords, t0, nt, tile);
    __syncthreads();
#pragma unroll 2
    for (uint32_t r = 0; r < nt; ++r) {
      float ref[D];
      load_ref<D>(tile, r, ref);
#pragma unroll
      for (int a = 0; a < Q; ++a) {
        const float d = dist2_canon<D>(q[a], ref);
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) cnt[a][rr] += (d < rad2.v[rr]) ? 1u : 0u;
      }
    }
  }
  // The sweep counted the self pair iff d2(i,i) < rad2 (d2(i,i) is +0, or NaN for non-finite
  // rows); the reference never evaluates it and starts every population at 1 (:132-134).
#pragma unroll
  for (int a = 0; a < Q; ++a) {
    if (qi[a] < i_to) {
      const float dself = dist2_canon<D>(q[a], q[a]);
#pragma unroll
      for (int rr = 0; rr < NR; ++rr)
        if (rr < n_rad)   // an instance with NR slots also serves fewer radii (unused slots = -1)
          pops[(size_t)rr * n_rows + qi[a]] = cnt[a][rr] + 1u - ((dself < rad2.v[rr]) ? 1u : 0u);
    }
  }
}

// -----------------------------------------------------------------------------------------
// nearest neighbour / nearest neighbour with lower free energy
// -----------------------------------------------------------------------------------------
template <int D, int Q, bool DIAG>
__device__ __forceinline__ void nn_tile(const float* tile, const float* tile_fe, uint32_t t0,
                                        uint32_t nt, const float (&q)[Q][D],
                                        const uint32_t (&qi)[Q], const float (&qfe)[Q],
                                        float (&best)[Q], uint32_t (&bj)[Q], float (&bhd)[Q],
                                        uint32_t (&bjhd)[Q]) {
#pragma unroll 2
  for (uint32_t r = 0; r < nt; ++r) {
    float ref[D];
    load_ref<D>(tile, r, ref);
    const float rfe = tile_fe[r];
    const uint32_t j = t0 + r;
#pragma unroll
    for (int a = 0; a < Q; ++a) {
      const float d = dist2_canon<D>(q[a], ref);
      bool lt = d < best[a];                       // strict: first (lowest) j wins ties, :270
      bool lh = (rfe < qfe[a]) && (d < bhd[a]);    // :275-276
      if (DIAG) {
        const bool other = (j != qi[a]);           // :262
        lt = lt && other;
        lh = lh && other;
      }
      best[a] = lt ? d : best[a];
      bj[a] = lt ? j : bj[a];
      bhd[a] = lh ? d : bhd[a];
      bjhd[a] = lh ? j : bjhd[a];
    }
  }
}

template <int D>
__global__ __launch_bounds__(kBlock) void nn_direct_kernel(
    const float* __restrict__ coords, uint32_t n_rows, const float* __restrict__ fe,
    uint32_t i_from, uint32_t i_to, uint32_t* __restrict__ nn_idx, float* __restrict__ nn_d2,
    uint32_t* __restrict__ hd_idx, float* __restrict__ hd_d2, const uint32_t* __restrict__ gate) {
  constexpr int S = Cfg<D>::S, Q = Cfg<D>::Q;
  if (gate && gate[1] == 0) return;
  __shared__ __attribute__((aligned(16))) float tile[kTile * S];
  __shared__ float tile_fe[kTile];
  const uint32_t qbase = i_from + blockIdx.x * (kBlock * Q);
  const uint32_t qend = min(qbase + kBlock * Q, i_to);   // this workgroup's rows: [qbase, qend)

  float q[Q][D], qfe[Q], best[Q], bhd[Q];
  uint32_t qi[Q], bj[Q], bjhd[Q];
#pragma unroll
  for (int a = 0; a < Q; ++a) {
    qi[a] = qbase + a * kBlock + threadIdx.x;
    const uint32_t row = qi[a] < i_to ? qi[a] : i_to - 1;
#pragma unroll
    for (int k = 0; k < D; ++k) q[a][k] = coords[(size_t)row * D + k];
    qfe[a] = fe[row];
    best[a] = FLT_MAX;       // :257-260
    bhd[a] = FLT_MAX;
    bj[a] = n_rows + 1;
    bjhd[a] = n_rows + 1;
  }

  for (uint32_t t0 = 0; t0 < n_rows; t0 += kTile) {
    const uint32_t nt = min((uint32_t)kTile, n_rows - t0);
    __syncthreads();
    stage_tile<D>(coords, t0, nt, tile);
    if (threadIdx.x < nt) tile_fe[threadIdx.x] = fe[t0 + threadIdx.x];
    __syncthreads();
    // only tiles that overlap this workgroup's own rows can contain a self pair
    if (t0 < qend && t0 + nt > qbase)
      nn_tile<D, Q, true>(tile, tile_fe, t0, nt, q, qi, qfe, best, bj, bhd, bjhd);
    else
      nn_tile<D, Q, false>(tile, tile_fe, t0, nt, q, qi, qfe, best, bj, bhd, bjhd);
  }
#pragma unroll
  for (int a = 0; a < Q; ++a) {
    if (qi[a] < i_to) {
      nn_idx[qi[a]] = bj[a];
      nn_d2[qi[a]] = best[a];
      hd_idx[qi[a]] = bjhd[a];
      hd_d2[qi[a]] = bhd[a];
    }
  }
}

// -----------------------------------------------------------------------------------------
// generic n_cols (33..kMaxColsGeneric): query rows live in LDS, lane-major ([k][lane], no bank
// conflicts); 64 lanes per workgroup, 32 reference rows per tile.  Same arithmetic.
// -----------------------------------------------------------------------------------------
constexpr int kGBlock = 64, kGTile = 32;

__global__ __launch_bounds__(kGBlock) void pop_generic_kernel(const float* __restrict__ coords,
                                                              uint32_t n_rows, uint32_t D,
                                                              uint32_t i_from, uint32_t i_to,
                                                              Rad2 rad2, int n_rad,
                                                              uint32_t* __restrict__ pops,
                                                              const uint32_t* __restrict__ gate) {
  if (gate && gate[1] == 0) return;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* qs = smem;                        // [D][kGBlock]
  float* tile = smem + (size_t)D * kGBlock;  // [kGTile][D]
  const uint32_t qi = i_from + blockIdx.x * kGBlock + threadIdx.x;
  const uint32_t row = qi < i_to ? qi : i_to - 1;
  for (uint32_t k = 0; k < D; ++k) qs[k * kGBlock + threadIdx.x] = coords[(size_t)row * D + k];
  uint32_t cnt[kMaxRadiiPerLaunch];
#pragma unroll
  for (int r = 0; r < kMaxRadiiPerLaunch; ++r) cnt[r] = 0;
  for (uint32_t t0 = 0; t0 < n_rows; t0 += kGTile) {
    const uint32_t nt = min((uint32_t)kGTile, n_rows - t0);
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < nt * D; e += kGBlock) tile[e] = coords[(size_t)t0 * D + e];
    __syncthreads();
    for (uint32_t r = 0; r < nt; ++r) {
      const float d = dist2_canon_rt(qs + threadIdx.x, kGBlock, tile + r * D, 1, (int)D);
#pragma unroll
      for (int rr = 0; rr < kMaxRadiiPerLaunch; ++rr) cnt[rr] += (d < rad2.v[rr]) ? 1u : 0u;
    }
  }
  if (qi < i_to) {
    const float dself = dist2_canon_rt(qs + threadIdx.x, kGBlock, qs + threadIdx.x, kGBlock, (int)D);
#pragma unroll
    for (int rr = 0; rr < kMaxRadiiPerLaunch; ++rr)
      if (rr < n_rad)
        pops[(size_t)rr * n_rows + qi] = cnt[rr] + 1u - ((dself < rad2.v[rr]) ? 1u : 0u);
  }
}

__global__ __launch_bounds__(kGBlock) void nn_generic_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t D, const float* __restrict__ fe,
    uint32_t i_from, uint32_t i_to, uint32_t* __restrict__ nn_idx, float* __restrict__ nn_d2,
    uint32_t* __restrict__ hd_idx, float* __restrict__ hd_d2, const uint32_t* __restrict__ gate) {
  if (gate && gate[1] == 0) return;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* qs = smem;
  float* tile = smem + (size_t)D * kGBlock;
  float* tile_fe = tile + (size_t)kGTile * D;
  const uint32_t qi = i_from + blockIdx.x * kGBlock + threadIdx.x;
  const uint32_t row = qi < i_to ? qi : i_to - 1;
  for (uint32_t k = 0; k < D; ++k) qs[k * kGBlock + threadIdx.x] = coords[(size_t)row * D + k];
  const float qfe = fe[row];
  float best = FLT_MAX, bhd = FLT_MAX;
  uint32_t bj = n_rows + 1, bjhd = n_rows + 1;
  for (uint32_t t0 = 0; t0 < n_rows; t0 += kGTile) {
    const uint32_t nt = min((uint32_t)kGTile, n_rows - t0);
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < nt * D; e += kGBlock) tile[e] = coords[(size_t)t0 * D + e];
    if (threadIdx.x < nt) tile_fe[threadIdx.x] = fe[t0 + threadIdx.x];
    __syncthreads();
    for (uint32_t r = 0; r < nt; ++r) {
      const uint32_t j = t0 + r;
      const float d = dist2_canon_rt(qs + threadIdx.x, kGBlock, tile + r * D, 1, (int)D);
      const bool other = (j != qi);
      const bool lt = other && (d < best);
      const bool lh = other && (tile_fe[r] < qfe) && (d < bhd);
      best = lt ? d : best;
      bj = lt ? j : bj;
      bhd = lh ? d : bhd;
      bjhd = lh ? j : bjhd;
    }
  }
  if (qi < i_to) {
    nn_idx[qi] = bj;
    nn_d2[qi] = best;
    hd_idx[qi] = bjhd;
    hd_d2[qi] = bhd;
  }
}

// -----------------------------------------------------------------------------------------
// small helpers
// -----------------------------------------------------------------------------------------
__global__ void nn_init_kernel(uint32_t n_rows, uint32_t* nn_idx, float* nn_d2, uint32_t* hd_idx,
                               float* hd_d2) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_rows) {
    nn_idx[i] = n_rows + 1;
    hd_idx[i] = n_rows + 1;
    nn_d2[i] = FLT_MAX;
    hd_d2[i] = FLT_MAX;
  }
}

__global__ void fe_gather_kernel(const uint32_t* __restrict__ pops, uint32_t n_rows,
                                 const float* __restrict__ table, float* __restrict__ fe) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_rows) fe[i] = table[pops[i]];
}

// Free energies on the device, with the host libm as the referee.  The reference computes
// fe = (float)(-log((double)((float)pop * (1.0f / max_pop)))) with glibc's double log
// (density_clustering.cpp:201-209 as compiled).  The device's double log differs from
// it by at most a few ulp(double); the two can only round to different floats when the double value lies
// within that distance of a float rounding boundary (a midpoint of neighbouring floats).  Such rows -- one
// in 2^22 with the margin below -- are listed for the host, which recomputes them with its libm.
// q, the reciprocal (double division rounded to float = correctly rounded float division, 53 >= 2*24+2)
// and the final rounding are IEEE operations, identical on both sides.
// state (device, kFeStateWords words): two slots of (max population, number of listed rows) used by alternate calls --
// slot = call parity.  The slot of the NEXT call is cleared by this call's kernel (its first thread), so that a step needs
// no fill kernels: max_u32_kernel and the rows' atomicAdd start from zeros.  (A first version cleared its own slot from
// the last block to finish, found with a ticket behind a __threadfence(): 245 us instead of 7 -- a device-scope release
// per block is an L2 write-back per block.)
__global__ void fe_log_kernel(const uint32_t* __restrict__ pops, uint32_t n_rows, uint32_t* __restrict__ state, uint32_t slot,
                              float* __restrict__ fe, uint2* __restrict__ flag_list, uint32_t flag_cap, double tol_rel) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    state[2u * (slot ^ 1u)] = 0u;
    state[2u * (slot ^ 1u) + 1u] = 0u;
  }
  if (i >= n_rows) return;
  const float rec = (float)(1.0 / (double)(float)(state[2u * slot]));
  const uint32_t pop = pops[i];
  const float q = (float)pop * rec;
  const double y = -log((double)q);
  const float f = (float)y;
  fe[i] = f;
  if (fabs(y) <= 1.0e300 && y != 0.0) {
    const uint32_t fb = __float_as_uint(f);
    // neighbours of f by magnitude (f != 0, finite: fe is at most -log(2^-24 / 1) = 16.6)
    const double up = (double)__uint_as_float(fb + 1u), dn = (double)__uint_as_float(fb - 1u);
    const double tol = tol_rel * fabs(y);   // default 64 ulp(double): device log <= 2 ulp, glibc log <= 1 ulp
    if (fabs(y - 0.5 * ((double)f + up)) < tol || fabs(y - 0.5 * ((double)f + dn)) < tol) {
      const uint32_t k = atomicAdd(state + 2u * slot + 1u, 1u);
      if (k < flag_cap) flag_list[k] = make_uint2(i, pop);
    }
  }
}

// neighbour results as order-preserving 64-bit words (d2 bits << 32 | index): d2 >= 0, so the words of a
// row order like the lexicographic (d2, index) and partial results merge with an integer minimum
__global__ void nn_pack_kernel(const uint32_t* __restrict__ nn_idx, const float* __restrict__ nn_d2,
                               const uint32_t* __restrict__ hd_idx, const float* __restrict__ hd_d2,
                               uint32_t n_rows, unsigned long long* __restrict__ words) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  words[i] = ((unsigned long long)__float_as_uint(nn_d2[i]) << 32) | nn_idx[i];
  words[(size_t)n_rows + i] = ((unsigned long long)__float_as_uint(hd_d2[i]) << 32) | hd_idx[i];
}
__global__ void nn_unpack_kernel(const unsigned long long* __restrict__ words, uint32_t n_rows,
                                 uint32_t* __restrict__ nn_idx, float* __restrict__ nn_d2,
                                 uint32_t* __restrict__ hd_idx, float* __restrict__ hd_d2) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const unsigned long long a = words[i], b = words[(size_t)n_rows + i];
  nn_idx[i] = (uint32_t)a;
  nn_d2[i] = __uint_as_float((uint32_t)(a >> 32));
  hd_idx[i] = (uint32_t)b;
  hd_d2[i] = __uint_as_float((uint32_t)(b >> 32));
}

__global__ void max_u32_kernel(const uint32_t* __restrict__ v, uint32_t n, uint32_t* out) {
  __shared__ uint32_t wave_max[4];
  uint32_t m = 0;
  // 16 bytes per lane and step (cudaMalloc-aligned arrays; the tail by the word)
  const uint32_t n4 = ((reinterpret_cast<uintptr_t>(v) & 15u) == 0) ? n / 4 : 0;
  const uint4* v4 = reinterpret_cast<const uint4*>(v);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    const uint4 w = v4[i];
    m = max(max(m, max(w.x, w.y)), max(w.z, w.w));
  }
  for (uint32_t i = 4 * n4 + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = max(m, v[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  // one atomic per block, and only if it can raise the word (thousands of atomics on one word serialise: 35 us at 10^6 rows)
  if (threadIdx.x == 0) {
    m = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    if (m > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, m);
  }
}

// ---- dispatch tables over n_cols ---------------------------------------------------------
using PopLaunch = void (*)(const float*, uint32_t, uint32_t, uint32_t, const Rad2&, int, uint32_t*,
                           const uint32_t*, hipStream_t);
using NnLaunch = void (*)(const float*, uint32_t, const float*, uint32_t, uint32_t, uint32_t*,
                          float*, uint32_t*, float*, const uint32_t*, hipStream_t);

template <int D, int NR>
void pop_launch(const float* c, uint32_t n, uint32_t i_from, uint32_t i_to, const Rad2& rad2,
                int n_rad, uint32_t* pops, const uint32_t* gate, hipStream_t s) {
  const uint32_t per_block = kBlock * Cfg<D>::Q;
  const uint32_t grid = (i_to - i_from + per_block - 1) / per_block;
  hipLaunchKernelGGL((pop_direct_kernel<D, NR>), dim3(grid), dim3(kBlock), 0, s, c, n, i_from,
                     i_to, rad2, n_rad, pops, gate);
}

template <int D>
void nn_launch(const float* c, uint32_t n, const float* fe, uint32_t i_from, uint32_t i_to,
               uint32_t* nn_idx, float* nn_d2, uint32_t* hd_idx, float* hd_d2, const uint32_t* gate,
               hipStream_t s) {
  const uint32_t per_block = kBlock * Cfg<D>::Q;
  const uint32_t grid = (i_to - i_from + per_block - 1) / per_block;
  hipLaunchKernelGGL((nn_direct_kernel<D>), dim3(grid), dim3(kBlock), 0, s, c, n, fe, i_from, i_to,
                     nn_idx, nn_d2, hd_idx, hd_d2, gate);
}

template <int NR, int... Ds>
constexpr auto make_pop_table(std::integer_sequence<int, Ds...>) {
  return std::array<PopLaunch, sizeof...(Ds)>{&pop_launch<Ds + 1, NR>...};
}
template <int... Ds>
constexpr auto make_nn_table(std::integer_sequence<int, Ds...>) {
  return std::array<NnLaunch, sizeof...(Ds)>{&nn_launch<Ds + 1>...};
}

using DSeq = std::make_integer_sequence<int, kMaxColsTemplated>;
const auto kPop1 = make_pop_table<1>(DSeq{});
const auto kPop4 = make_pop_table<4>(DSeq{});
const auto kPop8 = make_pop_table<8>(DSeq{});
const auto kNn = make_nn_table(DSeq{});

}  // namespace

bool launch_pop_direct(const float* d_coords, uint32_t n_rows, uint32_t n_cols, uint32_t i_from,
                       uint32_t i_to, const Rad2& rad2, int n_rad, uint32_t* d_pops,
                       const uint32_t* gate, hipStream_t stream) {
  if (i_to <= i_from || n_rad <= 0) return true;
  if (n_cols >= 1 && n_cols <= (uint32_t)kMaxColsTemplated) {
    // instances exist for 1, 4 and 8 radius slots; unused slots hold -1 ("d < -1" is never true)
    const auto& tab = (n_rad == 1) ? kPop1 : (n_rad <= 4 ? kPop4 : kPop8);
    tab[n_cols - 1](d_coords, n_rows, i_from, i_to, rad2, n_rad, d_pops, gate, stream);
    return true;
  }
  if (n_cols > (uint32_t)kMaxColsTemplated && n_cols <= (uint32_t)kMaxColsGeneric) {
    const uint32_t grid = (i_to - i_from + kGBlock - 1) / kGBlock;
    const size_t smem = sizeof(float) * (size_t)n_cols * (kGBlock + kGTile);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pop_generic_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(pop_generic_kernel, dim3(grid), dim3(kGBlock), smem, stream, d_coords,
                       n_rows, n_cols, i_from, i_to, rad2, n_rad, d_pops, gate);
    return true;
  }
  return false;
}

bool launch_nn_direct(const float* d_coords, uint32_t n_rows, uint32_t n_cols, const float* d_fe,
                      uint32_t i_from, uint32_t i_to, uint32_t* d_nn_idx, float* d_nn_d2,
                      uint32_t* d_hd_idx, float* d_hd_d2, const uint32_t* gate, hipStream_t stream) {
  if (i_to <= i_from) return true;
  if (n_cols >= 1 && n_cols <= (uint32_t)kMaxColsTemplated) {
    kNn[n_cols - 1](d_coords, n_rows, d_fe, i_from, i_to, d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2,
                    gate, stream);
    return true;
  }
  if (n_cols > (uint32_t)kMaxColsTemplated && n_cols <= (uint32_t)kMaxColsGeneric) {
    const uint32_t grid = (i_to - i_from + kGBlock - 1) / kGBlock;
    const size_t smem = sizeof(float) * ((size_t)n_cols * (kGBlock + kGTile) + kGTile);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nn_generic_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(nn_generic_kernel, dim3(grid), dim3(kGBlock), smem, stream, d_coords,
                       n_rows, n_cols, d_fe, i_from, i_to, d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2, gate);
    return true;
  }
  return false;
}

void launch_nn_init(uint32_t n_rows, uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx,
                    float* d_hd_d2, hipStream_t stream) {
  if (n_rows == 0) return;
  hipLaunchKernelGGL(nn_init_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, stream, n_rows,
                     d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2);
}

void launch_fe_gather(const uint32_t* d_pops, uint32_t n_rows, const float* d_table, float* d_fe,
                      hipStream_t stream) {
  if (n_rows == 0) return;
  hipLaunchKernelGGL(fe_gather_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, stream, d_pops,
                     n_rows, d_table, d_fe);
}

// (the call's slot of the state is zero on entry: see fe_log_kernel)
void launch_fe_log(const uint32_t* d_pops, uint32_t n_rows, uint32_t* d_state, uint32_t slot, float* d_fe, uint32_t* d_flag_list,
                   uint32_t flag_cap, double tol_rel, hipStream_t stream) {
  if (n_rows == 0) return;
  const uint32_t grid = min((n_rows + 1023u) / 1024u, 256u);
  hipLaunchKernelGGL(max_u32_kernel, dim3(grid), dim3(256), 0, stream, d_pops, n_rows, d_state + 2u * slot);
  hipLaunchKernelGGL(fe_log_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, stream, d_pops, n_rows, d_state, slot, d_fe,
                     (uint2*)d_flag_list, flag_cap, tol_rel);
}

void launch_nn_pack(const uint32_t* d_nn_idx, const float* d_nn_d2, const uint32_t* d_hd_idx,
                    const float* d_hd_d2, uint32_t n_rows, unsigned long long* d_words, hipStream_t stream) {
  if (n_rows == 0) return;
  hipLaunchKernelGGL(nn_pack_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, stream, d_nn_idx, d_nn_d2,
                     d_hd_idx, d_hd_d2, n_rows, d_words);
}
void launch_nn_unpack(const unsigned long long* d_words, uint32_t n_rows, uint32_t* d_nn_idx, float* d_nn_d2,
                      uint32_t* d_hd_idx, float* d_hd_d2, hipStream_t stream) {
  if (n_rows == 0) return;
  hipLaunchKernelGGL(nn_unpack_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, stream, d_words, n_rows,
                     d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2);
}

void launch_max_u32(const uint32_t* d_pops, uint32_t n_rows, uint32_t* d_out, hipStream_t stream) {
  (void)hipMemsetAsync(d_out, 0, sizeof(uint32_t), stream);
  if (n_rows == 0) return;
  const uint32_t grid = min((n_rows + 1023u) / 1024u, 256u);
  hipLaunchKernelGGL(max_u32_kernel, dim3(grid), dim3(256), 0, stream, d_pops, n_rows, d_out);
}

}  // namespace dc
