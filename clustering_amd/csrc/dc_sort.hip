// dc_sort.hip -- stable key/value radix sort that orders the frames (by grid cell, by (cell, free energy), by free
// energy) for the matrix-core sweeps.
//
// Rounds 1 - 4 called rocPRIM's Onesweep sort here.  At the sizes of this path (10^5 ... 10^7 pairs, 15 - 32 key bits)
// its cost is not the data -- 8 MB in, 8 MB out per pass at C3 -- but its shape: per call a histogram and a scan kernel,
// per 8-bit pass a decoupled-look-back kernel of 27 - 30 us (some 250 blocks chained through 256-entry look-back records)
// and two buffer fills of 5 us each: 120 us for the 16-bit population keys of C3, 160 us for the 24-bit neighbour keys
// (kernel trace of round 5), i.e. a third of the preparation every rank of a sharded run repeats.
// This is the classic three-kernel LSD pass instead, written for exactly this use (uint32 keys and values, at most 2^32
// items, 8 bits per pass, only the passes the caller's key bits need):
//   sort_hist_kernel     a block = 4 096 consecutive items: digit histogram in LDS -> table[digit][block]
//   sort_scan_kernel     one block per digit: exclusive scan of its table row in place, row total -> totals[digit]
//   sort_scatter_kernel  the same 4 096 items again: global base of every digit (exclusive scan of the totals + the
//                        block's table entry), the four waves' shares of the block, then every wave walks its 1 024
//                        consecutive items 64 at a time -- lanes with equal digits find each other with 8 ballots,
//                        rank = number of equal lanes below, the lowest of them advances the wave's LDS counter -- and
//                        writes key and value to their place.  Items keep their order within a digit at every level
//                        (lanes, steps, waves, blocks): stable, hence deterministic, hence the SAME order on every rank
//                        of a sharded run (the layout header of the neighbour blocks checks exactly that).
// No fills, no look-back, no library: 2 x 3 short launches for the population keys.
#include "dc_mfma.hpp"

#include <cstring>

namespace dc {

namespace {

constexpr uint32_t kSortTile = 2048;   // items per block: 4 waves x 8 steps x 64 lanes (4 096: one block per CU at C3, nothing to hide its latencies behind -- 16 us per pass against 11)
constexpr uint32_t kSortBins = 256;    // 8 bits per pass

__global__ __launch_bounds__(256) void sort_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t mask,
                                                        uint32_t* __restrict__ table, uint32_t n_blocks) {
  __shared__ uint32_t hist[kSortBins];
  hist[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile;
#pragma unroll
  for (uint32_t k = 0; k < kSortTile / 256u; k += 4) {   // four loads in flight
    uint32_t key[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      const uint32_t i = base + (k + j) * 256u + threadIdx.x;
      key[j] = (i < n) ? keys[i] : 0u;
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
      if (base + (k + j) * 256u + threadIdx.x < n) atomicAdd(&hist[(key[j] >> shift) & mask], 1u);
  }
  __syncthreads();
  table[(size_t)threadIdx.x * n_blocks + blockIdx.x] = hist[threadIdx.x];
}

// exclusive scan of table[digit][0 .. n_blocks) in place, one block per digit; the row's total -> totals[digit]
__global__ __launch_bounds__(256) void sort_scan_kernel(uint32_t* __restrict__ table, uint32_t n_blocks,
                                                        uint32_t* __restrict__ totals) {
  __shared__ uint32_t wave_sum[4];
  uint32_t* row = table + (size_t)blockIdx.x * n_blocks;
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (uint32_t j0 = 0; j0 < n_blocks; j0 += 256u) {
    const uint32_t j = j0 + threadIdx.x;
    const uint32_t v = (j < n_blocks) ? row[j] : 0u;
    uint32_t incl = v;   // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
      if ((int)lane >= off) incl += up;
    }
    if (lane == 63u) wave_sum[w] = incl;
    __syncthreads();
    uint32_t before = carry;
    for (uint32_t k = 0; k < w; ++k) before += wave_sum[k];
    if (j < n_blocks) row[j] = before + incl - v;
    carry += wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// REMAP (the last pass of an ordering): the sorted list is cut into segments [seg_start[s], seg_start[s + 1]) -- the
// components of the frames -- and segment s goes to the positions seg_base[s] + ... of a PADDED order (every component
// starts at a whole query group): only the values are written, and the tag of a tile (32 positions) is the segment of its
// first position.  Positions no item lands on keep what the caller preset them with.
template <bool REMAP>
__global__ __launch_bounds__(256) void sort_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                           const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
                                                           uint32_t* __restrict__ vals_out, uint32_t n, uint32_t shift, uint32_t mask,
                                                           const uint32_t* __restrict__ table, const uint32_t* __restrict__ totals,
                                                           uint32_t n_blocks, SortRemap remap) {
  __shared__ uint32_t cnt[4][kSortBins];   // per wave: its items of every digit, then the place of its next item of the digit
  __shared__ uint32_t scan_tmp[4];
  __shared__ uint32_t seg_s[kSortMaxSegments + 1], base_s[kSortMaxSegments + 1];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6, t = threadIdx.x;
  if constexpr (REMAP) {
    if (t <= remap.n_seg) {
      seg_s[t] = remap.seg_start[t];
      base_s[t] = remap.seg_base[t];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) cnt[k][t] = 0u;
  __syncthreads();
  // the wave's 1 024 consecutive items: step s holds items base + 64 s + lane
  const uint32_t base = blockIdx.x * kSortTile + w * (kSortTile / 4u);
  constexpr int kSteps = kSortTile / 4 / 64;
  uint32_t key[kSteps];
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    const uint32_t i = base + 64u * (uint32_t)s + lane;
    key[s] = (i < n) ? keys_in[i] : 0u;
  }
#pragma unroll
  for (int s = 0; s < kSteps; ++s)
    if (base + 64u * (uint32_t)s + lane < n) atomicAdd(&cnt[w][(key[s] >> shift) & mask], 1u);
  // global base of digit t: exclusive scan of the totals over the digits + what the blocks before this one hold of it
  const uint32_t tot = totals[t];
  uint32_t incl = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
    if ((int)lane >= off) incl += up;
  }
  if (lane == 63u) scan_tmp[w] = incl;
  __syncthreads();   // (also: every wave's counts are in)
  uint32_t place = incl - tot + table[(size_t)t * n_blocks + blockIdx.x];
  for (uint32_t k = 0; k < w; ++k) place += scan_tmp[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) {   // the waves' shares of the block's items of digit t, in wave order
    const uint32_t c = cnt[k][t];
    cnt[k][t] = place;
    place += c;
  }
  __syncthreads();
  volatile uint32_t* mine = cnt[w];
  const uint64_t below = (lane == 0u) ? 0ull : (~0ull >> (64u - lane));
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    const uint32_t i = base + 64u * (uint32_t)s + lane;
    const bool live = i < n;
    const uint32_t d = (key[s] >> shift) & mask;
    uint64_t peers = __builtin_amdgcn_ballot_w64(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t has = __builtin_amdgcn_ballot_w64(live && bit);
      peers &= bit ? has : ~has;
    }
    const uint32_t rank = (uint32_t)__builtin_popcountll(peers & below);
    uint32_t at = 0;
    if (live) at = mine[d];
    __builtin_amdgcn_wave_barrier();
    if (live && rank == 0u) mine[d] = at + (uint32_t)__builtin_popcountll(peers);
    __builtin_amdgcn_wave_barrier();
    if (live) {
      const uint32_t dst = at + rank;
      if constexpr (REMAP) {
        // the last segment that starts at or before dst (empty segments share their start with the next one)
        uint32_t lo = 0, hi = remap.n_seg;   // invariant: seg_s[lo] <= dst, hi: first candidate beyond
        while (hi - lo > 1u) {
          const uint32_t mid = (lo + hi) >> 1;
          if (seg_s[mid] <= dst) lo = mid; else hi = mid;
        }
        const uint32_t pos = base_s[lo] + (dst - seg_s[lo]);
        vals_out[pos] = vals_in[i];
        if ((pos & 31u) == 0u) remap.tags[pos >> 5] = lo;
      } else {
        keys_out[dst] = key[s];
        vals_out[dst] = vals_in[i];
      }
    }
  }
}

inline uint32_t sort_blocks(size_t n) { return (uint32_t)((n + kSortTile - 1) / kSortTile); }
inline size_t sort_table_bytes(size_t n) { return (((size_t)kSortBins * sort_blocks(n) + kSortBins) * sizeof(uint32_t) + 255) & ~(size_t)255; }

}  // namespace

// temp: the digit table + totals, then one more (key, value) buffer pair for the passes in between
size_t sort_temp_bytes(size_t n) { return sort_table_bytes(n) + 2 * (((n * sizeof(uint32_t)) + 255) & ~(size_t)255); }

int sort_pairs_u32(uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t n,
                   void* temp, size_t temp_bytes, hipStream_t stream, unsigned key_bits, const SortRemap* remap) {
  if (remap && (remap->n_seg == 0 || remap->n_seg > kSortMaxSegments)) return -1;
  if (n == 0) return 0;
  if (n > 0xFFFFFFFFull || temp_bytes < sort_temp_bytes(n)) return -1;
  if (key_bits > 32u) key_bits = 32u;
  const unsigned passes = key_bits == 0 ? 1u : (key_bits + 7u) / 8u;
  char* tp = (char*)temp;
  uint32_t* table = (uint32_t*)tp;
  const uint32_t nb = sort_blocks(n);
  uint32_t* totals = table + (size_t)kSortBins * nb;
  const size_t buf = ((n * sizeof(uint32_t)) + 255) & ~(size_t)255;
  uint32_t* tkeys = (uint32_t*)(tp + sort_table_bytes(n));
  uint32_t* tvals = (uint32_t*)(tp + sort_table_bytes(n) + buf);
  const uint32_t* src_k = keys_in;
  const uint32_t* src_v = vals_in;
  for (unsigned p = 0; p < passes; ++p) {
    // the last pass writes the caller's output -- and ONLY the last: with a remap the output is a padded order whose
    // untouched positions must keep their presets; the passes before alternate between the temp pair and the INPUT
    // buffers (scratch of the callers: their contents are lost)
    const bool last = p + 1u == passes;
    uint32_t* dst_k = last ? keys_out : (src_k == keys_in ? tkeys : keys_in);
    uint32_t* dst_v = last ? vals_out : (src_v == vals_in ? tvals : vals_in);
    const uint32_t shift = 8u * p;
    // (only the low key_bits bits order the items: the last pass may hold fewer than 8 of them)
    const uint32_t left = key_bits > shift ? key_bits - shift : 0u, mask = left >= 8u ? 0xFFu : ((1u << left) - 1u);
    hipLaunchKernelGGL(sort_hist_kernel, dim3(nb), dim3(256), 0, stream, src_k, (uint32_t)n, shift, mask, table, nb);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(kSortBins), dim3(256), 0, stream, table, nb, totals);
    if (remap && p + 1u == passes)
      hipLaunchKernelGGL(sort_scatter_kernel<true>, dim3(nb), dim3(256), 0, stream, src_k, src_v, dst_k, dst_v, (uint32_t)n, shift,
                         mask, (const uint32_t*)table, (const uint32_t*)totals, nb, *remap);
    else
      hipLaunchKernelGGL(sort_scatter_kernel<false>, dim3(nb), dim3(256), 0, stream, src_k, src_v, dst_k, dst_v, (uint32_t)n, shift,
                         mask, (const uint32_t*)table, (const uint32_t*)totals, nb, SortRemap{nullptr, nullptr, 0u, nullptr});
    src_k = dst_k;
    src_v = dst_v;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace dc
