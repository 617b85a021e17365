// dc_prep.hpp -- the preparation passes of the PRUNED sweeps, round 5 (included by dc_mfma.hip inside namespace
// dc::{anonymous}, behind the kernels of rounds 1 - 4 it still shares code with).
//
// Every rank of a sharded run repeats the preparation of a sweep in full -- statistics, components, ordering keys, the
// sort, the padded order, the gathered rows, the operand images -- while its sweep shrinks with the rank count: at an
// eighth of C3 the preparation was 0.72 ms of a 4.06 ms step (round 4), some forty launches of which half ran for the
// 5 us a launch costs whatever it does.  This file is the same preparation in a dozen launches:
//   stats_kernel        ONE pass over the coordinates: column sums, bounding box of columns 0/1, the flag for non-finite
//                       or overflow-prone data, the content fingerprint (colsum + mean + rowstats + a fill before)
//   fine_mark_kernel    (unchanged) occupancy bitmap of the sub-cells
//   components_kernel   + the coarse cells' boxes in front, the column means and the fine cell grids behind: one launch
//                       for four
//   order_key_kernel    ordering key and value of every row (compkey_kernel's arithmetic) + the rows per component +
//                       the extents max |x - origin(component)|^2 and max |x - mean|^2 (from order_rows / rowstats) +
//                       the pad presets of the order
//   order_meta_kernel   where every component starts in the sorted list and in the padded order (comp_start + comp_ranges)
//                       and the scale of the sweep (scale_kernel)
//   the sort            dc_sort.hip: its last pass writes straight into the PADDED order (pad_scatter and a fill before)
//   order_rows2_kernel  rows gathered into the order, tile boxes, free-energy ranges, AND the operand images of the
//                       tiles -- A form of every tile, B form of the query groups of this launch's segment -- built from
//                       the rows while they sit in LDS (order_rows + scale + two image launches)
// The arithmetic of every value is what the kernels of rounds 1 - 4 computed; only who computes it when has changed.
#pragma once

constexpr float kStatsLimit = 5.0e16f;   // |x_k| beyond this: |x - mean|^2 could pass kNormLimit = 1e36 (64 columns x (2 x 5e16)^2 = 6.4e35)

// ONE pass over the coordinates, element-wise (the thread's column is fixed: the stride is a multiple of n_cols): column
// sums (double), extent of columns 0/1 (header words 8..11), the flag for non-finite / overflow-prone data (word 1,
// bit 0), the content fingerprint, the cookie.  The header was zero-filled before.
// Many short blocks (the pass is bound by the latency of its loads: 37 us with 512 blocks of 19 trips, four loads each),
// and NO same-address atomics at their ends: every block leaves its column sums and its share of the fingerprint in a
// table [kStatsRow][blocks] doubles (the sort's temp region, free until the sort), which one workgroup adds up right behind
// it (stats_reduce_kernel).  The extents of columns 0/1 go the same way (raised with
// atomicMax on the four header words they cost 20 us at 10^5 rows: a thousand blocks queueing on four addresses).
constexpr uint32_t kStatsRow = kMaxCols + 3;   // per block: kMaxCols column sums, the fingerprint share, the extents of columns 0 and 1 (64-bit slots)
constexpr uint32_t kStatsMaxBlocks = 2048;
__global__ __launch_bounds__(256) void stats_kernel(const float* __restrict__ coords, uint32_t n_rows, uint32_t D,
                                                    uint32_t* __restrict__ hdr, uint32_t cookie, double* __restrict__ table) {
  __shared__ double part[kMaxCols];
  __shared__ uint32_t wave_max[4];
  __shared__ unsigned long long fp_part[4];
  if (threadIdx.x < (uint32_t)kMaxCols) part[threadIdx.x] = 0.0;
  __syncthreads();
  const uint32_t nthreads = gridDim.x * blockDim.x;
  const uint32_t used = (nthreads / D) * D;
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n_rows * D;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(coords);
  const uint32_t col = id % D;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  uint32_t m_lo = 0, m_hi = 0;
  unsigned long long fp = 0;
  bool bad = false;
  auto take = [&](uint32_t bits, size_t e, double& s) {
    const float v = __uint_as_float(bits);
    fp += fp_term(bits, e);
    const bool fin = fabsf(v) <= kStatsLimit;
    bad = bad | !fin;
    if (fin) {
      s += (double)v;
      m_lo = max(m_lo, ~fkey(v));
      m_hi = max(m_hi, fkey(v));
    }
  };
  if (id < used) {
    size_t e = id;
    for (; e + 3 * (size_t)used < total; e += 4 * (size_t)used) {   // four loads in flight
      const uint32_t b0 = w[e], b1 = w[e + used], b2 = w[e + 2 * (size_t)used], b3 = w[e + 3 * (size_t)used];
      take(b0, e, s0);
      take(b1, e + used, s1);
      take(b2, e + 2 * (size_t)used, s2);
      take(b3, e + 3 * (size_t)used, s3);
    }
    for (; e < total; e += used) take(w[e], e, s0);
    atomicAdd(&part[col], (s0 + s1) + (s2 + s3));
  }
  // the block's share of the fingerprint (wrap-around sum: any grouping gives the same total)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)fp, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(fp >> 32), off, 64);
    fp += ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & 63) == 0) fp_part[threadIdx.x >> 6] = fp;
  __syncthreads();
  if (threadIdx.x < D) table[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = part[threadIdx.x];
  if (threadIdx.x == 0)
    reinterpret_cast<unsigned long long*>(table)[(size_t)kMaxCols * gridDim.x + blockIdx.x] = fp_part[0] + fp_part[1] + fp_part[2] + fp_part[3];
  if (bad) atomicOr(hdr + 1, 1u);
  {  // the block's extents of columns 0 / 1: ~key(min), key(max) each (0: the block saw nothing of the column)
    const bool c0 = id < used && col == 0u, c1 = id < used && col == 1u;
    uint32_t e[4] = {c0 ? m_lo : 0u, c0 ? m_hi : 0u, c1 ? m_lo : 0u, c1 ? m_hi : 0u};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) e[q] = max(e[q], (uint32_t)__shfl_xor((int)e[q], off, 64));
    }
    __shared__ uint32_t ext_s[4][4];
    if ((threadIdx.x & 63) == 0)
      for (int q = 0; q < 4; ++q) ext_s[q][threadIdx.x >> 6] = e[q];
    __syncthreads();
    if (threadIdx.x < 2) {
      const uint32_t q = 2u * threadIdx.x;
      const uint32_t lo = max(max(ext_s[q][0], ext_s[q][1]), max(ext_s[q][2], ext_s[q][3]));
      const uint32_t hi = max(max(ext_s[q + 1][0], ext_s[q + 1][1]), max(ext_s[q + 1][2], ext_s[q + 1][3]));
      reinterpret_cast<unsigned long long*>(table)[(size_t)(kMaxCols + 1 + threadIdx.x) * gridDim.x + blockIdx.x] =
          ((unsigned long long)hi << 32) | lo;
    }
  }
  (void)wave_max;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    hdr[kHdrCookie] = cookie;   // whose statistics these are (DC_FLAG_STATS_VALID is checked against it)
    hdr[kHdrStatsBlocks] = gridDim.x;   // (rows of the table still to be added up: stats_reduce clears the word)
  }
}
// The table of stats_kernel added up by one workgroup: column sums -> header (kHdrSums), extents -> words 8..11,
// fingerprint -> kHdrFp, column means -> kHdrMeans (mean_kernel's arithmetic).  A header whose statistics are already
// complete (kHdrStatsBlocks == 0: the statistics of an earlier call, DC_FLAG_STATS_VALID) is left alone.
__device__ void stats_reduce(uint32_t* __restrict__ hdr, const double* __restrict__ table, uint32_t n_rows, uint32_t D) {
  const uint32_t nb = hdr[kHdrStatsBlocks];
  __syncthreads();   // (every thread has read the word before it is cleared)
  if (nb == 0u) return;
  // one wave per column (the fingerprint slot is column kMaxCols): lane l adds the blocks l, l + 64, ... in that order, the
  // lanes meet in a fixed tree.  (The table's entries themselves are sums of LDS double atomics in arrival order:
  // the column sums are reproducible to rounding, not bit for bit -- they only set the origin, which no result depends on.)
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  for (uint32_t c = wave; c <= D + 2u; c += n_waves) {
    const uint32_t col = (c >= D) ? (uint32_t)kMaxCols + (c - D) : c;
    if (col > (uint32_t)kMaxCols) {   // extent of column 0 / 1: (key(max) << 32 | ~key(min)), both maxima
      const unsigned long long* t = reinterpret_cast<const unsigned long long*>(table) + (size_t)col * nb;
      uint32_t lo = 0, hi = 0;
      for (uint32_t b = lane; b < nb; b += 64u) {
        const unsigned long long v = t[b];
        lo = max(lo, (uint32_t)v);
        hi = max(hi, (uint32_t)(v >> 32));
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        lo = max(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
      }
      if (lane == 0) {
        const uint32_t which = col - (uint32_t)kMaxCols - 1u;
        if (which == 1u && D == 1u) {   // (a second column of zeros, as the sweeps treat it)
          lo = ~fkey(0.0f);
          hi = fkey(0.0f);
        }
        hdr[8 + 2 * which] = lo;
        hdr[9 + 2 * which] = hi;
      }
    } else if (col == (uint32_t)kMaxCols) {
      const unsigned long long* t = reinterpret_cast<const unsigned long long*>(table) + (size_t)col * nb;
      unsigned long long f = 0;
      for (uint32_t b0 = lane; b0 < nb; b0 += 512u) {   // (eight loads in flight: one per trip was 10 us of latency)
        unsigned long long v[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) v[q] = (b0 + 64u * q < nb) ? t[b0 + 64u * q] : 0ull;
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) f += v[q];
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)f, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(f >> 32), off, 64);
        f += ((unsigned long long)hi << 32) | lo;
      }
      if (lane == 0) *reinterpret_cast<unsigned long long*>(hdr + kHdrFp) = f;
    } else {
      const double* t = table + (size_t)col * nb;
      double sum = 0.0;
      for (uint32_t b0 = lane; b0 < nb; b0 += 512u) {
        double v[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) v[q] = (b0 + 64u * q < nb) ? t[b0 + 64u * q] : 0.0;
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) sum += v[q];
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const long long bits = __double_as_longlong(sum);
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)bits, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)((unsigned long long)bits >> 32), off, 64);
        sum += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
      }
      if (lane == 0) {
        reinterpret_cast<double*>(reinterpret_cast<char*>(hdr) + kHdrSums)[col] = sum;
        float muf = (float)(sum / (double)n_rows);   // column means as the float the centring subtracts
        if (!(fabsf(muf) <= FLT_MAX)) muf = 0.0f;
        reinterpret_cast<float*>(reinterpret_cast<char*>(hdr) + kHdrMeans)[col] = muf;
      }
    }
  }
  if (threadIdx.x == 0) hdr[kHdrStatsBlocks] = 0u;
  __syncthreads();
}

// (its own launch, between the statistics pass and fine_mark_kernel, which needs the extents)
__global__ __launch_bounds__(1024) void stats_reduce_kernel(uint32_t* __restrict__ hdr, const double* __restrict__ table,
                                                           uint32_t n_rows, uint32_t D) {
  stats_reduce(hdr, table, n_rows, D);
}

// ---- DC_FLAG_STATS_VALID: the claim checked in two launches (were five: reset, fingerprint, guard, free-energy range,
// component guard) ---------------------------------------------------------------------------------------------------
// claim_pre_kernel: every block leaves its share of the content fingerprint of the coordinates and -- fe != nullptr, a
// neighbour call -- the extremes of the free energies it saw in a table (no atomics, nothing to clear before):
//   tab[b] fingerprint share, tab[B + b] = key of the largest finite free energy << 32 | ~key of the smallest, tab[2 B + b] NaN seen
constexpr uint32_t kClaimBlocks = 1024;
__global__ __launch_bounds__(256) void claim_pre_kernel(const float* __restrict__ coords, size_t total, const float* __restrict__ fe,
                                                        uint32_t n_rows, unsigned long long* __restrict__ tab) {
  __shared__ unsigned long long part[4];
  __shared__ uint32_t wmax[2][4];
  __shared__ uint32_t nan_s;
  if (threadIdx.x == 0) nan_s = 0u;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(coords);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long f = 0;
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; e + 3 * stride < total; e += 4 * stride) {   // four loads in flight
    const uint32_t v0 = w[e], v1 = w[e + stride], v2 = w[e + 2 * stride], v3 = w[e + 3 * stride];
    f += fp_term(v0, e) + fp_term(v1, e + stride) + fp_term(v2, e + 2 * stride) + fp_term(v3, e + 3 * stride);
  }
  for (; e < total; e += stride) f += fp_term(w[e], e);
  uint32_t inv = 0, top = 0;
  bool nan = false;
  if (fe) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += stride) {
      const float v = fe[i];
      nan = nan | (v != v);
      const uint32_t u = __float_as_uint(v);
      const uint32_t key = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
      inv = max(inv, ~key);
      top = max(top, (fabsf(v) <= FLT_MAX) ? key : 0u);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)f, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(f >> 32), off, 64);
    f += ((unsigned long long)hi << 32) | lo;
    inv = max(inv, (uint32_t)__shfl_xor((int)inv, off, 64));
    top = max(top, (uint32_t)__shfl_xor((int)top, off, 64));
  }
  __syncthreads();
  if (nan) nan_s = 1u;
  if ((threadIdx.x & 63) == 0) {
    part[threadIdx.x >> 6] = f;
    wmax[0][threadIdx.x >> 6] = inv;
    wmax[1][threadIdx.x >> 6] = top;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t B = gridDim.x;
    tab[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
    const uint32_t i4 = max(max(wmax[0][0], wmax[0][1]), max(wmax[0][2], wmax[0][3]));
    const uint32_t t4 = max(max(wmax[1][0], wmax[1][1]), max(wmax[1][2], wmax[1][3]));
    tab[B + blockIdx.x] = ((unsigned long long)t4 << 32) | i4;
    tab[2 * B + blockIdx.x] = nan_s;
  }
}
// claim_guard_kernel (one workgroup): the shares added up; the claim holds iff cookie (array address, shape) and fingerprint
// equal what the statistics pass stored -- otherwise the data is flagged (flag word 1: bit 0 = non-finite / overflow-prone
// coordinates, a statistic: it stays; bit 1 = the claim failed; bit 2 = NaN free energies) and the matrix-core kernels
// stand down: slow, never wrong.  The per-sweep words of the header start over (sweep_words_reset_kernel); the free-energy
// range of a neighbour call goes to words 12 / 13 (fe_key_kernel); and, comp != nullptr, the component partition of the
// workspace is checked for this sweep and its fine grids are formed (comp_guard_kernel).
__device__ void comp_guard_body(const uint32_t* __restrict__ hdr, const float* __restrict__ means, uint32_t D,
                                uint32_t* __restrict__ comp, uint32_t cookie, uint32_t n_rows, float fine_frames_per_cell,
                                uint32_t fine_bits);
__global__ __launch_bounds__(1024) void claim_guard_kernel(uint32_t* __restrict__ hdr, uint32_t cookie,
                                                          const unsigned long long* __restrict__ tab, uint32_t B, int pruned,
                                                          int have_fe, uint32_t D, uint32_t* __restrict__ comp, uint32_t n_rows,
                                                          float fine_frames_per_cell, uint32_t fine_bits) {
  __shared__ unsigned long long part[16];
  __shared__ uint32_t wmax[3][16];
  unsigned long long f = 0;
  uint32_t inv = 0, top = 0, nan = 0;
  for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) {
    f += tab[b];
    const unsigned long long x = tab[B + b];
    inv = max(inv, (uint32_t)x);
    top = max(top, (uint32_t)(x >> 32));
    nan |= (uint32_t)tab[2 * B + b];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)f, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(f >> 32), off, 64);
    f += ((unsigned long long)hi << 32) | lo;
    inv = max(inv, (uint32_t)__shfl_xor((int)inv, off, 64));
    top = max(top, (uint32_t)__shfl_xor((int)top, off, 64));
    nan |= (uint32_t)__shfl_xor((int)nan, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    part[threadIdx.x >> 6] = f;
    wmax[0][threadIdx.x >> 6] = inv;
    wmax[1][threadIdx.x >> 6] = top;
    wmax[2][threadIdx.x >> 6] = nan;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t nw = blockDim.x >> 6;
    f = 0;
    inv = top = nan = 0;
    for (uint32_t k = 0; k < nw; ++k) {
      f += part[k];
      inv = max(inv, wmax[0][k]);
      top = max(top, wmax[1][k]);
      nan |= wmax[2][k];
    }
    const bool same = hdr[kHdrCookie] == cookie && *reinterpret_cast<const unsigned long long*>(hdr + kHdrFp) == f;
    *reinterpret_cast<unsigned long long*>(hdr + kHdrFp + 2) = f;
    hdr[1] = (hdr[1] & 1u) | (same ? 0u : 2u) | ((have_fe && nan) ? 4u : 0u);
    if (!same) hdr[kHdrCookie] = 0u;   // (the component partition in the workspace is not this array's either)
    // the per-sweep words: evaluated-tile and MFMA counters, the extents this sweep forms again, the verdict of the last
    // block unpack (kHdrLayoutBad = kHdrFp + 4) and the spare word behind it
    if (pruned) hdr[0] = 0u;
    for (uint32_t k = 2; k <= 7; ++k) hdr[k] = 0u;
    hdr[kHdrMfmaNn] = 0u;
    hdr[kHdrMfmaNn + 1] = 0u;
    hdr[kHdrMloc] = 0u;
    hdr[kHdrFp + 4] = 0u;
    hdr[kHdrFp + 5] = 0u;
    hdr[12] = have_fe ? inv : 0u;
    hdr[13] = have_fe ? top : 0u;
  }
  if (comp) {
    __syncthreads();
    comp_guard_body(hdr, reinterpret_cast<const float*>(reinterpret_cast<const char*>(hdr) + kHdrMeans), D, comp, cookie, n_rows,
                    fine_frames_per_cell, fine_bits);
  }
}

// ---- one workgroup: coarse boxes -> components -> fine grids ----------------------------------------------------
// (components_kernel and fine_grid_kernel of dc_mfma.hip do the work; these are their launches fused)
__device__ void fine_grid_body(const uint32_t* __restrict__ hdr, uint32_t n_rows, float frames_per_cell, uint32_t fine_bits,
                               uint32_t* __restrict__ comp);

// Ordering key and value of the rows [i_from, i_to) (compkey_kernel's arithmetic), and in the same pass:
//   counts        rows of every component among the block's rows: a table [block][kMaxComp], added up by order_meta_kernel
//   measure       the extents of ALL rows: max |x - origin(component of x)|^2 -> hdr[kHdrMloc], max |x - mean|^2 ->
//                 hdr[0] (float accumulation with a margin, as order_rows_kernel formed the first one in rounds 3 - 4)
//   presets       the padded order of n_pos positions: every position kInvalidFrame, every tile the all-pad component
//                 (the sort's last pass writes the real entries over them)
static inline size_t order_key_smem(uint32_t n_cols, bool measure) {
  return measure ? sizeof(float) * ((size_t)kMaxComp + 1) * n_cols : 0;
}
__global__ __launch_bounds__(256) void order_key_kernel(
    const float* __restrict__ coords, uint32_t D, uint32_t* __restrict__ hdr, const uint32_t* __restrict__ comp,
    uint32_t fine_bits, uint32_t i_from, uint32_t i_to, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
    uint32_t n_total, const float* __restrict__ fe, uint32_t fe_bits, uint32_t* __restrict__ counts, int measure,
    uint32_t* __restrict__ perm, uint32_t* __restrict__ tile_comp, uint32_t n_pos) {
  // A block walks tiles of 256 rows (grid-stride) and ends with ONE atomic per component and extent word: with a block per
  // tile the 4 000 blocks of C3 queued 10 000 atomics on three counters -- 50 of the pass's 60 us.  A row per lane, read
  // where it lies (the loads of a wave touch the same 2.5 KB and meet in the vector cache); the origins and the means in LDS.
  extern __shared__ float org_s[];   // measure: [kMaxComp][D] origins, then the means [D]
  __shared__ uint32_t cnt_s[kMaxComp];
  __shared__ float blk_max[2][4];
  // the tables a key goes through, in LDS: component of a coarse cell (a byte each, the grid's ncx * ncy cells), fine grid
  // of a component -- from global memory they were three dependent round trips per row
  constexpr uint32_t kCellCap = 8192;
  __shared__ unsigned char cellc_s[kCellCap];
  __shared__ uint32_t par_s[kMaxComp][8];   // bits(lo0), bits(lo1), bits(c0), bits(c1), nby, first cell, first cell of the next
  if (threadIdx.x < (uint32_t)kMaxComp) cnt_s[threadIdx.x] = 0u;
  const uint32_t n_comp = min(comp[kCompGrid + 5], (uint32_t)kMaxComp);
  const uint32_t n_cells = comp[kCompGrid + 3] * comp[kCompGrid + 4];
  const bool cells_in_lds = n_cells <= kCellCap;
  if (cells_in_lds)
    for (uint32_t e = threadIdx.x; e < n_cells; e += 256u) cellc_s[e] = (unsigned char)min(comp[kCompCellComp + e], (uint32_t)kMaxComp - 1u);
  if (threadIdx.x < (uint32_t)kMaxComp) {
    const uint32_t c = threadIdx.x;
    const uint32_t* f = comp + kCompFine + 4 * (size_t)c;
    par_s[c][0] = f[0];
    par_s[c][1] = f[1];
    par_s[c][2] = f[2];
    par_s[c][3] = f[3];
    par_s[c][4] = comp[kCompNby + c];
    par_s[c][5] = comp[kCompCellOff + c];
    par_s[c][6] = comp[kCompCellOff + c + 1];
  }
  if (measure) {
    const float* origins = reinterpret_cast<const float*>(comp + kCompOrigin);
    const float* mu = reinterpret_cast<const float*>(reinterpret_cast<const char*>(hdr) + kHdrMeans);
    for (uint32_t e = threadIdx.x; e < n_comp * D; e += 256u) org_s[e] = origins[(size_t)(e / D) * kMaxCols + (e - (e / D) * D)];
    for (uint32_t k = threadIdx.x; k < D; k += 256u) org_s[(size_t)kMaxComp * D + k] = mu[k];
  }
  __syncthreads();
  const CoarseGrid g = coarse_grid(hdr, comp_r_conn(comp), n_total);
  const float fe_lo = fe ? fkey_inv(~hdr[12]) : 0.0f, fe_hi = fe ? fkey_inv(hdr[13]) : 0.0f;
  const uint32_t n_items = max(i_to - i_from, n_pos);
  float ext_loc = 0.0f, ext_glob = 0.0f;
  for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n_items; j += gridDim.x * 256u) {
    if (j < n_pos) perm[j] = kInvalidFrame;
    if (j < n_pos / 32u) tile_comp[j] = kMaxComp;
    const uint32_t i = i_from + j;
    if (i >= i_to) continue;
    const float* row = coords + (size_t)i * D;
    const float x = row[0], y = (D > 1) ? row[1] : 0.0f;
    uint32_t c = 0, bx = 0, by = 0, nby = 1;
    if (fabsf(x) <= FLT_MAX && fabsf(y) <= FLT_MAX) {
      const uint32_t cell = coarse_cell_of_point(g, x, y);
      c = min(cells_in_lds ? (uint32_t)cellc_s[min(cell, kCellCap - 1u)] : comp[kCompCellComp + cell], (uint32_t)kMaxComp - 1u);
      const float lo0 = __uint_as_float(par_s[c][0]), lo1 = __uint_as_float(par_s[c][1]);
      const float c0 = __uint_as_float(par_s[c][2]), c1 = __uint_as_float(par_s[c][3]);
      nby = par_s[c][4];
      const float fx = fminf(fmaxf((x - lo0) / c0, 0.0f), 4001.0f), fy = fminf(fmaxf((y - lo1) / c1, 0.0f), 4001.0f);
      bx = (uint32_t)fx;
      by = min((uint32_t)fy, nby - 1u);
    }
    const uint32_t lo = par_s[c][5], hi = par_s[c][6];
    // (the cells of a component are numbered column by column, every other column downwards: consecutive cells are
    //  always neighbours.  Numbered upwards in every column, a query group that began at the top of one column and ended at
    //  the bottom of the next had a box as tall as the component -- gap 0 to every tile of two columns, a first ring of
    //  hundreds of tiles in index order: the handful of such groups were the last waves of every sharded neighbour sweep,
    //  2 ms each at C3 where the mean wave takes 0.16)
    const uint32_t by_s = (bx & 1u) ? nby - 1u - by : by;
    uint32_t key = min(lo + bx * nby + by_s, hi - (hi > lo ? 1u : 0u));
    if (fe) {
      const float span = fe_hi - fe_lo;
      float u = (span > 0.0f && span <= FLT_MAX) ? (fe[i] - fe_lo) / span : 0.0f;
      u = fminf(fmaxf(u, 0.0f), 1.0f);                                // (+inf -> 1, -inf / NaN -> 0)
      const uint32_t levels = (1u << fe_bits) - 1u;
      const uint32_t level = (uint32_t)((double)u * (double)levels);
      key = (key << fe_bits) | level;
    }
    keys[j] = key;
    vals[j] = i;
    atomicAdd(&cnt_s[min(c, (uint32_t)kMaxComp - 1u)], 1u);
    if (measure) {
      const float* a = org_s + (size_t)min(c, n_comp ? n_comp - 1u : 0u) * D;
      const float* mu = org_s + (size_t)kMaxComp * D;
      float el = 0.0f, eg = 0.0f;
      for (uint32_t k0 = 0; k0 < D; k0 += 4) {   // (four columns per step, their loads issued together)
        float xv[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) xv[q] = row[min(k0 + q, D - 1u)];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q)
          if (k0 + q < D) {
            const float vl = xv[q] - a[k0 + q], vg = xv[q] - mu[k0 + q];
            el += vl * vl;
            eg += vg * vg;
          }
      }
      el = el * 1.0001f + FLT_MIN;
      eg = eg * 1.0001f + FLT_MIN;
      if (el <= FLT_MAX) ext_loc = fmaxf(ext_loc, el);     // (non-finite rows: the data is flagged, the sweep stands down)
      if (eg <= FLT_MAX) ext_glob = fmaxf(ext_glob, eg);
    }
  }
  if (measure) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ext_loc = fmaxf(ext_loc, __shfl_xor(ext_loc, off, 64));
      ext_glob = fmaxf(ext_glob, __shfl_xor(ext_glob, off, 64));
    }
    if ((threadIdx.x & 63u) == 0) {
      blk_max[0][threadIdx.x >> 6] = ext_loc;
      blk_max[1][threadIdx.x >> 6] = ext_glob;
    }
  }
  __syncthreads();
  if (measure && threadIdx.x < 2) {
    const float m = fmaxf(fmaxf(blk_max[threadIdx.x][0], blk_max[threadIdx.x][1]), fmaxf(blk_max[threadIdx.x][2], blk_max[threadIdx.x][3]));
    uint32_t* dst = threadIdx.x == 0 ? hdr + kHdrMloc : hdr;
    const uint32_t bits = __float_as_uint(m);
    if (bits > __atomic_load_n(dst, __ATOMIC_RELAXED)) atomicMax(dst, bits);
  }
  // the block's rows per component: a row of the table [block][kMaxComp] (order_meta_kernel adds the rows up: three
  // atomics per block on the same three words were a third of this pass at C3)
  if (threadIdx.x < (uint32_t)kMaxComp) counts[(size_t)blockIdx.x * kMaxComp + threadIdx.x] = cnt_s[threadIdx.x];
}

// (what the kernels of this file assume about the component count: a wave's lane indexes a component -- part[l][threadIdx & 63],
//  table[b * kMaxComp + c] --, order_key_kernel keeps components in an unsigned char, and the sort's remapped last pass takes
//  kMaxComp segments)
static_assert(kMaxComp == 64 && kMaxComp <= (int)kSortMaxSegments && kMaxComp <= 256, "order_key_kernel / order_meta_kernel / SortRemap are written for 64 components");

// One workgroup: the rows per component (the table order_key_kernel left: [n_blocks][kMaxComp]) added up, then -- one thread --
// the first sorted index of every component (start), the tile range of every component in the padded order (range; entry
// kMaxComp: the empty range of the all-pad tiles) and the first position of every component (base): comp_start_kernel +
// comp_ranges_kernel of rounds 3 - 4.  do_scale: also the scale of the sweep that follows (scale_kernel): r2max < 0 the
// neighbour rule, else the population rule.
__global__ __launch_bounds__(1024) void order_meta_kernel(uint32_t* __restrict__ hdr, uint32_t* __restrict__ comp,
                                                         const uint32_t* __restrict__ table, uint32_t n_blocks,
                                                         uint32_t* __restrict__ start /* [kMaxComp + 1] */,
                                                         uint32_t* __restrict__ range /* [kMaxComp + 1][2] */,
                                                         uint32_t* __restrict__ base /* [kMaxComp + 1] */, uint32_t n,
                                                         uint32_t group_rows, int do_scale, float r2max, uint32_t D,
                                                         int shift_steps = 0) {
  __shared__ uint32_t part[16][kMaxComp];
  __shared__ uint32_t cnt[kMaxComp];
  {
    const uint32_t c = threadIdx.x & 63u, l = threadIdx.x >> 6;   // 16 groups of 64 threads: component c, blocks l, l + 16, ..
    uint32_t sum = 0;
    for (uint32_t b0 = l; b0 < n_blocks; b0 += 64u) {   // (four loads in flight)
      uint32_t v[4];
#pragma unroll
      for (uint32_t q = 0; q < 4; ++q) v[q] = (b0 + 16u * q < n_blocks) ? table[(size_t)(b0 + 16u * q) * kMaxComp + c] : 0u;
      sum += (v[0] + v[1]) + (v[2] + v[3]);
    }
    part[l][c] = sum;
  }
  __syncthreads();
  if (threadIdx.x < (uint32_t)kMaxComp) {
    uint32_t sum = 0;
    for (int l = 0; l < 16; ++l) sum += part[l][threadIdx.x];
    cnt[threadIdx.x] = sum;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  uint32_t run = 0, s = 0;
  for (int c = 0; c < kMaxComp; ++c) {
    const uint32_t k = cnt[c];
    start[c] = run;
    const uint32_t padded = ((k + group_rows - 1) / group_rows) * group_rows;
    range[2 * c] = s / 32;
    range[2 * c + 1] = (s + padded) / 32;
    base[c] = s;
    run += k;
    s += padded;
  }
  start[kMaxComp] = n;   // (= run)
  base[kMaxComp] = s;
  range[2 * kMaxComp] = 0;
  range[2 * kMaxComp + 1] = 0;
  if (do_scale) {
    float M = __uint_as_float(hdr[0]);
    if (comp[kCompGrid + 5] > 1u) M = fminf(__uint_as_float(hdr[kHdrMloc]), fmaxf(M, 0.0f) * 4.0f + FLT_MIN);
    hdr[kHdrMused] = __float_as_uint(M);
    hdr[kHdrOpen] = 0u;
    const ScaleExp e = (r2max < 0.0f) ? pick_scale_nn(M) : pick_scale_pop(M, r2max, (int)D, shift_steps);
    hdr[kHdrShift] = (r2max < 0.0f) ? 0u : (uint32_t)shift_steps;   // (what the band of this scale pays for: dc_mfma_msym.hpp)
    hdr[kHdrScale + 0] = __float_as_uint(e.c);
    hdr[kHdrScale + 1] = __float_as_uint(e.s2);
    hdr[kHdrScale + 2] = (uint32_t)e.g;
    hdr[kHdrScale + 3] = (uint32_t)e.a;
    hdr[kHdrScale + 4] = (uint32_t)e.rounded;
  }
}

// ---- operand image of ONE tile from its rows in LDS (one wave) ------------------------------------------------------
// rows [32][Dp] original coordinates of the tile's positions, origin [D], frames [32] (kInvalidFrame: a dead row).  form
// 0: A form, 1: B form (pieces of -2x''), 2: the A form with the pieces of the row's own |x''|^2 / 2^a in the constant
// slots (nn_pruned_kernel's folded reference operand).  Same values as image_kernel, organised the other way round: a
// lane takes a (row, column) and forms the column's three pieces ONCE (image_kernel's lanes each formed the eight slots
// of their fragment, i.e. every column three times over: 250 instructions per fragment, 50 us per image at C3, ALU-bound),
// parks them as fp16 in an LDS copy of the tile's K rows, and the fragments are copied out 16 bytes per lane.
// lds_img: [32][img_stride] halves, img_stride = 16 NM + 2 (rows on different banks).
__device__ __forceinline__ uint32_t img_stride_halves(uint32_t NM) { return 16u * NM + 2u; }
__device__ __forceinline__ void build_tile_image(const float* __restrict__ rows, uint32_t Dp, const float* __restrict__ origin,
                                                 const uint32_t* __restrict__ frames, uint32_t D, uint32_t NM, const Scale& sc,
                                                 int form, unsigned short* __restrict__ lds_img, uint4* __restrict__ img_out,
                                                 float* __restrict__ norms_out, int lane) {
  const uint32_t stride = img_stride_halves(NM);
  const bool b_form = form == 1, fold = form == 2;
  const float s1 = b_form ? sc.sb : sc.sa;
  // zero the tile's K rows (padding slots, dead rows)
  {
    uint32_t* z = reinterpret_cast<uint32_t*>(lds_img);
    for (uint32_t e = (uint32_t)lane; e < 32u * stride / 2u; e += 64u) z[e] = 0u;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint32_t r = (uint32_t)lane & 31u;
  const bool live = frames[r] != kInvalidFrame;
  double nrm = 0.0;
  for (uint32_t k = (uint32_t)lane >> 5; k < D; k += 2u) {   // lane: row r, columns of its parity
    const float v = live ? (rows[r * Dp + k] - origin[k]) * s1 : 0.0f;   // x'' = c fl(x - mu)
    nrm += (double)v * (double)v;
    if (live) {
      const Pieces pc = split2(b_form ? -2.0f * v : v, sc.up, sc.dn);
      unsigned short* dst = lds_img + r * stride + (uint32_t)kConstSlots + k;
      dst[0] = (unsigned short)pc.hi;                                     // hi x hi
      dst[D] = (unsigned short)(b_form ? pc.hi_dn : pc.mid);              // A: mid 2^g   B: hi 2^-g
      dst[2u * D] = (unsigned short)(b_form ? pc.mid : pc.hi_dn);         // A: hi 2^-g   B: mid 2^g
    }
  }
  {  // the two column parities of a row meet (double, in a fixed order: deterministic)
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)__double_as_longlong(nrm), 32, 64);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)((unsigned long long)__double_as_longlong(nrm) >> 32), 32, 64);
    const double other = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    nrm = (lane < 32) ? nrm + other : other + nrm;
  }
  if (lane < 32) {
    unsigned short* dst = lds_img + r * stride;
    if (fold) {
      if (live) {
        const Pieces pn = split2((float)nrm * sc.cinv);   // (as load_query splits c_q)
        dst[0] = (unsigned short)pn.hi;
        dst[1] = (unsigned short)pn.mid;
      } else {
        dst[0] = 0x7BFFu;   // pad row: 65504 * 2^a, far above every threshold
      }
    } else if (live && !b_form) {
      dst[0] = (unsigned short)const_a_bits(sc.a);
      dst[1] = (unsigned short)const_a_bits(sc.a);
    }
    if (norms_out) norms_out[r] = live ? (float)nrm : INFINITY;   // pad rows can never be "inside"
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // fragment of lane l of MFMA m: slots 16 m + 8 (l >> 5) .. + 7 of row l & 31
  const uint32_t* src = reinterpret_cast<const uint32_t*>(lds_img + r * stride) + 4u * ((uint32_t)lane >> 5);
  for (uint32_t m = 0; m < NM; ++m) {
    const uint32_t* f = src + 8u * m;
    img_out[m * 64u + (uint32_t)lane] = make_uint4(f[0], f[1], f[2], f[3]);
  }
  __builtin_amdgcn_wave_barrier();
}

// dynamic LDS of order_rows2_kernel: the rows of 256 positions + per wave the K rows of one tile + its origin
static inline size_t order_rows_smem(uint32_t n_cols) {
  const uint32_t NM = (uint32_t)nm_for((int)n_cols);
  return sizeof(float) * 256 * (n_cols | 1u) + 4 * (sizeof(unsigned short) * 32 * (16 * NM + 2) + sizeof(float) * n_cols);
}

// The rows of an order in one pass, 256 positions = 8 tiles per block: gathered from the coordinates (element-wise:
// 40-byte runs of the source, coalesced stores to coords_o when asked for) and parked in LDS, from where every row's
// lane takes what the tile boxes and the free-energy ranges need (fe != nullptr: the neighbour sweep's order -- fe_s,
// invpos, ferange, and the hash of the order for the layout header of the all-gather blocks), and every wave builds the
// operand images of two of the block's tiles:
//   img_a != nullptr   A form of every tile (a_form 0 / 2) + the rows' norms
//   img_b != nullptr   B form of the tiles whose query group (grp_tq tiles) this launch's segment owns (+ norms_b)
// origin of a tile: origins[tile_comp[t]] (the all-pad tiles: the means).  The scale is in the header (order_meta_kernel).
__global__ __launch_bounds__(256) void order_rows2_kernel(
    const float* __restrict__ coords, uint32_t D, uint32_t NM, const uint32_t* __restrict__ perm, uint32_t T,
    float* __restrict__ coords_o, float4* __restrict__ boxes, const float* __restrict__ fe, float* __restrict__ fe_s,
    uint32_t* __restrict__ invpos, float2* __restrict__ ferange, const uint32_t* __restrict__ tile_comp,
    const float* __restrict__ origins, uint32_t* __restrict__ hdr, uint4* __restrict__ img_a, int a_form,
    float* __restrict__ norms_a, uint4* __restrict__ img_b, float* __restrict__ norms_b, uint32_t grp_tq, QSeg grp,
    unsigned long long* __restrict__ hash_slots, uint32_t* __restrict__ zero_pos, uint32_t zero_planes) {
  // (zero_pos: zero_planes arrays of n_pos words cleared by the way -- the counts by position of a symmetric population sweep)
  extern __shared__ float or_tile[];            // [256][D | 1], then per wave: K rows of a tile, origin
  __shared__ uint32_t s_frame[256];
  __shared__ unsigned long long fp_part[4];
  const uint32_t Dp = D | 1u;
  const uint32_t pos0 = blockIdx.x * 256u, pos = pos0 + threadIdx.x, n_pos = 32u * T;
  const uint32_t frame = (pos < n_pos) ? perm[pos] : kInvalidFrame;
  s_frame[threadIdx.x] = frame;
  if (zero_pos && pos < n_pos)
    for (uint32_t z = 0; z < zero_planes; ++z) zero_pos[(size_t)z * n_pos + pos] = 0u;
  __syncthreads();
  const size_t base = (size_t)pos0 * D, total = (size_t)n_pos * D;
  // (four elements per thread and step, their loads issued together)
  for (uint32_t e0 = threadIdx.x; e0 < 256u * D; e0 += 1024u) {
    float v[4];
    uint32_t off[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      const uint32_t e = e0 + 256u * j;
      const uint32_t r = min(e / D, 255u), k = e - (e / D) * D;
      const uint32_t i = s_frame[r];
      off[j] = r * Dp + k;
      v[j] = (e < 256u * D && i != kInvalidFrame) ? coords[(size_t)i * D + k] : 0.0f;   // (pad positions of a padded order)
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      const uint32_t e = e0 + 256u * j;
      if (e < 256u * D) {
        or_tile[off[j]] = v[j];
        if (coords_o && base + e < total) coords_o[base + e] = v[j];
      }
    }
  }
  __syncthreads();
  const bool in_range = pos < n_pos, live = frame != kInvalidFrame;
  const uint32_t t = min(pos >> 5, T - 1);
  const float* row = or_tile + threadIdx.x * Dp;
  const float x = live ? row[0] : 0.0f, y = (live && D > 1) ? row[1] : 0.0f;
  float lo0 = live ? x : INFINITY, hi0 = live ? x : -INFINITY;
  float lo1 = live ? y : INFINITY, hi1 = live ? y : -INFINITY;
  float flo = INFINITY, fhi = -INFINITY;
  if (fe) {
    const float f = live ? fe[frame] : INFINITY;
    if (in_range) fe_s[pos] = f;
    if (live) {
      invpos[frame] = pos;
      flo = f;
      fhi = f;
    }
    // hash of the order, what the layout header of a neighbour block carries: a wrap-around sum, left in 64 slots of the
    // component region (zero before; one word took 4 096 same-address 64-bit atomics at C3: ~20 us at the launch's end),
    // added up by nn_block_pack_kernel
    fp_publish(in_range ? fp_term(frame, pos) : 0ull, hash_slots + (blockIdx.x & 63u), fp_part);
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    lo0 = fminf(lo0, __shfl_xor(lo0, off, 64));
    hi0 = fmaxf(hi0, __shfl_xor(hi0, off, 64));
    lo1 = fminf(lo1, __shfl_xor(lo1, off, 64));
    hi1 = fmaxf(hi1, __shfl_xor(hi1, off, 64));
    flo = fminf(flo, __shfl_xor(flo, off, 64));
    fhi = fmaxf(fhi, __shfl_xor(fhi, off, 64));
  }
  if ((pos & 31u) == 0 && in_range) {
    boxes[t] = make_float4(lo0, hi0, lo1, hi1);   // empty tile: (+inf, -inf, ..): infinitely far
    if (ferange) ferange[t] = make_float2(flo, fhi);
  }
  // ---- operand images: wave w builds tiles 2 w and 2 w + 1 of the block
  const int lane = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6);
  unsigned short* lds_img = reinterpret_cast<unsigned short*>(or_tile + 256u * Dp) + (size_t)wv * (32u * img_stride_halves(NM) + 2u * D);
  float* org = reinterpret_cast<float*>(lds_img + 32u * img_stride_halves(NM));
  const Scale sc = load_scale(hdr);
  const float* means = reinterpret_cast<const float*>(reinterpret_cast<const char*>(hdr) + kHdrMeans);
  for (uint32_t tt = 0; tt < 2u; ++tt) {
    const uint32_t lt = 2u * (uint32_t)wv + tt, tile = blockIdx.x * 8u + lt;
    if (tile >= T) break;   // (wave-uniform)
    const uint32_t tc = tile_comp[tile];
    const float* o = (tc < (uint32_t)kMaxComp) ? origins + (size_t)tc * kMaxCols : means;
    for (uint32_t k = (uint32_t)lane; k < D; k += 64u) org[k] = o[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float* trows = or_tile + (size_t)lt * 32u * Dp;
    const uint32_t* tframes = s_frame + lt * 32u;
    if (img_a)
      build_tile_image(trows, Dp, org, tframes, D, NM, sc, a_form, lds_img, img_a + (size_t)tile * NM * 64u,
                       norms_a ? norms_a + (size_t)tile * 32u : nullptr, lane);
    if (img_b && seg_owns(tile / grp_tq, grp))
      build_tile_image(trows, Dp, org, tframes, D, NM, sc, 1, lds_img, img_b + (size_t)tile * NM * 64u,
                       norms_b ? norms_b + (size_t)tile * 32u : nullptr, lane);
  }
}
