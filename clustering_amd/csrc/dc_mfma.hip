// dc_mfma.hip -- host side of the MFMA variants: workspace layout, operand-image kernels and the
// switch over the per-K-step translation units (kernels: dc_mfma_kernels.hpp).
#include "dc_mfma_kernels.hpp"

#include <algorithm>
#include <cstring>
#include <atomic>
#include <mutex>

#ifndef DC_STEP_MASK
#define DC_STEP_MASK 0xFFFFu   // bit (n-1) set <=> dc_mfma_step.hip was built with -DDC_STEP=n
#endif

namespace dc {

namespace {

#include "dc_mfma32.hpp"

// ---------------------------------------------------------------------------------------------
// operand images
// ---------------------------------------------------------------------------------------------
__global__ void colsum_kernel(const float* __restrict__ coords, uint32_t n_rows, uint32_t D,
                              double* __restrict__ sums) {
  __shared__ double part[kMaxCols];
  if (threadIdx.x < (uint32_t)kMaxCols) part[threadIdx.x] = 0.0;
  __syncthreads();
  const uint32_t nthreads = gridDim.x * blockDim.x;
  const uint32_t used = (nthreads / D) * D;             // stride is a multiple of D: fixed column
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n_rows * D;
  if (id < used) {
    // four loads in flight per thread (one was bound by the latency of its load: 27 us for 40 MB); the sum of the
    // four partial sums is a sum in another order -- the mean is an origin, any value near it serves
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    size_t e = id;
    for (; e + 3 * (size_t)used < total; e += 4 * (size_t)used) {
      const float v0 = coords[e], v1 = coords[e + used], v2 = coords[e + 2 * (size_t)used], v3 = coords[e + 3 * (size_t)used];
      if (fabsf(v0) <= FLT_MAX) s0 += (double)v0;       // non-finite entries do not poison the mean
      if (fabsf(v1) <= FLT_MAX) s1 += (double)v1;
      if (fabsf(v2) <= FLT_MAX) s2 += (double)v2;
      if (fabsf(v3) <= FLT_MAX) s3 += (double)v3;
    }
    for (; e < total; e += used) {
      const float v = coords[e];
      if (fabsf(v) <= FLT_MAX) s0 += (double)v;
    }
    atomicAdd(&part[id % D], (s0 + s1) + (s2 + s3));
  }
  __syncthreads();
  if (threadIdx.x < D) atomicAdd(&sums[threadIdx.x], part[threadIdx.x]);
}

// column means as the float the centring subtracts (one division per column, not per element)
__global__ void mean_kernel(const double* __restrict__ sums, uint32_t n_rows, uint32_t D,
                            float* __restrict__ means) {
  const uint32_t k = threadIdx.x;
  if (k >= D) return;
  float muf = (float)(sums[k] / (double)n_rows);
  if (!(fabsf(muf) <= FLT_MAX)) muf = 0.0f;
  means[k] = muf;
}

// block-wide maximum (256 threads) -> one global atomicMax per block, and only when it would change
// the word (a plain read of a hot word is an L2 hit; thousands of same-address atomics are not)
__device__ __forceinline__ void publish_max(uint32_t* addr, uint32_t v, uint32_t* wave_max /* LDS [4] */) {
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

// Content fingerprint of a coordinate array (header words kHdrFp..+1, 64 bits): the wrap-around sum over all elements of
// (bits + c) * (2 * element index + 1) -- order-independent, so any kernel that reads every element can form it with
// integer atomics.  The statistics pass stores it; a call that CLAIMS the statistics are still valid
// (DC_FLAG_STATS_VALID) recomputes it in one streaming pass and the guard compares: an array rewritten in place, or a
// new array of the same shape at the same address, no longer passes for the old one.
__device__ __forceinline__ unsigned long long fp_term(uint32_t bits, unsigned long long e) {
  return (unsigned long long)(bits + 0x9E3779B9u) * (2ull * e + 1ull);
}
__device__ __forceinline__ void fp_publish(unsigned long long v, unsigned long long* dst, unsigned long long* part /* LDS [4] */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off, 64);
    v += ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dst, part[0] + part[1] + part[2] + part[3]);
  __syncthreads();
}
// Header pass over the frames in natural order: max |x'|^2 (word 0), non-finite / overflow flag
// (word 1) and the extent of columns 0/1 (words 8..11: ~key(min col0), key(max col0), ~key(min col1),
// key(max col1), all maintained with atomicMax).  |x'|^2 is formed exactly as image_kernel forms it,
// so word 0 bounds every norm of every operand image built from these coordinates.
__global__ void rowstats_kernel(const float* __restrict__ coords, uint32_t n_rows, uint32_t D,
                                const float* __restrict__ means, uint32_t* __restrict__ hdr, uint32_t cookie) {
  __shared__ uint32_t wave_max[4];
  __shared__ unsigned long long fp_part[4];
  __shared__ float mu[kMaxCols];
  if (threadIdx.x < D) mu[threadIdx.x] = means[threadIdx.x];
  __syncthreads();
  uint32_t m_norm = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0;
  unsigned long long fp = 0;   // content fingerprint of the rows this thread reads (fp_term)
  bool bad = false;
  // a row per lane, read where it lies: the ten loads of a wave touch the same 2.5 KB and meet in the vector cache
  const bool pairs = (D % 2u == 0) && ((reinterpret_cast<uintptr_t>(coords) & 7u) == 0);
  for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < n_rows; row += gridDim.x * blockDim.x) {
    const float* x = coords + (size_t)row * D;
    double nrm = 0.0;
    float c0 = 0.0f, c1 = 0.0f;
    if (pairs) {
      const float2* x2 = reinterpret_cast<const float2*>(x);
      const float2 first = x2[0];
      c0 = first.x;
      c1 = first.y;
      for (uint32_t k = 0; k < D; k += 2) {
        const float2 v = x2[k >> 1];
        fp += fp_term(__float_as_uint(v.x), (unsigned long long)row * D + k) + fp_term(__float_as_uint(v.y), (unsigned long long)row * D + k + 1);
        const float a = v.x - mu[k], b = v.y - mu[k + 1];
        nrm += (double)a * (double)a;
        nrm += (double)b * (double)b;
      }
    } else {
      c0 = x[0];
      if (D > 1) c1 = x[1];
      for (uint32_t k = 0; k < D; ++k) {
        fp += fp_term(__float_as_uint(x[k]), (unsigned long long)row * D + k);
        const float v = x[k] - mu[k];
        nrm += (double)v * (double)v;
      }
    }
    const float nf = (float)nrm;
    const bool ok = nf <= kNormLimit;
    bad = bad | !ok;   // NaN / inf / overflow-prone row: MFMA kernels stand down
    m_norm = max(m_norm, ok ? __float_as_uint(nf) : 0u);
    const bool fin = (fabsf(c0) <= FLT_MAX) && (fabsf(c1) <= FLT_MAX);
    m0 = max(m0, fin ? ~fkey(c0) : 0u);
    m1 = max(m1, fin ? fkey(c0) : 0u);
    m2 = max(m2, fin ? ~fkey(c1) : 0u);
    m3 = max(m3, fin ? fkey(c1) : 0u);
  }
  if (bad) atomicOr(hdr + 1, 1u);
  if (threadIdx.x == 0) hdr[kHdrCookie] = cookie;   // whose statistics these are (DC_FLAG_STATS_VALID is checked against it)
  publish_max(hdr, m_norm, wave_max);
  publish_max(hdr + 8, m0, wave_max);
  publish_max(hdr + 9, m1, wave_max);
  publish_max(hdr + 10, m2, wave_max);
  publish_max(hdr + 11, m3, wave_max);
  fp_publish(fp, reinterpret_cast<unsigned long long*>(hdr + kHdrFp), fp_part);
}

// dynamic LDS of image_kernel: per wave of the 256-thread block the 32 rows of its tile and their origin
static inline size_t image_smem(uint32_t n_cols) { return sizeof(float) * 4 * 33 * (size_t)n_cols; }

// operand image of the (centred, scaled) coordinates in the fp16x2 slot layout (dc_mfma_kernels.hpp), rows
// in natural order (perm == nullptr) or gathered through perm (an ordered frame list).  One thread
// writes the 16-byte fragment of one lane of one MFMA of one tile; the threads of MFMA 0, half 0
// also write the squared norm of their row (norms != nullptr).  b_form 1: query-side pieces (-2x'); 2: the A form
// with the pieces of the row's own |x''|^2 / 2^a in the two constant slots (65504 there for a pad row), the folded
// reference operand of nn_pruned_kernel.
__global__ void image_kernel(const float* __restrict__ coords, uint32_t n_total, uint32_t n_rows,
                             uint32_t D, uint32_t NM, uint32_t T, const float* __restrict__ means,
                             const uint32_t* __restrict__ perm, int b_form, uint4* __restrict__ img,
                             float* __restrict__ norms, const uint32_t* __restrict__ hdr,
                             uint32_t grp_tq = 1, QSeg grp = QSeg{1u, 0u, 1u},
                             const uint32_t* __restrict__ tile_comp = nullptr,
                             const float* __restrict__ origins = nullptr,
                             const uint32_t* __restrict__ valid = nullptr) {
  // tile_comp / origins: the origin of tile t is origins[tile_comp[t]][.] instead of the column means (the components
  // of the pruned population sweeps); valid: frame of every row of the order, kInvalidFrame for its pad rows
  // n_total: frames in the data set (divisor of the centring mean); n_rows: rows of this image.
  // (A variant that decodes the 16 slots of a block once into LDS was measured slower: the column
  //  loads then hang on the table look-ups instead of being issued together.)
  (void)n_total;
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = (uint32_t)(id & 63), m = (uint32_t)((id >> 6) % NM);
  // tile: the launch covers the tiles of every grp.stride-th group of grp_tq tiles (all tiles: {1, 0})
  const uint32_t tc = (uint32_t)((id >> 6) / NM);
  const uint32_t t = seg_group(tc / grp_tq, grp) * grp_tq + tc % grp_tq;
  if (t >= T) return;
  const uint32_t row = 32 * t + (lane & 31), h = lane >> 5;
  bool live = row < n_rows;
  uint32_t src = live ? (perm ? perm[row] : row) : 0u;
  if (live && perm && src == kInvalidFrame) live = false;
  if (live && valid && valid[row] == kInvalidFrame) live = false;
  src = live ? src : 0u;
  if (tile_comp) means = origins + (size_t)tile_comp[t] * kMaxCols;
  // The tile's rows and its origin go through LDS, a slice per wave: 32 x D floats read coalesced (or gathered by frame,
  // four loads in flight) instead of eight 4-byte loads per lane at a stride of D words, which kept the address units
  // busier than the 64 MB the kernel writes (50 us at 10^6 x 10).
  extern __shared__ float img_lds[];
  float* xs = img_lds + (size_t)(threadIdx.x >> 6) * (33u * D);
  float* org = xs + 32u * D;
  {
    const uint32_t rows_here = (32u * t < n_rows) ? min(32u, n_rows - 32u * t) : 0u;
    if (perm) {
      stage_query_rows(xs, nullptr, coords, src, live, D, (int)lane);
    } else {
      const float* contig = coords + (size_t)32 * t * D;
      const uint32_t total = rows_here * D;
      for (uint32_t e0 = lane; e0 < total; e0 += 256u) {
        float v4[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) v4[j] = (e0 + 64u * j < total) ? contig[e0 + 64u * j] : 0.0f;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j)
          if (e0 + 64u * j < total) xs[e0 + 64u * j] = v4[j];
      }
    }
    for (uint32_t k = lane; k < D; k += 64u) org[k] = means[k];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const float* x = xs + (size_t)(lane & 31u) * D;
  means = org;
  const Scale sc = load_scale(hdr);   // (the sweep's scale: scale_kernel ran before)
  const bool fold = b_form == 2;
  b_form = (b_form == 1) ? 1 : 0;
  const float s1 = b_form ? sc.sb : sc.sa;
  auto col = [&](uint32_t k) -> float { return (x[k] - means[k]) * s1; };   // x'' = 2^k fl(x - mu)
  uint32_t w[4] = {0u, 0u, 0u, 0u};
  double nrm = 0.0;
  if ((norms || fold) && m == 0 && h == 0) {
    // |x''|^2 of the SCALED coordinates (double accumulate, rounded once); a power-of-two scale commutes with it
    // (four columns per step, their loads issued together; summed in column order)
    for (uint32_t k0 = 0; k0 < D; k0 += 4) {
      float xv[4], mv[4];
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t k = min(k0 + j, D - 1u);
        xv[j] = x[k];
        mv[j] = means[k];
      }
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j)
        if (k0 + j < D) {
          const float v = live ? (xv[j] - mv[j]) * s1 : 0.0f;   // (= col(k))
          nrm += (double)v * (double)v;
        }
    }
  }
  if (!live && fold && m == 0 && h == 0) w[0] = 0x7BFFu;   // pad row: 65504 * 2^a, far above every threshold
  if (live) {
    // the 8 consecutive slots of this fragment (slot_value's layout, with ONE division for the first
    // coordinate slot instead of one per slot: the divisions were most of this kernel's time)
    const uint32_t s0 = 16 * m + 8 * h;
    uint32_t G = 0, k = 0;
    if (s0 >= (uint32_t)kConstSlots) {
      G = (s0 - kConstSlots) / D;
      k = (s0 - kConstSlots) - G * D;
    }
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
      uint32_t v;
      if (s0 + j < (uint32_t)kConstSlots) {
        v = b_form ? 0u : const_a_bits(sc.a);
        if (fold) {
          const Pieces pn = split2((float)nrm * sc.cinv);   // (as load_query splits c_q)
          v = (j == 0) ? pn.hi : pn.mid;
        }
      } else {
        v = 0u;
        if (G < (uint32_t)kPieceGroups) {
          const Pieces pc = split2(b_form ? -2.0f * col(k) : col(k), sc.up, sc.dn);
          const bool mid = b_form ? (G == 2u) : (G == 1u);
          v = (G == 0u) ? pc.hi : (mid ? pc.mid : pc.hi_dn);
        }
        if (++k == D) {
          k = 0;
          ++G;
        }
      }
      w[j >> 1] |= (v & 0xFFFFu) << (16 * (j & 1));
    }
  }
  img[((size_t)t * NM + m) * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);   // pad rows: all zero
  if (norms && m == 0 && h == 0) {
    norms[row] = live ? (float)nrm : INFINITY;   // pad rows can never be "inside"
  }
}

// the scale of the sweep that follows (dc_mfma_kernels.hpp "scale of a SWEEP") -> header words 20..24, read by
// the image builder and by the kernels.  r2max < 0: the neighbour rule; otherwise the population rule for
// a call whose largest squared radius is r2max.
__global__ void scale_kernel(uint32_t* __restrict__ hdr, float r2max, uint32_t D,
                             const uint32_t* __restrict__ comp = nullptr) {
  float M = __uint_as_float(hdr[0]);   // (final: rowstats_kernel ran before)
  // several components (pruned population sweeps): every row is measured from its component's origin (the maximum
  // over the rows: order_rows2_kernel)
  if (comp && comp[kCompGrid + 5] > 1u) M = fminf(__uint_as_float(hdr[kHdrMloc]), fmaxf(M, 0.0f) * 4.0f + FLT_MIN);
  hdr[kHdrMused] = __float_as_uint(M);
  hdr[kHdrOpen] = 0u;
  const ScaleExp e = (r2max < 0.0f) ? pick_scale_nn(M) : pick_scale_pop(M, r2max, (int)D);
  hdr[kHdrShift] = 0u;   // (no in-place threshold shifts under this scale)
  hdr[kHdrScale + 0] = __float_as_uint(e.c);
  hdr[kHdrScale + 1] = __float_as_uint(e.s2);
  hdr[kHdrScale + 2] = (uint32_t)e.g;
  hdr[kHdrScale + 3] = (uint32_t)e.a;
  hdr[kHdrScale + 4] = (uint32_t)e.rounded;
}

// ---- free-energy ordering of the reference frames (neighbour sweep) -----------------------------
// sortable key of a float (ascending); NaN free energies raise the flag (direct kernels take over)
__global__ void fe_key_kernel(const float* __restrict__ fe, uint32_t n_rows,
                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                              uint32_t* __restrict__ hdr) {
  __shared__ uint32_t wave_max[4];
  uint32_t inv = 0, top = 0;
  bool nan = false;
  // grid-stride: a block publishes its extrema once (two block reductions), however many rows it sees
  auto take = [&](float f, uint32_t i) {
    nan = nan | (f != f);
    const uint32_t u = __float_as_uint(f);
    const uint32_t key = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
    if (keys) {
      keys[i] = key;
      vals[i] = i;
    }
    inv = max(inv, ~key);
    top = max(top, (fabsf(f) <= FLT_MAX) ? key : 0u);
  };
  // 16 bytes per lane and step where the array allows it (one word per trip of the loop waited for every load)
  const uint32_t n4 = ((reinterpret_cast<uintptr_t>(fe) & 15u) == 0) ? n_rows / 4 : 0;
  const float4* fe4 = reinterpret_cast<const float4*>(fe);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    const float4 f = fe4[i];
    take(f.x, 4 * i);
    take(f.y, 4 * i + 1);
    take(f.z, 4 * i + 2);
    take(f.w, 4 * i + 3);
  }
  for (uint32_t i = 4 * n4 + blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += gridDim.x * blockDim.x) take(fe[i], i);
  if (nan) atomicOr(hdr + 1, 4u);
  // global minimum free energy -> header word 12 (as ~key, maintained with atomicMax), largest FINITE
  // one -> word 13 (as key); at most one atomic per block and word
  publish_max(hdr + 12, inv, wave_max);
  publish_max(hdr + 13, top, wave_max);
}


__global__ void fe_scatter_kernel(const uint32_t* __restrict__ perm, const float* __restrict__ fe,
                                  uint32_t n_rows, uint32_t T, uint32_t* __restrict__ invpos,
                                  float* __restrict__ fe_s) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= 32 * T) return;
  const uint32_t i = (p < n_rows) ? perm[p] : kInvalidFrame;   // (n_rows: the positions that hold a perm entry)
  if (i != kInvalidFrame) {
    invpos[i] = p;
    fe_s[p] = fe[i];
  } else {
    fe_s[p] = INFINITY;
  }
}

// pq[i] = #{ p : fe_s[p] < fe[i] }  (float comparison, exactly the reference's "fe[j] < fe[i]")
__global__ void fe_rank_kernel(const float* __restrict__ fe, const float* __restrict__ fe_s,
                               uint32_t n_rows, uint32_t* __restrict__ pq) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const float f = fe[i];
  uint32_t lo = 0, hi = n_rows;   // first position whose value is not < f
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (fe_s[mid] < f) lo = mid + 1; else hi = mid;
  }
  pq[i] = lo;
}


// Upper bound, known on the host, of the bits a cell key needs: the grid has about K = n / frames_per_cell cells
// (auto_cell), at most 4002 per dimension; (x + 1)(y + 1) with x y <= K and x, y <= 4001 is at most
// K + 4001 + K / 4001 + 1.  The radix sort (dc_sort.hip) costs one pass (~25 us at 10^6 frames) per 8 key bits: C3's population
// ordering needs 15 bits, not kCellKeyBits = 24.  (Should a data set ever exceed the bound, the sort would ignore the
// top bits of its keys: a worse ordering, i.e. less pruning -- never a different result.)
static unsigned cell_key_bits(uint32_t n_rows, float frames_per_cell) {
  const double K = (double)n_rows / (double)frames_per_cell;
  const double bound = K + 4002.0 + K / 4001.0 + 8.0;
  unsigned bits = 1;
  while (bits < kCellKeyBits && (double)(1u << bits) < bound) ++bits;
  return bits;
}
// ... and of the combined (cell, free energy) key of the neighbour sweep: the cell bits + at least kFeKeyBits of
// quantised free energy, rounded up to whole sort passes
constexpr unsigned kFeKeyBits = 10;
static unsigned cellfe_key_bits(uint32_t n_rows, float frames_per_cell) {
  const unsigned want = cell_key_bits(n_rows, frames_per_cell) + kFeKeyBits;
  const unsigned rounded = (want + 7u) & ~7u;
  return rounded > 32u ? 32u : rounded;
}



// ---- components of the pruned population sweeps (dc_mfma_kernels.hpp "components") -------------------------------
struct CoarseGrid {
  float gc, min0, min1;
  uint32_t ncx, ncy;
};
// coarse occupancy grid over the bounding box of columns 0/1: cells of half the connectivity length, at most
// kCoarseDim per dimension
// r_max < 0 (the neighbour sweep has no radius): -r_max times the cell edge of its ordering for n_rows frames
__device__ __forceinline__ CoarseGrid coarse_grid(const uint32_t* __restrict__ hdr, float r_max, uint32_t n_rows = 0) {
  CoarseGrid g;
  if (r_max < 0.0f) r_max = -r_max * auto_cell(hdr, n_rows, kNnCellFrames);
  g.min0 = fkey_inv(~hdr[8]);
  g.min1 = fkey_inv(~hdr[10]);
  float e0 = fkey_inv(hdr[9]) - g.min0, e1 = fkey_inv(hdr[11]) - g.min1;
  if (!(e0 >= 0.0f) || !(e0 <= FLT_MAX)) e0 = 0.0f;
  if (!(e1 >= 0.0f) || !(e1 <= FLT_MAX)) e1 = 0.0f;
  if (!(fabsf(g.min0) <= FLT_MAX)) g.min0 = 0.0f;
  if (!(fabsf(g.min1) <= FLT_MAX)) g.min1 = 0.0f;
  // (half the connectivity length: a quarter labelled four times the cells for boxes that the sub-cells make tight anyway)
  const float half_r = (r_max <= FLT_MAX) ? 0.5f * r_max : FLT_MAX;
  g.gc = fmaxf(half_r, fmaxf(e0, e1) / (float)(kCoarseDim - 1));
  if (!(g.gc > 0.0f)) g.gc = 1.0f;
  g.ncx = min((uint32_t)fminf(e0 / g.gc, (float)kCoarseDim) + 1u, (uint32_t)kCoarseDim);
  g.ncy = min((uint32_t)fminf(e1 / g.gc, (float)kCoarseDim) + 1u, (uint32_t)kCoarseDim);
  return g;
}
__device__ __forceinline__ uint32_t coarse_cell(const CoarseGrid& g, float x, float y) {
  // (NaN: fminf / fmaxf return the other operand -> cell 0)
  const uint32_t cx = (uint32_t)fminf(fmaxf((x - g.min0) / g.gc, 0.0f), (float)(g.ncx - 1));
  const uint32_t cy = (uint32_t)fminf(fmaxf((y - g.min1) / g.gc, 0.0f), (float)(g.ncy - 1));
  return cx * g.ncy + cy;
}

// (the connectivity length a component partition was built with lives in its region: comp[kCompGrid + 7])
__device__ __forceinline__ float comp_r_conn(const uint32_t* __restrict__ comp) { return __uint_as_float(comp[kCompGrid + 7]); }

// occupancy bitmap of the sub-cells (kFineSub x kFineSub per coarse cell): plain byte stores, no atomics
__global__ void fine_mark_kernel(const float* __restrict__ coords, uint32_t n_rows, uint32_t D,
                                 const uint32_t* __restrict__ hdr, float r_max, uint32_t* __restrict__ comp) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const float x = coords[(size_t)i * D], y = (D > 1) ? coords[(size_t)i * D + 1] : 0.0f;
  if (!(fabsf(x) <= FLT_MAX) || !(fabsf(y) <= FLT_MAX)) return;   // (flagged data: the sweep stands down anyway)
  const CoarseGrid g = coarse_grid(hdr, r_max, n_rows);
  const float gf = g.gc / (float)kFineSub;
  const uint32_t fx = (uint32_t)fminf(fmaxf((x - g.min0) / gf, 0.0f), (float)(g.ncx * kFineSub - 1));
  const uint32_t fy = (uint32_t)fminf(fmaxf((y - g.min1) / gf, 0.0f), (float)(g.ncy * kFineSub - 1));
  reinterpret_cast<unsigned char*>(comp + kCompBitmap)[(size_t)fx * (kCoarseDim * kFineSub) + fy] = 1;
}

// coarse cell of a point, through its sub-cell (the same arithmetic as fine_mark_kernel)
__device__ __forceinline__ uint32_t coarse_cell_of_point(const CoarseGrid& g, float x, float y) {
  const float gf = g.gc / (float)kFineSub;
  const uint32_t fx = (uint32_t)fminf(fmaxf((x - g.min0) / gf, 0.0f), (float)(g.ncx * kFineSub - 1));
  const uint32_t fy = (uint32_t)fminf(fmaxf((y - g.min1) / gf, 0.0f), (float)(g.ncy * kFineSub - 1));
  return (fx / kFineSub) * g.ncy + (fy / kFineSub);
}

// Labels the occupied coarse cells (one workgroup): cells whose point boxes are closer than r_max are connected --
// two frames closer than r_max in the (col 0, col 1) plane then always share a component, so frames of DIFFERENT
// components are at least r_max apart in full dimension as well.  Minimum-label propagation over the (2R + 1)^2
// neighbourhood with pointer jumping; labels are cell indices (the result does not depend on the order in which the
// cells were listed).  Writes the component of every cell, the components' origins and fine cell grids.
__device__ void fine_grid_body(const uint32_t* __restrict__ hdr, uint32_t n_rows, float frames_per_cell, uint32_t fine_bits,
                               uint32_t* __restrict__ comp);
// Round 5: ONE launch for what were four -- the boxes of the coarse cells (coarse_box_kernel) are formed here, straight
// from the occupancy bitmap into the labelling's tables; the column means (mean_kernel: means_out != nullptr) in front; the
// fine cell grids of the components (fine_grid_kernel) behind, by the first thread.
__global__ __launch_bounds__(1024) void components_kernel(const uint32_t* __restrict__ hdr,
                                                          const float* __restrict__ means, uint32_t D,
                                                          float r_max, uint32_t n_rows, float frames_per_cell,
                                                          uint32_t fine_bits, uint32_t* __restrict__ comp,
                                                          int force_single, float r_true, uint32_t cookie,
                                                          const double* __restrict__ stats_table = nullptr,
                                                          float fine_frames_per_cell = 0.0f) {
  // r_max: the connectivity length rho (what the coarse grid was built for); r_true: the largest radius itself;
  // cookie: whose partition this is (comp_guard_kernel)
  (void)frames_per_cell;
  const float r_conn_param = r_max;
#define DC_STAMP(k) do {} while (0)
  DC_STAMP(0);
  (void)stats_table;   // (the statistics table is added up by its own launch: stats_reduce_kernel)
  DC_STAMP(1);
  __shared__ uint32_t occ[kMaxOccupied], label[kMaxOccupied];
  __shared__ float4 obox[kMaxOccupied];
  __shared__ uint32_t n_occ_s, changed_s, n_comp_s, n_sub_s;
  // cell -> index in occ[] (0xFFFF: empty) during the labelling: in LDS, kCoarseDim^2 half-words (the look-ups of the
  // neighbourhood went to global memory before: 11 dependent round trips per cell and round, 110 of the kernel's 130 us)
  extern __shared__ uint16_t cell_idx[];
  __shared__ uint32_t root_cell[kMaxComp];
  __shared__ uint32_t cbox[kMaxComp][4];
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  const CoarseGrid g = coarse_grid(hdr, r_max, n_rows);
  if (r_max < 0.0f) r_max = g.gc * 2.0f;   // (the connectivity length the grid was built for: two coarse cells, or more)
  const uint32_t n_cells = g.ncx * g.ncy;
  uint32_t* cell_comp = comp + kCompCellComp;   // during the labelling: cell -> index in occ[] (0xFFFFFFFF: empty)
  if (tid == 0) {
    n_occ_s = 0;
    n_comp_s = 0;
    n_sub_s = 0;
  }
  __syncthreads();
  const float gf_sub = g.gc / (float)kFineSub, slack_sub = 1.0e-3f * gf_sub;
  for (uint32_t c = tid; c < n_cells; c += nt) {
    uint32_t idx = 0xFFFFFFFFu;
    // box of the occupied sub-cells of the cell (a little wider than the sub-cells: every frame of the cell lies inside
    // whatever the rounding of its sub-cell index did); lo0 > hi0 marks an empty cell.  NB: a frame is assigned to the
    // COARSE cell of its sub-cell (fx / kFineSub), see coarse_cell_of_point.
    const uint32_t cx = c / g.ncy, cy = c % g.ncy;
    uint32_t rows4[kFineSub];
#pragma unroll
    for (int sx = 0; sx < kFineSub; ++sx)   // (kFineSub = 4 bytes of a bitmap row: one word)
      rows4[sx] = comp[kCompBitmap + ((size_t)(cx * kFineSub + sx) * (kCoarseDim * kFineSub) + (size_t)cy * kFineSub) / 4];
    int x0 = kFineSub, x1 = -1, y0 = kFineSub, y1 = -1;
    uint32_t n_sub = 0;
#pragma unroll
    for (int sx = 0; sx < kFineSub; ++sx)
#pragma unroll
      for (int sy = 0; sy < kFineSub; ++sy)
        if ((rows4[sx] >> (8 * sy)) & 0xFFu) {
          x0 = min(x0, sx);
          x1 = max(x1, sx);
          y0 = min(y0, sy);
          y1 = max(y1, sy);
          ++n_sub;
        }
    float4 bx = make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
    if (x1 >= 0) {
      atomicAdd(&n_sub_s, n_sub);   // (the occupied sub-cells of the whole grid: what the frames cover of the plane)
      bx.x = g.min0 + (float)(cx * kFineSub + x0) * gf_sub - slack_sub;
      bx.y = g.min0 + (float)(cx * kFineSub + x1 + 1) * gf_sub + slack_sub;
      bx.z = g.min1 + (float)(cy * kFineSub + y0) * gf_sub - slack_sub;
      bx.w = g.min1 + (float)(cy * kFineSub + y1 + 1) * gf_sub + slack_sub;
    }
    if (bx.x <= bx.y) {
      idx = atomicAdd(&n_occ_s, 1u);
      if (idx < (uint32_t)kMaxOccupied) {
        occ[idx] = c;
        obox[idx] = bx;
      }
    }
    cell_idx[c] = (uint16_t)min(idx, 0xFFFFu);
  }
  __syncthreads();
  DC_STAMP(2);
  const uint32_t n_occ = n_occ_s;
  bool single = force_single != 0 || n_occ > (uint32_t)kMaxOccupied || n_occ <= 1u || !(r_max <= FLT_MAX);
  const float r2c = r_max * r_max * 1.0002f;
  if (!single) {
    // cells d apart (in either direction) keep their boxes at least (d - 1) gc - 2 slack apart: beyond R they cannot be
    // within reach (gc >= r_max / 2: R = 3; round 4 looked one ring further, 81 cells instead of 49)
    const int R = min((int)floorf((r_max * 1.0002f + 2.0f * slack_sub) / g.gc) + 1, 5);
    for (uint32_t i = tid; i < n_occ; i += nt) label[i] = occ[i];
    __syncthreads();
    // The neighbourhood of a cell does not change from round to round: the first round tests the boxes and leaves a bit
    // per neighbouring cell within reach (at most 11 x 11 of them: four words per cell, a thread owns at most two cells),
    // the later rounds only follow the bits -- a look-up of the cell and its label each (the box tests of every round
    // were 8 us per round and three rounds at C3: 25 of the kernel's 43 us).
    static_assert(kMaxOccupied <= 2048, "a thread of the 1024 owns at most two occupied cells");
    uint32_t nbr[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    int iter = 0;
    for (;; ++iter) {
      if (tid == 0) changed_s = 0;
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint32_t i = tid + (uint32_t)s * nt;
        if (i >= n_occ) continue;
        const uint32_t c = occ[i];
        const int cx = (int)(c / g.ncy), cy = (int)(c % g.ncy);
        uint32_t m = label[i];
        if (iter == 0) {
          const float4 bi = obox[i];
          for (int dx = -R; dx <= R; ++dx) {
            const int nx = cx + dx;
            if (nx < 0 || nx >= (int)g.ncx) continue;
            // (the look-ups of one row of the neighbourhood are independent: issued together, not one latency each)
            uint32_t jj[11];
#pragma unroll
            for (int k = 0; k < 11; ++k) {
              const int ny = cy + k - 5;
              const bool in = (k - 5 >= -R) && (k - 5 <= R) && ny >= 0 && ny < (int)g.ncy;
              jj[k] = in ? (uint32_t)cell_idx[(uint32_t)nx * g.ncy + (uint32_t)ny] : 0xFFFFu;
            }
            uint32_t row_bits = 0;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
              const uint32_t j = jj[k];
              if (j >= n_occ) continue;
              if (box_gap2(bi, obox[j]) <= r2c) {
                m = min(m, label[j]);
                row_bits |= 1u << k;
              }
            }
            const uint32_t at = (uint32_t)(dx + R) * 11u;   // bit index of the row's first cell
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const int sh = (int)at - 32 * w;
              if (sh >= 0 && sh < 32) nbr[s][w] |= row_bits << sh;
              if (sh < 0 && sh > -11) nbr[s][w] |= row_bits >> (-sh);
            }
          }
        } else {
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            uint32_t bits = nbr[s][w];
            while (bits) {
              const uint32_t idx = 32u * (uint32_t)w + (uint32_t)__builtin_ctz(bits);
              bits &= bits - 1u;
              const uint32_t dxi = idx / 11u, k = idx - dxi * 11u;
              const uint32_t j = cell_idx[(uint32_t)(cx + (int)dxi - R) * g.ncy + (uint32_t)(cy + (int)k - 5)];
              m = min(m, label[j]);
            }
          }
        }
        if (m < label[i]) {
          atomicMin(&label[i], m);
          changed_s = 1;
        }
      }
      __syncthreads();
      // pointer jumping: the label of my label's cell
      for (int hop = 0; hop < 4; ++hop) {
        for (uint32_t i = tid; i < n_occ; i += nt) {
          const uint32_t l = label[cell_idx[label[i]]];
          if (l < label[i]) {
            atomicMin(&label[i], l);
            changed_s = 1;
          }
        }
        __syncthreads();
      }
      const uint32_t ch = changed_s;
      __syncthreads();
      if (ch == 0 || iter >= 128) break;
    }
    if (iter >= 128) single = true;   // (not settled: one component is always right)
  }
  DC_STAMP(3);
  if (!single) {
    // roots = cells that are their own label; component id = rank of the root's cell index
    for (uint32_t i = tid; i < n_occ; i += nt)
      if (label[i] == occ[i]) {
        const uint32_t k = atomicAdd(&n_comp_s, 1u);
        if (k < (uint32_t)kMaxComp) root_cell[k] = occ[i];
      }
    __syncthreads();
    if (n_comp_s > (uint32_t)kMaxComp || n_comp_s <= 1u) single = true;
  }
  __syncthreads();
  uint32_t n_comp = single ? 1u : n_comp_s;
  if (tid < (uint32_t)kMaxComp) {
    cbox[tid][0] = 0;
    cbox[tid][1] = 0;
    cbox[tid][2] = 0;
    cbox[tid][3] = 0;
  }
  __syncthreads();
  if (single) {
    for (uint32_t c = tid; c < n_cells; c += nt) cell_comp[c] = 0u;
  } else {
    // (sort the few roots by cell index: id = number of roots with a smaller index)
    for (uint32_t i = tid; i < n_occ; i += nt) {
      const uint32_t root = label[i];
      uint32_t id = 0;
      for (uint32_t k = 0; k < n_comp; ++k) id += (root_cell[k] < root) ? 1u : 0u;
      label[i] = id;
      const float4 bx = obox[i];
      atomicMax(&cbox[id][0], ~fkey(bx.x));
      atomicMax(&cbox[id][1], fkey(bx.y));
      atomicMax(&cbox[id][2], ~fkey(bx.z));
      atomicMax(&cbox[id][3], fkey(bx.w));
    }
    __syncthreads();
    for (uint32_t c = tid; c < n_cells; c += nt) cell_comp[c] = 0u;
    __syncthreads();
    for (uint32_t i = tid; i < n_occ; i += nt) cell_comp[occ[i]] = label[i];
  }
  __syncthreads();
  DC_STAMP(4);
  // origins, boxes, adjacency
  const float gmin0 = fkey_inv(~hdr[8]), gmax0 = fkey_inv(hdr[9]), gmin1 = fkey_inv(~hdr[10]), gmax1 = fkey_inv(hdr[11]);
  if (tid < n_comp) {
    const float4 mine = single ? make_float4(gmin0, gmax0, gmin1, gmax1)
                               : make_float4(fkey_inv(~cbox[tid][0]), fkey_inv(cbox[tid][1]), fkey_inv(~cbox[tid][2]),
                                             fkey_inv(cbox[tid][3]));
    float* a = reinterpret_cast<float*>(comp + kCompOrigin) + (size_t)tid * kMaxCols;
    for (uint32_t k = 0; k < D; ++k) a[k] = means[k];
    if (!single) {
      a[0] = 0.5f * mine.x + 0.5f * mine.y;
      if (D > 1) a[1] = 0.5f * mine.z + 0.5f * mine.w;
    }
    reinterpret_cast<float4*>(comp + kCompBox)[tid] = mine;
    // which components come closer than the largest radius (cross pairs: pop_cross_kernel)
    unsigned long long adj = 0;
    const float r2t = r_true * r_true * 1.0002f;
    if (!single && r_true <= FLT_MAX)
      for (uint32_t k = 0; k < n_comp; ++k) {
        if (k == tid) continue;
        const float4 other = make_float4(fkey_inv(~cbox[k][0]), fkey_inv(cbox[k][1]), fkey_inv(~cbox[k][2]),
                                         fkey_inv(cbox[k][3]));
        if (box_gap2(mine, other) <= r2t) adj |= 1ull << k;
      }
    comp[kCompAdj + 2 * tid] = (uint32_t)adj;
    comp[kCompAdj + 2 * tid + 1] = (uint32_t)(adj >> 32);
  }
  if (tid == 0) {
    comp[kCompGrid + 0] = __float_as_uint(g.gc);
    comp[kCompGrid + 1] = single ? 0u : n_occ;
    comp[kCompGrid + 2] = n_sub_s;
    comp[kCompGrid + 3] = g.ncx;
    comp[kCompGrid + 4] = g.ncy;
    comp[kCompGrid + 5] = n_comp;
    comp[kCompGrid + 6] = cookie;
    comp[kCompGrid + 7] = __float_as_uint(r_conn_param);
  }
  DC_STAMP(5);
  if (fine_frames_per_cell > 0.0f) {   // the fine cell grids of the components for the sweep that follows
    __syncthreads();                   // (boxes, counts: this workgroup's own writes)
    if (tid == 0) fine_grid_body(hdr, n_rows, fine_frames_per_cell, fine_bits, comp);
  }
  DC_STAMP(6);
}

// dynamic LDS of components_kernel (the cell map); with its static arrays the workgroup needs 81 KB of the CU's 160
static size_t components_smem() {
  constexpr size_t bytes = sizeof(uint16_t) * kCoarseDim * kCoarseDim;
  // (per device: set on every call, a host-side table update)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(components_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes);
  return bytes;
}

// A partition left in the workspace by an earlier sweep over the same coordinates serves this sweep too (any partition
// does: what it cannot see across components, the exact cross passes look at) -- unless the cookie says it belongs to
// other data: then one component, the column means as its origin.
// Round 5: the fine cell grids of the sweep that follows are formed in the same launch (fine_frames_per_cell > 0), and the
// rows-per-component counters of both orders start from zero (order_key_kernel adds to them).
__device__ void comp_guard_body(const uint32_t* __restrict__ hdr, const float* __restrict__ means, uint32_t D,
                                uint32_t* __restrict__ comp, uint32_t cookie, uint32_t n_rows, float fine_frames_per_cell,
                                uint32_t fine_bits) {
  if (threadIdx.x < 2u * ((uint32_t)kMaxComp + 1u)) comp[kCompStart + threadIdx.x] = 0u;
  if (threadIdx.x < 128u) comp[kCompHash + threadIdx.x] = 0u;   // (shares of the order's hash: order_rows2_kernel adds to them)
  const bool keep = comp[kCompGrid + 6] == cookie && hdr[kHdrCookie] == cookie;   // (a failed claim: nothing of the workspace is this array's)
  if (!keep)
    for (uint32_t c = threadIdx.x; c < (uint32_t)kCoarseCells; c += blockDim.x) comp[kCompCellComp + c] = 0u;
  if (!keep && threadIdx.x == 0) {
    float* a = reinterpret_cast<float*>(comp + kCompOrigin);
    for (uint32_t k = 0; k < D; ++k) a[k] = means[k];
    reinterpret_cast<float4*>(comp + kCompBox)[0] =
        make_float4(fkey_inv(~hdr[8]), fkey_inv(hdr[9]), fkey_inv(~hdr[10]), fkey_inv(hdr[11]));
    comp[kCompAdj] = 0;
    comp[kCompAdj + 1] = 0;
    comp[kCompGrid + 1] = 0;
    comp[kCompGrid + 5] = 1;
    comp[kCompGrid + 7] = __float_as_uint(1.0f);
  }
  if (fine_frames_per_cell > 0.0f) {
    __syncthreads();
    if (threadIdx.x == 0) fine_grid_body(hdr, n_rows, fine_frames_per_cell, fine_bits, comp);
  }
}
// the fine cell grids of the components for one sweep (frames_per_cell is the sweep's own): one cell size for all
// components (at most 4001 cells per dimension and component), the cells of all components numbered consecutively --
// fewer than 2^fine_bits of them, so the ordering keys stay short
__device__ void fine_grid_body(const uint32_t* __restrict__ hdr, uint32_t n_rows, float frames_per_cell,
                               uint32_t fine_bits, uint32_t* __restrict__ comp) {
  const uint32_t n_comp = min(comp[kCompGrid + 5], (uint32_t)kMaxComp);
  const float gmin0 = fkey_inv(~hdr[8]), gmax0 = fkey_inv(hdr[9]), gmin1 = fkey_inv(~hdr[10]), gmax1 = fkey_inv(hdr[11]);
  float cell = auto_cell(hdr, n_rows, frames_per_cell);
  if (n_comp > 1u) {
    // sparse data: the frames cover a small part of the bounding box -- size the fine cells by the occupied area (the
    // occupied sub-cells of the coarse grid, x 2.7: what the bounding box of the reference workload is of the area its
    // blobs cover, i.e. the same cells there), so that the cells do not grow with the empty space between clusters
    const double gf = (double)__uint_as_float(comp[kCompGrid + 0]) / (double)kFineSub;
    const double a_box = (double)fmaxf(gmax0 - gmin0, 0.0f) * (double)fmaxf(gmax1 - gmin1, 0.0f);
    const double a_occ = 2.7 * (double)comp[kCompGrid + 2] * gf * gf;
    if (a_occ > 0.0 && a_occ < a_box)
      cell = (float)sqrt(a_occ * (double)frames_per_cell / (double)(n_rows ? n_rows : 1u));
  }
  const float4* cb = reinterpret_cast<const float4*>(comp + kCompBox);
  auto ext = [&](uint32_t c, int dim) {
    const float4 b = cb[c];
    const float e = dim ? b.w - b.z : b.y - b.x;
    return (e >= 0.0f && e <= FLT_MAX) ? e : 0.0f;
  };
  const float limit = ldexpf(1.0f, (int)fine_bits) - 2.0f;
  if (!(cell > 0.0f) || !(cell <= FLT_MAX)) cell = 1.0f;
  for (int round = 0; round < 8; ++round) {
    float total = 0.0f;
    for (uint32_t c = 0; c < n_comp; ++c) {
      const float c0 = fmaxf(cell, ext(c, 0) / 4000.0f), c1 = fmaxf(cell, ext(c, 1) / 4000.0f);
      total += (floorf(fminf(ext(c, 0) / c0, 4001.0f)) + 1.0f) * (floorf(fminf(ext(c, 1) / c1, 4001.0f)) + 1.0f);
    }
    if (total <= limit) break;
    cell *= sqrtf(total / limit) * 1.05f;
  }
  uint32_t off = 0;
  for (uint32_t c = 0; c < n_comp; ++c) {
    float c0 = fmaxf(cell, ext(c, 0) / 4000.0f), c1 = fmaxf(cell, ext(c, 1) / 4000.0f);
    if (!(c0 > 0.0f) || !(c0 <= FLT_MAX)) c0 = 1.0f;
    if (!(c1 > 0.0f) || !(c1 <= FLT_MAX)) c1 = 1.0f;
    const uint32_t nx = (uint32_t)fminf(ext(c, 0) / c0, 4001.0f) + 1u, ny = (uint32_t)fminf(ext(c, 1) / c1, 4001.0f) + 1u;
    const float4 b = cb[c];
    uint32_t* f = comp + kCompFine + 4 * (size_t)c;
    f[0] = __float_as_uint((fabsf(b.x) <= FLT_MAX) ? b.x : 0.0f);
    f[1] = __float_as_uint((fabsf(b.z) <= FLT_MAX) ? b.z : 0.0f);
    f[2] = __float_as_uint(c0);
    f[3] = __float_as_uint(c1);
    comp[kCompNby + c] = ny;
    comp[kCompCellOff + c] = off;
    off = min(off + nx * ny, (1u << fine_bits) - 1u);
  }
  for (uint32_t c = n_comp; c <= (uint32_t)kMaxComp; ++c) comp[kCompCellOff + c] = off;
}

// Frame pairs between ADJACENT components (rho = r_max / 2: components may come closer than the largest radius): the
// matrix-core sweep never meets them -- a query group only scans its own component -- so they are counted here, in the
// canonical arithmetic, one wave per query group: the group's box against the boxes of the adjacent components, then
// against their tiles, and every frame of a surviving tile against every query of the group.  Rare by construction
// (only the outskirts of two clusters face each other across less than r_max); each direction is counted from its own
// query side, so both the symmetric and the one-sided sweeps just add these counts.
// out: by_position == 1: counts[rr * stride + query position] (the symmetric sweeps' pops_pos), 2: counts[tile][stride][32]
// (the multi-radius symmetric sweep), 0: [rr * stride + frame]
__global__ __launch_bounds__(64) void pop_cross_kernel(
    const float* __restrict__ coords, uint32_t n_cols, const float* __restrict__ coords_r,
    const uint32_t* __restrict__ perm_r, const float4* __restrict__ box_r, const uint32_t* __restrict__ perm_q,
    const float4* __restrict__ box_q, const uint32_t* __restrict__ tile_comp_q, const uint32_t* __restrict__ comp,
    uint32_t T_q, uint32_t group_tiles, QSeg q_seg, int q_in_ref_order, Rad2 rad2, int n_rad, float r2max,
    const uint32_t* __restrict__ hdr, uint32_t* __restrict__ out, size_t stride, int by_position) {
  __shared__ uint32_t cnt[kMaxGroupRows * kMaxRadiiPerLaunch];
  __shared__ uint32_t list[64];
  if (hdr[1] != 0) return;
  const int lane = threadIdx.x;
  const uint32_t group = seg_group(blockIdx.x, q_seg);
  const uint32_t qt0 = group * group_tiles;
  if (qt0 >= T_q) return;
  const uint32_t my_comp = tile_comp_q[qt0];
  if (my_comp >= (uint32_t)kMaxComp) return;
  const unsigned long long adj = ((unsigned long long)comp[kCompAdj + 2 * my_comp + 1] << 32) | comp[kCompAdj + 2 * my_comp];
  if (adj == 0) return;
  const uint32_t group_rows = 32u * group_tiles;
  float4 gbox = make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
  for (uint32_t t = qt0; t < min(qt0 + group_tiles, T_q); ++t) {
    const float4 b = box_q[t];
    gbox.x = fminf(gbox.x, b.x);
    gbox.y = fmaxf(gbox.y, b.y);
    gbox.z = fminf(gbox.z, b.z);
    gbox.w = fmaxf(gbox.w, b.w);
  }
  const float far2 = r2max * 1.0001f;
  for (uint32_t k = lane; k < group_rows * (uint32_t)n_rad; k += 64) cnt[k] = 0;
  __syncthreads();
  bool any = false;
  const uint32_t* range = comp + kCompRange;
  for (uint32_t c2 = 0; c2 < (uint32_t)kMaxComp; ++c2) {
    if (!((adj >> c2) & 1ull)) continue;
    if (!(box_gap2(gbox, reinterpret_cast<const float4*>(comp + kCompBox)[c2]) < far2)) continue;
    const uint32_t t_lo = range[2 * c2], t_hi = range[2 * c2 + 1];
    for (uint32_t base = t_lo; base < t_hi; base += 64) {
      const uint32_t t = base + lane;
      const bool ok = t < t_hi && box_gap2(gbox, box_r[t]) < far2;
      const uint64_t m = __builtin_amdgcn_ballot_w64(ok);
      if (m == 0) continue;
      if (ok) list[__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = t;
      __syncthreads();
      const uint32_t n_list = (uint32_t)__builtin_popcountll(m);
      any = true;
      for (uint32_t s = 0; s < n_list; ++s) {
        const uint32_t tr = list[s];
        for (uint32_t r = 0; r < 32; ++r) {
          const uint32_t pr = 32u * tr + r;
          if (perm_r[pr] == kInvalidFrame) continue;   // (wave-uniform)
          const float* xr = coords_r + (size_t)pr * n_cols;
          for (uint32_t k = lane; k < group_rows; k += 64) {
            const uint32_t pq = 32u * qt0 + k;
            const uint32_t fq = (pq < 32u * T_q) ? perm_q[pq] : kInvalidFrame;
            if (fq == kInvalidFrame) continue;
            const float* yq = q_in_ref_order ? coords_r + (size_t)pq * n_cols : coords + (size_t)fq * n_cols;
            const float d2 = dist2_canon_rows(yq, xr, (int)n_cols);
            for (int rr = 0; rr < n_rad; ++rr)
              if (d2 < rad2.v[rr]) cnt[(size_t)rr * group_rows + k] += 1u;   // (query k is this lane's alone)
          }
        }
      }
      __syncthreads();
    }
  }
  if (!any) return;
  for (uint32_t k = lane; k < group_rows; k += 64) {
    const uint32_t pq = 32u * qt0 + k;
    const uint32_t fq = (pq < 32u * T_q) ? perm_q[pq] : kInvalidFrame;
    if (fq == kInvalidFrame) continue;
    for (int rr = 0; rr < n_rad; ++rr) {
      const uint32_t v = cnt[(size_t)rr * group_rows + k];
      if (!v) continue;
      if (by_position == 2)   // counts [tile][stride = radii per sweep][32] (dc_mfma_msym.hpp)
        atomicAdd(&out[(size_t)(pq >> 5) * (stride * 32) + (size_t)rr * 32 + (pq & 31u)], v);
      else
        atomicAdd(&out[(size_t)rr * stride + (by_position ? pq : fq)], v);
    }
  }
}

#include "dc_prep.hpp"

}  // namespace

// dynamic LDS of the measuring order_key_kernel (dc_prep.hpp)
static size_t order_key_smem_set(uint32_t n_cols) {
  const size_t bytes = order_key_smem(n_cols, true);
  if (bytes > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(order_key_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return bytes;
}
// dynamic LDS of order_rows2_kernel (dc_prep.hpp); beyond 64 KB (wide rows) the launch has to ask for it
static size_t order_rows_smem_set(uint32_t n_cols) {
  const size_t bytes = order_rows_smem(n_cols);
  if (bytes > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(order_rows2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return bytes;
}

#define DC_FOR_EACH_S(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
DC_FOR_EACH_S(DC_DECLARE_STEP)

// ---- sweep timers (see dc_mfma.hpp) -------------------------------------------------------------------
namespace {
struct SweepTimer {
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool started = false;
};
std::atomic<bool> g_timing{false};
SweepTimer g_timers[16][2];   // [device & 15][kind]
std::mutex g_timer_mutex;
}  // namespace

void sweep_timer_enable(bool on) { g_timing.store(on); }

void sweep_timer_mark(int kind, bool begin, hipStream_t s) {
  if (!g_timing.load(std::memory_order_relaxed)) return;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lock(g_timer_mutex);
  SweepTimer& t = g_timers[dev & 15][kind & 1];
  if (!t.ev[0]) {
    if (hipEventCreate(&t.ev[0]) != hipSuccess || hipEventCreate(&t.ev[1]) != hipSuccess) return;
  }
  if (begin) {
    if (!t.started) (void)hipEventRecord(t.ev[0], s);
    t.started = true;
  } else {
    (void)hipEventRecord(t.ev[1], s);
  }
}

int sweep_timer_read(int kind, float* ms) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  std::lock_guard<std::mutex> lock(g_timer_mutex);
  SweepTimer& t = g_timers[dev & 15][kind & 1];
  if (!t.started || !t.ev[0]) return -1;
  t.started = false;
  if (hipEventSynchronize(t.ev[1]) != hipSuccess) return -1;
  return hipEventElapsedTime(ms, t.ev[0], t.ev[1]) == hipSuccess ? 0 : -1;
}

void launch_pop_cross(const float* coords, uint32_t n_cols, const float* coords_r, const uint32_t* perm_r,
                      const float4* box_r, const uint32_t* perm_q, const float4* box_q, const uint32_t* tile_comp_q,
                      const uint32_t* comp, uint32_t T_q, uint32_t group_tiles, QSeg q_seg, int q_in_ref_order,
                      const Rad2& rad2, int n_rad, const uint32_t* hdr, uint32_t* out, size_t stride, int by_position,
                      hipStream_t s) {
  const uint32_t groups = seg_groups((T_q + group_tiles - 1) / group_tiles, q_seg);
  if (groups == 0 || n_rad <= 0 || 32u * group_tiles > (uint32_t)kMaxGroupRows) return;
  float r2max = rad2.v[0];
  for (int r = 1; r < n_rad; ++r) r2max = std::max(r2max, rad2.v[r]);
  hipLaunchKernelGGL(pop_cross_kernel, dim3(groups), dim3(64), 0, s, coords, n_cols, coords_r, perm_r, box_r, perm_q, box_q,
                     tile_comp_q, comp, T_q, group_tiles, q_seg, q_in_ref_order, rad2, n_rad, r2max, hdr, out, stride,
                     by_position);
}

// diagnostics of the last pruned population sweep in a workspace: components, global and component-wise extent
int components_info(const void* d_ws, size_t n_rows, size_t n_cols, uint32_t* n_comp, float* m_global, float* m_local,
                    float* scale, hipStream_t stream) {
  const Layout L = make_layout(n_rows, n_cols);
  const char* p = (const char*)d_ws;
  uint32_t hdr[32], grid[8];
  if (hipMemcpyAsync(hdr, p, sizeof(hdr), hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
  if (hipMemcpyAsync(grid, p + L.off_comp, sizeof(grid), hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
  if (hipStreamSynchronize(stream) != hipSuccess) return -1;
  *n_comp = grid[5];
  memcpy(m_global, &hdr[0], 4);
  *m_local = (grid[5] > 1u) ? __builtin_bit_cast(float, hdr[kHdrMloc]) : *m_global;
  memcpy(scale, &hdr[kHdrScale + 1], 4);
  return 0;
}

bool mfma_supports(size_t n_cols) {
  if (n_cols < 1 || n_cols > (size_t)kMaxCols) return false;
  return ((DC_STEP_MASK >> (nm_for((int)n_cols) - 1)) & 1u) != 0;
}
size_t mfma_workspace_bytes(size_t n_rows, size_t n_cols) {
  if (!mfma_supports(n_cols) || n_rows == 0) return 0;
  return make_layout(n_rows, n_cols).fixed_end + sort_temp_bytes(n_rows + kOrderPadRows);   // (the padded orders)
}

// whose statistics / components a workspace holds: the array (address) and its shape
static uint32_t data_cookie(const float* d_coords, uint32_t n_rows, uint32_t n_cols) {
  return (0x5354A7u ^ (n_rows * 2654435761u) ^ (n_cols * 40503u) ^ (uint32_t)((uintptr_t)d_coords >> 4)) | 1u;
}

// DC_POP_CELL_FRAMES / DC_NN_CELL_FRAMES: measurement overrides of the frames per cell of the orderings
static float cell_frames(bool nn) {
  static const float v[2] = {[] { const char* e = getenv("DC_POP_CELL_FRAMES"); return (e && e[0]) ? (float)atof(e) : kPopCellFrames; }(),
                             [] { const char* e = getenv("DC_NN_CELL_FRAMES"); return (e && e[0]) ? (float)atof(e) : kNnCellFrames; }()};
  return v[nn ? 1 : 0];
}
static float nn_cell_frames() { return cell_frames(true); }

int mfma_prepare(const float* d_coords, uint32_t n_rows, uint32_t n_cols, void* d_ws,
                 bool natural_image, hipStream_t stream, bool stats_valid, bool pruned, const float* d_fe) {
  // pruned: a pruned sweep follows -- the statistics come from ONE pass (stats_kernel, dc_prep.hpp); the column means and
  // the extents max |x - mean|^2 / max |x - origin|^2 follow in the passes of its preparation that read the rows anyway
  // (components_kernel, order_key_kernel).  The full sweeps and the fp32-MFMA instance need means and max norm before
  // their first image: the three passes of rounds 1 - 4.
  char* p = (char*)d_ws;
  const uint32_t cookie = data_cookie(d_coords, n_rows, n_cols);
  if (stats_valid) {
    // DC_FLAG_STATS_VALID: means, max norm, flag and bounding box of an earlier sweep over the same coordinates
    // stay; the claim is checked on the device (content fingerprint, cookie) and the per-sweep words start over -- two
    // launches (dc_prep.hpp: claim_pre_kernel / claim_guard_kernel; five in rounds 3 - 4).  d_fe (a pruned neighbour call):
    // the range of the free energies and the component guard + fine grids of that sweep ride along.
    const Layout L = make_layout(n_rows, n_cols);
    const size_t total = (size_t)n_rows * n_cols;
    const uint32_t B = (uint32_t)std::max<size_t>(1, std::min<size_t>(kClaimBlocks, (total + 1023) / 1024));
    unsigned long long* tab = (unsigned long long*)(p + L.fixed_end);   // (the sort's temp region: free until the sort)
    const bool nn_reuse = pruned && d_fe != nullptr;
    hipLaunchKernelGGL(claim_pre_kernel, dim3(B), dim3(256), 0, stream, d_coords, total, d_fe, n_rows, tab);
    hipLaunchKernelGGL(claim_guard_kernel, dim3(1), dim3(1024), 0, stream, (uint32_t*)p, cookie, (const unsigned long long*)tab, B,
                       pruned ? 1 : 0, d_fe != nullptr ? 1 : 0, n_cols, nn_reuse ? (uint32_t*)(p + L.off_comp) : (uint32_t*)nullptr,
                       n_rows, nn_reuse ? nn_cell_frames() : 0.0f, cell_key_bits(n_rows, kNnCellFrames) + 1u);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  if (pruned) {
    const Layout L = make_layout(n_rows, n_cols);
    // header and component region in ONE fill (the layout puts them side by side), then one pass over the coordinates
    if (hipMemsetAsync(p, 0, L.off_img, stream) != hipSuccess) return -1;   // (a whole number of 256-byte lines: one fill kernel)
    // (its per-block table: the sort's temp region, free until the sort; components_kernel adds it up)
    const size_t room = sort_temp_bytes(n_rows + kOrderPadRows) / (sizeof(double) * kStatsRow);
    const size_t want = ((size_t)n_rows * n_cols + 1023) / 1024;
    const uint32_t blocks_s = (uint32_t)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(kStatsMaxBlocks, want), room));
    hipLaunchKernelGGL(stats_kernel, dim3(blocks_s), dim3(256), 0, stream, d_coords, n_rows, n_cols, (uint32_t*)p, cookie,
                       (double*)(p + L.fixed_end));
    hipLaunchKernelGGL(stats_reduce_kernel, dim3(1), dim3(1024), 0, stream, (uint32_t*)p, (const double*)(p + L.fixed_end), n_rows,
                       n_cols);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  if (hipMemsetAsync(p, 0, kHdrBytes, stream) != hipSuccess) return -1;
  const uint32_t blocks = (uint32_t)std::min<size_t>(512, ((size_t)n_rows * n_cols + 255) / 256);
  hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, stream, d_coords, n_rows, n_cols,
                     (double*)(p + kHdrSums));
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, stream, (const double*)(p + kHdrSums), n_rows,
                     n_cols, (float*)(p + kHdrMeans));
  // (512 blocks: every block ends with five same-address atomics -- at 2 048 blocks they were half of the kernel's 46 us,
  //  scratch/pb/rowstats_bench.hip)
  hipLaunchKernelGGL(rowstats_kernel, dim3(std::min<uint32_t>((n_rows + 255) / 256, 512u)), dim3(256), 0, stream, d_coords, n_rows, n_cols,
                     (const float*)(p + kHdrMeans), (uint32_t*)p, cookie);
  (void)natural_image;   // (the full sweeps build their natural-order images themselves, at their own scale)
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The matrix-core population kernels take ONE radius per sweep: with the threshold folded into the
// accumulator a radius costs 16 v_alignbit + 8 v_min3 per 1024 pairs, less than the subtract /
// sign / min triple per radius of a multi-radius epilogue -- and the MFMAs of a repeated sweep hide
// behind that epilogue.  Radii are therefore swept one after the other (the pruned sweep then also
// runs on its own pruned survivor lists).
static Rad2 single_radius(const Rad2& rad2, int r) {
  Rad2 one;
  for (int k = 0; k < kMaxRadiiPerLaunch; ++k) one.v[k] = -1.0f;
  one.v[0] = rad2.v[r];
  return one;
}

static float max_radius2(const Rad2& rad2, int n_rad) {
  float m = 0.0f;
  for (int r = 0; r < n_rad; ++r) m = std::max(m, rad2.v[r]);
  return m;
}
// operand images of the frames in natural order (the full sweeps): A form + norms and / or B form
static void natural_images(const float* d_coords, uint32_t n_rows, uint32_t n_cols, void* d_ws, bool a_form,
                           bool b_form, hipStream_t stream) {
  const Layout L = make_layout(n_rows, n_cols);
  char* p = (char*)d_ws;
  const dim3 grid_img((uint32_t)(((size_t)L.T * L.NM * 64 + 255) / 256));
  if (a_form)
    hipLaunchKernelGGL(image_kernel, grid_img, dim3(256), image_smem(n_cols), stream, d_coords, n_rows, n_rows, n_cols,
                       L.NM, L.T, (const float*)(p + kHdrMeans), (const uint32_t*)nullptr, 0,
                       (uint4*)(p + L.off_img), (float*)(p + L.off_norm), (const uint32_t*)p);
  if (b_form)
    hipLaunchKernelGGL(image_kernel, grid_img, dim3(256), image_smem(n_cols), stream, d_coords, n_rows, n_rows, n_cols,
                       L.NM, L.T, (const float*)(p + kHdrMeans), (const uint32_t*)nullptr, 1,
                       (uint4*)(p + L.off_img_b), a_form ? (float*)nullptr : (float*)(p + L.off_norm),
                       (const uint32_t*)p);
}

void launch_pop_mfma(const float* d_coords, uint32_t n_rows, uint32_t n_cols, uint32_t i_from,
                     uint32_t i_to, const Rad2& rad2, int n_rad, uint32_t* d_pops, void* d_ws,
                     hipStream_t stream) {
  hipLaunchKernelGGL(scale_kernel, dim3(1), dim3(1), 0, stream, (uint32_t*)d_ws, max_radius2(rad2, n_rad), n_cols);
  natural_images(d_coords, n_rows, n_cols, d_ws, true, true, stream);
  for (int r = 0; r < n_rad; ++r) {
    const Rad2 one = single_radius(rad2, r);
    uint32_t* out = d_pops + (size_t)r * n_rows;
    switch (nm_for((int)n_cols)) {
#define X(SV)                                                                                   \
  case SV:                                                                                      \
    if ((DC_STEP_MASK >> (SV - 1)) & 1u)                                                        \
      pop_mfma_step_##SV(d_coords, n_rows, n_cols, d_ws, i_from, i_to, one, 1, out, stream);     \
    break;
      DC_FOR_EACH_S(X)
#undef X
      default:
        break;
    }
  }
}

// rows answered by a pruned sweep: the row range [i_from, i_to), or -- n_segments > 0 -- segment
// `segment` of the spatial order cut into n_segments runs of whole query groups
struct QuerySel {
  uint32_t i_from, i_to, segment, n_segments;
};
static int tq_of(uint32_t n_cols) { return nm_for((int)n_cols) <= 5 ? 4 : 2; }   // = tq_for<NM>
static int tq_nn_of(uint32_t n_cols) { return nm_for((int)n_cols) <= 2 ? DC_NN_TQ_SMALL : tq_of(n_cols); }   // = tq_nn_for<NM>
static int tq_pop_of(uint32_t n_cols) { return nm_for((int)n_cols) <= 2 ? 6 : tq_of(n_cols); }   // = tq_pop_for<NM>
// query tiles per group of the population sweep that will run: the unit segments are dealt out in and the
// query image is built for (pop_shared_kernel: the four waves of a workgroup form one group)
static uint32_t pop_group_tiles(uint32_t n_rows, uint32_t n_cols, bool sink, int n_rad) {
  if (!sink && pop_shared_wanted(n_rows, n_cols, n_rad)) return 4u * (uint32_t)tq_shared_of(n_cols, n_rad);
  return (uint32_t)tq_pop_of(n_cols);
}
// radii per sweep: one (the folded-threshold sweeps), except the shared-operand sweep of wide rows: up to eight
static bool pop_multi_radius(uint32_t n_rows, uint32_t n_cols, int n_rad) {
  const int nm = nm_for((int)n_cols);
  return n_rad > 1 && nm >= 3 && nm <= 8 && pop_shared_wanted(n_rows, n_cols, n_rad);
}

// r2_scale: the largest squared radius the prepared images have to serve (the radii of the whole call)
// comp_clean: the component region of the workspace has been zero-filled by this call already (mfma_prepare)
static void pop_pruned_one(const float* d_coords, uint32_t n_rows, uint32_t n_cols, const QuerySel& qs,
                           const Rad2& rad2, int n_rad, uint32_t* d_pops, void* d_ws,
                           const EdgeSink* sink, hipStream_t stream, float r2_scale, bool prep = true, bool comp_clean = false);

void launch_pop_pruned(const float* d_coords, uint32_t n_rows, uint32_t n_cols, uint32_t i_from,
                       uint32_t i_to, const Rad2& rad2, int n_rad, uint32_t* d_pops, void* d_ws,
                       hipStream_t stream, bool comp_clean) {
  if (pop_multi_radius(n_rows, n_cols, n_rad)) {
    pop_pruned_one(d_coords, n_rows, n_cols, QuerySel{i_from, i_to, 0, 0}, rad2, n_rad, d_pops, d_ws, nullptr, stream,
                   max_radius2(rad2, n_rad), true, comp_clean);
    return;
  }
  for (int r = 0; r < n_rad; ++r)
    pop_pruned_one(d_coords, n_rows, n_cols, QuerySel{i_from, i_to, 0, 0}, single_radius(rad2, r), 1,
                   d_pops + (size_t)r * n_rows, d_ws, nullptr, stream, max_radius2(rad2, n_rad),
                   r == 0, comp_clean);   // one preparation for all radii
}

void launch_pop_pruned_segment(const float* d_coords, uint32_t n_rows, uint32_t n_cols,
                               uint32_t segment, uint32_t n_segments, const Rad2& rad2, int n_rad,
                               uint32_t* d_pops, void* d_ws, hipStream_t stream, bool comp_clean) {
  if (pop_multi_radius(n_rows, n_cols, n_rad)) {
    pop_pruned_one(d_coords, n_rows, n_cols, QuerySel{0, n_rows, segment, n_segments}, rad2, n_rad, d_pops, d_ws,
                   nullptr, stream, max_radius2(rad2, n_rad), true, comp_clean);
    return;
  }
  for (int r = 0; r < n_rad; ++r)
    pop_pruned_one(d_coords, n_rows, n_cols, QuerySel{0, n_rows, segment, n_segments},
                   single_radius(rad2, r), 1, d_pops + (size_t)r * n_rows, d_ws, nullptr, stream,
                   max_radius2(rad2, n_rad), r == 0, comp_clean);
}

// positions of the sweep's spatial order -> frame ids, for the pairs actually written; a flagged
// data set (the matrix-core kernel stood down) reports count = ~0
__global__ void edges_to_frames_kernel(uint2* __restrict__ edges, const unsigned long long* __restrict__ count,
                                       unsigned long long capacity, const uint32_t* __restrict__ perm) {
  const unsigned long long n = *count < capacity ? *count : capacity;
  for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (unsigned long long)gridDim.x * blockDim.x) {
    const uint2 e = edges[k];
    edges[k] = make_uint2(perm[e.x], perm[e.y]);
  }
}
__global__ void edges_flag_kernel(const uint32_t* __restrict__ hdr, unsigned long long* __restrict__ count) {
  if (hdr[1] != 0) *count = ~0ull;
}

// per-frame values into the sweep's order (component ids, ranks)
__global__ void gather_u32_kernel(const uint32_t* __restrict__ by_frame, const uint32_t* __restrict__ perm,
                                  uint32_t n, uint32_t* __restrict__ by_pos) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) by_pos[p] = (perm[p] != kInvalidFrame) ? by_frame[perm[p]] : 0xFFFFFFFFu;   // (pad positions of the order)
}

// number of unordered pairs from the populations: sum(pop - 1) / 2 (every pair is counted at both ends)
__global__ void pairs_from_pops_kernel(const uint32_t* __restrict__ pops, uint32_t n_rows,
                                       unsigned long long* __restrict__ twice) {
  unsigned long long s = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += gridDim.x * blockDim.x)
    s += pops[i] - 1u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(twice, s);
}
__global__ void halve_kernel(unsigned long long* __restrict__ v) { *v >>= 1; }

void launch_radius_pairs(const float* d_coords, uint32_t n_rows, uint32_t n_cols, float r2,
                         uint32_t* d_pops, uint2* d_pairs, unsigned long long capacity,
                         unsigned long long* d_count, void* d_ws, hipStream_t stream) {
  const Layout L = make_layout(n_rows, n_cols);
  Rad2 one;
  for (int k = 0; k < kMaxRadiiPerLaunch; ++k) one.v[k] = -1.0f;
  one.v[0] = r2;
  (void)hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream);
  if (d_pairs && capacity) {
    const EdgeSink sink{d_pairs, d_count, capacity, nullptr, nullptr, nullptr};
    pop_pruned_one(d_coords, n_rows, n_cols, QuerySel{0, n_rows, 0, 0}, one, 1, d_pops, d_ws, &sink, stream, r2, true, true);
    hipLaunchKernelGGL(edges_to_frames_kernel, dim3(1024), dim3(256), 0, stream, d_pairs,
                       (const unsigned long long*)d_count, capacity,
                       (const uint32_t*)((char*)d_ws + L.off_perm_p));
  } else {
    // counting only: the plain population sweep knows the answer
    pop_pruned_one(d_coords, n_rows, n_cols, QuerySel{0, n_rows, 0, 0}, one, 1, d_pops, d_ws, nullptr, stream, r2, true, true);
    hipLaunchKernelGGL(pairs_from_pops_kernel, dim3(256), dim3(256), 0, stream, (const uint32_t*)d_pops,
                       n_rows, d_count);
    hipLaunchKernelGGL(halve_kernel, dim3(1), dim3(1), 0, stream, d_count);
  }
  hipLaunchKernelGGL(edges_flag_kernel, dim3(1), dim3(1), 0, stream, (const uint32_t*)d_ws, d_count);
}

void launch_radius_min_edge(const float* d_coords, uint32_t n_rows, uint32_t n_cols, float r2,
                            const uint32_t* d_comp, const uint32_t* d_rank, unsigned long long* d_best,
                            uint32_t* d_pops, void* d_ws, hipStream_t stream, uint32_t segment,
                            uint32_t n_segments) {
  Rad2 one;
  for (int k = 0; k < kMaxRadiiPerLaunch; ++k) one.v[k] = -1.0f;
  one.v[0] = r2;
  (void)hipMemsetAsync(d_best, 0xFF, sizeof(unsigned long long) * n_rows, stream);
  // comp / rank arrive per FRAME; pop_pruned_one gathers them into the sweep's order
  const EdgeSink sink{nullptr, nullptr, 0, d_comp, d_rank, d_best};
  pop_pruned_one(d_coords, n_rows, n_cols, QuerySel{0, n_rows, segment, n_segments}, one, 1, d_pops, d_ws,
                 &sink, stream, r2, true, true);
}

uint32_t seg_block(uint32_t n_segments) { return n_segments <= 1u ? 1u : kSegBlockGroups; }

// DC_POP_COMPONENTS=0: one component whatever the data looks like (measurements, tests)
static bool components_off() {
  static const bool off = [] {
    const char* v = getenv("DC_POP_COMPONENTS");
    return v && v[0] == '0';
  }();
  return off;
}

static void pop_pruned_one(const float* d_coords, uint32_t n_rows, uint32_t n_cols, const QuerySel& qs,
                           const Rad2& rad2, int n_rad, uint32_t* d_pops, void* d_ws,
                           const EdgeSink* sink_in, hipStream_t stream, float r2_scale, bool prep, bool comp_clean) {
  // prep == false: the orderings, images and boxes of the previous call (same coordinates, same query
  // selection) are still in the workspace -- the further radii of one populations call
  const uint32_t i_from = qs.i_from, i_to = qs.i_to;
  const Layout L = make_layout(n_rows, n_cols);
  char* p = (char*)d_ws;
  uint32_t* hdr = (uint32_t*)p;
  uint32_t* keys_in = (uint32_t*)(p + L.off_keys_in);
  uint32_t* keys_out = (uint32_t*)(p + L.off_keys_out);
  uint32_t* vals_in = (uint32_t*)(p + L.off_vals_in);
  uint32_t* perm_p = (uint32_t*)(p + L.off_perm_p);
  uint32_t* perm_q = (uint32_t*)(p + L.off_perm_q);
  uint32_t* comp = (uint32_t*)(p + L.off_comp);
  uint32_t* tile_comp = (uint32_t*)(p + L.off_tile_comp);
  uint32_t* tile_comp_q = (uint32_t*)(p + L.off_tile_comp_q);
  const float* origins = (const float*)(comp + kCompOrigin);
  constexpr float kCellFramesHere = kPopCellFrames;
  const dim3 blk(256), grid_n((n_rows + 255) / 256);
  const bool full = (i_from == 0 && i_to == n_rows);
  uint32_t n_q = i_to - i_from;
  int q_mode = full ? kQueryAll : kQueryOwnOrder;
  QSeg q_seg{1u, 0u, 1u};
  if (qs.n_segments > 0) {   // one segment of a sharded run: every n_segments-th query group of all rows
    q_mode = kQueryAll;
    n_q = n_rows;
    q_seg = QSeg{qs.n_segments, qs.segment, seg_block(qs.n_segments)};
  }
  // The orders are PADDED: every component of the frames (dc_mfma_kernels.hpp "components") starts at a whole query
  // group; the sort's last pass writes its values straight to the padded positions (SortRemap: segment starts -> bases),
  // the pad positions in between keep the kInvalidFrame that order_meta_kernel's presets left there.
  const uint32_t tq = pop_group_tiles(n_rows, n_cols, sink_in != nullptr, n_rad), group_rows = 32u * tq;
  const uint32_t T_r = (n_rows + (uint32_t)kMaxComp * (group_rows - 1u) + 31u) / 32u;
  const uint32_t T_q = (n_q + (uint32_t)kMaxComp * (group_rows - 1u) + 31u) / 32u;
  const size_t tmp_bytes = sort_temp_bytes(n_rows + kOrderPadRows);
  const float r_max = sqrtf(fmaxf(r2_scale, 0.0f)) * 1.0001f;
  // connectivity length of the components: the largest radius itself for the sweeps that list pairs (no pair between
  // components), half of it for the plain sweeps (pairs between adjacent components: pop_cross_kernel)
  const float r_conn = sink_in ? r_max : 0.5f * r_max;
  // (the cells of all components are numbered consecutively: fewer than 2^fine_bits keys)
  const unsigned fine_bits = cell_key_bits(n_rows, kPopCellFrames) + 1u;
  const unsigned key_bits = fine_bits;
  EdgeSink sink_local;
  const EdgeSink* sink = sink_in;
  if (sink_in && sink_in->best) {   // (component ids and ranks in the sweep's order: gathered below)
    sink_local = *sink_in;
    sink_local.comp = keys_in;
    sink_local.rank = keys_out;
    sink = &sink_local;
  }
  // the counts by position of the one-radius symmetric per-wave sweep (the pq region): cleared by this call's preparation
  const bool pos_clean = prep && sink_in == nullptr && n_rad == 1 && !pop_shared_wanted(n_rows, n_cols, n_rad) &&
                         pop_sym_wanted(false, q_mode, q_seg, n_rows, n_rad) && tq <= 6u;
  // the multi-radius symmetric sweep takes its thresholds off the accumulator in place, one MFMA per radius: the band of
  // the scale has to pay for those steps (guard_shift; NR - 1 of them, NR = 4 or 8 radii per sweep)
  const int shift_steps = (sink_in == nullptr && pop_multi_radius(n_rows, n_cols, n_rad)) ? (n_rad > 4 ? 7 : 3) : 0;
  if (prep) {
    // (round 5: the passes of dc_prep.hpp -- twelve launches for the thirty of rounds 3 - 4, same values)
    const uint32_t cookie = data_cookie(d_coords, n_rows, n_cols);
    uint32_t* start_r = comp + kCompStart, *start_q = comp + kCompStart + (kMaxComp + 1);
    uint32_t* range_r = comp + kCompRange, *range_q = comp + kCompRange + kCompRangeStride;
    uint32_t* base_r = comp + kCompBase, *base_q = comp + kCompBase + (kMaxComp + 1);
    // components of the frames for this call's largest radius, their origins, the column means and the fine grids
    if (!comp_clean) (void)hipMemsetAsync(comp, 0, sizeof(uint32_t) * kCompWords, stream);
    hipLaunchKernelGGL(fine_mark_kernel, grid_n, blk, 0, stream, d_coords, n_rows, n_cols, (const uint32_t*)hdr, r_conn,
                       comp);
    hipLaunchKernelGGL(components_kernel, dim3(1), dim3(1024), components_smem(), stream, (const uint32_t*)hdr,
                       (const float*)(p + kHdrMeans), n_cols, r_conn, n_rows, kPopCellFrames, fine_bits, comp,
                       components_off() ? 1 : 0, r_max, cookie, (const double*)(p + L.fixed_end), cell_frames(false));
    // order all frames by (component, fine cell): keys, rows per component, the extents, the pad presets ...
    const uint32_t kb_r = std::min<uint32_t>((std::max(n_rows, 32u * T_r) + 255) / 256, 1024u);
    uint32_t* cnt_tab = (uint32_t*)(p + L.fixed_end);   // (rows per component and block: the sort's temp region, free until the sort)
    hipLaunchKernelGGL(order_key_kernel, dim3(kb_r), blk, order_key_smem_set(n_cols), stream, d_coords, n_cols, hdr,
                       (const uint32_t*)comp, (uint32_t)fine_bits, 0u, n_rows, keys_in, vals_in, n_rows, (const float*)nullptr, 0u,
                       cnt_tab, 1, perm_p, tile_comp, 32u * T_r);
    // ... where the components start, the scale of the sweep (it follows the components' extents) ...
    hipLaunchKernelGGL(order_meta_kernel, dim3(1), dim3(1024), 0, stream, hdr, comp, (const uint32_t*)cnt_tab, kb_r, start_r, range_r,
                       base_r, n_rows, group_rows, 1, fmaxf(r2_scale, 0.0f), n_cols, shift_steps);
    // ... the sort, whose last pass moves every component to a whole query group of the padded order ...
    {
      const SortRemap remap{start_r, base_r, (uint32_t)kMaxComp, tile_comp};
      if (sort_pairs_u32(keys_in, keys_out, vals_in, perm_p, n_rows, p + L.fixed_end, tmp_bytes, stream, key_bits, &remap))
        return;
    }
    // ... and the rows in that order (the deferred exact path reads them without a permutation look-up), the tile boxes
    // and the operand images: A form of every tile, B form of the query groups of this segment (queries in the
    // reference order)
    {
      const bool ref_queries = q_mode != kQueryOwnOrder;
      hipLaunchKernelGGL(order_rows2_kernel, dim3((32 * T_r + 255) / 256), blk, order_rows_smem_set(n_cols), stream, d_coords, n_cols,
                         L.NM, (const uint32_t*)perm_p, T_r, (float*)(p + L.off_coords_p), (float4*)(p + L.off_box_p),
                         (const float*)nullptr, (float*)nullptr, (uint32_t*)nullptr, (float2*)nullptr, (const uint32_t*)tile_comp,
                         origins, hdr, (uint4*)(p + L.off_img_p), 0, (float*)(p + L.off_norm_p),
                         ref_queries ? (uint4*)(p + L.off_img_q) : (uint4*)nullptr, (float*)nullptr, tq, q_seg,
                         (unsigned long long*)(comp + kCompHash), pos_clean ? (uint32_t*)(p + L.off_pq) : (uint32_t*)nullptr, 1u);
    }
    // lightest-outgoing-pair variant: component ids and ranks in the sweep's order (the sort's key
    // buffers are free again)
    if (sink_in && sink_in->best) {
      hipLaunchKernelGGL(gather_u32_kernel, dim3((32 * T_r + 255) / 256), blk, 0, stream, sink_in->comp,
                         (const uint32_t*)perm_p, 32u * T_r, keys_in);
      hipLaunchKernelGGL(gather_u32_kernel, dim3((32 * T_r + 255) / 256), blk, 0, stream, sink_in->rank,
                         (const uint32_t*)perm_p, 32u * T_r, keys_out);
    }
    if (q_mode == kQueryOwnOrder) {
      // query rows of this call: the same ordering restricted to [i_from, i_to)
      const uint32_t kb_q = std::min<uint32_t>((std::max(n_q, 32u * T_q) + 255) / 256, 1024u);
      hipLaunchKernelGGL(order_key_kernel, dim3(kb_q), blk, 0, stream, d_coords, n_cols, hdr,
                         (const uint32_t*)comp, (uint32_t)fine_bits, i_from, i_to, keys_in, vals_in, n_rows, (const float*)nullptr, 0u,
                         cnt_tab, 0, perm_q, tile_comp_q, 32u * T_q);
      hipLaunchKernelGGL(order_meta_kernel, dim3(1), dim3(1024), 0, stream, hdr, comp, (const uint32_t*)cnt_tab, kb_q, start_q, range_q,
                         base_q, n_q, group_rows, 0, 0.0f, n_cols);
      const SortRemap remap{start_q, base_q, (uint32_t)kMaxComp, tile_comp_q};
      if (sort_pairs_u32(keys_in, keys_out, vals_in, perm_q, n_q, p + L.fixed_end, tmp_bytes, stream, key_bits, &remap))
        return;
      hipLaunchKernelGGL(order_rows2_kernel, dim3((32 * T_q + 255) / 256), blk, order_rows_smem_set(n_cols), stream, d_coords, n_cols,
                         L.NM, (const uint32_t*)perm_q, T_q, (float*)nullptr, (float4*)(p + L.off_box_q),
                         (const float*)nullptr, (float*)nullptr, (uint32_t*)nullptr, (float2*)nullptr, (const uint32_t*)tile_comp_q,
                         origins, hdr, (uint4*)nullptr, 0, (float*)nullptr, (uint4*)(p + L.off_img_q), (float*)(p + L.off_norm_q), 1u,
                         QSeg{1u, 0u, 1u}, (unsigned long long*)nullptr, (uint32_t*)nullptr, 0u);
    }
  }
  (void)kCellFramesHere;
  const uint32_t n_pos_q = 32u * ((q_mode == kQueryOwnOrder) ? T_q : T_r);
  switch (nm_for((int)n_cols)) {
#define X(SV)                                                                                 \
  case SV:                                                                                    \
    if ((DC_STEP_MASK >> (SV - 1)) & 1u)                                                      \
      pop_pruned_step_##SV(d_coords, n_rows, n_cols, d_ws, T_r, n_pos_q, q_mode, q_seg, rad2, n_rad, \
                           d_pops, sink, stream, pos_clean);                                  \
    break;
    DC_FOR_EACH_S(X)
#undef X
    default:
      break;
  }
}

static void nn_pruned_sel(const float* d_coords, uint32_t n_rows, uint32_t n_cols, const float* d_fe,
                          const QuerySel& qs, uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx,
                          float* d_hd_d2, void* d_ws, hipStream_t stream, bool reuse_components, bool comp_clean);

void launch_nn_pruned(const float* d_coords, uint32_t n_rows, uint32_t n_cols, const float* d_fe,
                      uint32_t i_from, uint32_t i_to, uint32_t* d_nn_idx, float* d_nn_d2,
                      uint32_t* d_hd_idx, float* d_hd_d2, void* d_ws, hipStream_t stream, bool reuse_components) {
  // (a neighbour call that does not reuse the components follows a mfma_prepare that zero-filled their region)
  nn_pruned_sel(d_coords, n_rows, n_cols, d_fe, QuerySel{i_from, i_to, 0, 0}, d_nn_idx, d_nn_d2,
                d_hd_idx, d_hd_d2, d_ws, stream, reuse_components, !reuse_components);
}

void launch_nn_pruned_segment(const float* d_coords, uint32_t n_rows, uint32_t n_cols,
                              const float* d_fe, uint32_t segment, uint32_t n_segments,
                              uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx, float* d_hd_d2,
                              void* d_ws, hipStream_t stream, bool reuse_components) {
  nn_pruned_sel(d_coords, n_rows, n_cols, d_fe, QuerySel{0, n_rows, segment, n_segments}, d_nn_idx,
                d_nn_d2, d_hd_idx, d_hd_d2, d_ws, stream, reuse_components, !reuse_components);
}

// Neighbours that lie in ANOTHER component than their query (dc_mfma_kernels.hpp "components"): the matrix-core sweep
// only ever looks inside the query's own component, so afterwards every query checks whether a frame of another
// component could still beat (or tie) what it has -- the squared gap between the query and the component's box in
// columns 0/1 bounds every distance into it from below -- and the few queries for which that is so (the frames at the
// edge of a cluster; the lowest free energies of a cluster, whose lower-free-energy neighbour is in another one)
// search those components exactly: tiles whose boxes are within the current incumbent, every row of a surviving tile
// in the canonical arithmetic, lexicographic minimum on (d2, frame).  One wave per 64 positions of the query order;
// an open query is served by the whole wave.
__device__ __forceinline__ float point_box_gap2(float x, float y, const float4& b) {
  const float dx = fmaxf(0.0f, fmaxf(b.x - x, x - b.y));
  const float dy = fmaxf(0.0f, fmaxf(b.z - y, y - b.w));
  return (dx * dx + dy * dy) * 0.99999f;   // (a lower bound of every d2 into the box, its own rounding included)
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off, 64);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off, 64);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    v = o < v ? o : v;
  }
  return v;
}
// Which queries can have a neighbour in ANOTHER component: their incumbent (or, where a lower-energy neighbour is possible
// at all, their lower-energy incumbent) reaches another component's box.  By FRAME when the queries are the rows of the
// reference order (every read coalesced: the by-position form spent 145 us on its six scattered reads per query), by
// position of the query order otherwise.  The component of a frame is looked up as order_key_kernel assigned it.
__global__ void nn_open_kernel(const float* __restrict__ coords, uint32_t n_rows, uint32_t D, const float* __restrict__ fe,
                               const uint32_t* __restrict__ invpos_r, const uint32_t* __restrict__ perm_q, uint32_t n_items,
                               const uint32_t* __restrict__ comp, uint32_t group_rows, QSeg q_seg,
                               const uint32_t* __restrict__ hdr, const float* __restrict__ nn_d2,
                               const float* __restrict__ hd_d2, const uint32_t* __restrict__ nn_idx,
                               const uint32_t* __restrict__ hd_idx,
                               unsigned long long* __restrict__ merge64, uint32_t* __restrict__ list,
                               uint32_t* __restrict__ count) {
  // merge64 [2][n_rows] by frame: the packed (d2, frame) incumbents of the listed queries, which the search lowers
  // with 64-bit atomic minima (several waves per query) and nn_cross_write_kernel hands back.
  if (hdr[1] != 0) return;
  const uint32_t n_comp = comp[kCompGrid + 5];
  if (n_comp <= 1u) return;   // (one component: the sweep saw everything)
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  bool live = j < n_items;
  uint32_t i = j;
  if (perm_q) {   // (a position of the query order)
    i = live ? perm_q[j] : kInvalidFrame;
    live = live && i != kInvalidFrame && seg_owns(j / group_rows, q_seg);
  } else if (live) {
    live = q_seg.stride <= 1 || seg_owns(invpos_r[i] / group_rows, q_seg);
  }
  bool open = false;
  if (live) {
    const float x = coords[(size_t)i * D], y = (D > 1) ? coords[(size_t)i * D + 1] : 0.0f;
    const float inc_nn = nn_d2[i], inc_hd = hd_d2[i];
    const bool hd_possible = fkey_inv(~hdr[12]) < fe[i];
    const CoarseGrid g = coarse_grid(hdr, comp_r_conn(comp), n_rows);
    uint32_t c = 0;
    if (fabsf(x) <= FLT_MAX && fabsf(y) <= FLT_MAX) c = comp[kCompCellComp + coarse_cell_of_point(g, x, y)];
    const float4* cbox = reinterpret_cast<const float4*>(comp + kCompBox);
    for (uint32_t c2 = 0; c2 < n_comp; ++c2) {
      if (c2 == c) continue;
      const float g2 = point_box_gap2(x, y, cbox[c2]);
      open = open || (g2 <= inc_nn) || (hd_possible && g2 <= inc_hd);
    }
  }
  // one atomic per block
  __shared__ uint32_t wave_n[4], wave_base[4];
  const uint64_t m = __builtin_amdgcn_ballot_w64(open);
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
  if ((threadIdx.x & 63u) == 0) wave_n[threadIdx.x >> 6] = (uint32_t)__builtin_popcountll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t total = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
    uint32_t b = total ? atomicAdd(count, total) : 0u;
    for (int w = 0; w < 4; ++w) {
      wave_base[w] = b;
      b += wave_n[w];
    }
  }
  __syncthreads();
  const uint32_t base = wave_base[threadIdx.x >> 6];
  if (open) {
    list[base + rank] = i;
    merge64[i] = ((unsigned long long)__float_as_uint(nn_d2[i]) << 32) | nn_idx[i];
    merge64[(size_t)n_rows + i] = ((unsigned long long)__float_as_uint(hd_d2[i]) << 32) | hd_idx[i];
  }
}

// The open queries: exact search of the other components' tiles (boxes and free-energy ranges first, rows of the
// surviving tiles in the canonical order).  A work item is (query, share of the order's tiles) and takes a wave: the few
// queries of well-separated clusters (the free-energy minimum of each looks through ALL tiles of the others for its
// lower-energy neighbour: 77 us for one wave at C3) are spread over kCrossShares waves each; with many open queries
// (touching clusters) a query is one item.  Incumbents meet in merge64 (64-bit atomic minima).
constexpr uint32_t kCrossShares = 32;
__global__ __launch_bounds__(64) void nn_cross_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols, const float* __restrict__ fe,
    const float* __restrict__ coords_r, const uint32_t* __restrict__ perm_r, const float4* __restrict__ box_r,
    const float2* __restrict__ ferange_r, const float* __restrict__ fe_c, const uint32_t* __restrict__ comp,
    const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ open_list,
    const uint32_t* __restrict__ open_count, uint32_t T, unsigned long long* __restrict__ merge64) {
  __shared__ uint32_t list[256];
  __shared__ float list_gap[256];
  if (hdr[1] != 0) return;
  const uint32_t n_comp = comp[kCompGrid + 5];
  const int lane = threadIdx.x;
  const uint32_t n_open = *open_count;
  const uint32_t shares = (n_open >= (1u << 17)) ? 1u : kCrossShares;
  const uint32_t share_tiles = (T + shares - 1) / shares;
  const float4* cbox = reinterpret_cast<const float4*>(comp + kCompBox);
  const uint32_t* range = comp + kCompRange;
  const CoarseGrid g = coarse_grid(hdr, comp_r_conn(comp), n_rows);
  const float fe_floor = fkey_inv(~hdr[12]);
  for (uint32_t item = blockIdx.x; item < n_open * shares; item += gridDim.x) {
    const uint32_t e = item / shares, share = item - e * shares;
    const uint32_t s_lo = share * share_tiles, s_hi = min(s_lo + share_tiles, T);
    const uint32_t q_frame = open_list[e];
    const float* q_row = coords + (size_t)q_frame * n_cols;
    const float qx0 = q_row[0], qx1 = (n_cols > 1) ? q_row[1] : 0.0f;
    uint32_t q_comp = 0;
    if (fabsf(qx0) <= FLT_MAX && fabsf(qx1) <= FLT_MAX) q_comp = comp[kCompCellComp + coarse_cell_of_point(g, qx0, qx1)];
    const float q_fe = fe[q_frame];
    const bool q_hd = fe_floor < q_fe;
    // (what the other shares of this query have found by now is an upper bound like any other)
    unsigned long long best_nn = __atomic_load_n(&merge64[q_frame], __ATOMIC_RELAXED);
    unsigned long long best_hd = __atomic_load_n(&merge64[(size_t)n_rows + q_frame], __ATOMIC_RELAXED);
    const unsigned long long start_nn = best_nn, start_hd = best_hd;
    for (uint32_t c2 = 0; c2 < n_comp; ++c2) {
      if (c2 == q_comp) continue;   // (its own component: the sweep's business)
      {
        const float inc_nn = __uint_as_float((uint32_t)(best_nn >> 32)), inc_hd = __uint_as_float((uint32_t)(best_hd >> 32));
        const float g2 = point_box_gap2(qx0, qx1, cbox[c2]);
        if (!(g2 <= inc_nn) && !(q_hd && g2 <= inc_hd)) continue;
      }
      const uint32_t t_lo = max(range[2 * c2], s_lo), t_hi = min(range[2 * c2 + 1], s_hi);
      // 256 tiles per step: a lane tests four boxes whose loads are independent
      for (uint32_t base = t_lo; base < t_hi; base += 256) {
        const float inc_nn = __uint_as_float((uint32_t)(best_nn >> 32)), inc_hd = __uint_as_float((uint32_t)(best_hd >> 32));
        float4 bx[4];
        float flo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t t = base + 64u * (uint32_t)k + (uint32_t)lane;
          bx[k] = (t < t_hi) ? box_r[t] : make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
          flo[k] = (t < t_hi) ? ferange_r[t].x : INFINITY;
        }
        uint32_t n_list = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t t = base + 64u * (uint32_t)k + (uint32_t)lane;
          const float g2 = point_box_gap2(qx0, qx1, bx[k]);
          const bool ok = (t < t_hi) && ((g2 <= inc_nn) | (q_hd & (g2 <= inc_hd) & (flo[k] < q_fe)));
          const uint64_t m = __builtin_amdgcn_ballot_w64(ok);
          if (ok) {
            const uint32_t slot = n_list + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            list[slot] = t;
            list_gap[slot] = g2;
          }
          n_list += (uint32_t)__builtin_popcountll(m);
        }
        if (n_list == 0) continue;
        __syncthreads();
        // eight tiles per step: the half-waves take four each, a lane one row of each -- the four row fetches of a
        // lane are independent and overlap (one tile pair per step was bound by the latency of its two loads).  The
        // incumbents are renewed after every step and listed tiles they exclude by then are passed over: a query without
        // a lower-energy neighbour so far (the minimum of its component) lists every tile with a lower energy at first
        for (uint32_t s0 = 0; s0 < n_list; s0 += 8) {
          unsigned long long my_nn = ~0ull, my_hd = ~0ull;
          const float cur_nn = __uint_as_float((uint32_t)(best_nn >> 32)), cur_hd = __uint_as_float((uint32_t)(best_hd >> 32));
          uint32_t pr[4], jr[4];
          bool any = false;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t si = s0 + 2u * (uint32_t)k + (uint32_t)(lane >> 5);
            bool take = si < n_list;
            if (take) {
              const float g2 = list_gap[si];
              take = (g2 <= cur_nn) | (q_hd & (g2 <= cur_hd));
            }
            pr[k] = take ? 32u * list[si] + (uint32_t)(lane & 31) : 0xFFFFFFFFu;
            any |= take;
          }
          if (__builtin_amdgcn_ballot_w64(any) == 0) continue;
#pragma unroll
          for (int k = 0; k < 4; ++k) jr[k] = (pr[k] != 0xFFFFFFFFu) ? perm_r[pr[k]] : kInvalidFrame;
          float d2[4], fr[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool use = jr[k] != kInvalidFrame && jr[k] != q_frame;
            const size_t row = use ? (size_t)pr[k] : 0;
            d2[k] = dist2_canon_rows(q_row, coords_r + row * n_cols, (int)n_cols);
            fr[k] = fe_c[row];
            if (!use) jr[k] = kInvalidFrame;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (jr[k] != kInvalidFrame) {
              const unsigned long long key = ((unsigned long long)__float_as_uint(d2[k]) << 32) | jr[k];
              my_nn = key < my_nn ? key : my_nn;
              if (fr[k] < q_fe) my_hd = key < my_hd ? key : my_hd;
            }
          const unsigned long long w_nn = wave_min_u64(my_nn), w_hd = wave_min_u64(my_hd);
          best_nn = w_nn < best_nn ? w_nn : best_nn;
          best_hd = w_hd < best_hd ? w_hd : best_hd;
        }
      }
    }
    if (lane == 0) {
      if (best_nn < start_nn) atomicMin(&merge64[q_frame], best_nn);
      if (best_hd < start_hd) atomicMin(&merge64[(size_t)n_rows + q_frame], best_hd);
    }
    __syncthreads();
  }
}

// the answers of the listed queries back into the caller's arrays
__global__ void nn_cross_write_kernel(const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ comp,
                                      const uint32_t* __restrict__ open_list, const uint32_t* __restrict__ open_count,
                                      const unsigned long long* __restrict__ merge64, uint32_t n_rows,
                                      uint32_t* __restrict__ nn_idx, float* __restrict__ nn_d2,
                                      uint32_t* __restrict__ hd_idx, float* __restrict__ hd_d2) {
  (void)comp;
  if (hdr[1] != 0) return;
  const uint32_t n_open = *open_count;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n_open; e += gridDim.x * blockDim.x) {
    const uint32_t i = open_list[e];
    const unsigned long long a = merge64[i], b = merge64[(size_t)n_rows + i];
    nn_idx[i] = (uint32_t)a;
    nn_d2[i] = __uint_as_float((uint32_t)(a >> 32));
    hd_idx[i] = (uint32_t)b;
    hd_d2[i] = __uint_as_float((uint32_t)(b >> 32));
  }
}

static void nn_pruned_sel(const float* d_coords, uint32_t n_rows, uint32_t n_cols, const float* d_fe,
                          const QuerySel& qs, uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx,
                          float* d_hd_d2, void* d_ws, hipStream_t stream, bool reuse_components, bool comp_clean) {
  const uint32_t i_from = qs.i_from, i_to = qs.i_to;
  const Layout L = make_layout(n_rows, n_cols);
  char* p = (char*)d_ws;
  uint32_t* hdr = (uint32_t*)p;
  uint32_t* keys_in = (uint32_t*)(p + L.off_keys_in);
  uint32_t* keys_out = (uint32_t*)(p + L.off_keys_out);
  uint32_t* vals_in = (uint32_t*)(p + L.off_vals_in);
  uint32_t* perm_p = (uint32_t*)(p + L.off_perm_p);
  uint32_t* perm_q = (uint32_t*)(p + L.off_perm_q);
  uint32_t* comp = (uint32_t*)(p + L.off_comp);
  uint32_t* tile_comp = (uint32_t*)(p + L.off_tile_comp);
  uint32_t* tile_comp_q = (uint32_t*)(p + L.off_tile_comp_q);
  const float* origins = (const float*)(comp + kCompOrigin);
  const dim3 blk(256), grid_n((n_rows + 255) / 256);
  const size_t tmp_bytes = sort_temp_bytes(n_rows + kOrderPadRows);
  const bool full = (i_from == 0 && i_to == n_rows);
  uint32_t n_q = i_to - i_from;
  int q_mode = full ? kQueryAll : kQueryOwnOrder;
  QSeg q_seg{1u, 0u, 1u};
  if (qs.n_segments > 0) {   // one segment of a sharded run: every n_segments-th query group of all rows
    q_mode = kQueryAll;
    n_q = n_rows;
    q_seg = QSeg{qs.n_segments, qs.segment, seg_block(qs.n_segments)};
  }
  // (query tiles per group: a wave's, or with the shared-operand sweep the workgroup's; the orders are padded so that
  //  every component starts at a whole group -- see pop_pruned_one)
  const uint32_t tq = nn_shared_wanted(n_rows, n_cols) ? 4u * (uint32_t)tq_of(n_cols) : (uint32_t)tq_nn_of(n_cols), group_rows = 32u * tq;
  const uint32_t T_r = (n_rows + (uint32_t)kMaxComp * (group_rows - 1u) + 31u) / 32u;
  const uint32_t T_q = (n_q + (uint32_t)kMaxComp * (group_rows - 1u) + 31u) / 32u;
  // ordering key: (cell number over all components, quantised free energy) in whole sort passes
  const unsigned fine_bits = cell_key_bits(n_rows, kNnCellFrames) + 1u;
  const unsigned key_bits = (fine_bits + 9u <= 24u) ? 24u : 32u;
  unsigned fe_bits = key_bits > fine_bits ? std::min(key_bits - fine_bits, 16u) : 0u;
  {  // DC_NN_FE_BITS: measurements (0 = the frames of a cell in any order)
    static const int forced = [] { const char* e = getenv("DC_NN_FE_BITS"); return (e && e[0]) ? atoi(e) : -1; }();
    if (forced >= 0) fe_bits = std::min((unsigned)forced, fe_bits);
  }
  const float r_conn = -8.0f;   // components: connected over 8 cells of the ordering (no radius in this sweep)
  const uint32_t cookie = data_cookie(d_coords, n_rows, n_cols);
  uint32_t* start_r = comp + kCompStart, *start_q = comp + kCompStart + (kMaxComp + 1);
  uint32_t* range_r = comp + kCompRange, *range_q = comp + kCompRange + kCompRangeStride;
  uint32_t* base_r = comp + kCompBase, *base_q = comp + kCompBase + (kMaxComp + 1);
  if (reuse_components) {
    // the partition an earlier sweep over these coordinates left in the workspace (DC_FLAG_STATS_VALID: the
    // populations -> neighbours pair): checked on the device, with the range of the free energies and the fine grids of THIS
    // sweep, by the claim's guard (mfma_prepare: claim_guard_kernel)
  } else {
    // the pass over the free energies finds their range (and raises the flag for NaNs)
    hipLaunchKernelGGL(fe_key_kernel, dim3(std::min<uint32_t>(grid_n.x, 256u)), blk, 0, stream, d_fe, n_rows, (uint32_t*)nullptr,
                       (uint32_t*)nullptr, hdr);
    if (!comp_clean) (void)hipMemsetAsync(comp, 0, sizeof(uint32_t) * kCompWords, stream);
    hipLaunchKernelGGL(fine_mark_kernel, grid_n, blk, 0, stream, d_coords, n_rows, n_cols, (const uint32_t*)hdr, r_conn, comp);
    hipLaunchKernelGGL(components_kernel, dim3(1), dim3(1024), components_smem(), stream, (const uint32_t*)hdr,
                       (const float*)(p + kHdrMeans), n_cols, r_conn, n_rows, kNnCellFrames, fine_bits, comp,
                       components_off() ? 1 : 0, 0.0f, cookie, (const double*)(p + L.fixed_end), cell_frames(true));
  }
  // frames by (component, cell, free energy): ONE sort on a combined key, its last pass writes the padded order
  const uint32_t kb_r = std::min<uint32_t>((std::max(n_rows, 32u * T_r) + 255) / 256, 1024u);
  uint32_t* cnt_tab = (uint32_t*)(p + L.fixed_end);   // (rows per component and block: the sort's temp region, free until the sort)
  hipLaunchKernelGGL(order_key_kernel, dim3(kb_r), blk, order_key_smem_set(n_cols), stream, d_coords, n_cols, hdr,
                     (const uint32_t*)comp, (uint32_t)fine_bits, 0u, n_rows, keys_in, vals_in, n_rows, d_fe, (uint32_t)fe_bits, cnt_tab, 1,
                     perm_p, tile_comp, 32u * T_r);
  hipLaunchKernelGGL(order_meta_kernel, dim3(1), dim3(1024), 0, stream, hdr, comp, (const uint32_t*)cnt_tab, kb_r, start_r, range_r,
                     base_r, n_rows, group_rows, 1, -1.0f, n_cols);   // (the neighbour scale)
  {
    const SortRemap remap{start_r, base_r, (uint32_t)kMaxComp, tile_comp};
    if (sort_pairs_u32(keys_in, keys_out, vals_in, perm_p, n_rows, p + L.fixed_end, tmp_bytes, stream, fine_bits + fe_bits, &remap))
      return;
  }
  const float* coords_p = (const float*)(p + L.off_coords_p);
  // rows, boxes, free-energy ranges, and the operand images: the neighbour sweeps take the reference norms through the
  // operand image (dc_mfma_kernels.hpp "reference norms folded": a_form 2), the queries' B form for this segment's groups
  {
    const bool ref_queries = q_mode != kQueryOwnOrder;
    hipLaunchKernelGGL(order_rows2_kernel, dim3((32 * T_r + 255) / 256), blk, order_rows_smem_set(n_cols), stream, d_coords, n_cols,
                       L.NM, (const uint32_t*)perm_p, T_r, (float*)(p + L.off_coords_p), (float4*)(p + L.off_box_p), d_fe,
                       (float*)(p + L.off_fe_s), (uint32_t*)(p + L.off_invpos), (float2*)(p + L.off_ferange_p),
                       (const uint32_t*)tile_comp, origins, hdr, (uint4*)(p + L.off_img_p), 2, (float*)(p + L.off_norm_p),
                       ref_queries ? (uint4*)(p + L.off_img_q) : (uint4*)nullptr, (float*)nullptr, tq, q_seg,
                         (unsigned long long*)(comp + kCompHash), (uint32_t*)nullptr, 0u);
  }
  if (q_mode == kQueryOwnOrder) {
    // query rows of this call: the cell ordering restricted to [i_from, i_to)
    const uint32_t kb_q = std::min<uint32_t>((std::max(n_q, 32u * T_q) + 255) / 256, 1024u);
    hipLaunchKernelGGL(order_key_kernel, dim3(kb_q), blk, 0, stream, d_coords, n_cols, hdr,
                       (const uint32_t*)comp, (uint32_t)fine_bits, i_from, i_to, keys_in, vals_in, n_rows, (const float*)nullptr, 0u,
                       cnt_tab, 0, perm_q, tile_comp_q, 32u * T_q);
    hipLaunchKernelGGL(order_meta_kernel, dim3(1), dim3(1024), 0, stream, hdr, comp, (const uint32_t*)cnt_tab, kb_q, start_q, range_q,
                       base_q, n_q, group_rows, 0, 0.0f, n_cols);
    const SortRemap remap{start_q, base_q, (uint32_t)kMaxComp, tile_comp_q};
    if (sort_pairs_u32(keys_in, keys_out, vals_in, perm_q, n_q, p + L.fixed_end, tmp_bytes, stream, fine_bits, &remap))
      return;
    hipLaunchKernelGGL(order_rows2_kernel, dim3((32 * T_q + 255) / 256), blk, order_rows_smem_set(n_cols), stream, d_coords, n_cols,
                       L.NM, (const uint32_t*)perm_q, T_q, (float*)nullptr, (float4*)(p + L.off_box_q), (const float*)nullptr,
                       (float*)nullptr, (uint32_t*)nullptr, (float2*)nullptr, (const uint32_t*)tile_comp_q, origins, hdr,
                       (uint4*)nullptr, 0, (float*)nullptr, (uint4*)(p + L.off_img_q), (float*)(p + L.off_norm_q), 1u,
                       QSeg{1u, 0u, 1u}, (unsigned long long*)nullptr, (uint32_t*)nullptr, 0u);
  }
  const bool own = q_mode == kQueryOwnOrder;
  const uint32_t n_pos_q = 32u * (own ? T_q : T_r);
  switch (nm_for((int)n_cols)) {
#define X(SV)                                                                                   \
  case SV:                                                                                      \
    if ((DC_STEP_MASK >> (SV - 1)) & 1u)                                                        \
      nn_pruned_step_##SV(d_coords, n_rows, n_cols, d_fe, d_ws, T_r, n_pos_q, q_mode, q_seg, -1.0f, \
                          d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2, stream);                        \
    break;
    DC_FOR_EACH_S(X)
#undef X
    default:
      break;
  }
  // what lies in other components than the query: exact, for the few queries that can have a neighbour there (listed
  // in the sort's key buffer; their incumbents go through the merge buffer, both free again; the counter is header
  // word kHdrOpen, zeroed by scale_kernel)
  uint32_t* open_list = keys_in;
  uint32_t* open_count = hdr + kHdrOpen;
  unsigned long long* merge64 = (unsigned long long*)(p + L.off_merge64);
  const uint32_t n_items = own ? 32u * T_q : n_rows;
  hipLaunchKernelGGL(nn_open_kernel, dim3((n_items + 255) / 256), blk, 0, stream, d_coords, n_rows, n_cols, d_fe,
                     (const uint32_t*)(p + L.off_invpos), own ? (const uint32_t*)perm_q : (const uint32_t*)nullptr, n_items,
                     (const uint32_t*)comp, 32u * tq, q_seg, (const uint32_t*)hdr, (const float*)d_nn_d2,
                     (const float*)d_hd_d2, (const uint32_t*)d_nn_idx, (const uint32_t*)d_hd_idx,
                     merge64, open_list, open_count);
  hipLaunchKernelGGL(nn_cross_kernel, dim3(8192), dim3(64), 0, stream, d_coords, n_rows, n_cols, d_fe, coords_p,
                     (const uint32_t*)perm_p, (const float4*)(p + L.off_box_p), (const float2*)(p + L.off_ferange_p),
                     (const float*)(p + L.off_fe_s), (const uint32_t*)comp, (const uint32_t*)hdr,
                     (const uint32_t*)open_list, (const uint32_t*)open_count, T_r, merge64);
  hipLaunchKernelGGL(nn_cross_write_kernel, dim3(64), blk, 0, stream, (const uint32_t*)hdr, (const uint32_t*)comp,
                     (const uint32_t*)open_list, (const uint32_t*)open_count, (const unsigned long long*)merge64, n_rows,
                     d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2);
}

// ---- blocks of a sharded neighbour sweep (all-gather merge) ---------------------------------------------------
// A segment's rows are scattered over the trajectory (every G-th query group of the sweep's order); compacted by
// LOCAL POSITION they form a dense block that an all-gather can move -- half the bytes of the all-reduce(min) of packed
// words, and no reduction.  Local position l of segment g: group l / (32 tq) of the segment = group (l / (32 tq)) G + g
// of the order, i.e. position p(l); frame perm[p].  A flagged data set (the direct kernels answered for the row block
// of the segment) uses the row block: the choice is made on the device from the same flag the sweeps looked at.
__device__ __forceinline__ uint32_t block_none(uint32_t c, uint32_t n_rows) {
  return (c & 1u) ? __float_as_uint(FLT_MAX) : n_rows + 1u;
}
// Layout header: the last kBlockHdrRows entries of plane 0 of a block say under which layout it was packed (by position
// or row block, positions of the padded order, group size, segment count, groups per deal, and a 64-bit hash of the
// order itself): every rank must have derived the SAME order, or its rows would be scattered to the wrong frames
// without any sign.  The hosts compare the headers of the gathered blocks (clustering_amd/distributed.py, dc_session.hip).
constexpr uint32_t kBlockHdrRows = 32, kBlockMagic = 0x6E6E4200u;   // "nnB" | by_position
__global__ void nn_block_pack_kernel(const uint32_t* __restrict__ nn_idx, const float* __restrict__ nn_d2,
                                     const uint32_t* __restrict__ hd_idx, const float* __restrict__ hd_d2,
                                     uint32_t n_rows, uint32_t n_pos /* positions of the padded order */,
                                     uint32_t gsize /* 32 tq, 0: row blocks only */, uint32_t seg,
                                     uint32_t G, uint32_t seg_blk, uint32_t block_rows, const uint32_t* __restrict__ perm,
                                     const uint32_t* __restrict__ hdr, uint32_t* __restrict__ block,
                                     const unsigned long long* __restrict__ order_hash) {
  const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= block_rows) return;
  const bool by_position = gsize != 0u && hdr[1] == 0u;
  const uint32_t payload = block_rows - kBlockHdrRows;
  if (l >= payload) {
    const uint32_t k = l - payload;
    uint32_t w = 0;
    if (k == 0) w = kBlockMagic | (by_position ? 1u : 0u);
    if (k == 1) w = by_position ? n_pos : n_rows;
    if (k == 2) w = by_position ? gsize : 0u;
    if (k == 3) w = G;
    if (k == 4) w = by_position ? seg_blk : 0u;
    if ((k == 5 || k == 6) && by_position) {   // hash of the order: the 64 shares order_rows2_kernel left (wrap-around sum)
      unsigned long long hsum = 0;
      for (int q = 0; q < 64; ++q) hsum += order_hash[q];
      w = (k == 5) ? (uint32_t)hsum : (uint32_t)(hsum >> 32);
    }
    block[l] = w;
    for (uint32_t c = 1; c < 4; ++c) block[c * (size_t)block_rows + l] = block_none(c, n_rows);
    return;
  }
  uint32_t i = 0xFFFFFFFFu;
  if (by_position) {
    const unsigned long long p = (unsigned long long)seg_group(l / gsize, QSeg{G, seg, seg_blk}) * gsize + l % gsize;
    if (p < n_pos) i = perm[p];   // (kInvalidFrame for the pad positions of the order)
  } else {
    const uint32_t rng = n_rows / G, lo = seg * rng, hi = (seg == G - 1u) ? n_rows : lo + rng;
    if (lo + l < hi) i = lo + l;
  }
  const bool live = i != 0xFFFFFFFFu;
  block[0 * (size_t)block_rows + l] = live ? nn_idx[i] : block_none(0, n_rows);
  block[1 * (size_t)block_rows + l] = live ? __float_as_uint(nn_d2[i]) : block_none(1, n_rows);
  block[2 * (size_t)block_rows + l] = live ? hd_idx[i] : block_none(2, n_rows);
  block[3 * (size_t)block_rows + l] = live ? __float_as_uint(hd_d2[i]) : block_none(3, n_rows);
}
__global__ void nn_block_unpack_kernel(const uint32_t* __restrict__ blocks /* [G][4][block_rows] */, uint32_t n_rows,
                                       uint32_t n_pos, uint32_t gsize, uint32_t G, uint32_t seg_blk, uint32_t block_rows,
                                       const uint32_t* __restrict__ perm, const uint32_t* __restrict__ hdr,
                                       uint32_t* __restrict__ nn_idx, float* __restrict__ nn_d2,
                                       uint32_t* __restrict__ hd_idx, float* __restrict__ hd_d2,
                                       uint32_t* __restrict__ layout_bad) {
  // Every rank must have packed under the SAME layout (words 0..6 of the blocks' headers): compared here, by every
  // workgroup before it scatters anything -- on a mismatch nothing is written and the workspace's flag is raised
  // (dc_hip_workspace_layout_status_dev): a host need not synchronise in front of the unpack to look at the headers.
  {
    __shared__ uint32_t bad_s;
    if (threadIdx.x == 0) bad_s = 0u;
    __syncthreads();
    const uint32_t payload = block_rows - kBlockHdrRows;
    for (uint32_t e = threadIdx.x; e < 7u * G; e += blockDim.x) {
      const uint32_t r = e / 7u, k = e - 7u * r;
      if (blocks[(size_t)r * 4u * block_rows + payload + k] != blocks[payload + k]) bad_s = 1u;
    }
    __syncthreads();
    if (layout_bad) {   // (no workspace to flag in: the host's own comparison is the check)
      if (threadIdx.x == 0 && blockIdx.x == 0) *layout_bad = bad_s;   // THIS unpack's verdict (the next unpack or sweep in the workspace replaces it)
      if (bad_s != 0u) return;
    }
  }
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool by_position = gsize != 0u && hdr[1] == 0u;
  if (p >= (by_position ? n_pos : n_rows)) return;
  uint32_t i, r, l;
  if (by_position) {
    const uint32_t grp = p / gsize;
    r = (grp / seg_blk) % G;
    l = seg_unit(grp, QSeg{G, r, seg_blk}) * gsize + p % gsize;
    i = perm[p];
    if (i == kInvalidFrame) return;   // (a pad position of the order)
  } else {
    const uint32_t rng = n_rows / G;
    i = p;
    r = rng ? min(p / rng, G - 1u) : G - 1u;
    l = p - r * rng;
  }
  const uint32_t* b = blocks + (size_t)r * 4u * block_rows;
  nn_idx[i] = b[l];
  nn_d2[i] = __uint_as_float(b[(size_t)block_rows + l]);
  hd_idx[i] = b[2 * (size_t)block_rows + l];
  hd_d2[i] = __uint_as_float(b[3 * (size_t)block_rows + l]);
}

static uint32_t nn_group_rows(uint32_t n_rows, uint32_t n_cols) {
  return 32u * (nn_shared_wanted(n_rows, n_cols) ? 4u * (uint32_t)tq_of(n_cols) : (uint32_t)tq_nn_of(n_cols));
}
// tiles of the neighbour sweep's padded order (every component starts at a whole query group)
static uint32_t nn_order_tiles(uint32_t n_rows, uint32_t n_cols) {
  return (n_rows + (uint32_t)kMaxComp * (nn_group_rows(n_rows, n_cols) - 1u) + 31u) / 32u;
}
size_t nn_block_rows(size_t n_rows, size_t n_cols, size_t n_segments) {
  if (n_segments == 0 || n_rows == 0) return 0;
  const size_t row_block = n_rows - (n_segments - 1) * (n_rows / n_segments);   // the last (largest) row block
  if (!mfma_supports(n_cols)) return row_block + kBlockHdrRows;
  const size_t gs = nn_group_rows((uint32_t)n_rows, (uint32_t)n_cols);
  const size_t groups = ((size_t)32 * nn_order_tiles((uint32_t)n_rows, (uint32_t)n_cols) + gs - 1) / gs;
  // (segment 0 owns the most groups of a block-cyclic deal)
  const size_t most = seg_groups((uint32_t)groups, QSeg{(uint32_t)n_segments, 0u, seg_block((uint32_t)n_segments)});
  return std::max(row_block, most * gs) + kBlockHdrRows;   // (+ the layout header, nn_block_pack_kernel)
}
void launch_nn_block_pack(const uint32_t* d_nn_idx, const float* d_nn_d2, const uint32_t* d_hd_idx,
                          const float* d_hd_d2, uint32_t n_rows, uint32_t n_cols, uint32_t segment,
                          uint32_t n_segments, bool pruned, const void* d_ws, uint32_t* d_block, hipStream_t stream) {
  const uint32_t rows = (uint32_t)nn_block_rows(n_rows, n_cols, n_segments);
  const Layout L = make_layout(n_rows, n_cols);
  const char* p = (const char*)d_ws;
  // (the hash of the order the block is packed by -- 64 shares in the component region, kCompHash -- was formed with the
  //  order itself: order_rows2_kernel of the neighbour call that ran in this workspace)
  hipLaunchKernelGGL(nn_block_pack_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, d_nn_idx, d_nn_d2, d_hd_idx,
                     d_hd_d2, n_rows, 32u * nn_order_tiles(n_rows, n_cols), pruned ? nn_group_rows(n_rows, n_cols) : 0u, segment, n_segments, seg_block(n_segments), rows,
                     pruned ? (const uint32_t*)(p + L.off_perm_p) : nullptr, pruned ? (const uint32_t*)p : nullptr, d_block,
                     (const unsigned long long*)(p + L.off_comp + sizeof(uint32_t) * kCompHash));
}
void launch_nn_block_unpack(const uint32_t* d_blocks, uint32_t n_rows, uint32_t n_cols, uint32_t n_segments,
                            bool pruned, const void* d_ws, uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx,
                            float* d_hd_d2, hipStream_t stream) {
  const uint32_t rows = (uint32_t)nn_block_rows(n_rows, n_cols, n_segments);
  const Layout L = make_layout(n_rows, n_cols);
  const char* p = (const char*)d_ws;
  const uint32_t n_pos = 32u * nn_order_tiles(n_rows, n_cols);
  hipLaunchKernelGGL(nn_block_unpack_kernel, dim3((n_pos + 255) / 256), dim3(256), 0, stream, d_blocks, n_rows, n_pos,
                     pruned ? nn_group_rows(n_rows, n_cols) : 0u, n_segments, seg_block(n_segments), rows,
                     pruned ? (const uint32_t*)(p + L.off_perm_p) : nullptr, pruned ? (const uint32_t*)p : nullptr,
                     d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2,
                     // (the verdict word lives in the header of a PRUNED sweep's workspace, whose size the caller's entry point
                     //  has checked; any other workspace is never written)
                     (pruned && p) ? (uint32_t*)(const_cast<char*>(p) + 4 * kHdrLayoutBad) : (uint32_t*)nullptr);
}

void launch_nn_mfma(const float* d_coords, uint32_t n_rows, uint32_t n_cols, const float* d_fe,
                    uint32_t i_from, uint32_t i_to, uint32_t* d_nn_idx, float* d_nn_d2,
                    uint32_t* d_hd_idx, float* d_hd_d2, void* d_ws, hipStream_t stream) {
  // order the reference frames by free energy and build their operand image
  const Layout L = make_layout(n_rows, n_cols);
  char* p = (char*)d_ws;
  uint32_t* keys_in = (uint32_t*)(p + L.off_keys_in);
  uint32_t* keys_out = (uint32_t*)(p + L.off_keys_out);
  uint32_t* vals_in = (uint32_t*)(p + L.off_vals_in);
  uint32_t* perm = (uint32_t*)(p + L.off_perm);
  const dim3 blk(256), grid_n((n_rows + 255) / 256), grid_t((32 * L.T + 255) / 256);
  auto grid_img = [&](uint32_t tiles) { return dim3((uint32_t)(((size_t)tiles * L.NM * 64 + 255) / 256)); };
  hipLaunchKernelGGL(scale_kernel, dim3(1), dim3(1), 0, stream, (uint32_t*)p, -1.0f, n_cols);   // the neighbour scale
  natural_images(d_coords, n_rows, n_cols, d_ws, false, true, stream);   // queries: B form + norms
  hipLaunchKernelGGL(fe_key_kernel, dim3(std::min<uint32_t>(grid_n.x, 1024u)), blk, 0, stream, d_fe, n_rows, keys_in, vals_in,
                     (uint32_t*)p);
  if (sort_pairs_u32(keys_in, keys_out, vals_in, perm, n_rows, p + L.fixed_end,
                     sort_temp_bytes(n_rows), stream) != 0)
    return;
  hipLaunchKernelGGL(fe_scatter_kernel, grid_t, blk, 0, stream, perm, d_fe, n_rows, L.T,
                     (uint32_t*)(p + L.off_invpos), (float*)(p + L.off_fe_s));
  hipLaunchKernelGGL(fe_rank_kernel, grid_n, blk, 0, stream, d_fe, (const float*)(p + L.off_fe_s),
                     n_rows, (uint32_t*)(p + L.off_pq));
  hipLaunchKernelGGL(image_kernel, grid_img(L.T), blk, image_smem(n_cols), stream, d_coords, n_rows, n_rows, n_cols,
                     L.NM, L.T, (const float*)(p + kHdrMeans), (const uint32_t*)perm, 0,
                     (uint4*)(p + L.off_img_s), (float*)(p + L.off_norm_s), (const uint32_t*)p);
  switch (nm_for((int)n_cols)) {
#define X(SV)                                                                                \
  case SV:                                                                                   \
    if ((DC_STEP_MASK >> (SV - 1)) & 1u)                                                     \
      nn_mfma_step_##SV(d_coords, n_rows, n_cols, d_ws, i_from, i_to, d_nn_idx, d_nn_d2, d_hd_idx, \
                        d_hd_d2, stream);                                                    \
    break;
    DC_FOR_EACH_S(X)
#undef X
    default:
      break;
  }
}

// ---- DC_VARIANT_MFMA32: the fp32-input MFMA instance (dc_mfma32.hpp) ---------------------------------------------
bool mfma32_supports(size_t n_cols) { return n_cols == 9 || n_cols == 10; }
static uint32_t wpb32() {
  static const uint32_t v = [] { const char* e = getenv("DC_MFMA32_WPB"); const int k = (e && e[0]) ? atoi(e) : 1; return (k == 1 || k == 2 || k == 4) ? (uint32_t)k : 1u; }();
  return v;
}

void launch_pop_mfma32(const float* d_coords, uint32_t n_rows, uint32_t n_cols, uint32_t i_from, uint32_t i_to,
                       const Rad2& rad2, int n_rad, uint32_t* d_pops, void* d_ws, hipStream_t stream) {
  const Layout L = make_layout(n_rows, n_cols);
  char* p = (char*)d_ws;
  float* img = (float*)(p + L.off_img);
  float* norms = (float*)(p + L.off_norm);
  // the frames in the order of their 2-D cells (cell32_key_kernel; about 64 frames per cell of the bounding box of columns
  // 0 / 1, at most 256 x 256 cells: 16-bit keys, two passes of the library's sort)
  uint32_t* keys_in = (uint32_t*)(p + L.off_keys_in);
  uint32_t* keys_out = (uint32_t*)(p + L.off_keys_out);
  uint32_t* vals_in = (uint32_t*)(p + L.off_vals_in);
  uint32_t* perm = (uint32_t*)(p + L.off_perm);
  uint32_t G = 1, g_bits = 0;
  while (G < 256u && (size_t)(2u * G) * (2u * G) * 64u <= (size_t)n_rows) { G *= 2u; ++g_bits; }
  hipLaunchKernelGGL(cell32_key_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, stream, d_coords, n_rows, n_cols,
                     (const uint32_t*)p, G, keys_in, vals_in);
  if (sort_pairs_u32(keys_in, keys_out, vals_in, perm, n_rows, p + L.fixed_end, sort_temp_bytes(n_rows), stream, 2u * g_bits) != 0) return;
  // (ONE image for all radii of the call, scaled for the largest: the band of a smaller radius is narrower than 1)
  float r2max = 0.0f;
  for (int r = 0; r < n_rad; ++r) r2max = std::max(r2max, rad2.v[r]);
  hipLaunchKernelGGL(image32_kernel, dim3((32 * L.T + 255) / 256), dim3(256), 0, stream, d_coords, n_rows, n_cols, L.T,
                     (const float*)(p + kHdrMeans), (const uint32_t*)perm, (const uint32_t*)p, r2max, img, norms);
#ifdef DC_MFMA32_TQ
  constexpr int kTQ = DC_MFMA32_TQ;
#else
  constexpr int kTQ = 8;
#endif
  // (two waves per SIMD at eight query tiles per wave: 2 048 wave slots; DC_MFMA32_CHUNKS for measurements)
  static const uint32_t env_chunks = [] { const char* v = getenv("DC_MFMA32_CHUNKS"); return (v && v[0]) ? (uint32_t)atoi(v) : 0u; }();
  // (waves per workgroup: DC_MFMA32_WPB, measurements)
  const uint32_t wpb = wpb32();
  const uint32_t blocks = ((L.T + kTQ - 1) / kTQ + wpb - 1) / wpb;   // (all positions: the rows of a range are scattered over the order)
  const uint32_t chunks = env_chunks ? std::min(env_chunks, L.T) : chunks32(blocks * wpb, L.T, 2048u);
  const dim3 grid(blocks, chunks), block(64 * wpb);
  for (int r = 0; r < n_rad; ++r) {
    if (chunks > 1)
      (void)hipMemsetAsync(d_pops + (size_t)r * n_rows + i_from, 0, sizeof(uint32_t) * (size_t)(i_to - i_from), stream);
    sweep_timer_mark(0, true, stream);
    hipLaunchKernelGGL((pop_mfma32_kernel<kS32, kTQ>), grid, block, wpb * sizeof(float) * 2 * kNormBatch32 * 32, stream, d_coords, n_rows, n_cols, (const float*)img,
                       (const float*)norms, (const uint32_t*)perm, (const uint32_t*)p, L.T, i_from, i_to, rad2.v[r], r2max, d_pops + (size_t)r * n_rows);
    sweep_timer_mark(0, false, stream);
  }
}

void launch_nn_mfma32(const float* d_coords, uint32_t n_rows, uint32_t n_cols, const float* d_fe, uint32_t i_from,
                      uint32_t i_to, uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx, float* d_hd_d2, void* d_ws,
                      hipStream_t stream) {
  // the reference frames ordered by free energy, exactly as launch_nn_mfma orders them; fp32 images instead of fp16 x 2
  const Layout L = make_layout(n_rows, n_cols);
  char* p = (char*)d_ws;
  uint32_t* keys_in = (uint32_t*)(p + L.off_keys_in);
  uint32_t* keys_out = (uint32_t*)(p + L.off_keys_out);
  uint32_t* vals_in = (uint32_t*)(p + L.off_vals_in);
  uint32_t* perm = (uint32_t*)(p + L.off_perm);
  const dim3 blk(256), grid_n((n_rows + 255) / 256), grid_t((32 * L.T + 255) / 256);
  float* img = (float*)(p + L.off_img);
  float* norms = (float*)(p + L.off_norm);
  float* img_s = (float*)(p + L.off_img_s);
  float* norms_s = (float*)(p + L.off_norm_s);
  hipLaunchKernelGGL(image32_kernel, grid_t, blk, 0, stream, d_coords, n_rows, n_cols, L.T, (const float*)(p + kHdrMeans),
                     (const uint32_t*)nullptr, (const uint32_t*)p, -1.0f, img, norms);
  hipLaunchKernelGGL(fe_key_kernel, dim3(std::min<uint32_t>(grid_n.x, 1024u)), blk, 0, stream, d_fe, n_rows, keys_in, vals_in,
                     (uint32_t*)p);
  if (sort_pairs_u32(keys_in, keys_out, vals_in, perm, n_rows, p + L.fixed_end, sort_temp_bytes(n_rows), stream) != 0) return;
  hipLaunchKernelGGL(fe_scatter_kernel, grid_t, blk, 0, stream, perm, d_fe, n_rows, L.T, (uint32_t*)(p + L.off_invpos),
                     (float*)(p + L.off_fe_s));
  hipLaunchKernelGGL(fe_rank_kernel, grid_n, blk, 0, stream, d_fe, (const float*)(p + L.off_fe_s), n_rows,
                     (uint32_t*)(p + L.off_pq));
  hipLaunchKernelGGL(image32_kernel, grid_t, blk, 0, stream, d_coords, n_rows, n_cols, L.T, (const float*)(p + kHdrMeans),
                     (const uint32_t*)perm, (const uint32_t*)p, -1.0f, img_s, norms_s);
  constexpr int kTQ = 4;
  static const uint32_t env_chunks = [] { const char* v = getenv("DC_MFMA32_CHUNKS"); return (v && v[0]) ? (uint32_t)atoi(v) : 0u; }();
  // ONE wave per workgroup: a workgroup holds its wave slots until its last wave is done, and the waves of this sweep
  // differ in length (those that start from published bounds skip most of the candidate path)
  const uint32_t wpb = wpb32();
  const uint32_t blocks = (grid_for(i_from, i_to, kTQ) * 4u + wpb - 1) / wpb;
  const uint32_t chunks = env_chunks ? std::min(env_chunks, L.T) : chunks32(blocks * wpb, L.T, 2048u);
  unsigned long long* merge64 = (unsigned long long*)(p + L.off_merge64);
  if (chunks > 1)
    hipLaunchKernelGGL(nn_merge_fill_kernel, dim3((2 * n_rows + 255) / 256), dim3(256), 0, stream, merge64, n_rows);
  sweep_timer_mark(1, true, stream);
  const size_t smem = wpb * sizeof(uint32_t) * (2 * kNormBatch32 * 32 + 2 * kWaveQueue + 4 * kTQ * 32 + kTQ * 32 * (size_t)n_cols);
  hipLaunchKernelGGL((nn_mfma32_kernel<kS32, kTQ>), dim3(blocks, chunks), dim3(64 * wpb), smem, stream, d_coords, n_rows, n_cols,
                     (const float*)img, (const float*)norms, (const float*)img_s, (const float*)norms_s, (const uint32_t*)perm,
                     (const uint32_t*)(p + L.off_invpos), (const uint32_t*)(p + L.off_pq), (const uint32_t*)p, L.T, i_from, i_to,
                     merge64, d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2);
  sweep_timer_mark(1, false, stream);
  if (chunks > 1)
    hipLaunchKernelGGL(nn32_unpack_kernel, dim3((i_to - i_from + 255) / 256), dim3(256), 0, stream,
                       (const unsigned long long*)merge64, n_rows, i_from, i_to, d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2);
}

}  // namespace dc
