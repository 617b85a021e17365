// dc_mfma32.hpp -- the literal fp32-MFMA variant of the two sweeps (v_mfma_f32_32x32x2_f32), every pair evaluated
// (included by dc_mfma.hip after dc_mfma_kernels.hpp, inside namespace dc::{anonymous}; DC_VARIANT_MFMA32).
//
// BASELINE.json's north star names "an MFMA-tiled fp32 variant whose utilisation is evidenced by rocprof" with a target
// of 60 % of the fp32 MFMA roofline (157.3 TFLOP/s: on gfx950 the fp32-input MFMA runs at the VECTOR rate and, measured,
// does not overlap VALU work -- DESIGN.md 4.2).  The default path therefore moved to fp16 pieces on the 16-bit pipe
// (16x the rate); this file keeps the fp32 instance that round 1 measured at 58 - 61 % of that roof (tag fp32-mfma-r1)
// buildable, covered by the parity tests and benchable (bench.py --variant mfma32), for n_cols 9 .. 10 (five K-steps of
// two columns).  Same classifier idea as the fp16 kernels, with the fp32 band of round 1:
//     acc = |y'|^2 - 2 x'.y'  (A = centred reference coordinates, B = -2 centred query coordinates, C = |y'|^2)
//     eps = 1.25 u [(4 K + 10) M + (D / 4 + 12) d2cap],  K = 2 S multiply-adds per chain, M = max |x'|^2
// (the fp32 MFMA is a chain of fmaf: products exact to one rounding each, 2 S + 2 roundings around values <= 4 M).
// Operand image: img32[(t * 64 + l) * 8 + s] = y'[32 t + (l & 31)][2 s + (l >> 5)], norms32[32 t + c] = |y'|^2.
//
// Round 6 -- what the fp32-input MFMA wants from the code around it (scratch/ub/mfma32_sched.hip, one MI355X, ns per
// chain and SIMD at two waves per SIMD; five dependent MFMAs alone: 137.4):
//   * it does not overlap with VALU work at all, so dealing the epilogue to the MFMA slots (the fp16 kernels' software
//     pipelining) buys nothing and every MFMA <-> VALU alternation costs a few cycles: 48 VALU dealt 208.7, the same 48 in
//     one clump behind the chains of FOUR query tiles 202.2; the neighbour epilogue (12) 174.1 -> 159.5;
//   * the epilogue itself: populations keep TWO bits per element like the fp16 kernels (round 2) -- the image is scaled
//     so that the guard band of the call's largest radius is just below 1 and the threshold is folded as
//     t = acc - ((S r^2 - 1) - |x''|^2): sign = inside, bit 30 (t >= 2) = outside, neither = band -> canonical re-check:
//     one v_pk_add_f32 per two elements and one v_alignbit(.., 30) per element, no minimum (24 VALU: 186.4);
//   * ONE wave-level test per reference tile (the four chains' band strings OR-ed; the neighbour sweep's four tile minima
//     against cached thresholds) instead of one or two scalar hand-offs per chain.
// Scaled band: with x'' = fl(c x') every term of the unscaled bound scales by S = c^2; the scaling adds 2 u (M + d2)
// (rounding of c x'), 2 u thr (fl(c c), fl(S r^2)) and u (M + 2 thr) / 2 (the two subtractions that form the per-query
// constant): 1.25 u [(4 K + 13) M + (D / 4 + 18) r^2] in data units, and c is the largest scale at which that is < 1.
constexpr int kS32 = 5;   // K-steps: n_cols 9 .. 10
constexpr int kLane32 = 8;    // floats per lane and tile in the operand image (S of them used: one dwordx4 + one dword)
constexpr int kNormBatch32 = 8;   // tiles whose row norms one LDS-DMA of a wave fetches (64 lanes x 16 B = 8 x 32 floats)
constexpr int kClump32 = 4;   // chains (query tiles) whose MFMAs are issued back to back in front of their epilogues

__host__ __device__ inline float guard_eps32(float M, float d2cap, int K, int D) {
  const double u = 5.9604644775390625e-8;
  const double cap = (d2cap > 0.0f) ? (double)d2cap : 0.0;
  return next_up((float)(1.25 * u * ((4.0 * K + 10.0) * (double)M + (0.25 * D + 12.0) * cap)));
}

// scale c of the population image: the largest one at which the band of the scaled chain (see the header) stays below 1
// for squared radii up to r2max.  Every thread of the image pass and of the sweep evaluates this from the same words.
__host__ __device__ inline float scale32_pop(float M, float r2max, int K, int D) {
  const double u = 5.9604644775390625e-8;
  const double cap = (r2max > 0.0f) ? (double)r2max : 0.0;
  const double eps1 = 1.25 * u * ((4.0 * K + 13.0) * (double)M + (0.25 * D + 18.0) * cap);   // band at scale 1
  if (!(eps1 > 1.0e-76)) return 1.8446744e19f;                                               // (all rows equal, r = 0: 2^64)
  const float c = (float)(sqrt((1.0 - 1.0 / 65536.0) / eps1) * (1.0 - 1.0 / 1048576.0));
  return c < 1.8446744e19f ? c : 1.8446744e19f;   // (M <= 1e36 and r2 <= FLT_MAX keep c above 2^-56: no lower clamp needed)
}

// Spatial order of the fp32 population sweep (round 6): the frames by the cell of a G x G grid on columns 0 / 1 (their
// extents: header words 8..11).  Every pair is still evaluated on the matrix pipe -- what the order buys is that most
// 32 x 32 chains then hold NO pair within the radius, which one tile minimum per chain shows (pop_mfma32_kernel).
__global__ void cell32_key_kernel(const float* __restrict__ coords, uint32_t n_rows, uint32_t D, const uint32_t* __restrict__ hdr,
                                  uint32_t G, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const float lo0 = fkey_inv(~hdr[8]), hi0 = fkey_inv(hdr[9]), lo1 = fkey_inv(~hdr[10]), hi1 = fkey_inv(hdr[11]);
  auto cell = [&](float v, float lo, float hi) -> uint32_t {
    const float u = (hi > lo) ? (v - lo) / (hi - lo) * (float)G : 0.0f;
    return (u >= 0.0f) ? min((uint32_t)fminf(u, 1.0e6f), G - 1u) : 0u;   // (NaN / -inf: cell 0 -- flagged data never reaches the sweep)
  };
  keys[i] = cell(coords[(size_t)i * D], lo0, hi0) * G + ((D > 1) ? cell(coords[(size_t)i * D + 1], lo1, hi1) : 0u);
  vals[i] = i;
}

// fp32 operand image of rows in natural order (perm == nullptr) or gathered through perm; r2max >= 0: scaled by
// scale32_pop (population sweeps), r2max < 0: unscaled (neighbour sweep, whose band is relative)
__global__ void image32_kernel(const float* __restrict__ coords, uint32_t n_rows, uint32_t D, uint32_t T,
                               const float* __restrict__ means, const uint32_t* __restrict__ perm,
                               const uint32_t* __restrict__ hdr, float r2max, float* __restrict__ img,
                               float* __restrict__ norms) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= 32 * T) return;
  const uint32_t t = row >> 5, c = row & 31;
  const bool live = row < n_rows;
  const uint32_t src = live ? (perm ? perm[row] : row) : 0u;
  const bool scaled = r2max >= 0.0f;
  const float cs = scaled ? scale32_pop(__uint_as_float(hdr[0]), r2max, 2 * kS32, (int)D) : 1.0f;
  double nrm = 0.0;
  for (uint32_t k = 0; k < 2 * (uint32_t)kS32; ++k) {
    float v = 0.0f;
    if (live && k < D) {
      v = coords[(size_t)src * D + k] - means[k];
      if (scaled) v = cs * v;
    }
    nrm += (double)v * (double)v;
    img[((size_t)t * 64 + c + 32 * (k & 1)) * kLane32 + (k >> 1)] = v;   // (lane-contiguous: see load_frag32)
  }
  norms[row] = live ? (float)nrm : INFINITY;   // pad rows: never inside, never a candidate
}

// What a reference tile costs a wave besides its chains is the ISSUE of its loads: a vector-memory instruction among
// fp32 MFMAs costs about 37 cycles wherever it stands (nine of them per tile -- five fragments, four quarters of the row
// norms -- were 330 of a tile's 2 x 1 280 MFMA cycles at four query tiles per wave; dealt one per MFMA instead of in
// front of the tile: 190 -> 215 ms).  So the image keeps a lane's S fragments side by side (one dwordx4 + one dword),
// and the row norms -- the same 16 values for all lanes of a half-wave -- come through a wave-private LDS ring that ONE
// LDS-DMA per kNormBatch32 tiles fills (ds_read_b128 costs next to nothing among MFMAs): 2 1/8 instead of 9 per tile.
template <int S>
__device__ __forceinline__ void load_frag32(const float* __restrict__ img, uint32_t t, int lane, float (&a)[S]) {
  static_assert(S == 5, "one dwordx4 and one dword per lane");
  const float* ip = img + ((size_t)t * 64 + lane) * kLane32;
  const float4 v = *reinterpret_cast<const float4*>(ip);
  a[0] = v.x;
  a[1] = v.y;
  a[2] = v.z;
  a[3] = v.w;
  a[4] = ip[4];
}
// the wave's ring of row norms: two batches of kNormBatch32 tiles; batch i of the chunk (tiles tb + 8 i ...) in slot i & 1
struct Norms32 {
  float* ring;          // [2][kNormBatch32 * 32] floats of LDS, private to the wave
  const float* norms;   // + 32 * first tile of the chunk
  __device__ __forceinline__ void fetch(uint32_t batch, int lane) const {   // (runs past the chunk's last tile by up to 7 tiles: inside the workspace, never read)
    lds_dma16(norms + (size_t)batch * (kNormBatch32 * 32) + 4 * lane,
              (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(ring + (batch & 1u) * (kNormBatch32 * 32))));   // (wave-uniform: M0)
  }
  __device__ __forceinline__ void read(uint32_t k, int h, float4 (&nv)[4]) const {   // tile k of the chunk
    const float4* np = reinterpret_cast<const float4*>(ring + (k & (2 * kNormBatch32 - 1)) * 32 + 4 * h);
#pragma unroll
    for (int g = 0; g < 4; ++g) nv[g] = np[2 * g];        // rows 8g + 4h .. +3  <->  registers 4g .. 4g+3
  }
};

// the chains of NC query tiles against one reference tile, their MFMAs interleaved (five dependent MFMAs per chain)
template <int S, int NC>
__device__ __forceinline__ void chains32(const float (&a)[S], const float (*b)[S], const f32x16& c0, f32x16 (&acc)[NC]) {
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int q = 0; q < NC; ++q)
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[q][s], s == 0 ? c0 : acc[q], 0, 0, 0);
}

// two bits per element of t = acc - lo (element r at bits 31 - 2 r, 30 - 2 r: inside_of / band_of / element_of)
// (nlo: the NEGATED per-query constant in both halves of a register pair; the packed add is spelt out because hipcc
//  otherwise falls back to sixteen v_sub_f32)
__device__ __forceinline__ uint32_t pop32_string(const f32x16& acc, f32x2 nlo) {
  f32x2 t[8];
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const f32x2 v = {acc[r], acc[r + 1]};
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(t[r / 2]) : "v"(v), "v"(nlo));
  }
  uint32_t bits = 0;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t[r / 2].x), 30);
    bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t[r / 2].y), 30);
  }
  return bits;
}

// rare: exact re-check of the band pairs of one accumulator tile (by value: see pop_fix)
__device__ __attribute__((noinline)) uint32_t pop32_fix(const float* __restrict__ coords, const uint32_t* __restrict__ perm,
                                                        uint32_t n_rows, uint32_t n_cols, float r2, uint32_t band, uint32_t jq,
                                                        uint32_t t, int h) {
  uint32_t m = band, out = 0;   // (band_of(string): element r at bit 31 - 2 r; jq: the query's FRAME, the tile's rows are positions)
  while (__builtin_amdgcn_ballot_w64(m != 0) != 0) {
    if (m != 0) {
      const int r = element_of(31 - __builtin_clz(m));
      const uint32_t row = tile_row(t, r, h);
      if (row < n_rows) out += (exact_d2(coords, n_cols, jq, perm[row]) < r2) ? 1u : 0u;
      m &= ~(0x80000000u >> (2 * r));
    }
  }
  return out;
}

template <int S, int TQ>
__global__ __launch_bounds__(256, 2) void pop_mfma32_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols, const float* __restrict__ img,
    const float* __restrict__ norms, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ hdr, uint32_t T,
    uint32_t i_from, uint32_t i_to, float r2, float r2max, uint32_t* __restrict__ pops) {
  static_assert(TQ % kClump32 == 0, "query tiles are handled in clumps");
  extern __shared__ __attribute__((aligned(16))) float pop32_lds[];   // per wave: the ring of row norms (Norms32)
  if (hdr[1] != 0) return;   // non-finite / overflow-prone data: the gated direct kernel runs instead
  const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
  const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  // (queries and references are POSITIONS of the spatial order -- perm: position -> frame; the rows [i_from, i_to) of a
  //  row range are wherever the order put them)
  const uint32_t qt0 = wave * TQ;
  if (qt0 >= T) return;   // whole wave leaves; no barriers in this kernel
  // scaled units (the image was built with the same scale): inside <=> t < 0, outside <=> t >= 2, else band
  const float cs = scale32_pop(__uint_as_float(hdr[0]), r2max, 2 * S, (int)n_cols);
  const float thr = (cs * cs) * r2 - 1.0f;

  float b[TQ][S];
  f32x2 nlo[TQ];
  uint32_t cnt[TQ], jq[TQ];
  uint64_t livemask[TQ];
  uint64_t any_live = 0;
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const uint32_t tile = qt0 + qt, pos = tile * 32 + c;
    jq[qt] = (tile < T && pos < n_rows) ? perm[pos] : 0xFFFFFFFFu;
    const bool live = (jq[qt] >= i_from) && (jq[qt] < i_to);   // (0xFFFFFFFF: beyond every range)
    livemask[qt] = __builtin_amdgcn_ballot_w64(live);
    any_live |= livemask[qt];
    const uint32_t tl = tile < T ? tile : T - 1;
#pragma unroll
    for (int s = 0; s < S; ++s) b[qt][s] = -2.0f * img[((size_t)tl * 64 + lane) * kLane32 + s];
    const float lo = thr - (live ? norms[tl * 32 + c] : INFINITY);   // -inf for an idle lane: t = +inf, outside, no band
    nlo[qt] = f32x2{-lo, -lo};
    cnt[qt] = 0;
  }
  if (any_live == 0) return;   // (a row range: none of this wave's positions belongs to it)
  // reference tiles [tb, te) of this wave's chunk (gridDim.y chunks: the launcher sizes the grid so that its waves fill
  // the chip's wave slots a whole number of times; partial counts merge by atomicAdd into zero-filled rows)
  const uint32_t per = (T + gridDim.y - 1) / gridDim.y, tb = blockIdx.y * per, te = min(T, tb + per);
  if (tb >= te) return;
  const uint32_t nt = te - tb;
  const Norms32 N{pop32_lds + (size_t)(threadIdx.x >> 6) * (2 * kNormBatch32 * 32), norms + (size_t)tb * 32};
  N.fetch(0, lane);
  N.fetch(1, lane);
  float a0[S], a1[S];
  float4 n0[4], n1[4];
  load_frag32<S>(img, tb, lane, a0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the first two batches of norms have landed)
  N.read(0, h, n0);
  // `prefetch`: the loads of the NEXT tile, issued right behind the first clump's MFMAs -- the results of a clump cannot
  // be read for some twenty issue slots anyway (the compiler pads them with s_nop), and there the loads' own issue cost
  // disappears as well
  auto tile_body = [&](const float (&a)[S], const float4 (&nv)[4], uint32_t t, auto&& prefetch) {
    const f32x16 c0 = frag16(nv);   // (+inf for a pad row: t = +inf)
#pragma unroll
    for (int g = 0; g < TQ; g += kClump32) {
      f32x16 acc[kClump32];
      chains32<S, kClump32>(a, &b[g], c0, acc);
      __builtin_amdgcn_sched_barrier(0);
      prefetch(g / kClump32);   // (part 0: the fragments, part 1: the row norms)
      __builtin_amdgcn_sched_barrier(0);
      // Most chains of a spatially ordered sweep hold no pair inside or in the band: the smallest t of the lane's 16
      // elements says so (8 v_min3 + 2 per chain against the 27 of the strings), ONE wave-level test for the clump.
      float tm[kClump32], dmin = INFINITY;
#pragma unroll
      for (int q = 0; q < kClump32; ++q) {
        tm[q] = INFINITY;
        tile_min<0, 16>(acc[q], tm[q]);
        tm[q] += nlo[g + q].x;                 // smallest t = acc - lo of the lane (+inf: idle lane, pad rows)
        dmin = fminf(dmin, tm[q]);
      }
      if (__builtin_amdgcn_ballot_w64(dmin < 2.0f) != 0) {
#pragma unroll
        for (int q = 0; q < kClump32; ++q) {
          if (__builtin_amdgcn_ballot_w64(tm[q] < 2.0f) == 0) continue;   // (every pair of this chain is outside)
          const uint32_t bits = pop32_string(acc[q], nlo[g + q]);
          cnt[g + q] += __builtin_popcount(inside_of(bits));
          const uint32_t bq = band_of(bits);
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(bq != 0) != 0, 0))
            cnt[g + q] += pop32_fix(coords, perm, n_rows, n_cols, r2, bq, jq[g + q], t, h);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // (every chain in the three-address form: for the chain at which c0 dies hipcc 7.2 otherwise accumulates INTO c0's
    //  registers and copies the tile back afterwards -- and its first v_mov of that copy is issued before the wait states
    //  the last MFMA needs: element 15 of the pending tile was stale, found by the parity test at TQ = 4)
    keep_alive(c0);
  };
  // (tile k of the chunk: its fragments and norms were asked for during tile k - 1; every kNormBatch32 tiles the batch
  //  after the next one is ordered -- the slot it lands in was read for the last time a tile ago)
  for (uint32_t k = 0; k < nt; k += 2) {
    tile_body(a0, n0, tb + k, [&](int part) {
      if (part == 0) load_frag32<S>(img, tb + min(k + 1, nt - 1), lane, a1);
      if (part == TQ / kClump32 - 1) N.read(k + 1, h, n1);
    });
    if (k + 1 < nt) {
      tile_body(a1, n1, tb + k + 1, [&](int part) {
        if (part == 0) {
          if (((k + 2) & (kNormBatch32 - 1)) == 0) N.fetch((k + 2) / kNormBatch32 + 1, lane);
          load_frag32<S>(img, tb + min(k + 2, nt - 1), lane, a0);
        }
        if (part == TQ / kClump32 - 1) N.read(k + 2, h, n0);
      });
    }
  }
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    uint32_t total = cnt[qt] + (uint32_t)__shfl_xor((int)cnt[qt], 32, 64);
    if (h == 0 && ((livemask[qt] >> lane) & 1)) {
      if (blockIdx.y == 0) {
        // the sweep met the self pair and counted it iff d2(i,i) < rad2; the reference starts at 1 (:132-134)
        const float dself = exact_d2(coords, n_cols, jq[qt], jq[qt]);
        total += 1u - ((dself < r2) ? 1u : 0u);
      }
      if (gridDim.y == 1)
        pops[jq[qt]] = total;
      else
        atomicAdd(&pops[jq[qt]], total);
    }
  }
}

// Reference chunks of the fp32-input sweeps: every wave costs the same (all pairs are evaluated), so a launch whose
// waves do not fill the chip's wave slots a whole number of times idles at its end -- 7 813 waves of four query tiles on
// 2 048 slots: 3.81 rounds, 5 % lost; on 3 072 (three waves per SIMD) 2.54 rounds, 15 % lost.  The smallest chunk count
// (<= 16, >= 512 reference tiles per chunk) whose wave count comes within 1.5 % of a whole number of rounds.
inline uint32_t chunks32(uint32_t waves_q, uint32_t T, uint32_t slots) {
  uint32_t best = 1;
  double best_eff = 0.0;
  for (uint32_t r = 1; r <= 16 && (r == 1 || T / r >= 512u); ++r) {
    const double x = (double)waves_q * r / slots, eff = x / ceil(x);
    if (eff > best_eff + 1e-9) { best_eff = eff; best = r; }
    if (eff >= 0.985) return r;
  }
  return best;
}

// candidate elements of one accumulator tile: values below the bands of the query's running minima, the query itself,
// pad rows and -- for the lower-free-energy minimum -- the frames at or above the query's free energy left out
// (element r -> bit r of the nn mask, bit 16 + r of the hd mask); by value: see pop_fix
__device__ __attribute__((noinline)) uint32_t nn32_masks(f32x16 acc, float bn, float bh, uint32_t spos, uint32_t pq,
                                                         uint32_t n_rows, uint32_t t, int h) {
  uint32_t m = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t pos = tile_row(t, r, h);
    const bool other = (pos != spos) & (pos < n_rows);
    m |= (other & (acc[r] < bn)) ? (1u << r) : 0u;
    m |= (other & (pos < pq) & (acc[r] < bh)) ? (0x10000u << r) : 0u;
  }
  return m;
}

// evaluates the parked candidates of a wave, 64 at a time (one per lane): x = sorted position | nn-flag << 30 |
// hd-flag << 31, y = query slot; the exact incumbents are order-preserving words (d2 bits << 32 | frame id) in LDS, shared
// by the two half-wave lanes of a query (see nn_wave_flush)
__device__ __attribute__((noinline)) void nn32_wave_flush(const uint2* queue, uint32_t qn, const float* qrows,
                                                          unsigned long long* best, uint32_t n_queries,
                                                          const float* __restrict__ coords, const uint32_t* __restrict__ perm,
                                                          uint32_t n_cols, int lane) {
  for (uint32_t k0 = 0; k0 < qn; k0 += 64) {
    if (k0 + lane < qn) {
      const uint2 ent = queue[k0 + lane];
      const uint32_t j = perm[ent.x & kQueuePosMask], qidx = ent.y;
      const float d2c = dist2_canon_rows(qrows + (size_t)qidx * n_cols, coords + (size_t)j * n_cols, (int)n_cols);
      const unsigned long long key = ((unsigned long long)__float_as_uint(d2c) << 32) | j;
      if ((ent.x >> 30) & 1u) atomicMin(&best[qidx], key);
      if ((ent.x >> 31) & 1u) atomicMin(&best[n_queries + qidx], key);
    }
  }
}

// reference frames ORDERED BY FREE ENERGY (nn_mfma_kernel's scheme: whole tiles below / above / straddling)
template <int S, int TQ>
__global__ __launch_bounds__(256, 2) void nn_mfma32_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols, const float* __restrict__ img,
    const float* __restrict__ norms, const float* __restrict__ img_s, const float* __restrict__ norms_s,
    const uint32_t* __restrict__ perm, const uint32_t* __restrict__ invpos, const uint32_t* __restrict__ pq_of,
    const uint32_t* __restrict__ hdr, uint32_t T, uint32_t i_from, uint32_t i_to,
    unsigned long long* __restrict__ merge64, uint32_t* __restrict__ nn_idx, float* __restrict__ nn_d2, uint32_t* __restrict__ hd_idx,
    float* __restrict__ hd_d2) {
  static_assert(TQ == kClump32, "one clump of chains per reference tile");
  // dynamic LDS, per wave: the ring of row norms (Norms32), candidate list [kWaveQueue] x 8 B, exact incumbents
  // [2][TQ*32] x 8 B, query rows [TQ*32][n_cols]
  extern __shared__ __attribute__((aligned(16))) float nn32_lds[];
  if (hdr[1] != 0) return;
  const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
  const uint32_t wib = threadIdx.x >> 6;
  const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + wib;
  const uint32_t qt0 = i_from / 32 + wave * TQ;
  if (qt0 * 32 >= i_to) return;
  const float M = __uint_as_float(hdr[0]);
  const float eps2 = 2.5f * guard_eps32(M, 4.0f * M, 2 * S, (int)n_cols);   // candidates can be as far apart as 2 sqrt(M)
  uint32_t* wave_lds = reinterpret_cast<uint32_t*>(nn32_lds) + (size_t)wib * (2 * kNormBatch32 * 32 + 2 * kWaveQueue + 4 * TQ * 32 + TQ * 32 * n_cols);
  float* norm_ring = reinterpret_cast<float*>(wave_lds);
  wave_lds += 2 * kNormBatch32 * 32;
  uint2* cand = reinterpret_cast<uint2*>(wave_lds);
  unsigned long long* best64 = reinterpret_cast<unsigned long long*>(wave_lds + 2 * kWaveQueue);
  float* qrows = reinterpret_cast<float*>(wave_lds + 2 * kWaveQueue + 4 * TQ * 32);
  uint32_t qn = 0;   // parked candidates (wave-uniform)

  float b[TQ][S];
  // per query tile and lane: running minima of the MFMA values over this lane's reference rows (acc = |y'|^2 - 2 x'.y' =
  // d2 - |x'|^2: the minima carry the same offset, the band does not care; idle lanes sit at -inf and never trigger),
  // the cached candidate thresholds m + eps2 (they only move in the rare path: a value below a running minimum is below
  // its band), the query's sorted position, the number of frames of lower free energy and the number of reference tiles
  // that hold one
  float m_nn[TQ], m_hd[TQ], thr_nn[TQ], thr_hd[TQ];
  uint32_t spos[TQ], pq[TQ], t_low[TQ];
  uint32_t jq[TQ];
  uint64_t livemask[TQ];
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const uint32_t tile = qt0 + qt;
    jq[qt] = tile * 32 + c;
    const bool live = (tile < T) && (jq[qt] >= i_from) && (jq[qt] < i_to);
    livemask[qt] = __builtin_amdgcn_ballot_w64(live);
    const uint32_t tl = tile < T ? tile : T - 1;
#pragma unroll
    for (int s = 0; s < S; ++s) b[qt][s] = -2.0f * img[((size_t)tl * 64 + lane) * kLane32 + s];
    const uint32_t jl = live ? jq[qt] : (n_rows - 1);
    pq[qt] = live ? pq_of[jl] : 0u;
    spos[qt] = live ? invpos[jl] : 0xFFFFFFFFu;
    t_low[qt] = (pq[qt] + 31u) >> 5;
    m_nn[qt] = live ? INFINITY : -INFINITY;
    m_hd[qt] = live ? INFINITY : -INFINITY;
    if (gridDim.y > 1 && live) {
      // what the waves of earlier reference chunks have published for this query (exact d2, FLT_MAX: nothing yet): the
      // MFMA value of that pair is at most d2 - |x'|^2 + guard, so the running minima may start there instead of at
      // +inf -- only the first chunk of a query group learns its thresholds from nothing
      const float d_nn = __uint_as_float((uint32_t)(merge64[jq[qt]] >> 32)), d_hd = __uint_as_float((uint32_t)(merge64[(size_t)n_rows + jq[qt]] >> 32));
      const float cq = norms[tl * 32 + c], guard = 0.4f * eps2;
      if (d_hd < FLT_MAX) m_hd[qt] = round_up(round_up(d_hd - cq) + guard);
      if (d_nn < FLT_MAX) m_nn[qt] = round_up(round_up(d_nn - cq) + guard);
      m_nn[qt] = fminf(m_nn[qt], m_hd[qt]);
    }
    thr_nn[qt] = m_nn[qt] + eps2;
    thr_hd[qt] = m_hd[qt] + eps2;
    stage_query_rows(qrows + (size_t)qt * 32 * n_cols, nullptr, coords, jl, live, n_cols, lane);
    if (h == 0) {
      best64[qt * 32 + c] = ((unsigned long long)__float_as_uint(FLT_MAX) << 32) | (n_rows + 1);
      best64[TQ * 32 + qt * 32 + c] = ((unsigned long long)__float_as_uint(FLT_MAX) << 32) | (n_rows + 1);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // (query rows and incumbents: written by the h = 0 lanes)
  // reference tiles [tb, te) of this wave's chunk (see chunks32; partial results merge by a 64-bit atomic minimum of
  // (d2 bits, frame id) -- the lexicographic order of the reference's scan -- in merge64, nn32_unpack_kernel writes them out)
  const uint32_t per = (T + gridDim.y - 1) / gridDim.y, tb = blockIdx.y * per, te = min(T, tb + per);
  const uint32_t nt = te > tb ? te - tb : 0u;
  const Norms32 N{norm_ring, norms_s + (size_t)(nt ? tb : 0u) * 32};
  N.fetch(0, lane);
  N.fetch(1, lane);
  float a0[S], a1[S];
  float4 n0[4], n1[4];
  load_frag32<S>(img_s, nt ? tb : 0u, lane, a0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the first two batches of norms have landed)
  N.read(0, h, n0);
  auto flush = [&]() {
    nn32_wave_flush(cand, qn, qrows, best64, TQ * 32, coords, perm, n_cols, lane);
    qn = 0;
  };
  // a chain whose tile minimum came below its query's threshold: masked minima where the tile holds the query itself or
  // straddles its free energy, candidates parked for the exact path, running minima and thresholds renewed
  auto rare_chain = [&](const f32x16& acc, auto qi_c, float tmin, uint32_t t) {
    constexpr int qi = decltype(qi_c)::value;
    const bool special = (t == (spos[qi] >> 5)) | ((t == (pq[qi] >> 5)) & ((pq[qi] & 31u) != 0u));
    float hmin = (t < (pq[qi] >> 5)) ? tmin : INFINITY;
    if (__builtin_amdgcn_ballot_w64(special) != 0) {
      const NnMin g = nn_special(acc, t, h, spos[qi], pq[qi]);   // valid for every lane, just slower
      tmin = g.tmin;
      hmin = g.hmin;
    }
    const bool trig = (tmin < m_nn[qi] + eps2) | (hmin < m_hd[qi] + eps2);
    const float new_nn = fminf(m_nn[qi], tmin), new_hd = fminf(m_hd[qi], hmin);
    if (__builtin_amdgcn_ballot_w64(trig) != 0) {
      uint32_t m = nn32_masks(acc, new_nn + eps2, new_hd + eps2, spos[qi], pq[qi], n_rows, t, h);
      if (!((livemask[qi] >> lane) & 1)) m = 0;
      for (;;) {
        const uint64_t have = __builtin_amdgcn_ballot_w64(m != 0);
        if (have == 0) break;
        const uint32_t n_new = (uint32_t)__builtin_popcountll(have);
        if (qn + n_new > (uint32_t)kWaveQueue) flush();
        if (m != 0) {
          const int r = __builtin_ctz(m | (m >> 16));
          const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(have >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)have, 0));
          cand[slot] = make_uint2(tile_row(t, r, h) | (((m >> r) & 1u) << 30) | (((m >> (16 + r)) & 1u) << 31), (uint32_t)(qi * 32 + c));
          m &= ~(0x10001u << r);
        }
        qn += n_new;
      }
      if (qn >= 64u) flush();
    }
    m_nn[qi] = new_nn;
    m_hd[qi] = new_hd;
    thr_nn[qi] = new_nn + eps2;
    thr_hd[qi] = new_hd + eps2;
  };
  auto tile_body = [&](const float (&a)[S], const float4 (&nv)[4], uint32_t t, auto&& prefetch) {
    const f32x16 c0 = frag16(nv);
    f32x16 acc[TQ];
    chains32<S, TQ>(a, b, c0, acc);
    __builtin_amdgcn_sched_barrier(0);
    prefetch();   // (the next tile's loads in the issue slots that wait for the chains' results: see pop_mfma32_kernel; in
                  //  front of the chains instead: 172.6 against 173.0 ms, the same)
    __builtin_amdgcn_sched_barrier(0);
    // Common path: the raw tile minimum (the query itself included: it only ever makes the test pass) against ONE
    // threshold per chain -- thr_hd >= thr_nn when the tile holds a frame of lower free energy, thr_nn otherwise; what
    // the masked minima and the two bands of the old per-chain test could trigger is a subset of this
    float tm[TQ], dmin = INFINITY;
#pragma unroll
    for (int qi = 0; qi < TQ; ++qi) {
      tm[qi] = INFINITY;
      tile_min<0, 16>(acc[qi], tm[qi]);
      dmin = fminf(dmin, tm[qi] - ((t < t_low[qi]) ? thr_hd[qi] : thr_nn[qi]));   // (inf - inf: NaN, ignored by the minimum)
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(dmin < 0.0f) != 0, 0)) {
      constexpr_for_all<TQ>([&](auto qi_c) {
        constexpr int qi = decltype(qi_c)::value;
        const float thr_c = (t < t_low[qi]) ? thr_hd[qi] : thr_nn[qi];
        if (__builtin_amdgcn_ballot_w64(tm[qi] < thr_c) != 0) rare_chain(acc[qi], qi_c, tm[qi], t);
      });
    }
    __builtin_amdgcn_sched_barrier(0);
    keep_alive(c0);   // (see pop_mfma32_kernel)
  };
  for (uint32_t k = 0; k < nt; k += 2) {   // (see pop_mfma32_kernel)
    tile_body(a0, n0, tb + k, [&]() {
      load_frag32<S>(img_s, tb + min(k + 1, nt - 1), lane, a1);
      N.read(k + 1, h, n1);
    });
    if (k + 1 < nt) {
      tile_body(a1, n1, tb + k + 1, [&]() {
        if (((k + 2) & (kNormBatch32 - 1)) == 0) N.fetch((k + 2) / kNormBatch32 + 1, lane);
        load_frag32<S>(img_s, tb + min(k + 2, nt - 1), lane, a0);
        N.read(k + 2, h, n0);
      });
    }
  }
  flush();
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    if (h == 0 && ((livemask[qt] >> lane) & 1)) {
      const unsigned long long w_nn = best64[qt * 32 + c], w_hd = best64[TQ * 32 + qt * 32 + c];
      if (gridDim.y == 1) {
        nn_idx[jq[qt]] = (uint32_t)w_nn;
        nn_d2[jq[qt]] = __uint_as_float((uint32_t)(w_nn >> 32));
        hd_idx[jq[qt]] = (uint32_t)w_hd;
        hd_d2[jq[qt]] = __uint_as_float((uint32_t)(w_hd >> 32));
      } else {
        atomicMin(&merge64[jq[qt]], w_nn);
        atomicMin(&merge64[(size_t)n_rows + jq[qt]], w_hd);
      }
    }
  }
}

__global__ void nn32_unpack_kernel(const unsigned long long* __restrict__ merge64, uint32_t n_rows, uint32_t i_from,
                                   uint32_t i_to, uint32_t* __restrict__ nn_idx, float* __restrict__ nn_d2,
                                   uint32_t* __restrict__ hd_idx, float* __restrict__ hd_d2) {
  const uint32_t i = i_from + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= i_to) return;
  const unsigned long long a = merge64[i], b = merge64[(size_t)n_rows + i];
  nn_idx[i] = (uint32_t)a;
  nn_d2[i] = __uint_as_float((uint32_t)(a >> 32));
  hd_idx[i] = (uint32_t)b;
  hd_d2[i] = __uint_as_float((uint32_t)(b >> 32));
}
