// dc_mfma_msym.hpp -- SYMMETRIC population sweep for SEVERAL radii (included by dc_mfma_kernels.hpp, inside namespace
// dc::{anonymous}, after dc_mfma_shared.hpp).
//
// The reference counts every unordered frame pair once for all radii and credits both frames
// (density_clustering.cpp:170-188: j > i only, descending radii with an early break).  pop_shared_kernel<NM, 2, NR> is
// one-sided: it evaluates every ORDERED tile pair, because crediting the reference side of a chain is a sum ACROSS the
// 32 query lanes per reference row and radius -- with the round-2 form of that sum (ref_credit: a DPP tree and one
// 128-byte global atomic per reference tile, WAVE and radius) the eight-radius sweep of one C5 rank went from 797 to
// 2 062 ms (2 * 10^9 atomics).  This kernel keeps the structure of pop_shared_kernel -- a workgroup of four waves, two
// resident query tiles per wave, the surviving reference tiles streamed once per workgroup through an LDS ring -- and
// moves the lane sum out of the chains:
//   in the chain loop   per reference tile and radius a wave adds the sign strings of its two chains bit-sliced
//                       (2-bit counts), spreads them to 4-bit fields (two words: odd / even elements) and adds those to
//                       a LANE-PRIVATE accumulator of the tile's ring slot in LDS with ONE ds_add_u64 -- no lane
//                       reduction, no conflict (address = slot, radius, lane), 7 VALU instructions per radius and tile
//                       against the 2 x 32 of the second chain pair it replaces.  Four waves x two tiles: fields <= 8.
//   two windows later   (every wave has long left the tile) the four waves share out the (tile, radius) accumulators of
//                       the retired window: fields to bytes, four DPP steps inside the 16-lane rows, the row sums
//                       staged in LDS, lane i picks the two bytes of reference row i -- and ONE 256-byte atomic per
//                       tile and PAIR of radii into counts laid out [tile][radius][32 rows]: 4 atomics per reference
//                       tile and WORKGROUP instead of 32.
// Ownership of the unordered pairs is that of the one-radius symmetric sweeps (dc_mfma_kernels.hpp "symmetric
// population sweep"): groups of 4 * TQ = 8 tiles on a circle, a group meets its own tiles in both orders (query side
// only) and the half of the other groups that lies ahead of it; the rule does not depend on the rank that runs a
// group, so segments of a sharded run produce PARTIAL counts of all rows that merge by summation.
#ifndef DC_MS_WIN
#define DC_MS_WIN 3
#endif
constexpr int kMsWin = DC_MS_WIN;       // reference tiles per window: one barrier per window (LDS: two workgroups per CU at 3)
constexpr int kMsRing = 2 * kMsWin;     // operand slots: the window in use and the one in flight
constexpr int kMsAccSlots = 2 * kMsWin; // accumulator slots: the window in use and the one being reduced
constexpr int kMsTQ = 2;

// position -> word of the counts [tile][NR][32]
template <int NR>
__device__ __forceinline__ size_t ms_index(uint32_t pos, int rr) {
  return (size_t)(pos >> 5) * (NR * 32) + (size_t)rr * 32 + (pos & 31u);
}

template <int NR>
__device__ __attribute__((noinline)) void pop_wave_flush_ms(const uint2* queue, uint32_t qn, const uint32_t* jq_tab,
                                                            uint32_t* fix_tab, uint32_t n_queries,
                                                            const float* __restrict__ coords,
                                                            const float* __restrict__ coords_r, uint32_t n_cols, Rad2 rad2,
                                                            int lane, uint32_t* __restrict__ pops_pos, uint32_t group_tiles,
                                                            uint32_t own_group) {
  for (uint32_t k0 = 0; k0 < qn; k0 += 64) {
    if (k0 + lane < qn) {
      const uint2 ent = queue[k0 + lane];
      const uint32_t qidx = ent.y & 0xFFu, flags = ent.y >> 8;
      const float d2c = dist2_canon_rows(coords + (size_t)jq_tab[qidx] * n_cols, coords_r + (size_t)ent.x * n_cols, (int)n_cols);
      const bool both = (ent.x >> 5) / group_tiles != own_group;   // (this workgroup alone evaluates the pair)
#pragma unroll
      for (int rr = 0; rr < NR; ++rr)
        if (((flags >> rr) & 1u) && d2c < rad2.v[rr]) {
          atomicAdd(&fix_tab[rr * n_queries + qidx], 1u);
          if (both) atomicAdd(&pops_pos[ms_index<NR>(ent.x, rr)], 1u);
        }
    }
  }
}

// ---- radii that hold nothing of a chain -------------------------------------------------------------------------
// With ASCENDING radii (delta_r = r_r^2 - r_0^2 >= 0, non-decreasing) a chain whose smallest accumulator value is
// >= 2 + delta_r has all 1 024 pairs outside radius r -- and outside every smaller one.  The strings of such radii are
// the all-outside pattern (sign 0, bit 30 set: 0x55555555 -- no count, no band, nothing for the reference side), and
// their 24 epilogue instructions each need not be issued.  At C5 (radii 0.30 ... 0.65 against a typical intra-cluster
// distance of 0.62) the smallest radius is empty in 99.5 % of the chains, the two smallest in 87.7 %.
// What the decision may cost decides its form: the lane minima (8 v_min3) and ONE wave-level compare -- "the first
// kMsSkip<NR> radii are all empty, or none is skipped" -- with one scalar hand-off and two code variants: one rank of C5
// 412 -> 397 ms; finer decisions lose what they find (three levels 398, five levels 402 ms without the bookkeeping:
// about 100 cycles per skipped radius and chain against 150 for a five-way decision; forced skips of two / three radii
// with no decision at all: 368 / 338 ms).  Radii in any other order: nothing is skipped.
constexpr uint32_t kAllOutside = 0x55555555u;
template <int NR>
constexpr int kMsSkip = NR / 4;   // the leading radii skipped together: 2 of 8, 1 of 4
template <int NR, int K0>
__device__ __forceinline__ void mr_begin_k(MrAcc<NR>& e) {
#pragma unroll
  for (int rr = 0; rr < NR; ++rr) e.bits[rr] = (rr < K0) ? kAllOutside : 0u;
}
template <int NR, int K0, int R0, int R1>   // elements [R0, R1), both even; radii K0 .. NR-1
__device__ __forceinline__ void mr_epi_k(const f32x16& acc, const PopDeltas<NR>& dl, MrAcc<NR>& e) {
  static_assert(R0 % 2 == 0 && R1 % 2 == 0, "elements are handled in pairs");
  // (the subtraction of the next element pair in front of the current pair's two v_alignbit: see mr_epi)
#pragma unroll
  for (int rr = K0; rr < NR; ++rr) {
    if (rr == 0) {
#pragma unroll
      for (int r = R0; r < R1; ++r) e.bits[0] = __builtin_amdgcn_alignbit(e.bits[0], __float_as_uint(acc[r]), 30);
    } else if constexpr (R1 > R0) {
      const f32x2 d2 = {dl.d[rr], dl.d[rr]};
      f32x2 t = f32x2{acc[R0], acc[R0 + 1]} - d2;
#pragma unroll
      for (int r = R0; r < R1; r += 2) {
        f32x2 tn = t;
        if (r + 2 < R1) tn = f32x2{acc[r + 2], acc[r + 3]} - d2;
        e.bits[rr] = __builtin_amdgcn_alignbit(e.bits[rr], __float_as_uint(t.x), 30);
        e.bits[rr] = __builtin_amdgcn_alignbit(e.bits[rr], __float_as_uint(t.y), 30);
        t = tn;
      }
    }
  }
}
template <int NM, int NR, int K0, int MI = 0>
__device__ __forceinline__ void mr_chain_k(const s16x8 (&a)[NM], const s16x8 (&b)[NM], const f32x16& c0,
                                           f32x16& acc_new, const f32x16& acc_old, const PopDeltas<NR>& dl,
                                           MrAcc<NR>& e) {
  if constexpr (MI < NM) {
    if constexpr (MI == 0)
      acc_new = mfma16(a[0], b[0], c0);
    else
      acc_new = mfma16(a[MI], b[MI], acc_new);
    mr_epi_k<NR, K0, 2 * ((8 * MI) / NM), 2 * ((8 * (MI + 1)) / NM)>(acc_old, dl, e);
    mr_chain_k<NM, NR, K0, MI + 1>(a, b, c0, acc_new, acc_old, dl, e);
  }
}

// INPL: the instance that takes the thresholds off the accumulator in place (below).  Both instances are launched; which
// one runs is decided on the device, from the scale and the radii (the other returns at once, like the gated direct kernel).
template <int NM, int NR, bool INPL>
__global__ __launch_bounds__(256, 2) void pop_msym_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols,
    const uint4* __restrict__ img_r, const float* __restrict__ norms_r,
    const float4* __restrict__ box_r, const float* __restrict__ coords_r, uint32_t T,
    const uint4* __restrict__ img_q, const float* __restrict__ norms_q,
    const uint32_t* __restrict__ perm_q, const float4* __restrict__ box_q, uint32_t n_q, QSeg q_seg,
    const uint32_t* __restrict__ hdr, unsigned long long* __restrict__ chain_counter, Rad2 rad2, int n_rad,
    CompView CV, uint32_t* __restrict__ pops_pos) {
  constexpr int TQ = kMsTQ;
  static_assert(NR == 4 || NR == 8, "radius pairs per reducing wave");
  static_assert(4 * TQ <= 15, "4-bit fields of the lane-private accumulators hold a workgroup's chains on one tile");
  __shared__ uint32_t lists[4][kShareSub];
  __shared__ uint32_t list_cnt[4];
  __shared__ float4 wave_box[4];
  __shared__ uint32_t flush_flag[2];   // "some wave's queue is filling up": all four flush at the next window (parity of the window)
  __shared__ __attribute__((aligned(16))) uint32_t stage[4][(kMsWin * (NR / 2) + 3) / 4][2][16];   // per wave and unit of a window: the row sums of the unit's two radii (lanes 15/31/47/63)
  // dynamic LDS: operand ring [kMsRing][kTileUnits] x 16 B, the accumulators [kMsAccSlots][NR][64] x 8 B, then per wave
  // the compact queue of deferred exact evaluations [kWaveQueue] x 8 B, the positions of its queries [TQ*32] and their
  // exact-path counts [NR][TQ*32]
  extern __shared__ __attribute__((aligned(16))) float shared_dyn[];
  if (hdr[1] != 0) return;   // flagged data: the gated direct kernel runs instead
  constexpr int kUnits = kTileUnits<NM>;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31, wib = tid >> 6;
  const uint32_t TQT = (n_q + 31) / 32;
  const uint32_t n_groups = (TQT + 4u * TQ - 1u) / (4u * TQ);
  const uint32_t blk_unit = xcd_block(seg_groups(n_groups, q_seg));
  if (blk_unit == 0xFFFFFFFFu) return;   // (pad block of the grid: the whole workgroup leaves)
  const uint32_t group = seg_group(blk_unit, q_seg);
  const uint32_t chunk = blockIdx.y, n_chunks = gridDim.y;
  if (group * (4u * TQ) >= TQT) return;   // whole workgroup leaves
  const uint32_t qt0 = (group * 4u + (uint32_t)wib) * TQ;
  const bool wave_live = qt0 < TQT;       // (a wave without tiles keeps loading, reducing and meeting the barriers)
  uint4* ring = reinterpret_cast<uint4*>(shared_dyn);
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(shared_dyn + kMsRing * kUnits * 4);   // [slot][NR][64]
  uint32_t* wave_lds = reinterpret_cast<uint32_t*>(acc + kMsAccSlots * NR * 64) + (size_t)wib * shared_wave_words(TQ, NR);
  uint2* queue = reinterpret_cast<uint2*>(wave_lds);   // (reference position, query | radius flags << 8)
  uint32_t* jq_tab = wave_lds + 2 * kWaveQueue;        // position of query (qt, c) in the order
  uint32_t* fix_tab = jq_tab + TQ * 32;                // [NR][TQ*32]: band pairs the exact path found inside
  uint32_t qn = 0;                                     // queued entries (wave-uniform)

  const PopSetup<NR> P = pop_setup<NR>(hdr, rad2, n_cols);
  float r2max = rad2.v[0];
#pragma unroll
  for (int rr = 1; rr < NR; ++rr) r2max = fmaxf(r2max, rad2.v[rr]);
  const float far2 = r2max * 1.0001f;   // boxes at least this far apart (squared) hold no pair inside
  // radii that hold nothing of a chain (see mr_chain_k): ascending radii only
#ifdef DC_MS_ABL_NOSKIP
  bool radii_ascending = false;
#else
  bool radii_ascending = n_rad >= kMsSkip<NR>;
#endif
#pragma unroll
  for (int rr = 1; rr < NR; ++rr) radii_ascending &= (rr >= n_rad) || (P.dl.d[rr] >= P.dl.d[rr - 1]);
  // (rounded UP: fl(2 + d) may lie below 2 + d -- by up to 2^-5 at d ~ 2^19 -- and a chain whose minimum equals it would
  //  skip radii for which the epilogue's own test, fl(acc - d) >= 2, sends the pair to the band; with t >= next_up(..)
  //  >= 2 + d the difference is >= 2 exactly, and rounding is monotone: never laxer than the epilogue)
  const float skip_thr = next_up(2.0f + P.dl.d[kMsSkip<NR> - 1]);   // (d[0] = 0)
  // (a wave that finds nothing to skip in a whole round of its survivor list -- fewer than a quarter of the chains --
  //  raises the threshold to +inf: its chains then go straight to the full epilogue.  Measured with radii that never
  //  skip, 0.50 ... 0.65: 451 -> 460 ms, +2 %, the price of the minima; with C5's radii 412 -> 397 ms.)
  float skip_thr_now = radii_ascending ? skip_thr : INFINITY;
  uint32_t skip_hits = 0;
  // number of leading radii (0 or kMsSkip<NR>) that hold nothing of the chain with accumulator `acc` (wave-uniform)
  auto skip_count = [&](const f32x16& acc) -> int {
    float tmin = INFINITY;
    tile_min<0, 16>(acc, tmin);
    const bool hit = __builtin_amdgcn_ballot_w64(tmin < skip_thr_now) == 0;
    skip_hits += hit ? 1u : 0u;
    return hit ? kMsSkip<NR> : 0;
  };
  auto with_skip = [&](int k, auto&& body) __attribute__((always_inline)) {
    switch (k) {
      case 0: body(std::integral_constant<int, 0>{}); break;
      default: body(std::integral_constant<int, kMsSkip<NR>>{}); break;
    }
  };

  // ---- thresholds taken off the accumulator IN PLACE (round 6) ------------------------------------------------------------
  // Radius k's string needs the top two bits of acc - delta_k: 8 v_pk_add_f32 per radius and chain in front of the 16
  // v_alignbit.  The matrix pipe is idle nine tenths of this sweep, and acc_k = acc_{k-1} - (delta_k - delta_{k-1}) is ONE
  // MFMA: A = ones in the first three slots, B = three fp16 pieces of the (negated) step there, C = the accumulator.
  // The two chains of a tile go through their radii in turn, so that the step of one runs under the 16 v_alignbit of the
  // other.  What the steps cost in accuracy -- four truncated addends and one rounding each -- is part of the band the
  // scale was chosen for (guard_shift, kHdrShift); a launch whose steps the band does not cover, or whose steps do not
  // fit fp16, keeps the subtractions on the vector unit.
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const uint32_t m32 = (lane < 32) ? 0xFFFFFFFFu : 0u;   // (the first eight K slots are the first half-wave's)
  uint32_t sh0[NR + 1], sh1[NR + 1];   // [k]: radius k-1 -> k (k >= 1); [NR]: radius 0 -> kMsSkip<NR>
  bool inplace = hdr[kHdrShift] >= (uint32_t)(NR - 1);
  {
    auto pieces = [&](float x, uint32_t& w0, uint32_t& w1) {
      auto fz = [](_Float16 v) { return (fabsf((float)v) < 6.103515625e-5f) ? (_Float16)0.0f : v; };   // (below 2^-14: stored as zero)
      const _Float16 p0 = fz((_Float16)x);
      const float r1 = x - (float)p0;
      const _Float16 p1 = fz((_Float16)r1);
      const _Float16 p2 = fz((_Float16)(r1 - (float)p1));
      w0 = (uint32_t)__builtin_bit_cast(unsigned short, p0) | ((uint32_t)__builtin_bit_cast(unsigned short, p1) << 16);
      w1 = (uint32_t)__builtin_bit_cast(unsigned short, p2);
      return fabsf(x) < 60000.0f;
    };
    sh0[0] = sh1[0] = 0u;
#pragma unroll
    for (int k = 1; k <= NR; ++k) {
      uint32_t w0, w1;
      // (radii the call does not use -- squared radius -1 -- repeat the string of the last one that it does: no step;
      //  their counts are never read and their bands are its bands)
      const float step = (k == NR) ? -P.dl.d[kMsSkip<NR>] : ((k < n_rad) ? -(P.dl.d[k] - P.dl.d[k - 1]) : 0.0f);
      inplace &= pieces(step, w0, w1);
      sh0[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w0);
      sh1[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w1);
    }
  }
#ifdef DC_MS_NO_INPLACE
  inplace = false;
#endif
  if (inplace != INPL) return;   // (the other instance's launch; before any barrier)
  const s16x8 ones_op = __builtin_bit_cast(s16x8, u32x4_t{0x3C003C00u & m32, 0x00003C00u & m32, 0u, 0u});
  auto shift_op = [&](int k) { return __builtin_bit_cast(s16x8, u32x4_t{sh0[k] & m32, sh1[k] & m32, 0u, 0u}); };
  uint32_t shifts = 0;   // MFMAs issued for the steps (the executed-flop figure of the bench line counts them)
  for (uint32_t k = tid; k < (uint32_t)(kMsAccSlots * NR * 64); k += 256) acc[k] = 0ull;
  if (tid < 2) flush_flag[tid] = 0u;

  s16x8 b[TQ][NM];
  uint32_t cnt_q[TQ][NR], jq[TQ];
  uint64_t livemask[TQ];
  float4 gbox = make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const uint32_t tile = qt0 + qt;
    const uint32_t tl = tile < TQT ? tile : TQT - 1;
    const uint32_t pos = tile * 32 + c;
    const uint32_t frame = ((tile < TQT) && (pos < n_q)) ? perm_q[pos] : kInvalidFrame;   // (pad positions: kInvalidFrame)
    const bool live = frame != kInvalidFrame;
    livemask[qt] = __builtin_amdgcn_ballot_w64(live);
    jq[qt] = live ? frame : 0u;
    const float cq = live ? norms_q[tl * 32 + c] - P.rad2e.v[0] : dead_const(P.sc);
    load_query<NM>(img_q, tl, lane, h, cq, P.sc, b[qt]);
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) cnt_q[qt][rr] = 0;
    if (h == 0) {
      jq_tab[qt * 32 + c] = pos;   // (the queries are rows of the reference order: their original coordinates sit in coords_r)
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) fix_tab[rr * (TQ * 32) + qt * 32 + c] = 0;
    }
    const float4 qb = (tile < TQT) ? box_q[tile] : make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
    gbox.x = fminf(gbox.x, qb.x);
    gbox.y = fmaxf(gbox.y, qb.y);
    gbox.z = fminf(gbox.z, qb.z);
    gbox.w = fmaxf(gbox.w, qb.w);
  }
  if (lane == 0) wave_box[wib] = gbox;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 4; ++w) {   // the workgroup's box: one survivor list for all four waves
    const float4 wb = wave_box[w];
    gbox.x = fminf(gbox.x, wb.x);
    gbox.y = fmaxf(gbox.y, wb.y);
    gbox.z = fminf(gbox.z, wb.z);
    gbox.w = fmaxf(gbox.w, wb.w);
  }

  uint32_t sb[NR][TQ];   // strings of the two chains on the current reference tile
  uint32_t next_par = 0;  // parity of the window after the current one
  auto flush = [&]() {
#ifndef DC_MS_ABL_NOFLUSH
    pop_wave_flush_ms<NR>(queue, qn, jq_tab, fix_tab, TQ * 32, coords_r, coords_r, n_cols, rad2, lane, pops_pos, 4u * TQ, group);
#endif
    qn = 0;
  };
  // the rest of an epilogue: query-side counts, band test, parking of the band pairs
  auto finish = [&](auto qi_c, const MrAcc<NR>& e, uint32_t t) {
    constexpr int qi = decltype(qi_c)::value;
    uint32_t decided = 0xFFFFFFFFu;   // bit 31 - 2 r: element r is inside or outside for EVERY radius
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      cnt_q[qi][rr] += __builtin_popcount(inside_of(e.bits[rr]));
      decided &= e.bits[rr] | (e.bits[rr] << 1);
      sb[rr][qi] = e.bits[rr];
    }
    uint32_t m = ~decided & kSignBits;
    if (__builtin_expect((__builtin_amdgcn_ballot_w64(m != 0) & livemask[qi]) != 0, 0)) {
      // Pad rows (acc = +inf) and idle lanes (acc ~ 2^22) are never in a band.
      uint32_t fl[NR];   // per radius: bit (31 - 2 r) set <=> element r sits in that radius' band
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) fl[rr] = band_of(e.bits[rr]);
      for (;;) {
        const uint64_t have = __builtin_amdgcn_ballot_w64(m != 0);
        if (have == 0) break;
        const uint32_t n_new = (uint32_t)__builtin_popcountll(have);
        if (qn + n_new > (uint32_t)kWaveQueue) flush();
        if (m != 0) {
          const int p = __builtin_ctz(m);
          uint32_t flags = 0;
#pragma unroll
          for (int rr = 0; rr < NR; ++rr) flags |= ((fl[rr] >> p) & 1u) << rr;
          const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(have >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)have, 0));
          queue[slot] = make_uint2(tile_row(t, element_of(p), h), (uint32_t)(qi * 32 + c) | (flags << 8));
          m &= m - 1;
        }
        qn += n_new;
      }
      // The exact evaluations of a wave are a few microseconds of memory latency during which its three partners end up
      // waiting at the next window barrier: a wave whose queue is filling asks for a flush of ALL four at the next window
      // (one stall instead of four; ablation: the flushes cost 24 of 393 ms, a flush batch itself 2 - 3 us); it only
      // flushes on its own when the queue is about to overflow.
      if (qn >= 40u && lane == 0) flush_flag[next_par] = 1u;
      if (qn >= 96u) flush();
    }
  };
  // reference side of the pending tile: per radius the two strings added bit-sliced (2-bit counts at the elements'
  // string positions), the counts spread to 4-bit fields -- odd elements in the low word, even ones in the high word --
  // and added to this lane's accumulator of the tile's slot
  auto credit = [&](uint32_t acc_slot) {
    unsigned long long* a_lane = acc + (size_t)acc_slot * (NR * 64) + lane;
    // (no "anything inside?" test per radius and no test for the radii the call does not use -- their strings are all-outside,
    //  the host leaves -1 in their squared radii: a wave-level test is a compare, a scalar hand-off and a branch, and the
    //  sixteen branches of a tile's credit cost more wave time than its eight LDS additions -- round 6, cycle stamps)
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      const uint32_t s0 = sb[rr][0], s1 = sb[rr][1];
      const uint32_t hi = s0 & s1 & kSignBits, lo = (s0 ^ s1) & kSignBits;
      const uint32_t x = hi | (lo >> 1);                       // element r: 0..2 at bits 31-2r, 30-2r
      const uint32_t A = x & 0x33333333u, B = (x >> 2) & 0x33333333u;
      __hip_atomic_fetch_add(a_lane + rr * 64, ((unsigned long long)B << 32) | A, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  // reduce the accumulators of a retired window: its (tile, pair of radii) units are dealt to the four waves;
  // entry_of(k): the k-th reference tile of the window as it comes out of LDS (not yet a scalar), n_tiles of them
  const uint32_t my_byte = ref_credit_byte(lane);   // byte (of the 16 a row-end lane stages) with the count of reference row lane & 31
  // (round 6: the wave's units of a window -- up to kMsUnits = ceil(kMsWin * NR / 2 / 4) -- go through the reducer TOGETHER,
  //  phase by phase: all accumulator reads, then the byte sums, then the staged gather, then the atomics.  One unit after the
  //  other was three dependent LDS round trips per unit, 16 % of a C5 rank's sweep.)
  constexpr uint32_t kMsUnits = (kMsWin * (NR / 2) + 3) / 4;
  // (full_c: the window holds kMsWin tiles -- every one but the last of a round: no unit is missing, nothing to test)
  auto reduce_window = [&](uint32_t win, auto&& entry_of, uint32_t n_tiles, auto full_c) {
    constexpr bool kFull = decltype(full_c)::value;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4 LdsU4;
    typedef __attribute__((address_space(3))) unsigned char LdsU8;
    constexpr uint32_t kPairs = NR / 2;
    const uint32_t n_units = n_tiles * kPairs;
    unsigned long long v[kMsUnits][2];
    uint32_t which[kMsUnits], rr0[kMsUnits], t_unit[kMsUnits];
    bool have[kMsUnits];
    // (no wave-level "is there anything?" tests: a workgroup's eight chains leave something inside nearly every radius of
    //  nearly every tile they reach, and a test is a compare, a scalar hand-off and a branch -- round 6, cycle stamps)
    // phase 1: the lane's fields of every unit and radius; the accumulators are cleared for the window after next
#pragma unroll
    for (uint32_t j = 0; j < kMsUnits; ++j) {
      const uint32_t p = (uint32_t)wib + 4u * j;
      // (kFull and a unit count that is a multiple of the four waves: every slot of every wave holds a unit)
      have[j] = (kFull && (kMsWin * kPairs) % 4u == 0u) || p < n_units;
      which[j] = have[j] ? p / kPairs : 0u;
      rr0[j] = have[j] ? 2u * (p % kPairs) : 0u;
      t_unit[j] = entry_of(which[j]);   // (the LDS read; a scalar in phase 4)
      const uint32_t slot = (win & 1u) * kMsWin + which[j];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        unsigned long long* w = acc + (size_t)slot * (NR * 64) + (size_t)(rr0[j] + u) * 64 + lane;
        v[j][u] = have[j] ? *w : 0ull;
      }
    }
#pragma unroll
    for (uint32_t j = 0; j < kMsUnits; ++j) {
      const uint32_t slot = (win & 1u) * kMsWin + which[j];
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (have[j]) acc[(size_t)slot * (NR * 64) + (size_t)(rr0[j] + u) * 64 + lane] = 0ull;
    }
    // phase 2: fields -> bytes -> sums over the 16 lanes of a row, left by the row-end lanes in the unit's stage
#pragma unroll
    for (uint32_t j = 0; j < kMsUnits; ++j) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t A = (uint32_t)v[j][u], B = (uint32_t)(v[j][u] >> 32);
        uint32_t W[4] = {A & 0x0F0F0F0Fu, B & 0x0F0F0F0Fu, (A >> 4) & 0x0F0F0F0Fu, (B >> 4) & 0x0F0F0F0Fu};
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // bytes <= 8 -> <= 128 over the 16 lanes of a row
          W[e] += dpp_take<0x111>(W[e]);         // row_shr:1
          W[e] += dpp_take<0x112>(W[e]);         // row_shr:2
          W[e] += dpp_take<0x114>(W[e]);         // row_shr:4
          W[e] += dpp_take<0x118>(W[e]);         // row_shr:8
        }
        if ((lane & 15) == 15) ((LdsU4*)stage[wib][j][u])[lane >> 4] = u32x4{W[0], W[1], W[2], W[3]};
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (one wave: its LDS operations execute in order)
    __builtin_amdgcn_wave_barrier();
    // phase 3: lane L: radius rr0 + (L >> 5), reference row L & 31 -- the row lies in half hh = bit 2 of the row index; the
    // sums of that half's 32 query lanes are the row-end lanes 2 hh and 2 hh + 1
    uint32_t cnt2[kMsUnits];
#pragma unroll
    for (uint32_t j = 0; j < kMsUnits; ++j) {
      const int u = lane >> 5;
      const uint32_t hh = my_byte >> 4, byte = my_byte & 15u;
      const volatile LdsU8* st = (const volatile LdsU8*)stage[wib][j][u];
      cnt2[j] = (uint32_t)st[(2u * hh) * 16u + byte] + (uint32_t)st[(2u * hh + 1u) * 16u + byte];
    }
    __builtin_amdgcn_wave_barrier();
    // phase 4: one 256-byte atomic per unit
#pragma unroll
    for (uint32_t j = 0; j < kMsUnits; ++j) {
      const uint32_t c2 = cnt2[j];
      const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane(t_unit[j]);
#ifdef DC_MS_ABL_NOATOMIC
      if (c2 == 0xFFFFFFFFu)
#else
      if (have[j] && c2 != 0u && rr0[j] + (uint32_t)(lane >> 5) < (uint32_t)n_rad && 32u * t + (uint32_t)(lane & 31) < CV.n_pos)
#endif
        atomicAdd(&pops_pos[(size_t)t * (NR * 32) + (size_t)(rr0[j] + (lane >> 5)) * 32 + (uint32_t)(lane & 31)], c2);
    }
  };

  uint32_t chains = 0;
  // (only the tiles of the workgroup's own COMPONENT: every other frame is at least r_max away -- CompView)
  const uint32_t my_comp = CV.tile_comp_q[group * (4u * TQ)];
  const uint32_t t_lo = CV.range_r[2 * my_comp], t_hi = min(CV.range_r[2 * my_comp + 1], T);
  const uint32_t u_lo = (t_lo > chunk) ? (t_lo - chunk + n_chunks - 1) / n_chunks : 0u;
  const uint32_t U = (t_hi > chunk) ? (t_hi - chunk + n_chunks - 1) / n_chunks : 0u;
  auto tile_of = [&](uint32_t u) { return chunk + u * n_chunks; };
  for (uint32_t base = u_lo; base < U; base += 4 * kShareSub) {
    // ---- scan: every wave tests its quarter of the round's boxes against the workgroup's box
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < kShareSub; k += 64) {
      const uint32_t u = base + (uint32_t)wib * kShareSub + k + lane;
      bool ok = false;
      uint32_t t = 0;
      if (u < U) {
        t = tile_of(u);
        ok = box_gap2(gbox, box_r[t]) < far2;
        // the workgroup's own group, or a group at most half the circle ahead (exactly half: the lower index)
        const uint32_t gt = t / (4u * TQ);
        const uint32_t ahead = (gt >= group) ? gt - group : gt + n_groups - group;
        ok = ok & ((2u * ahead < n_groups) | ((2u * ahead == n_groups) & (group < gt)));
      }
      const uint64_t m = __builtin_amdgcn_ballot_w64(ok);
      if (ok) lists[wib][cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = t;
      cnt += (uint32_t)__builtin_popcountll(m);
    }
    if (lane == 0) list_cnt[wib] = cnt;
    __syncthreads();
    // (wave-uniform values held in scalar registers: what the compiler cannot see of an LDS read)
    const uint32_t o1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)list_cnt[0]);
    const uint32_t o2 = o1 + (uint32_t)__builtin_amdgcn_readfirstlane((int)list_cnt[1]);
    const uint32_t o3 = o2 + (uint32_t)__builtin_amdgcn_readfirstlane((int)list_cnt[2]);
    const uint32_t total = o3 + (uint32_t)__builtin_amdgcn_readfirstlane((int)list_cnt[3]);
    if (total != 0) {
      // (scalar arithmetic, no branches: the list of wave w starts at w * kShareSub, its entries are o_w .. o_{w+1} - 1 of the round)
      const uint32_t d21 = o2 - o1, d32 = o3 - o2;
      // (entry_raw: the LDS read alone -- issued where the index is known, made a scalar (v_readfirstlane, which waits for it)
      //  where the tile number is needed: three dependent LDS round trips at the head of a window's fetch are one)
      auto entry_raw = [&](uint32_t i) {
        i = min(i, total - 1u);
        const uint32_t g1 = 0u - (uint32_t)(i >= o1), g2 = 0u - (uint32_t)(i >= o2), g3 = 0u - (uint32_t)(i >= o3);   // 0 / ~0
        const uint32_t first = (o1 & g1) + (d21 & g2) + (d32 & g3);             // o_w
        const uint32_t idx = ((uint32_t)kShareSub & g1) + ((uint32_t)kShareSub & g2) + ((uint32_t)kShareSub & g3) + (i - first);
        return (&lists[0][0])[idx];
      };
      // the tiles of a window are fetched fragment-wise: the NM + 1 pieces of a tile (its MFMA fragments and its 32
      // row norms) go round the four waves
      // (branch-free: a tile beyond the end of the list is the list's last tile once more, into a slot nobody reads; the
      //  pieces of tile k that this wave fetches are m = (wave - k) mod 4 and m + 4 -- a fragment, the norms (eight
      //  lanes) or nothing, by lane predicate.  As a loop over all pieces with a wave-level test each, a window's fetch
      //  was 21 branches.)
      auto fetch_window = [&](uint32_t i0) {
        uint32_t t_raw[kMsWin];
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)kMsWin; ++k) t_raw[k] = entry_raw(i0 + k);
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)kMsWin; ++k) {
          const uint32_t i = i0 + k;
          const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane(t_raw[k]);
          const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(ring + (i % kMsRing) * kUnits));
          const uint4* src = img_r + (size_t)t * (NM * 64) + lane;
          const uint32_t m_a = ((uint32_t)__builtin_amdgcn_readfirstlane(wib) - k) & 3u;   // (wave-uniform: a scalar for M0)
          const uint4* nrm = reinterpret_cast<const uint4*>(norms_r + (size_t)t * 32) + lane;
#pragma unroll
          for (uint32_t j = 0; j < ((uint32_t)NM + 4u) / 4u; ++j) {
            const uint32_t m = m_a + 4u * j;
            if (4u * j + 3u < (uint32_t)NM) {   // (a fragment whatever the wave: compile-time)
              lds_dma16(src + m * 64u, dst + m * 1024u);
            } else {
              const void* p_m = (m == (uint32_t)NM) ? (const void*)nrm : (const void*)(src + m * 64u);
              if ((m < (uint32_t)NM) | ((m == (uint32_t)NM) & (lane < 8))) lds_dma16(p_m, dst + m * 1024u);
            }
          }
        }
      };
      fetch_window(0);
      for (uint32_t i = 0; i < total; ++i) {
        if ((i % kMsWin) == 0) {
          __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of the window starting at i
          __syncthreads();                      // ... and every wave's adds to the previous window's accumulators have landed
          fetch_window(i + kMsWin);
          {
            const uint32_t par = (i / kMsWin) & 1u;
            next_par = par ^ 1u;
            if (flush_flag[par] != 0u) {        // (a late reader may miss a flag wave 0 has already cleared: a flush is never needed, only wanted)
              if (qn != 0u) flush();
              if (tid == 0) flush_flag[par] = 0u;
            }
          }
#ifndef DC_MS_ABL_NOREDUCE
          if (i >= (uint32_t)kMsWin) {
            const uint32_t j = i - kMsWin;
            reduce_window(j / kMsWin, [&](uint32_t k) { return entry_raw(j + k); }, (uint32_t)kMsWin, std::true_type{});
          }
#endif
        }
        const uint32_t t_raw = entry_raw(i);   // (made a scalar behind the first epilogue: nothing before it needs the tile's number)
        const uint4* slot = ring + (i % kMsRing) * kUnits;
        // (operands and accumulators are locals of the tile: kept alive across the reducer above -- as they are when
        //  the first chain of the next tile is started early -- they cost 16 - 44 spilled registers and 20 - 40 ms of 450)
        s16x8 a[NM];
        float4 nv[4];
#pragma unroll
        for (int m = 0; m < NM; ++m) a[m] = __builtin_bit_cast(s16x8, slot[m * 64 + lane]);
#pragma unroll
        for (int g = 0; g < 4; ++g) nv[g] = reinterpret_cast<const float4*>(slot + NM * 64)[2 * g + h];
        if (wave_live) {
          // The two chains of a tile, and nothing pending beyond the tile: the sweep is bound by its epilogues (8 radii x
          // 32 instructions against 6 MFMAs), so only the second chain's MFMAs sit in the shadow of the first one's
          // epilogue -- and the tile's reference-side counts are in LDS before the wave moves on, which is what lets
          // the window's accumulators be reduced one window later.
          const f32x16 c0 = frag16(nv);
          chains += TQ;
          f32x16 acc0 = gram_chain<NM>(a, b[0], c0), acc1;
          MrAcc<NR> e;
          uint32_t t;
          if constexpr (INPL) {
            acc1 = gram_chain<NM>(a, b[1], c0);
            keep_alive(c0);
            MrAcc<NR> e1;
            const int k0 = skip_count(acc0), k1 = skip_count(acc1);
            with_skip(min(k0, k1), [&](auto k_c) {   // (one decision for the tile: the radii BOTH chains hold nothing of)
              constexpr int K0 = decltype(k_c)::value;
              mr_begin_k<NR, K0>(e);
              mr_begin_k<NR, K0>(e1);
              if constexpr (K0 > 0) {
                acc0 = mfma16(ones_op, shift_op(NR), acc0);
                acc1 = mfma16(ones_op, shift_op(NR), acc1);
              }
#pragma unroll
              for (int rr = K0; rr < NR; ++rr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) e.bits[rr] = __builtin_amdgcn_alignbit(e.bits[rr], __float_as_uint(acc0[r]), 30);
                if (rr + 1 < NR) acc0 = mfma16(ones_op, shift_op(rr + 1), acc0);
#pragma unroll
                for (int r = 0; r < 16; ++r) e1.bits[rr] = __builtin_amdgcn_alignbit(e1.bits[rr], __float_as_uint(acc1[r]), 30);
                if (rr + 1 < NR) acc1 = mfma16(ones_op, shift_op(rr + 1), acc1);
              }
              shifts += 2u * (uint32_t)(NR - 1 - K0 + (K0 > 0 ? 1 : 0));
            });
            t = (uint32_t)__builtin_amdgcn_readfirstlane(t_raw);
            finish(std::integral_constant<int, 0>{}, e, t);
            finish(std::integral_constant<int, 1>{}, e1, t);
          } else {
            with_skip(skip_count(acc0), [&](auto k_c) {
              constexpr int K0 = decltype(k_c)::value;
              mr_begin_k<NR, K0>(e);
              mr_chain_k<NM, NR, K0>(a, b[1], c0, acc1, acc0, P.dl, e);
            });
            keep_alive(c0);
            t = (uint32_t)__builtin_amdgcn_readfirstlane(t_raw);
            finish(std::integral_constant<int, 0>{}, e, t);
            with_skip(skip_count(acc1), [&](auto k_c) {
              constexpr int K0 = decltype(k_c)::value;
              mr_begin_k<NR, K0>(e);
              mr_epi_k<NR, K0, 0, 16>(acc1, P.dl, e);
            });
            finish(std::integral_constant<int, 1>{}, e, t);
          }
#ifndef DC_MS_ABL_NOCREDIT
          if ((t / (4u * TQ)) != group)   // (not the workgroup's own group)
#else
          if (t == 0xFFFFFFFFu)
#endif
            credit((((i / kMsWin) & 1u) * kMsWin) + (i % kMsWin));
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the fetch beyond the list's end (into slots nobody reads) has landed too
      __syncthreads();   // every add of this round has landed
      {  // the last window
        const uint32_t n_win = (total + kMsWin - 1) / kMsWin, w = n_win - 1u, j = w * kMsWin;
        reduce_window(w, [&](uint32_t k) { return entry_raw(j + k); }, min((uint32_t)kMsWin, total - j), std::false_type{});
      }
      // (the skip test of this wave: worth its compare only where it finds something)
      if (wave_live && total >= 16u && skip_hits * 4u < (uint32_t)TQ * total) skip_thr_now = INFINITY;
      skip_hits = 0;
    }
    __syncthreads();   // lists, ring and accumulators are free for the next round
  }
  if (lane == 0 && chain_counter && wave_live) {
    atomicAdd(chain_counter, (unsigned long long)chains);
    atomicAdd(chain_counter + kMfmaCtrPop, (unsigned long long)chains * NM + shifts);
  }
  flush();

#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const bool live = (livemask[qt] >> lane) & 1;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      const uint32_t total = cnt_q[qt][rr] + (uint32_t)__shfl_xor((int)cnt_q[qt][rr], 32, 64) +
                             fix_tab[rr * (TQ * 32) + qt * 32 + c];
      if (h == 0 && live && rr < n_rad) {
        // the sweep met the self pair (box gap 0: never pruned) and counted it iff d2(i,i) < rad2; the reference
        // starts every population at 1 (:132-134): corrected once, by chunk 0
        uint32_t v = total;
        if (chunk == 0) {
          const float dself = exact_d2(coords, n_cols, jq[qt], jq[qt]);
          v += 1u - ((dself < rad2.v[rr]) ? 1u : 0u);
        }
        if (v != 0u) atomicAdd(&pops_pos[ms_index<NR>((qt0 + (uint32_t)qt) * 32u + (uint32_t)c, rr)], v);
      }
    }
  }
}

// counts [tile][NR][32] by position in the sweep's order -> populations [radius][frame]
template <int NR>
__global__ void pops_by_frame_ms_kernel(const uint32_t* __restrict__ pops_pos, const uint32_t* __restrict__ perm,
                                        uint32_t n_pos, uint32_t n_rows, int n_rad, const uint32_t* __restrict__ hdr,
                                        uint32_t* __restrict__ pops) {
  if (hdr[1] != 0) return;
  const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= n_pos) return;
  const uint32_t f = perm[pos];
  if (f == kInvalidFrame) return;
  for (int rr = 0; rr < n_rad; ++rr) pops[(size_t)rr * n_rows + f] = pops_pos[ms_index<NR>(pos, rr)];
}

// DC_POP_MSYM = 0 keeps the one-sided multi-radius sweep (pop_shared_kernel<NM, 2, NR>; tests, measurements)
inline bool pop_msym_wanted() {
  static const bool off = [] {
    const char* v = getenv("DC_POP_MSYM");
    return v && v[0] == '0';
  }();
  return !off;
}
template <int NM, int NR>
constexpr size_t msym_smem() {
  return (size_t)kMsRing * kTileUnits<NM> * 16 + (size_t)kMsAccSlots * NR * 64 * 8 + sizeof(uint32_t) * 4 * shared_wave_words(kMsTQ, NR);
}
