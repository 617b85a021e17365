// dc_mfma_shared.hpp -- population sweep with the reference operands SHARED THROUGH LDS by the four waves
// of a workgroup (included by dc_mfma_kernels.hpp, inside namespace dc::{anonymous}).
//
// pop_pruned_kernel lets every wave stream its own copy of the surviving reference tiles from L2: NM KB
// per tile for TQ chains.  That is fine while the operand image lives in the caches (C3: 64 MB), but at
// C5 (5M x 30: NM = 6, image 0.96 GB > Infinity Cache) the sweep moved 22 TB through the fabric at
// 7.4 TB/s and was bound by it (profiles/r2_c5_pmc.json).  Here the workgroup is the unit: its 4 x TQ query
// tiles share one survivor list, every surviving reference tile is fetched ONCE per workgroup, straight into
// an eight-slot LDS ring (`global_load_lds_dwordx4`: no staging registers), and read from there by the four
// waves -- a quarter of the L2 traffic per chain, and `ds_read_b128` instead of global loads in the hot loop.
// Wave w fetches the tiles i = w (mod 4) of the survivor sequence.  Every fourth tile:
//     s_waitcnt vmcnt(0) (the wave's own tile of this window has landed)  ->  barrier (all four tiles of the
//     window are visible, and everybody has left the previous window: its slots are free)  ->  issue the
//     loads of the NEXT window
// then four tiles of ds_read + TQ chains + epilogues (as pop_pruned_kernel) without further synchronisation:
// one barrier per four reference tiles, loads four tiles (25 KB per workgroup at NM = 6) ahead of their use.
// Counting, guard band, deferred exact re-check, reference shares (gridDim.y) and segments are those of
// pop_pruned_kernel (single radius, no pair sinks).
constexpr int kShareSub = 128;   // boxes scanned per wave and round
constexpr int kRing = 8;         // LDS slots: two windows of four reference tiles
// 32-bit words of LDS per wave behind the ring: queue (8 B entries), frame ids, exact-path counts per radius
constexpr size_t shared_wave_words(int tq, int nr) { return 2 * kWaveQueue + (size_t)(1 + nr) * tq * 32; }

template <int NM>
constexpr int kTileUnits = NM * 64 + 8;   // 16-byte units of one staged tile: NM fragments + 32 norms

// LDS-DMA: 16 (4) bytes per lane from a per-lane global address straight into LDS at lds + 16 (4) * lane.
// Written as asm on purpose: issued through __builtin_amdgcn_global_load_lds the compiler orders every later LDS
// read of the ring behind the load (s_waitcnt vmcnt(0) right after the issue: the fetching wave then sits out
// the memory latency of the tile it has just asked for, once per window); the loop waits for its DMAs itself,
// once per window, before the barrier that publishes them.  (Operations the compiler does not count only make its
// own vmcnt waits stricter: loads return in order.)
// (M0 carries the LDS address; it is saved and restored, so the compiler's own uses of it are not disturbed.)
__device__ __forceinline__ void lds_dma16(const void* gptr, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void lds_dma4(const void* gptr, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint32_t lds_address(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// ---- epilogue for NR radii per sweep ---------------------------------------------------------------------
// With five and more MFMAs per chain the distances are worth more than the compares: ONE sweep serves up to
// eight radii (C5: eight radii, NM = 6).  Per radius and element: t = acc - delta_r (two elements per
// v_pk_add_f32) and its top two bits -- sign = inside, bit 30 = outside, neither = band (the band is [0, 2) for
// every radius: pick_scale_pop sized the scale for the largest of them) -- into that radius' string (v_alignbit):
// 1.5 instructions per element and radius, one for the first radius.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NR>
using MrAcc = PopAcc<NR>;
template <int NR>
__device__ __forceinline__ void mr_begin(MrAcc<NR>& e) { pop_epi_begin<NR>(e); }
template <int NR, int R0, int R1>   // elements [R0, R1), both even
__device__ __forceinline__ void mr_epi(const f32x16& acc, const PopDeltas<NR>& dl, MrAcc<NR>& e) {
  static_assert(R0 % 2 == 0 && R1 % 2 == 0, "elements are handled in pairs");
  // (source order: the packed subtraction of the NEXT element pair stands in front of the two v_alignbit of the current
  //  one, and the scheduler keeps it there -- a v_alignbit that reads the result of the v_pk_add_f32 right before it costs
  //  an s_nop, one per pair and radius: a hundred per reference tile of the eight-radius sweeps, round 6)
#pragma unroll
  for (int rr = 0; rr < NR; ++rr) {
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
template <int NM, int NR, int MI = 0>
__device__ __forceinline__ void mr_chain(const s16x8 (&a)[NM], const s16x8 (&b)[NM], const f32x16& c0,
                                         f32x16& acc_new, const f32x16& acc_old, const PopDeltas<NR>& dl,
                                         MrAcc<NR>& e) {
  if constexpr (MI < NM) {
    if constexpr (MI == 0)
      acc_new = mfma16(a[0], b[0], c0);
    else
      acc_new = mfma16(a[MI], b[MI], acc_new);
    mr_epi<NR, 2 * ((8 * MI) / NM), 2 * ((8 * (MI + 1)) / NM)>(acc_old, dl, e);
    mr_chain<NM, NR, MI + 1>(a, b, c0, acc_new, acc_old, dl, e);
  }
}

// evaluates the queued band pairs of a wave, 64 at a time (one per lane), and credits the owners of the
// queries through their LDS counters (fix_tab[radius][query]).  Entry: x = reference position, y = query
// (tile * 32 + column) | radii whose band holds the pair << 8.  Out of line: the hot loop calls it once
// every few hundred chains.
// sym_tiles > 0 (symmetric sweep, dc_mfma_kernels.hpp "symmetric population sweep"): a pair whose reference tile
// lies outside the workgroup's own group of sym_tiles query tiles is evaluated here alone, so the reference frame is
// credited too (pops_pos: [NR][pos_stride] counts by position).
template <int NR>
__device__ __attribute__((noinline)) void pop_wave_flush(const uint2* queue, uint32_t qn,
                                                         const uint32_t* jq_tab, uint32_t* fix_tab,
                                                         uint32_t n_queries, const float* __restrict__ coords,
                                                         const float* __restrict__ coords_r, uint32_t n_cols,
                                                         Rad2 rad2, int lane, uint32_t* __restrict__ pops_pos = nullptr,
                                                         uint32_t pos_stride = 0, uint32_t sym_tiles = 0,
                                                         uint32_t own_group = 0) {
  for (uint32_t k0 = 0; k0 < qn; k0 += 64) {
    if (k0 + lane < qn) {
      const uint2 ent = queue[k0 + lane];
      const uint32_t qidx = ent.y & 0xFFu, flags = ent.y >> 8;
      const float d2c = dist2_canon_rows(coords + (size_t)jq_tab[qidx] * n_cols, coords_r + (size_t)ent.x * n_cols, (int)n_cols);
      const bool both = sym_tiles != 0u && (ent.x >> 5) / sym_tiles != own_group;
#pragma unroll
      for (int rr = 0; rr < NR; ++rr)
        if (((flags >> rr) & 1u) && d2c < rad2.v[rr]) {
          atomicAdd(&fix_tab[rr * n_queries + qidx], 1u);
          if (both) atomicAdd(&pops_pos[(size_t)rr * pos_stride + ent.x], 1u);
        }
    }
  }
}

// SYM: every unordered pair of query GROUPS (here: the 4 * TQ tiles of a workgroup) once, both frames credited --
// pop_pruned_kernel's symmetric form; the strings of the NR radii on the pending reference tile are kept per wave.
template <int NM, int TQ, int NR, bool SYM = false>
__global__ __launch_bounds__(256, 2) void pop_shared_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols,
    const uint4* __restrict__ img_r, const float* __restrict__ norms_r,
    const float4* __restrict__ box_r, const float* __restrict__ coords_r, uint32_t T,
    const uint4* __restrict__ img_q, const float* __restrict__ norms_q,
    const uint32_t* __restrict__ perm_q, const float4* __restrict__ box_q, uint32_t n_q, QSeg q_seg,
    const uint32_t* __restrict__ hdr, unsigned long long* __restrict__ chain_counter, Rad2 rad2, int n_rad,
    uint32_t* __restrict__ pops, int q_in_ref_order, CompView CV, uint32_t* __restrict__ pops_pos = nullptr) {
  static_assert(TQ % 2 == 0, "accumulator ping-pong needs an even number of query tiles");
  static_assert(NR >= 1 && NR <= 8 && TQ * 32 <= 256, "queue entries: 8 radius flags, 8 bits of query index");
  __shared__ uint32_t lists[4][kShareSub];
  __shared__ uint32_t credit_stage[4][8];
  __shared__ uint32_t list_cnt[4];
  __shared__ float4 wave_box[4];
  // dynamic LDS: operand ring [kRing][kTileUnits] x 16 B, then per wave the compact queue of deferred exact
  // evaluations [kWaveQueue] x 8 B, the frame ids of its queries [TQ*32] and their exact-path counts [NR][TQ*32].  (The
  // query rows of the exact path stay in global memory here: staged in LDS like pop_pruned_kernel does they
  // would take 61 KB at D = 30 with four tiles per wave and leave one workgroup per CU.)
  extern __shared__ __attribute__((aligned(16))) float shared_dyn[];
  if (hdr[1] != 0) return;   // flagged data: the gated direct kernel runs instead
  constexpr int kUnits = kTileUnits<NM>;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31, wib = tid >> 6;
  const uint32_t TQT = (n_q + 31) / 32;
  const uint32_t blk_unit = xcd_block(seg_groups((TQT + 4u * TQ - 1u) / (4u * TQ), q_seg));
  if (blk_unit == 0xFFFFFFFFu) return;   // (pad block of the grid: the whole workgroup leaves)
  const uint32_t group = seg_group(blk_unit, q_seg);
  const uint32_t chunk = blockIdx.y, n_chunks = gridDim.y;
  if (group * (4u * TQ) >= TQT) return;   // whole workgroup leaves
  const uint32_t qt0 = (group * 4u + (uint32_t)wib) * TQ;
  const bool wave_live = qt0 < TQT;       // (a wave without tiles keeps loading and meeting the barriers)
  uint4* ring = reinterpret_cast<uint4*>(shared_dyn);
  uint32_t* wave_lds = reinterpret_cast<uint32_t*>(shared_dyn + kRing * kUnits * 4) +
                       (size_t)wib * shared_wave_words(TQ, NR);
  uint2* queue = reinterpret_cast<uint2*>(wave_lds);   // (reference position, query | radius flags << 8)
  uint32_t* jq_tab = wave_lds + 2 * kWaveQueue;        // frame id of query (qt, c)
  uint32_t* fix_tab = jq_tab + TQ * 32;                // [NR][TQ*32]: band pairs the exact path found inside
  uint32_t qn = 0;                                     // queued entries (wave-uniform)

  const PopSetup<NR> P = pop_setup<NR>(hdr, rad2, n_cols);
  float r2max = rad2.v[0];
#pragma unroll
  for (int rr = 1; rr < NR; ++rr) r2max = fmaxf(r2max, rad2.v[rr]);
  const float far2 = r2max * 1.0001f;   // boxes at least this far apart (squared) hold no pair inside

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
      // row of the query for the exact path: its place in the reference order when the queries are the reference
      // rows (their original coordinates then sit next to each other in coords_r: the wave's TQ * 32 rows stay in
      // the caches from one flush to the next), else its frame id in the caller's matrix
      jq_tab[qt * 32 + c] = q_in_ref_order ? pos : jq[qt];
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

  // Deferred exact re-check, wave-wide.  A band pair is not evaluated where it appears (one or two lanes busy
  // for a dependent memory latency) and not kept per lane either (pop_pruned_kernel's per-lane queues flush
  // when ONE lane is full, with most lanes idle): the wave appends (reference position, query) to ONE compact
  // list and evaluates 64 entries at a time, one per lane -- two independent row fetches and one canonical
  // distance per lane and batch.  The owner of the query is credited through an LDS counter.  At C5's radii
  // (r^2 near the typical pair distance: a band pair every five chains) the per-lane scheme cost 30 - 40 %
  // of the sweep.
  // SYM: the groups on their circle, the strings of the chains on the pending reference tile (per radius and query
  // tile), whether that tile is credited (it lies outside the workgroup's own group), stride of a radius in pops_pos
  const uint32_t n_groups = (TQT + 4u * TQ - 1u) / (4u * TQ);
  const uint32_t pos_stride = 32u * T;
  uint32_t sb[NR][TQ];
  bool symB = false;
  const uint32_t my_byte = ref_credit_byte(lane);
  auto credit = [&](uint32_t t) {
#ifdef DC_ABL_SHARED_NOCREDIT
    if (t != 0xFFFFFFFFu) return;
#endif
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
      if (rr < n_rad) ref_credit<TQ>(sb[rr], t, CV.n_pos, pops_pos + (size_t)rr * pos_stride, credit_stage[wib], my_byte, lane);
  };
  const float* q_rows = q_in_ref_order ? coords_r : coords;
  auto flush = [&]() {
    if constexpr (SYM)
      pop_wave_flush<NR>(queue, qn, jq_tab, fix_tab, TQ * 32, q_rows, coords_r, n_cols, rad2, lane, pops_pos, pos_stride,
                         4u * TQ, group);
    else
      pop_wave_flush<NR>(queue, qn, jq_tab, fix_tab, TQ * 32, q_rows, coords_r, n_cols, rad2, lane);
    qn = 0;
  };
  // the rest of an epilogue: counts, band test, parking of the band pairs (positions fit the queue entries:
  // the launcher sends larger problems to pop_pruned_kernel)
  auto finish = [&](const f32x16& acc, auto qi_c, const MrAcc<NR>& e, uint32_t t) {
    constexpr int qi = decltype(qi_c)::value;
    uint32_t decided = 0xFFFFFFFFu;   // bit 31 - 2 r: element r is inside or outside for EVERY radius
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      cnt_q[qi][rr] += __builtin_popcount(inside_of(e.bits[rr]));
      decided &= e.bits[rr] | (e.bits[rr] << 1);
      if constexpr (SYM) sb[rr][qi] = e.bits[rr];
    }
    uint32_t m = ~decided & kSignBits;
    (void)acc;
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
      if (qn >= 64u) flush();
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
        if constexpr (SYM) {
          // the workgroup's own group, or a group at most half the circle ahead (exactly half: the lower index)
          const uint32_t gt = t / (4u * TQ);
          const uint32_t ahead = (gt >= group) ? gt - group : gt + n_groups - group;
          ok = ok & ((2u * ahead < n_groups) | ((2u * ahead == n_groups) & (group < gt)));
        }
      }
      const uint64_t m = __builtin_amdgcn_ballot_w64(ok);
      if (ok) lists[wib][cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = t;
      cnt += (uint32_t)__builtin_popcountll(m);
    }
    if (lane == 0) list_cnt[wib] = cnt;
    __syncthreads();
    // (wave-uniform values held in scalar registers, the survivor list addressed by scalar arithmetic: as compares and
    //  selects on what an LDS read returned this was four or five branches per call -- dc_mfma_msym.hpp, round 6)
    const uint32_t o1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)list_cnt[0]);
    const uint32_t o2 = o1 + (uint32_t)__builtin_amdgcn_readfirstlane((int)list_cnt[1]);
    const uint32_t o3 = o2 + (uint32_t)__builtin_amdgcn_readfirstlane((int)list_cnt[2]);
    const uint32_t total = o3 + (uint32_t)__builtin_amdgcn_readfirstlane((int)list_cnt[3]);
    if (total != 0) {
      const uint32_t d21 = o2 - o1, d32 = o3 - o2;
      auto entry = [&](uint32_t i) {
        i = min(i, total - 1u);
        const uint32_t g1 = 0u - (uint32_t)(i >= o1), g2 = 0u - (uint32_t)(i >= o2), g3 = 0u - (uint32_t)(i >= o3);   // 0 / ~0
        const uint32_t first = (o1 & g1) + (d21 & g2) + (d32 & g3);             // o_w: where the list of wave w starts in the round
        const uint32_t idx = ((uint32_t)kShareSub & g1) + ((uint32_t)kShareSub & g2) + ((uint32_t)kShareSub & g3) + (i - first);
        return (uint32_t)__builtin_amdgcn_readfirstlane((&lists[0][0])[idx]);
      };
      // reference tile t -> ring slot, by this wave alone: NM fragments of 1 KB (lane l lands at +16 l) and the
      // 32 row norms (lanes 0..7)
      auto fetch = [&](uint32_t t, uint32_t slot_id) {
        const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(ring + slot_id * kUnits));
        const uint4* src = img_r + (size_t)t * (NM * 64) + lane;
#pragma unroll
        for (int m = 0; m < NM; ++m) lds_dma16(src + m * 64, dst + (uint32_t)m * 1024u);
        if (lane < 8) lds_dma16(reinterpret_cast<const uint4*>(norms_r + (size_t)t * 32) + lane, dst + (uint32_t)NM * 1024u);
      };
      f32x16 accA, accB;   // accB: the chain whose epilogue is pending (+inf everywhere: contributes nothing)
#pragma unroll
      for (int r = 0; r < 16; ++r) accB[r] = INFINITY;
      uint32_t tB = 0;
      if ((uint32_t)wib < total) fetch(entry((uint32_t)wib), (uint32_t)wib);
      for (uint32_t i = 0; i < total; ++i) {
        if ((i & 3u) == 0) {
          __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's tile of the window starting at i
          __syncthreads();
          const uint32_t nxt = i + 4 + (uint32_t)wib;
          if (nxt < total) fetch(entry(nxt), nxt % kRing);
        }
        const uint32_t t = entry(i);
        const uint4* slot = ring + (i % kRing) * kUnits;
        s16x8 a[NM];
        float4 nv[4];
#pragma unroll
        for (int m = 0; m < NM; ++m) a[m] = __builtin_bit_cast(s16x8, slot[m * 64 + lane]);
#pragma unroll
        for (int g = 0; g < 4; ++g) nv[g] = reinterpret_cast<const float4*>(slot + NM * 64)[2 * g + h];
        if (wave_live) {
          const f32x16 c0 = frag16(nv);
          chains += TQ;
          constexpr_for_pairs<TQ>([&](auto qt_c) {
            constexpr int qt = decltype(qt_c)::value;
            constexpr int qb = (qt == 0) ? TQ - 1 : qt - 1;
            MrAcc<NR> e;
            mr_begin<NR>(e);
            mr_chain<NM, NR>(a, b[qt], c0, accA, accB, P.dl, e);
            finish(accB, std::integral_constant<int, qb>{}, e, (qt == 0) ? tB : t);
            if constexpr (SYM && qt == 0)   // the strings of tile tB are complete now
              if (symB) credit(tB);
            mr_begin<NR>(e);
            mr_chain<NM, NR>(a, b[qt + 1], c0, accB, accA, P.dl, e);
            finish(accA, std::integral_constant<int, qt>{}, e, t);
          });
          keep_alive(c0);
          tB = t;
          symB = (t / (4u * TQ)) != group;
        }
      }
      if (wave_live) {  // drain: epilogue of the last pending chain of this round
        MrAcc<NR> e;
        mr_begin<NR>(e);
        mr_epi<NR, 0, 16>(accB, P.dl, e);
        finish(accB, std::integral_constant<int, TQ - 1>{}, e, tB);
        if constexpr (SYM) {
          if (symB) credit(tB);
          symB = false;
        }
      }
    }
    __syncthreads();   // lists and ring are free for the next round
  }
  if (lane == 0 && chain_counter && wave_live) {
    atomicAdd(chain_counter, (unsigned long long)chains);
    atomicAdd(chain_counter + kMfmaCtrPop, (unsigned long long)chains * NM);
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
        if constexpr (SYM)
          atomicAdd(&pops_pos[(size_t)rr * pos_stride + (qt0 + (uint32_t)qt) * 32u + (uint32_t)c], v);   // (position = place in the order)
        else if (n_chunks == 1)
          pops[(size_t)rr * n_rows + jq[qt]] = v;
        else
          atomicAdd(&pops[(size_t)rr * n_rows + jq[qt]], v);   // pops was zero-filled by the caller
      }
    }
  }
}

// Which calls take the shared-operand sweep.  Measured (pop, one MI355X, per-wave streams -> shared), ONE radius:
// 5M x 30 one segment of eight 202 -> 166 ms (r = 0.35), 286 -> 204 ms (r = 0.6), all rows 2785 -> 1822 ms; 1M x 30
// 73.4 -> 59.6 ms; 1M x 40 97.3 -> 87.8 ms; 3M x 24 497 -> 482 ms; but 300k x 26 5.81 -> 5.94 ms, 2M x 20 181 -> 190 ms,
// 4M x 12 536 -> 563 ms, 1M x 10 23.7 -> 28.4 ms: with few MFMAs per chain a single-radius sweep is bound by its epilogue,
// not by the operand stream, and the workgroup-wide survivor list prunes less than a wave's own -- five or more MFMAs
// per chain and an operand image beyond the caches' comfortable reach.  SEVERAL radii in one call (one sweep for up to
// eight of them against one sweep per radius): 5M x 30 segment, 8 radii 1449 -> 845 ms; 300k x 26, 8 radii 39.1 -> 24.7 ms;
// 2M x 20, 8 radii 1390 -> 1068 ms; 1M x 16, 8 radii 300 -> 222 ms, 4 radii 153 -> 127 ms; 600k x 12, 8 radii 90 -> 77 ms
// (the distances are computed once; by the issue model the gain is (4*26 + 24*NM) R against 4*33 R + 24*NM cycles):
// three or more MFMAs per chain and three or more radii, or five or more MFMAs and two radii.
// DC_POP_SHARED = 0 / 1 forces it off / on (tests, measurements).
template <int NM, int NR>
constexpr int tq_shared_for = (NM <= 6 && NR == 1) ? 4 : 2;
inline int nr_shared_of(int n_rad) { return n_rad <= 1 ? 1 : (n_rad <= 4 ? 4 : 8); }   // radii per sweep instance
inline int tq_shared_of(uint32_t n_cols, int n_rad) {   // = tq_shared_for<NM, NR>
  return (nm_for((int)n_cols) <= 6 && nr_shared_of(n_rad) == 1) ? 4 : 2;
}
inline bool pop_shared_wanted(uint32_t n_rows, uint32_t n_cols, int n_rad) {
  static const int forced = [] {
    const char* v = getenv("DC_POP_SHARED");
    return (v && v[0]) ? atoi(v) : -1;
  }();
  const int nm = nm_for((int)n_cols);
  if (n_rows + 32u * kPadTiles > kPopQueueMaxRows || nm > 8) return false;
  if (forced >= 0) return forced != 0;
  const size_t image = (size_t)((n_rows + 31) / 32) * (size_t)nm * 1024;
  if (nm >= 5 && image > ((size_t)96 << 20)) return true;
  return n_rows >= 50000u && ((nm >= 3 && n_rad >= 3) || (nm >= 5 && n_rad >= 2));
}
