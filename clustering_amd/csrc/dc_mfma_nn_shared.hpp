// dc_mfma_nn_shared.hpp -- neighbour sweep with the reference operands shared through LDS by the four waves of a
// workgroup (included by dc_mfma_kernels.hpp, inside namespace dc::{anonymous}; generated from nn_pruned_kernel's
// logic, see there for the rings, the free-energy classes, the candidate path).  What changes against
// nn_pruned_kernel is what changed from pop_pruned_kernel to pop_shared_kernel (dc_mfma_shared.hpp): the workgroup's
// 4 x TQ query tiles share ONE ring schedule (the largest open incumbent of any of its queries sets the next
// radius) and ONE survivor list per round; every surviving reference tile is fetched once per workgroup with
// global_load_lds into an eight-slot ring (fragments, row norms, free-energy range) and read from there by the
// four waves; one barrier per four tiles.  Motivation as there: at 5M x 30 the per-wave streams of the neighbour
// sweep moved 0.5 - 1.3 TB per segment sweep at up to 5.7 TB/s (profiles/r2_c5_*_pmc.json).
constexpr int kNnRingExtra = 9;   // 16-byte units behind the NM fragments of a slot: 8 of row norms, 1 of free-energy range

// nn_wave_flush with the query rows in global memory (frame ids from an LDS table)
__device__ __attribute__((noinline)) void nn_wave_flush_rows(const uint2* queue, uint32_t qn,
                                                             const float* __restrict__ coords,
                                                             const uint32_t* jq_tab,
                                                             unsigned long long* best, uint32_t n_queries,
                                                             const float* __restrict__ coords_c,
                                                             const uint32_t* __restrict__ perm, uint32_t n_cols,
                                                             int lane) {
  for (uint32_t k0 = 0; k0 < qn; k0 += 64) {
    if (k0 + lane < qn) {
      const uint2 ent = queue[k0 + lane];
      const uint32_t pos = ent.x & kQueuePosMask, qidx = ent.y;
      const uint32_t j = perm[pos];
      const float d2c = dist2_canon_rows(coords + (size_t)jq_tab[qidx] * n_cols, coords_c + (size_t)pos * n_cols,
                                       (int)n_cols);
      const unsigned long long key = ((unsigned long long)__float_as_uint(d2c) << 32) | j;
      if ((ent.x >> 30) & 1u) atomicMin(&best[qidx], key);
      if ((ent.x >> 31) & 1u) atomicMin(&best[n_queries + qidx], key);
    }
  }
}

// which shapes take the shared-operand neighbour sweep (DC_NN_SHARED = 0 / 1 forces it off / on)
inline bool nn_shared_wanted(uint32_t n_rows, uint32_t n_cols) {
  static const int forced = [] {
    const char* v = getenv("DC_NN_SHARED");
    return (v && v[0]) ? atoi(v) : -1;
  }();
  const int nm = nm_for((int)n_cols);
  if (nm > 8 || n_rows >= (1u << 30)) return false;
  if (forced >= 0) return forced != 0;
  const size_t image = (size_t)((n_rows + 31) / 32) * (size_t)nm * 1024;
  return nm >= 5 && image > ((size_t)96 << 20);
}

// NB: the MFMAs of a chain that run before the early-out test (dc_mfma_kernels.hpp "early-out of the pruned
// neighbour sweep"): nn_coarse_for(n_cols); the fragments behind them are read from the ring only by the chains
// that go on.
// Only the NB coarse fragments of a tile go through the ring: 97 % of the chains stop after them, and the sweep ran at the
// memory side's pace (C5: 0.56 TB per launch at 5.7 TB/s).  A chain that goes on loads its remaining fragments straight
// from the image and waits for them -- one rank of C5 100 -> 87 ms.  (DC_NNS_GLOBAL_REST=0: the whole tile through the
// ring, as before; measurements.)
#ifndef DC_NNS_GLOBAL_REST
#define DC_NNS_GLOBAL_REST 1
#endif
#if DC_NNS_GLOBAL_REST
#define DC_NNS_RING_FRAGS NB
#else
#define DC_NNS_RING_FRAGS NM
#endif
template <int NM, int TQ, int NB>
__global__ __launch_bounds__(256, 2) void nn_shared_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols,
    const float* __restrict__ fe, const uint4* __restrict__ img_r,
    const float* __restrict__ norms_r, const uint32_t* __restrict__ perm_r,
    const float4* __restrict__ box_r, const float4* __restrict__ box_t,
    const float2* __restrict__ ferange_r,
    const float* __restrict__ fe_c, const float* __restrict__ coords_c,
    const uint32_t* __restrict__ invpos_r, uint32_t T,
    const uint4* __restrict__ img_q, const float* __restrict__ norms_q,
    const uint32_t* __restrict__ perm_q, const float4* __restrict__ box_q, uint32_t n_q, QSeg q_seg,
    int full_range, float cell2,
    const uint32_t* __restrict__ hdr, unsigned long long* __restrict__ chain_counter,
    unsigned long long* __restrict__ merge64, uint32_t* __restrict__ nn_idx,
    float* __restrict__ nn_d2, uint32_t* __restrict__ hd_idx, float* __restrict__ hd_d2, CompView CV) {
  __shared__ uint32_t lists[4][kShareSub];
  __shared__ uint32_t list_cnt[4];
  __shared__ float4 wave_box[4];
  __shared__ float red_need[4];
  __shared__ uint32_t red_flags[4];
  // dynamic LDS: operand ring [kRing][kNnUnits] x 16 B (NM fragments, 32 row norms, the tile's free-energy range),
  // then per wave the candidate list, the packed exact incumbents (as nn_pruned_kernel) and the frame ids of its
  // queries (the query rows of the exact path stay in global memory: in LDS they would leave one workgroup per CU)
  extern __shared__ __attribute__((aligned(16))) float qrows_all[];
  if (hdr[1] != 0) return;
  constexpr int kUnits = NM * 64 + kNnRingExtra;
  const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
  const int wib = threadIdx.x >> 6;
  const uint32_t TQT = (n_q + 31) / 32;
  const uint32_t blk_unit = xcd_block(seg_groups((TQT + 4u * TQ - 1u) / (4u * TQ), q_seg));
  if (blk_unit == 0xFFFFFFFFu) return;   // (pad block of the grid: the whole workgroup leaves)
  const uint32_t group = seg_group(blk_unit, q_seg);
  const uint32_t chunk = blockIdx.y, n_chunks = gridDim.y;   // reference tiles dealt round-robin
  if (group * (4u * TQ) >= TQT) return;   // whole workgroup leaves
  const uint32_t qt0 = (group * 4u + (uint32_t)wib) * TQ;   // (a wave without tiles keeps meeting the barriers)
  const bool wave_live = qt0 < TQT;
  uint4* ring = reinterpret_cast<uint4*>(qrows_all);
  uint32_t* queues = reinterpret_cast<uint32_t*>(qrows_all + (size_t)kRing * kUnits * 4) +
                     (size_t)wib * (TQ * kQueueCap * 64 + TQ * 32);
  uint32_t* jq_tab = queues + TQ * kQueueCap * 64;   // frame id of query (qt, c)
  // the wave's LDS behind the query rows (TQ * kQueueCap * 64 words): the compact candidate list (kWaveQueue
  // entries of 8 B) and the packed exact incumbents [2][TQ*32] of 8 B
  static_assert(TQ * kQueueCap * 64 >= 2 * kWaveQueue + 4 * TQ * 32, "candidate list + incumbents fit the queue region");
  uint2* cand = reinterpret_cast<uint2*>(queues);
  unsigned long long* best64 = reinterpret_cast<unsigned long long*>(queues + 2 * kWaveQueue);
  uint32_t qn = 0;   // queued candidates (wave-uniform)

  // (scaled units, like the accumulators and the running minima taken from them)
  const Scale sc = load_scale(hdr);   // (the neighbour scale: scale_kernel ran before the images were built)
  // (reference norms folded into the operand image, query norm outside the accumulator -- dc_mfma_kernels.hpp
  //  "reference norms folded": q[].m_nn / m_hd in d2 units, q[].bn / bh in the accumulators' units)
  const GuardBand gb = guard_band(__uint_as_float(hdr[kHdrMused]) * sc.s2, 0.0f, (int)n_cols, sc, true);   // (the extent the scale was chosen for)
  const float skipb = nn_skip_bound(__uint_as_float(hdr[kHdrMused]) * sc.s2);
  (void)norms_r;
  (void)cell2;   // (the first ring's floor: the cell edge of the query's own component, set below)

  s16x8 b[TQ][NM];
  NnPQ q[TQ];
  float cq[TQ];   // |x'|^2 of the lane's query (scaled units)
  uint32_t jq[TQ];
  uint64_t livemask[TQ];
  float4 qbox[TQ];
  float g_nn[TQ], g_hd[TQ];   // exact incumbents published by other reference chunks (FLT_MAX: none)
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
    load_query_folded<NM>(img_q, tl, lane, h, live, sc, b[qt]);
    cq[qt] = live ? norms_q[tl * 32 + c] : 0.0f;
    q[qt].feq = live ? fe[jq[qt]] : -INFINITY;
    q[qt].spos = live ? (full_range ? pos : invpos_r[jq[qt]]) : 0xFFFFFFFFu;
    if (h == 0) jq_tab[qt * 32 + c] = jq[qt];
    q[qt].m_nn = live ? INFINITY : -INFINITY;   // idle lanes can never trigger the exact path
    q[qt].m_hd = live ? INFINITY : -INFINITY;
    g_nn[qt] = FLT_MAX;
    g_hd[qt] = FLT_MAX;
    if (n_chunks > 1 && live) {
      // what the waves of other reference chunks have already published for this query: an exact
      // upper bound.  Only candidates that can still beat (or tie) it need to be looked at, i.e.
      // MFMA values below d2 + eps(d2); the band test adds its usual margin on top.
      g_nn[qt] = __uint_as_float((uint32_t)(merge64[jq[qt]] >> 32));
      g_hd[qt] = __uint_as_float((uint32_t)(merge64[(size_t)n_rows + jq[qt]] >> 32));
      const float s_nn = g_nn[qt] * sc.s2, s_hd = g_hd[qt] * sc.s2;   // (exact d2 -> scaled units)
      if (g_nn[qt] < FLT_MAX) q[qt].m_nn = s_nn + (gb.e0 + gb.kappa * s_nn);
      if (g_hd[qt] < FLT_MAX) q[qt].m_hd = s_hd + (gb.e0 + gb.kappa * s_hd);
    }
    q[qt].m_nn = fminf(q[qt].m_nn, q[qt].m_hd);   // (two reads of merge64 a moment apart: keep m_nn <= m_hd)
    q[qt].bn = nn_prime(nn_band(gb, q[qt].m_nn), cq[qt]);
    q[qt].bh = nn_prime(nn_band(gb, q[qt].m_hd), cq[qt]);
    q[qt].bd_nn = FLT_MAX;
    q[qt].bd_hd = FLT_MAX;
    q[qt].bj_nn = n_rows + 1;
    q[qt].bj_hd = n_rows + 1;
    qbox[qt] = (tile < TQT) ? box_q[tile] : make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
    gbox.x = fminf(gbox.x, qbox[qt].x);
    gbox.y = fmaxf(gbox.y, qbox[qt].y);
    gbox.z = fminf(gbox.z, qbox[qt].z);
    gbox.w = fmaxf(gbox.w, qbox[qt].w);
  }
  if (lane == 0) wave_box[wib] = gbox;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 4; ++w) {   // the workgroup's box: one ring schedule and one survivor list for all four waves
    const float4 wb = wave_box[w];
    gbox.x = fminf(gbox.x, wb.x);
    gbox.y = fmaxf(gbox.y, wb.y);
    gbox.z = fminf(gbox.z, wb.z);
    gbox.w = fmaxf(gbox.w, wb.w);
  }
  // Seeds: the frames next to each query in the sweep's order (same cell, neighbouring free energy; the
  // ones before it have a lower free energy) are evaluated exactly before the first ring.  They are
  // ordinary candidates; what they buy is finite running minima from the start -- without them the first
  // tiles of a sweep park every element, and query groups in sparse regions (wide boxes, long first ring)
  // spent a microsecond per chain in the candidate path.
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // (query rows: written by the h = 0 lanes)
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const bool live = (livemask[qt] >> lane) & 1;
    NnPQ& Q = q[qt];
    // (only the first reference share: the later ones start from what the earlier ones published)
    if (live && chunk == 0) {
      const float* qrow = coords + (size_t)jq[qt] * n_cols;
      for (int k = 1; k <= kSeedNeighbours; ++k) {
        const long long p2 = (long long)Q.spos + (h ? -k : k);
        if (p2 >= 0 && p2 < (long long)CV.n_pos && perm_r[p2] != kInvalidFrame) {
          const float d2c = dist2_canon_rt(qrow, 1, coords_c + (size_t)p2 * n_cols, 1, (int)n_cols);
          const uint32_t j = perm_r[p2];
          lexi_update(true, Q.bd_nn, Q.bj_nn, d2c, j, n_rows);
          lexi_update(fe_c[p2] < Q.feq, Q.bd_hd, Q.bj_hd, d2c, j, n_rows);
        }
      }
      // (the two half-wave lanes looked at the frames after / before the query: merge, then both hold the result)
      {
        float od = __shfl_xor(Q.bd_nn, 32, 64);
        uint32_t oj = (uint32_t)__shfl_xor((int)Q.bj_nn, 32, 64);
        lexi_update(oj <= n_rows, Q.bd_nn, Q.bj_nn, od, oj, n_rows);
        od = __shfl_xor(Q.bd_hd, 32, 64);
        oj = (uint32_t)__shfl_xor((int)Q.bj_hd, 32, 64);
        lexi_update(oj <= n_rows, Q.bd_hd, Q.bj_hd, od, oj, n_rows);
      }
      const float s_nn = Q.bd_nn * sc.s2, s_hd = Q.bd_hd * sc.s2;
      if (Q.bd_nn < FLT_MAX) Q.m_nn = fminf(Q.m_nn, s_nn + (gb.e0 + gb.kappa * s_nn));
      if (Q.bd_hd < FLT_MAX) Q.m_hd = fminf(Q.m_hd, s_hd + (gb.e0 + gb.kappa * s_hd));
      Q.bn = nn_prime(nn_band(gb, Q.m_nn), cq[qt]);
      Q.bh = nn_prime(nn_band(gb, Q.m_hd), cq[qt]);
      // published at once: the other shares of this group start while this wave is still sweeping
      if (n_chunks > 1) {
        if (Q.bd_nn < FLT_MAX)
          atomicMin(&merge64[jq[qt]], ((unsigned long long)__float_as_uint(Q.bd_nn) << 32) | Q.bj_nn);
        if (Q.bd_hd < FLT_MAX)
          atomicMin(&merge64[(size_t)n_rows + jq[qt]],
                    ((unsigned long long)__float_as_uint(Q.bd_hd) << 32) | Q.bj_hd);
      }
    }
  }
  // the exact incumbents of the wave's queries, as order-preserving words in LDS (see nn_wave_flush)
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt)
    if (h == 0) {
      best64[qt * 32 + c] = ((unsigned long long)__float_as_uint(q[qt].bd_nn) << 32) | q[qt].bj_nn;
      best64[TQ * 32 + qt * 32 + c] = ((unsigned long long)__float_as_uint(q[qt].bd_hd) << 32) | q[qt].bj_hd;
    }
  // lowest free energy of the whole data set (header word 12, ordered-integer key, written by the
  // ordering pass): a query at that level has no lower-FE neighbour
  const float fe_floor = fkey_inv(~hdr[12]);

  // evaluate and empty the candidate list (64 candidates at a time, one per lane)
  auto flush = [&]() {
    nn_wave_flush_rows(cand, qn, coords, jq_tab, best64, TQ * 32, coords_c, perm_r, n_cols, lane);
    qn = 0;
  };
  // the registers' copy of the exact incumbents (both half-wave lanes of a query hold the same)
  auto reload = [&]() {
#pragma unroll
    for (int qt = 0; qt < TQ; ++qt) {
      const unsigned long long a = best64[qt * 32 + c], b2 = best64[TQ * 32 + qt * 32 + c];
      q[qt].bd_nn = __uint_as_float((uint32_t)(a >> 32));
      q[qt].bj_nn = (uint32_t)a;
      q[qt].bd_hd = __uint_as_float((uint32_t)(b2 >> 32));
      q[qt].bj_hd = (uint32_t)b2;
    }
  };

  uint32_t chains = 0, visited = 0;
  uint32_t chains_on = 0;   // chains that went on behind the early-out test
  // this wave's share of the reference tiles: t = chunk + u * n_chunks, u = 0 .. U-1 (round-robin, so
  // every share sees every region; the scans only touch their own boxes)
  // (the tiles of the group's own COMPONENT only -- dc_mfma_kernels.hpp "components": what lies in other components is
  //  looked at afterwards, exactly, for the few queries whose neighbours may be there: nn_cross_kernel)
  const uint32_t my_comp = CV.tile_comp_q[group * (4u * TQ)];
  const uint32_t t_lo = CV.range_r[2 * my_comp], t_hi = min(CV.range_r[2 * my_comp + 1], T);
  const uint32_t u_lo = (t_lo > chunk) ? (t_lo - chunk + n_chunks - 1) / n_chunks : 0u;
  const uint32_t U = max((t_hi > chunk) ? (t_hi - chunk + n_chunks - 1) / n_chunks : 0u, u_lo);
  {
    const float cl = __uint_as_float(CV.comp[kCompFine + 4 * min(my_comp, (uint32_t)kMaxComp - 1u) + 2]);
    cell2 = cl * cl;
  }
  const uint32_t U_stride = (T + n_chunks - 1) / n_chunks;   // boxes of a share in box_t
  const float dgx = gbox.y - gbox.x, dgy = gbox.w - gbox.z;
  float r2_lo = -1.0f;                                     // rings: r2_lo <= gap2 < r2_hi
  float r2_hi = fmaxf(dgx * dgx + dgy * dgy, cell2);
  if (!(r2_hi > 0.0f)) r2_hi = FLT_MIN;
  for (;;) {
    for (uint32_t base = u_lo; base < U; base += 4 * kShareSub) {
      // ---- scan: every wave tests its quarter of the round's boxes against the workgroup's box and ring
      uint32_t cnt = 0;
      auto tile_of = [&](uint32_t u) { return chunk + u * n_chunks; };
      const float4* box_s = box_t + (size_t)chunk * U_stride;
#pragma unroll
      for (int k = 0; k < kShareSub; k += 64) {
        const uint32_t u = base + (uint32_t)wib * kShareSub + k + lane;
        bool ok = false;
        if (u < U) {
          const float g2 = box_gap2(gbox, box_s[u]);
          ok = (g2 < r2_hi) & (g2 >= r2_lo);
        }
        const uint64_t m = __builtin_amdgcn_ballot_w64(ok);
        if (ok) lists[wib][cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = tile_of(u);
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
      if (total == 0) {
        __syncthreads();   // (list_cnt is rewritten by the next round)
        continue;
      }
      visited += total;
      const uint32_t d21 = o2 - o1, d32 = o3 - o2;
      auto entry = [&](uint32_t i) {
        i = min(i, total - 1u);
        const uint32_t g1 = 0u - (uint32_t)(i >= o1), g2 = 0u - (uint32_t)(i >= o2), g3 = 0u - (uint32_t)(i >= o3);   // 0 / ~0
        const uint32_t first = (o1 & g1) + (d21 & g2) + (d32 & g3);             // o_w: where the list of wave w starts in the round
        const uint32_t idx = ((uint32_t)kShareSub & g1) + ((uint32_t)kShareSub & g2) + ((uint32_t)kShareSub & g3) + (i - first);
        return (uint32_t)__builtin_amdgcn_readfirstlane((&lists[0][0])[idx]);
      };
      // reference tile t -> ring slot, by this wave alone (see pop_shared_kernel): the coarse fragments of 1 KB each (the
      // row norms ride in their constant slots) and the tile's free-energy range (two dwords)
      auto fetch = [&](uint32_t t, uint32_t slot_id) {
        const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(ring + slot_id * kUnits));
        const uint4* src = img_r + (size_t)t * (NM * 64) + lane;
#pragma unroll
        for (int m = 0; m < DC_NNS_RING_FRAGS; ++m) lds_dma16(src + m * 64, dst + (uint32_t)m * 1024u);
        if (lane < 2) lds_dma4(reinterpret_cast<const float*>(ferange_r + t) + lane, dst + (uint32_t)NM * 1024u + 128u);
      };
      // the rest of an epilogue: free-energy classes, band test, parking of the candidates.
      // (t, fr) describe the reference tile the accumulator belongs to.
      auto finish = [&](const f32x16& acc, auto qi_c, float tmin, uint32_t t, float2 fr) __attribute__((always_inline)) {
        constexpr int qi = decltype(qi_c)::value;
        NnPQ& Q = q[qi];
        // Common path: two compares against the cached candidate thresholds.  "Lower free energy" is
        // taken conservatively here (the tile has SOME lower frame => its minimum might be one), and
        // the tile holding the query itself always passes (its own d2 ~ 0): whatever needs the
        // per-element treatment ends up in the rare path.  The running minima can only change there
        // too (a value below the minimum is below its band).  ONE wave-level test: the scalar
        // hand-off (v_cmp -> s_cbranch) is a pipeline bubble at two waves per SIMD.
        // (bh >= bn always -- the minimum over the lower-free-energy frames cannot undercut the minimum over all
        //  frames -- so a tile that has lower frames is tested against bh alone, any other against bn)
        const float thr = (fr.x < Q.feq) ? Q.bh : Q.bn;
        const bool rare = tmin < thr;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare) != 0, 0)) {
          const bool all_lower = fr.y < Q.feq;
          const bool mixed = (fr.x < Q.feq) & !all_lower;
          const bool special = mixed | (t == (Q.spos >> 5));
          float hmin = all_lower ? tmin : INFINITY;
          const bool any_special = __builtin_amdgcn_ballot_w64(special) != 0;
          if (any_special) {
            // masked per-element minima (the tile holds the query itself and/or straddles feq); the
            // free energies of the tile's frames are fetched only here
            float4 fv[4];
            load_frag(fe_c, t, h, fv);
            const f32x16 fef = frag16(fv);
            tmin = INFINITY;
            hmin = INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = (tile_row(t, r, h) != Q.spos) ? acc[r] : INFINITY;
              tmin = fminf(tmin, v);
              hmin = fminf(hmin, (fef[r] < Q.feq) ? v : INFINITY);
            }
          }
          // (the two half-wave lanes of a query see different rows of every tile: what either of them has
          //  found bounds the answer of both, so the running minima are shared whenever they move -- the
          //  records of one sequence over all rows instead of two over half of them each)
          float new_nn = fminf(Q.m_nn, nn_unprime(tmin, cq[qi])), new_hd = fminf(Q.m_hd, nn_unprime(hmin, cq[qi]));
          new_nn = fminf(new_nn, __shfl_xor(new_nn, 32, 64));
          new_hd = fminf(new_hd, __shfl_xor(new_hd, 32, 64));
          const float bn = nn_prime(nn_band(gb, new_nn), cq[qi]), bh = nn_prime(nn_band(gb, new_hd), cq[qi]);
          const bool trig = (tmin < bn) | (hmin < bh);
          if (__builtin_amdgcn_ballot_w64(trig) != 0) {
            // park this tile's candidates (values within the band of the running minima); element r
            // of the accumulator is bit (15 - r) of the masks
            uint32_t mn = 0, mh = 0;
            if (!any_special && t + 1 != T) {
              // plain tile: below-threshold sign strings (idle lanes have thresholds of -inf, pad
              // rows only exist in the last tile)
              uint32_t sn = 0, sh = 0;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                sn = __builtin_amdgcn_alignbit(sn, __float_as_uint(acc[r] - bn), 31);
                sh = __builtin_amdgcn_alignbit(sh, __float_as_uint(acc[r] - bh), 31);
              }
              mn = sn & 0xFFFFu;
              mh = all_lower ? (sh & 0xFFFFu) : 0u;
            } else {
              float4 fv[4];
              load_frag(fe_c, t, h, fv);
              const f32x16 fef = frag16(fv);
              const bool live = (livemask[qi] >> lane) & 1;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const uint32_t pos = tile_row(t, r, h);
                const bool other = live & (pos != Q.spos) & (pos < CV.n_pos);
                mn |= (other & (acc[r] < bn)) ? (0x8000u >> r) : 0u;
                mh |= (other & (acc[r] < bh) & (fef[r] < Q.feq)) ? (0x8000u >> r) : 0u;
              }
            }
            uint32_t m = mn | mh;
            for (;;) {
              const uint64_t have = __builtin_amdgcn_ballot_w64(m != 0);
              if (have == 0) break;
              const uint32_t n_new = (uint32_t)__builtin_popcountll(have);
              if (qn + n_new > (uint32_t)kWaveQueue) flush();
              if (m != 0) {
                const int p = __builtin_ctz(m);
                const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(have >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)have, 0));
                cand[slot] = make_uint2(tile_row(t, 15 - p, h) | (((mn >> p) & 1u) << 30) | (((mh >> p) & 1u) << 31),
                                        (uint32_t)(qi * 32 + c));
                m &= m - 1;
              }
              qn += n_new;
            }
            if (qn >= 64u) flush();
          }
          Q.m_nn = new_nn;
          Q.m_hd = new_hd;
          Q.bn = bn;
          Q.bh = bh;
        }
      };
      // Chains software-pipelined over two accumulator tiles WITHIN a reference tile: the coarse part of chain q + 1
      // runs while the coarse minimum of chain q is taken; a chain none of whose lanes can hold a candidate stops
      // there, the others are computed in full (remaining fragments from the ring slot) and go on to the epilogue.
      // (No chain stays pending across tiles: the slot is refilled at the next window.)
      f32x16 accA, accB;
      // (i: the tile's place in the round's survivor list -- its index is looked up only by a chain that goes on: the
      //  look-up is an LDS round trip that sat in front of every tile's operand reads)
      auto compute = [&](const s16x8 (&a)[NB], uint32_t i, float2 fr, const uint4* slot) {
        f32x16 c0;
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = 0.0f;   // (an inline constant of the first MFMA)
        chains += TQ;
        static_assert(TQ % 2 == 0, "accumulator ping-pong needs an even number of query tiles");
        // the coarse minima of the tile's TQ chains, tested together (one scalar hand-off per tile, as in
        // nn_pruned_kernel); a chain that goes on starts again from its first MFMA
        float tm[TQ], dmin = INFINITY;
        accA = mfma16(a[0], b[0][0], c0);
#pragma unroll
        for (int m = 1; m < NB; ++m) accA = mfma16(a[m], b[0][m], accA);
        constexpr_for_pairs<TQ>([&](auto qt_c) {
          constexpr int qt = decltype(qt_c)::value;
          tm[qt] = INFINITY;
          nn_chain_coarse<NM, NB, NB>(a, b[qt + 1], c0, accB, accA, tm[qt]);
          tm[qt + 1] = INFINITY;
          if constexpr (qt + 2 < TQ)
            nn_chain_coarse<NM, NB, NB>(a, b[qt + 2], c0, accA, accB, tm[qt + 1]);
          else
            tile_min<0, 16>(accB, tm[qt + 1]);
        });
#pragma unroll
        for (int qi = 0; qi < TQ; ++qi) dmin = fminf(dmin, tm[qi] - (((fr.x < q[qi].feq) ? q[qi].bh : q[qi].bn) + skipb));
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(dmin < 0.0f) != 0, 0)) {
          const uint32_t t = entry(i);
          constexpr_for_all<TQ>([&](auto qi_c) {
            constexpr int qi = decltype(qi_c)::value;
            const float thr_c = ((fr.x < q[qi].feq) ? q[qi].bh : q[qi].bn) + skipb;
            if (__builtin_amdgcn_ballot_w64(tm[qi] < thr_c) != 0) {
              chains_on += 1;
              f32x16 acc = mfma16(a[0], b[qi][0], c0);
#pragma unroll
              for (int m = 1; m < NB; ++m) acc = mfma16(a[m], b[qi][m], acc);
#if DC_NNS_GLOBAL_REST
              {
                const uint4* rest = img_r + (size_t)t * (NM * 64) + lane;
                s16x8 ar[NM - NB > 0 ? NM - NB : 1];
#pragma unroll
                for (int m = NB; m < NM; ++m) ar[m - NB] = __builtin_bit_cast(s16x8, rest[m * 64]);
#pragma unroll
                for (int m = NB; m < NM; ++m) acc = mfma16(ar[m - NB], b[qi][m], acc);
              }
#else
#pragma unroll
              for (int m = NB; m < NM; ++m) acc = mfma16(__builtin_bit_cast(s16x8, slot[m * 64 + lane]), b[qi][m], acc);
#endif
              float tmin = INFINITY;
              tile_min<0, 16>(acc, tmin);
              finish(acc, qi_c, tmin, t, fr);
            }
          });
        }
      };
      if ((uint32_t)wib < total) fetch(entry((uint32_t)wib), (uint32_t)wib);
      for (uint32_t i = 0; i < total; ++i) {
        if ((i & 3u) == 0) {
          __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's tile of the window starting at i
          __syncthreads();
          const uint32_t nxt = i + 4 + (uint32_t)wib;
          if (nxt < total) fetch(entry(nxt), nxt % kRing);
        }
        const uint4* slot = ring + (i % kRing) * kUnits;
        s16x8 a[NB];
#pragma unroll
        for (int m = 0; m < NB; ++m) a[m] = __builtin_bit_cast(s16x8, slot[m * 64 + lane]);
        const float2 fr = *reinterpret_cast<const float2*>(slot + NM * 64 + 8);
        if (wave_live) compute(a, i, fr, slot);
      }
      __syncthreads();   // lists and ring are free for the next round
    }
    flush();                                          // the settle test needs the exact incumbents
    reload();
    if (!(r2_hi <= FLT_MAX) || visited >= U - u_lo)
      break;   // every reference tile of this wave's share has been visited
    // settled: every unvisited frame is >= sqrt(r2_hi) away; the exact incumbents decide
    const float sure = r2_hi * 0.9999f;
    float need = 0.0f;      // largest incumbent that still has to be confirmed
    bool blind = false;     // some query has no candidate at all yet
#pragma unroll
    for (int qt = 0; qt < TQ; ++qt) {
      const bool live = (livemask[qt] >> lane) & 1;
      const bool hd_possible = fe_floor < q[qt].feq;
      // a query's incumbent is the better one of its two half-wave lanes
      const float inc_nn = fminf(g_nn[qt], fminf(q[qt].bd_nn, __shfl_xor(q[qt].bd_nn, 32, 64)));
      const float inc_hd = fminf(g_hd[qt], fminf(q[qt].bd_hd, __shfl_xor(q[qt].bd_hd, 32, 64)));
      const float want = fmaxf(inc_nn, hd_possible ? inc_hd : 0.0f);
      const bool open = live & !(want < sure);
      blind = blind | (open & !(want < FLT_MAX));
      need = fmaxf(need, open ? want : 0.0f);
    }
    // one ring schedule for the workgroup: the largest open incumbent of any of its queries
    const float wneed = wave_max(need);
    const uint32_t wflags = (__builtin_amdgcn_ballot_w64(need > 0.0f) != 0 ? 1u : 0u) |
                            (__builtin_amdgcn_ballot_w64(blind) != 0 ? 2u : 0u);
    if (lane == 0) {
      red_need[wib] = wneed;
      red_flags[wib] = wflags;
    }
    __syncthreads();
    const float gneed = fmaxf(fmaxf(red_need[0], red_need[1]), fmaxf(red_need[2], red_need[3]));
    const uint32_t gflags = red_flags[0] | red_flags[1] | red_flags[2] | red_flags[3];
    __syncthreads();   // (the reduction words are rewritten after the next ring)
    if ((gflags & 1u) == 0) break;
    r2_lo = r2_hi;
    if (gflags & 2u) {
      r2_hi = r2_hi * 4.0f;
    } else {
      r2_hi = fmaxf(gneed * 1.001f, r2_hi * 1.001f);
    }
    if (!(r2_hi < 1.0e37f)) r2_hi = INFINITY;
  }
  if (lane == 0 && chain_counter && wave_live) {
    atomicAdd(chain_counter, (unsigned long long)chains);
    atomicAdd(chain_counter + kMfmaCtrNn, (unsigned long long)chains * NB + (unsigned long long)chains_on * NM);
  }

#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    NnPQ& Q = q[qt];
    float od = __shfl_xor(Q.bd_nn, 32, 64);
    uint32_t oj = (uint32_t)__shfl_xor((int)Q.bj_nn, 32, 64);
    lexi_update(oj <= n_rows, Q.bd_nn, Q.bj_nn, od, oj, n_rows);
    od = __shfl_xor(Q.bd_hd, 32, 64);
    oj = (uint32_t)__shfl_xor((int)Q.bj_hd, 32, 64);
    lexi_update(oj <= n_rows, Q.bd_hd, Q.bj_hd, od, oj, n_rows);
    if (h == 0 && ((livemask[qt] >> lane) & 1)) {
      if (n_chunks == 1) {
        nn_idx[jq[qt]] = Q.bj_nn;
        nn_d2[jq[qt]] = Q.bd_nn;
        hd_idx[jq[qt]] = Q.bj_hd;
        hd_d2[jq[qt]] = Q.bd_hd;
      } else {
        // d2 >= 0, so (d2 bits << 32 | frame id) orders like the lexicographic (d2, id): the merge
        // over the chunks is a 64-bit atomic min (merge64 was filled with (FLT_MAX, n_rows+1))
        atomicMin(&merge64[jq[qt]],
                  ((unsigned long long)__float_as_uint(Q.bd_nn) << 32) | Q.bj_nn);
        atomicMin(&merge64[(size_t)n_rows + jq[qt]],
                  ((unsigned long long)__float_as_uint(Q.bd_hd) << 32) | Q.bj_hd);
      }
    }
  }
}

