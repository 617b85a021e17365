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
//     t   = acc - ((r^2 - eps) - |x'|^2):  sign -> inside, bits(t) <u bits(2 eps) -> band -> canonical re-check
//     eps = 1.25 u [(4 K + 10) M + (D / 4 + 12) d2cap],  K = 2 S multiply-adds per chain, M = max |x'|^2
// (the fp32 MFMA is a chain of fmaf: products exact to one rounding each, 2 S + 2 roundings around values <= 4 M).
// Operand image: img32[(t * S + s) * 64 + l] = y'[32 t + (l & 31)][2 s + (l >> 5)], norms32[32 t + c] = |y'|^2.
constexpr int kS32 = 5;   // K-steps: n_cols 9 .. 10

__host__ __device__ inline float guard_eps32(float M, float d2cap, int K, int D) {
  const double u = 5.9604644775390625e-8;
  const double cap = (d2cap > 0.0f) ? (double)d2cap : 0.0;
  return next_up((float)(1.25 * u * ((4.0 * K + 10.0) * (double)M + (0.25 * D + 12.0) * cap)));
}

// fp32 operand image of rows in natural order (perm == nullptr) or gathered through perm
__global__ void image32_kernel(const float* __restrict__ coords, uint32_t n_rows, uint32_t D, uint32_t T,
                               const float* __restrict__ means, const uint32_t* __restrict__ perm,
                               float* __restrict__ img, float* __restrict__ norms) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= 32 * T) return;
  const uint32_t t = row >> 5, c = row & 31;
  const bool live = row < n_rows;
  const uint32_t src = live ? (perm ? perm[row] : row) : 0u;
  double nrm = 0.0;
  for (uint32_t k = 0; k < 2 * (uint32_t)kS32; ++k) {
    float v = 0.0f;
    if (live && k < D) v = coords[(size_t)src * D + k] - means[k];
    nrm += (double)v * (double)v;
    img[((size_t)t * kS32 + (k >> 1)) * 64 + c + 32 * (k & 1)] = v;
  }
  norms[row] = live ? (float)nrm : INFINITY;   // pad rows: never inside, never a candidate
}

template <int S>
__device__ __forceinline__ void load_tile32(const float* __restrict__ img, const float* __restrict__ norms, uint32_t t,
                                            int lane, int h, float (&a)[S], float4 (&nv)[4]) {
  const float* ip = img + (size_t)t * (S * 64) + lane;
#pragma unroll
  for (int s = 0; s < S; ++s) a[s] = ip[s * 64];
  const float4* np = reinterpret_cast<const float4*>(norms + (size_t)t * 32 + 4 * h);
#pragma unroll
  for (int g = 0; g < 4; ++g) nv[g] = np[2 * g];        // rows 8g + 4h .. +3  <->  registers 4g .. 4g+3
}

// one-radius epilogue state of a chain: sign string and the unsigned minimum of bits(acc - lo)
struct Pop32Acc {
  uint32_t bits, tmin;
};
template <int R0, int R1>
__device__ __forceinline__ void pop32_epi(const f32x16& acc, float lo, Pop32Acc& e) {
#pragma unroll
  for (int r = R0; r < R1; ++r) {
    const uint32_t tb = __float_as_uint(acc[r] - lo);
    e.bits = __builtin_amdgcn_alignbit(e.bits, tb, 31);   // (bits << 1) | sign(t)
    e.tmin = min(e.tmin, tb);                             // negative t: huge unsigned
  }
}
template <int S, int SI = 0>
__device__ __forceinline__ void pop32_chain(const float (&a)[S], const float (&b)[S], const f32x16& c0, f32x16& acc_new,
                                            const f32x16& acc_old, float lo_old, Pop32Acc& e) {
  if constexpr (SI < S) {
    if constexpr (SI == 0)
      acc_new = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c0, 0, 0, 0);
    else
      acc_new = __builtin_amdgcn_mfma_f32_32x32x2f32(a[SI], b[SI], acc_new, 0, 0, 0);
    pop32_epi<(16 * SI) / S, (16 * (SI + 1)) / S>(acc_old, lo_old, e);
    pop32_chain<S, SI + 1>(a, b, c0, acc_new, acc_old, lo_old, e);
  }
}

// rare: exact re-check of the band pairs of one accumulator tile (by value: see pop_fix)
__device__ __attribute__((noinline)) uint32_t pop32_fix(const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols,
                                                        float r2, float lo, f32x16 acc, uint32_t wbits, uint32_t jq,
                                                        uint32_t t, int h) {
  uint32_t m = 0, out = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    m |= ((__float_as_uint(acc[r] - lo) < wbits) && (tile_row(t, r, h) < n_rows)) ? (1u << r) : 0u;
  while (__builtin_amdgcn_ballot_w64(m != 0) != 0) {
    if (m != 0) {
      const int r = __builtin_ctz(m);
      out += (exact_d2(coords, n_cols, jq, tile_row(t, r, h)) < r2) ? 1u : 0u;
      m &= m - 1;
    }
  }
  return out;
}

template <int S, int TQ>
__global__ __launch_bounds__(256, 2) void pop_mfma32_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols, const float* __restrict__ img,
    const float* __restrict__ norms, const uint32_t* __restrict__ hdr, uint32_t T, uint32_t i_from, uint32_t i_to,
    float r2, uint32_t* __restrict__ pops) {
  static_assert(TQ % 2 == 0, "accumulator ping-pong needs an even number of query tiles");
  if (hdr[1] != 0) return;   // non-finite / overflow-prone data: the gated direct kernel runs instead
  const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t qt0 = i_from / 32 + wave * TQ;
  if (qt0 * 32 >= i_to) return;   // whole wave leaves; no barriers in this kernel
  const float eps = guard_eps32(__uint_as_float(hdr[0]), r2, 2 * S, (int)n_cols);
  const uint32_t wbits = __float_as_uint(2.0f * eps) + 1u;   // band width 2 eps as an unsigned key, + 1 ulp
  const float r2e = r2 - eps;

  float b[TQ][S], lo[TQ];
  uint32_t cnt[TQ], jq[TQ];
  uint64_t livemask[TQ];
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const uint32_t tile = qt0 + qt;
    jq[qt] = tile * 32 + c;
    const bool live = (tile < T) && (jq[qt] >= i_from) && (jq[qt] < i_to);
    livemask[qt] = __builtin_amdgcn_ballot_w64(live);
    const uint32_t tl = tile < T ? tile : T - 1;
#pragma unroll
    for (int s = 0; s < S; ++s) b[qt][s] = -2.0f * img[((size_t)tl * S + s) * 64 + lane];
    lo[qt] = r2e - (live ? norms[tl * 32 + c] : INFINITY);   // threshold of the lane's query; -inf: nothing inside, no band
    cnt[qt] = 0;
  }
  f32x16 accA, accB;   // ping-pong; B starts as "+inf everywhere" = contributes nothing
#pragma unroll
  for (int r = 0; r < 16; ++r) accB[r] = INFINITY;
  uint32_t tB = 0;
  float a0[S], a1[S];
  float4 n0[4], n1[4];
  load_tile32<S>(img, norms, 0, lane, h, a0, n0);
  auto finish = [&](const f32x16& acc, auto qi_c, const Pop32Acc& e, uint32_t t) {
    constexpr int qi = decltype(qi_c)::value;
    cnt[qi] += __builtin_popcount(e.bits & 0xFFFFu);
    if (__builtin_expect((__builtin_amdgcn_ballot_w64(e.tmin < wbits) & livemask[qi]) != 0, 0)) {
      const uint32_t d = pop32_fix(coords, n_rows, n_cols, r2, lo[qi], acc, wbits, jq[qi], t, h);
      cnt[qi] += ((livemask[qi] >> lane) & 1) ? d : 0u;
    }
  };
  auto tile_body = [&](const float (&a)[S], const float4 (&nv)[4], uint32_t t) {
    const f32x16 c0 = frag16(nv);
    constexpr_for_pairs<TQ>([&](auto qt_c) {
      constexpr int qt = decltype(qt_c)::value;
      constexpr int qb = (qt == 0) ? TQ - 1 : qt - 1;
      Pop32Acc e{0u, 0xFFFFFFFFu};
      pop32_chain<S>(a, b[qt], c0, accA, accB, lo[qb], e);
      finish(accB, std::integral_constant<int, qb>{}, e, (qt == 0) ? tB : t);
      e = Pop32Acc{0u, 0xFFFFFFFFu};
      pop32_chain<S>(a, b[qt + 1], c0, accB, accA, lo[qt], e);
      finish(accA, std::integral_constant<int, qt>{}, e, t);
    });
    // (every chain in the three-address form: for the chain at which c0 dies hipcc 7.2 otherwise accumulates INTO c0's
    //  registers and copies the tile back afterwards -- and its first v_mov of that copy is issued before the wait states
    //  the last MFMA needs: element 15 of the pending tile was stale, found by the parity test at TQ = 4)
    keep_alive(c0);
    tB = t;
  };
  for (uint32_t t = 0; t < T; t += 2) {
    load_tile32<S>(img, norms, (t + 1 < T) ? t + 1 : t, lane, h, a1, n1);
    tile_body(a0, n0, t);
    if (t + 1 < T) {
      load_tile32<S>(img, norms, (t + 2 < T) ? t + 2 : t + 1, lane, h, a0, n0);
      tile_body(a1, n1, t + 1);
    }
  }
  {  // drain: epilogue of the last pending chain
    Pop32Acc e{0u, 0xFFFFFFFFu};
    pop32_epi<0, 16>(accB, lo[TQ - 1], e);
    finish(accB, std::integral_constant<int, TQ - 1>{}, e, tB);
  }
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const uint32_t total = cnt[qt] + (uint32_t)__shfl_xor((int)cnt[qt], 32, 64);
    if (h == 0 && ((livemask[qt] >> lane) & 1)) {
      // the sweep met the self pair and counted it iff d2(i,i) < rad2; the reference starts at 1 (:132-134)
      const float dself = exact_d2(coords, n_cols, jq[qt], jq[qt]);
      pops[jq[qt]] = total + 1u - ((dself < r2) ? 1u : 0u);
    }
  }
}

template <int S, int SI = 0>
__device__ __forceinline__ void nn32_chain(const float (&a)[S], const float (&b)[S], const f32x16& c0, f32x16& acc_new,
                                           const f32x16& acc_old, float& tmin) {
  if constexpr (SI < S) {
    if constexpr (SI == 0)
      acc_new = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c0, 0, 0, 0);
    else
      acc_new = __builtin_amdgcn_mfma_f32_32x32x2f32(a[SI], b[SI], acc_new, 0, 0, 0);
    tile_min<(16 * SI) / S, (16 * (SI + 1)) / S>(acc_old, tmin);
    nn32_chain<S, SI + 1>(a, b, c0, acc_new, acc_old, tmin);
  }
}

// reference frames ORDERED BY FREE ENERGY (nn_mfma_kernel's scheme: whole tiles below / above / straddling)
template <int S, int TQ>
__global__ __launch_bounds__(256, 2) void nn_mfma32_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols, const float* __restrict__ img,
    const float* __restrict__ norms, const float* __restrict__ img_s, const float* __restrict__ norms_s,
    const uint32_t* __restrict__ perm, const uint32_t* __restrict__ invpos, const uint32_t* __restrict__ pq_of,
    const uint32_t* __restrict__ hdr, uint32_t T, uint32_t i_from, uint32_t i_to, uint32_t* __restrict__ nn_idx,
    float* __restrict__ nn_d2, uint32_t* __restrict__ hd_idx, float* __restrict__ hd_d2) {
  static_assert(TQ % 2 == 0, "accumulator ping-pong needs an even number of query tiles");
  if (hdr[1] != 0) return;
  const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t qt0 = i_from / 32 + wave * TQ;
  if (qt0 * 32 >= i_to) return;
  const float M = __uint_as_float(hdr[0]);
  const float eps2 = 2.5f * guard_eps32(M, 4.0f * M, 2 * S, (int)n_cols);   // candidates can be as far apart as 2 sqrt(M)

  float b[TQ][S];
  NnQ q[TQ];
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
    for (int s = 0; s < S; ++s) b[qt][s] = -2.0f * img[((size_t)tl * S + s) * 64 + lane];
    const uint32_t jl = live ? jq[qt] : (n_rows - 1);
    q[qt].pq = live ? pq_of[jl] : 0u;
    q[qt].spos = live ? invpos[jl] : 0xFFFFFFFFu;
    q[qt].t_self = live ? (q[qt].spos >> 5) : 0xFFFFFFFFu;
    q[qt].t_full = q[qt].pq >> 5;
    q[qt].t_part = (q[qt].pq & 31u) ? (q[qt].pq >> 5) : 0xFFFFFFFFu;
    // acc = |y'|^2 - 2 x'.y' = d2 - |x'|^2: the running minima carry the same offset, the band does not care.
    // idle lanes start at -inf: they can never trigger the exact path and never change
    q[qt].m_nn = live ? INFINITY : -INFINITY;
    q[qt].m_hd = live ? INFINITY : -INFINITY;
    q[qt].bd_nn = FLT_MAX;
    q[qt].bd_hd = FLT_MAX;
    q[qt].bj_nn = n_rows + 1;
    q[qt].bj_hd = n_rows + 1;
  }
  (void)norms;
  f32x16 accA, accB;
#pragma unroll
  for (int r = 0; r < 16; ++r) accB[r] = INFINITY;
  uint32_t tB = 0;
  float a0[S], a1[S];
  float4 n0[4], n1[4];
  load_tile32<S>(img_s, norms_s, 0, lane, h, a0, n0);
  auto finish = [&](const f32x16& acc, auto qi_c, float tmin, uint32_t t) {
    constexpr int qi = decltype(qi_c)::value;
    NnQ& Q = q[qi];
    const bool special = (t == Q.t_self) | (t == Q.t_part);
    float hmin = (t < Q.t_full) ? tmin : INFINITY;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0, 0)) {
      const NnMin g = nn_special(acc, t, h, Q.spos, Q.pq);   // valid for every lane, just slower
      tmin = g.tmin;
      hmin = g.hmin;
    }
    const bool trig = (tmin < Q.m_nn + eps2) | (hmin < Q.m_hd + eps2);
    const float new_nn = fminf(Q.m_nn, tmin), new_hd = fminf(Q.m_hd, hmin);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(trig) != 0, 0)) {
      const bool live = (livemask[qi] >> lane) & 1;
      NnBest best{Q.bd_nn, Q.bd_hd, Q.bj_nn, Q.bj_hd};
      best = nn_fix(coords, perm, n_rows, n_cols, acc, new_nn + eps2, new_hd + eps2, best, jq[qi], Q.spos, Q.pq, t, h);
      Q.bd_nn = live ? best.bd_nn : Q.bd_nn;
      Q.bj_nn = live ? best.bj_nn : Q.bj_nn;
      Q.bd_hd = live ? best.bd_hd : Q.bd_hd;
      Q.bj_hd = live ? best.bj_hd : Q.bj_hd;
    }
    Q.m_nn = new_nn;
    Q.m_hd = new_hd;
  };
  auto tile_body = [&](const float (&a)[S], const float4 (&nv)[4], uint32_t t) {
    const f32x16 c0 = frag16(nv);
    constexpr_for_pairs<TQ>([&](auto qt_c) {
      constexpr int qt = decltype(qt_c)::value;
      constexpr int qb = (qt == 0) ? TQ - 1 : qt - 1;
      float tmin = INFINITY;
      nn32_chain<S>(a, b[qt], c0, accA, accB, tmin);
      finish(accB, std::integral_constant<int, qb>{}, tmin, (qt == 0) ? tB : t);
      tmin = INFINITY;
      nn32_chain<S>(a, b[qt + 1], c0, accB, accA, tmin);
      finish(accA, std::integral_constant<int, qt>{}, tmin, t);
    });
    keep_alive(c0);   // (see pop_mfma32_kernel)
    tB = t;
  };
  for (uint32_t t = 0; t < T; t += 2) {
    load_tile32<S>(img_s, norms_s, (t + 1 < T) ? t + 1 : t, lane, h, a1, n1);
    tile_body(a0, n0, t);
    if (t + 1 < T) {
      load_tile32<S>(img_s, norms_s, (t + 2 < T) ? t + 2 : t + 1, lane, h, a0, n0);
      tile_body(a1, n1, t + 1);
    }
  }
  {  // drain: epilogue of the last pending chain
    float tmin = INFINITY;
    tile_min<0, 16>(accB, tmin);
    finish(accB, std::integral_constant<int, TQ - 1>{}, tmin, tB);
  }
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    NnQ& Q = q[qt];
    // merge the two half-waves (disjoint reference rows of the same query)
    float od = __shfl_xor(Q.bd_nn, 32, 64);
    uint32_t oj = (uint32_t)__shfl_xor((int)Q.bj_nn, 32, 64);
    lexi_update(oj <= n_rows, Q.bd_nn, Q.bj_nn, od, oj, n_rows);
    od = __shfl_xor(Q.bd_hd, 32, 64);
    oj = (uint32_t)__shfl_xor((int)Q.bj_hd, 32, 64);
    lexi_update(oj <= n_rows, Q.bd_hd, Q.bj_hd, od, oj, n_rows);
    if (h == 0 && ((livemask[qt] >> lane) & 1)) {
      nn_idx[jq[qt]] = Q.bj_nn;
      nn_d2[jq[qt]] = Q.bd_nn;
      hd_idx[jq[qt]] = Q.bj_hd;
      hd_d2[jq[qt]] = Q.bd_hd;
    }
  }
}
