// dc_mfma.hip -- fp32-MFMA variants of the two pairwise sweeps (gfx950, v_mfma_f32_32x32x2_f32).
//
// Idea.  The N x D . D x N distance block is a dense contraction:
//     d2(x, y) = |x|^2 + |y|^2 - 2 x.y
// so a 32(reference) x 32(query) tile costs ceil(D/2) MFMAs instead of 32*32*(3D-1) VALU ops.
// But the Gram form does not round like the reference's direct-difference sum, and the outputs
// (integer populations, neighbour indices) must equal the reference's bit for bit.  So the MFMA
// result is used only as a CLASSIFIER with a rigorous guard band eps (derivation in DESIGN.md):
//     acc <  thr - eps   =>  canonical d2 <  thr      (decided by the MFMA value alone)
//     acc >= thr + eps   =>  canonical d2 >= thr
//     otherwise          =>  the pair is re-evaluated in the canonical order (dist2_canon_rt) from
//                            the ORIGINAL coordinates, in-kernel, by the lane that owns it.
// A handful of pairs per frame fall in the band (|d2 - r^2| < ~1e-5), so the re-check costs ~1 %.
// The nearest-neighbour sweep uses the same band around the running minimum: every reference frame
// whose MFMA distance is within 2 eps of the running minimum is evaluated exactly and merged
// lexicographically on (d2, index), which reproduces "lowest index wins ties" (:270) exactly.
//
// Mapping (one wave = TQ query tiles of 32 frames, swept against all reference tiles of 32 frames):
//   v_mfma_f32_32x32x2_f32:  D[i][j] = C[i][j] + sum_k A[i][k] B[k][j],  lane l holds A[l&31][l>>5],
//   B[l>>5][l&31]; D[i][j] sits in lane (j + 32*h), register r with i = (r&3) + 8*(r>>2) + 4*h.
//     A = centred reference coordinates   y'[i][k]            (streamed, one dword per K-step)
//     B = -2 * centred query coordinates  x'[j][k]            (resident in VGPRs for the whole sweep)
//     C = |y'_i|^2 broadcast along the row                     (initial accumulator, 4 x dwordx4)
//   => acc = |y'|^2 - 2 x'.y' ; the query norm moves into the per-lane threshold r^2 - |x'|^2 -+ eps.
// Queries sit on the lane axis, so populations / running minima are per-lane registers; the two
// half-waves (h = 0/1) see disjoint reference rows of the same 32 queries and are merged by one
// __shfl_xor(.., 32) at the very end.
//
// Operand images (built once per call by mfma_prepare in the caller's workspace):
//   img   [T][S][64]  A fragments in lane order: img[(t*S+s)*64 + l] = y'[32t + (l&31)][2s + (l>>5)]
//   norms [32T]       |y'|^2 (double accumulate, rounded once); +inf for the pad rows of the last tile
//   fe    [32T]       free energies padded with +inf (nearest-neighbour sweep only)
// Each wave streams the images with plain coalesced global loads (256 B per K-step); all waves of
// the chip walk the same 44 MB, which lives in L2 / Infinity Cache.
#pragma once
#include "dc_mfma.hpp"

#include <float.h>
#include <math.h>

namespace dc {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMaxSteps = 16;          // K-steps of two columns -> n_cols <= 32
constexpr size_t kHdrBytes = 1024;     // [0] max norm (float bits); [256..] column sums (double)
constexpr size_t kHdrSums = 256;

struct Layout {
  uint32_t T, S;
  size_t off_img, off_norm, off_fe, total;
};

inline Layout make_layout(size_t n_rows, size_t n_cols) {
  Layout L;
  L.T = (uint32_t)((n_rows + 31) / 32);
  L.S = (uint32_t)((n_cols + 1) / 2);
  L.off_img = kHdrBytes;
  L.off_norm = L.off_img + sizeof(float) * 64 * (size_t)L.T * L.S;
  L.off_fe = L.off_norm + sizeof(float) * 32 * (size_t)L.T;
  L.total = L.off_fe + sizeof(float) * 32 * (size_t)L.T;
  L.total = (L.total + 255) & ~(size_t)255;
  return L;
}

// ---------------------------------------------------------------------------------------------
// guard band (DESIGN.md "guard band"): |acc + |x'|^2 - d2_canonical| <= eps for every pair whose
// canonical d2 is <= d2cap, given M = max |x'|^2, K = 2S fused multiply-adds in the MFMA chain.
//   eps = 1.25 * u * [ (4K + 10) * M + (D/4 + 12) * d2cap ],  u = 2^-24, rounded up.
// Non-finite M  ->  +inf  ->  every pair takes the exact path (slow, still correct).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float guard_eps(float M, float d2cap, int K, int D) {
  const double u = 5.9604644775390625e-8;
  const double cap = (d2cap > 0.0f) ? (double)d2cap : 0.0;
  const double e = 1.25 * u * ((4.0 * K + 10.0) * (double)M + (0.25 * D + 12.0) * cap);
  const float f = (float)e;
  if (!(f <= FLT_MAX)) return INFINITY;                 // inf or NaN
  return __uint_as_float(__float_as_uint(f) + 1u);      // next float up (f >= 0)
}

template <int S>
__device__ __forceinline__ void load_tile(const float* __restrict__ img,
                                          const float* __restrict__ norms, uint32_t t, int lane,
                                          int h, float (&a)[S], float4 (&nv)[4]) {
  const float* ip = img + (size_t)t * (S * 64) + lane;
#pragma unroll
  for (int s = 0; s < S; ++s) a[s] = ip[s * 64];
  const float4* np = reinterpret_cast<const float4*>(norms + (size_t)t * 32 + 4 * h);
#pragma unroll
  for (int g = 0; g < 4; ++g) nv[g] = np[2 * g];        // rows 8g + 4h .. +3  <->  registers 4g .. 4g+3
}

__device__ __forceinline__ f32x16 frag16(const float4 (&v)[4]) {
  f32x16 o;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    o[4 * g + 0] = v[g].x;
    o[4 * g + 1] = v[g].y;
    o[4 * g + 2] = v[g].z;
    o[4 * g + 3] = v[g].w;
  }
  return o;
}

template <int S>
__device__ __forceinline__ f32x16 gram_tile(const float (&a)[S], const float (&b)[S],
                                            const f32x16& c0) {
  f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c0, 0, 0, 0);
#pragma unroll
  for (int s = 1; s < S; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
  return acc;
}

// row of reference tile t held by register r of a lane in half h
__device__ __forceinline__ uint32_t tile_row(uint32_t t, int r, int h) {
  return 32u * t + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * h);
}

// ---------------------------------------------------------------------------------------------
// population count
// ---------------------------------------------------------------------------------------------
template <int S, int NR, int TQ>
__global__ __launch_bounds__(256, 2) void pop_mfma_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols,
    const float* __restrict__ img, const float* __restrict__ norms,
    const uint32_t* __restrict__ maxnorm_bits, uint32_t T, uint32_t i_from, uint32_t i_to,
    Rad2 rad2, int n_rad, uint32_t* __restrict__ pops) {
  const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t qt0 = i_from / 32 + wave * TQ;
  if (qt0 * 32 >= i_to) return;   // whole wave leaves; no barriers in this kernel

  float r2max = rad2.v[0];
#pragma unroll
  for (int rr = 1; rr < NR; ++rr) r2max = fmaxf(r2max, rad2.v[rr]);
  const float eps = guard_eps(__uint_as_float(*maxnorm_bits), r2max, 2 * S, (int)n_cols);

  float b[TQ][S], lo[TQ][NR], hi[TQ][NR];
  uint32_t cnt[TQ][NR], jq[TQ];
  bool live[TQ];
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const uint32_t tile = qt0 + qt;
    jq[qt] = tile * 32 + c;
    live[qt] = (tile < T) && (jq[qt] >= i_from) && (jq[qt] < i_to);
    const uint32_t tl = tile < T ? tile : T - 1;
#pragma unroll
    for (int s = 0; s < S; ++s) b[qt][s] = -2.0f * img[((size_t)tl * S + s) * 64 + lane];
    const float nx = norms[tl * 32 + c];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      lo[qt][rr] = live[qt] ? (rad2.v[rr] - nx) - eps : -INFINITY;
      hi[qt][rr] = live[qt] ? (rad2.v[rr] - nx) + eps : -INFINITY;
      cnt[qt][rr] = 0;
    }
  }

  float a[S];
  float4 nv[4];
  load_tile<S>(img, norms, 0, lane, h, a, nv);
  for (uint32_t t = 0; t < T; ++t) {
    float an[S];
    float4 nvn[4];
    load_tile<S>(img, norms, (t + 1 < T) ? t + 1 : t, lane, h, an, nvn);   // prefetch
    const f32x16 c0 = frag16(nv);
#pragma unroll
    for (int qt = 0; qt < TQ; ++qt) {
      const f32x16 acc = gram_tile<S>(a, b[qt], c0);
      bool band = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
          const bool in = acc[r] < lo[qt][rr];
          const bool nout = !(acc[r] >= hi[qt][rr]);      // true for NaN: undecidable -> exact path
          cnt[qt][rr] += in ? 1u : 0u;
          band = band || (nout && !in);
        }
      }
      band = band && live[qt];
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(band) != 0, 0)) {
        // exact re-check of the pairs inside a guard band (rare)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          bool any = false;
#pragma unroll
          for (int rr = 0; rr < NR; ++rr)
            any = any || (!(acc[r] >= hi[qt][rr]) && !(acc[r] < lo[qt][rr]));
          const uint32_t i = tile_row(t, r, h);
          if (any && live[qt] && i < n_rows) {
            const float d2c = dist2_canon_rt(coords + (size_t)jq[qt] * n_cols, 1,
                                             coords + (size_t)i * n_cols, 1, (int)n_cols);
#pragma unroll
            for (int rr = 0; rr < NR; ++rr)
              if (!(acc[r] >= hi[qt][rr]) && !(acc[r] < lo[qt][rr]))
                cnt[qt][rr] += (d2c < rad2.v[rr]) ? 1u : 0u;
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) a[s] = an[s];
#pragma unroll
    for (int g = 0; g < 4; ++g) nv[g] = nvn[g];
  }

#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      const uint32_t total = cnt[qt][rr] + (uint32_t)__shfl_xor((int)cnt[qt][rr], 32, 64);
      if (h == 0 && live[qt] && rr < n_rad) {
        // the sweep met the self pair and counted it iff d2(i,i) < rad2; the reference starts at 1
        const float* x = coords + (size_t)jq[qt] * n_cols;
        const float dself = dist2_canon_rt(x, 1, x, 1, (int)n_cols);
        pops[(size_t)rr * n_rows + jq[qt]] = total + 1u - ((dself < rad2.v[rr]) ? 1u : 0u);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// nearest neighbour / nearest neighbour with lower free energy
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void lexi_update(bool cond, float& bd, uint32_t& bj, float d,
                                            uint32_t j, uint32_t n_rows) {
  // strict '<' on d2 scanning ascending j  ==  lexicographic min on (d2, j); a tie only counts
  // against a REAL incumbent (the initial FLT_MAX / n_rows+1 never loses a tie: :257-260, :270).
  // Written as selects on purpose: the branchy form of this update was mis-structurised by
  // hipcc 7.2 (the tie winner's index move was dropped), caught by the duplicate-frame test.
  const bool take = cond & ((d < bd) | ((d == bd) & (j < bj) & (bj <= n_rows)));
  bd = take ? d : bd;
  bj = take ? j : bj;
}

template <int S, int TQ>
__global__ __launch_bounds__(256, 2) void nn_mfma_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols,
    const float* __restrict__ img, const float* __restrict__ norms,
    const float* __restrict__ fe_pad, const uint32_t* __restrict__ maxnorm_bits, uint32_t T,
    uint32_t i_from, uint32_t i_to, uint32_t* __restrict__ nn_idx, float* __restrict__ nn_d2,
    uint32_t* __restrict__ hd_idx, float* __restrict__ hd_d2) {
  const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t qt0 = i_from / 32 + wave * TQ;
  if (qt0 * 32 >= i_to) return;

  const float M = __uint_as_float(*maxnorm_bits);
  // candidates can be as far apart as 2*sqrt(M): d2cap = 4M
  const float eps = guard_eps(M, 4.0f * M, 2 * S, (int)n_cols);
  const float eps2 = 2.5f * eps;

  float b[TQ][S], feq[TQ], m_nn[TQ], m_hd[TQ], bd_nn[TQ], bd_hd[TQ];
  uint32_t jq[TQ], bj_nn[TQ], bj_hd[TQ];
  bool live[TQ];
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const uint32_t tile = qt0 + qt;
    jq[qt] = tile * 32 + c;
    live[qt] = (tile < T) && (jq[qt] >= i_from) && (jq[qt] < i_to);
    const uint32_t tl = tile < T ? tile : T - 1;
#pragma unroll
    for (int s = 0; s < S; ++s) b[qt][s] = -2.0f * img[((size_t)tl * S + s) * 64 + lane];
    feq[qt] = fe_pad[tl * 32 + c];
    m_nn[qt] = INFINITY;
    m_hd[qt] = INFINITY;
    bd_nn[qt] = FLT_MAX;
    bd_hd[qt] = FLT_MAX;
    bj_nn[qt] = n_rows + 1;
    bj_hd[qt] = n_rows + 1;
  }

  float a[S];
  float4 nv[4], fv[4];
  load_tile<S>(img, norms, 0, lane, h, a, nv);
  {
    const float4* fp = reinterpret_cast<const float4*>(fe_pad + 4 * h);
#pragma unroll
    for (int g = 0; g < 4; ++g) fv[g] = fp[2 * g];
  }
  for (uint32_t t = 0; t < T; ++t) {
    float an[S];
    float4 nvn[4], fvn[4];
    const uint32_t tn = (t + 1 < T) ? t + 1 : t;
    load_tile<S>(img, norms, tn, lane, h, an, nvn);
    {
      const float4* fp = reinterpret_cast<const float4*>(fe_pad + (size_t)tn * 32 + 4 * h);
#pragma unroll
      for (int g = 0; g < 4; ++g) fvn[g] = fp[2 * g];
    }
    const f32x16 c0 = frag16(nv);
    const f32x16 fef = frag16(fv);
#pragma unroll
    for (int qt = 0; qt < TQ; ++qt) {
      f32x16 acc = gram_tile<S>(a, b[qt], c0);
      // nn: lanes also hold the norm |x'|^2 implicitly: acc = d2 - |x'|^2, same offset for every
      // reference of this lane, so minima and bands can be taken on acc directly.
      if (t == qt0 + qt) {   // the tile that contains the queries themselves: drop i == j (:262)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((r & 3) + 8 * (r >> 2) + 4 * h == c) acc[r] = INFINITY;
      }
      float tmin = acc[0], hmin = INFINITY;
#pragma unroll
      for (int r = 1; r < 16; ++r) tmin = fminf(tmin, acc[r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) hmin = fminf(hmin, (fef[r] < feq[qt]) ? acc[r] : INFINITY);
      const bool trig = live[qt] && ((tmin < m_nn[qt] + eps2) || (hmin < m_hd[qt] + eps2));
      const float new_nn = fminf(m_nn[qt], tmin), new_hd = fminf(m_hd[qt], hmin);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(trig) != 0, 0)) {
        const float bn = new_nn + eps2, bh = new_hd + eps2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool cn = acc[r] < bn;
          const bool ch = (fef[r] < feq[qt]) && (acc[r] < bh);
          const uint32_t i = tile_row(t, r, h);
          if ((cn || ch) && live[qt] && i < n_rows) {
            const float d2c = dist2_canon_rt(coords + (size_t)jq[qt] * n_cols, 1,
                                             coords + (size_t)i * n_cols, 1, (int)n_cols);
            lexi_update(cn, bd_nn[qt], bj_nn[qt], d2c, i, n_rows);
            lexi_update(ch, bd_hd[qt], bj_hd[qt], d2c, i, n_rows);
          }
        }
      }
      m_nn[qt] = new_nn;
      m_hd[qt] = new_hd;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) a[s] = an[s];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      nv[g] = nvn[g];
      fv[g] = fvn[g];
    }
  }

#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    // merge the two half-waves (disjoint reference rows of the same query)
    float od = __shfl_xor(bd_nn[qt], 32, 64);
    uint32_t oj = (uint32_t)__shfl_xor((int)bj_nn[qt], 32, 64);
    lexi_update(oj <= n_rows, bd_nn[qt], bj_nn[qt], od, oj, n_rows);
    od = __shfl_xor(bd_hd[qt], 32, 64);
    oj = (uint32_t)__shfl_xor((int)bj_hd[qt], 32, 64);
    lexi_update(oj <= n_rows, bd_hd[qt], bj_hd[qt], od, oj, n_rows);
    if (h == 0 && live[qt]) {
      nn_idx[jq[qt]] = bj_nn[qt];
      nn_d2[jq[qt]] = bd_nn[qt];
      hd_idx[jq[qt]] = bj_hd[qt];
      hd_d2[jq[qt]] = bd_hd[qt];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dispatch over the number of K-steps
// ---------------------------------------------------------------------------------------------
constexpr int kTQ = 4;

struct Ptrs {
  const uint32_t* maxnorm;
  const double* sums;
  const float* img;
  const float* norms;
  const float* fe;
};

inline Ptrs ws_ptrs(void* d_ws, const Layout& L) {
  char* p = (char*)d_ws;
  return Ptrs{(const uint32_t*)p, (const double*)(p + kHdrSums), (const float*)(p + L.off_img),
              (const float*)(p + L.off_norm), (const float*)(p + L.off_fe)};
}

inline uint32_t grid_for(uint32_t i_from, uint32_t i_to, int tq) {
  const uint32_t tiles = (i_to + 31) / 32 - i_from / 32;
  const uint32_t waves = (tiles + tq - 1) / tq;
  return (waves + 3) / 4;
}

template <int S>
void pop_dispatch(const float* coords, uint32_t n_rows, uint32_t n_cols, const Ptrs& P, uint32_t T,
                  uint32_t i_from, uint32_t i_to, const Rad2& rad2, int n_rad, uint32_t* pops,
                  hipStream_t s) {
  const dim3 grid(grid_for(i_from, i_to, kTQ)), block(256);
  if (n_rad == 1)
    hipLaunchKernelGGL((pop_mfma_kernel<S, 1, kTQ>), grid, block, 0, s, coords, n_rows, n_cols, P.img,
                       P.norms, P.maxnorm, T, i_from, i_to, rad2, n_rad, pops);
  else if (n_rad <= 4)
    hipLaunchKernelGGL((pop_mfma_kernel<S, 4, kTQ>), grid, block, 0, s, coords, n_rows, n_cols, P.img,
                       P.norms, P.maxnorm, T, i_from, i_to, rad2, n_rad, pops);
  else
    hipLaunchKernelGGL((pop_mfma_kernel<S, 8, kTQ>), grid, block, 0, s, coords, n_rows, n_cols, P.img,
                       P.norms, P.maxnorm, T, i_from, i_to, rad2, n_rad, pops);
}

template <int S>
void nn_dispatch(const float* coords, uint32_t n_rows, uint32_t n_cols, const Ptrs& P, uint32_t T,
                 uint32_t i_from, uint32_t i_to, uint32_t* nn_idx, float* nn_d2, uint32_t* hd_idx,
                 float* hd_d2, hipStream_t s) {
  const dim3 grid(grid_for(i_from, i_to, kTQ)), block(256);
  hipLaunchKernelGGL((nn_mfma_kernel<S, kTQ>), grid, block, 0, s, coords, n_rows, n_cols, P.img,
                     P.norms, P.fe, P.maxnorm, T, i_from, i_to, nn_idx, nn_d2, hd_idx, hd_d2);
}

}  // namespace

// one translation unit per K-step count (dc_mfma_step.hip, -DDC_STEP=n) so the instances build
// in parallel; dc_mfma.hip switches over them.
#define DC_DECLARE_STEP(SV)                                                                      \
  void pop_mfma_step_##SV(const float* coords, uint32_t n_rows, uint32_t n_cols, void* d_ws,     \
                          uint32_t i_from, uint32_t i_to, const Rad2& rad2, int n_rad,           \
                          uint32_t* pops, hipStream_t s);                                        \
  void nn_mfma_step_##SV(const float* coords, uint32_t n_rows, uint32_t n_cols, void* d_ws,      \
                         uint32_t i_from, uint32_t i_to, uint32_t* nn_idx, float* nn_d2,         \
                         uint32_t* hd_idx, float* hd_d2, hipStream_t s);

}  // namespace dc
