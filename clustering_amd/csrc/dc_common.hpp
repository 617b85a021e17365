// dc_common.hpp -- shared declarations of the HIP density kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// The canonical distance is a specification of individual float roundings
// (SURVEY.md Appendix B): never let the compiler fuse a multiply into an add.
#pragma clang fp contract(off)

namespace dc {

constexpr int kMaxColsTemplated = 32;   // n_cols with a register-resident kernel instance
constexpr int kMaxColsGeneric = 400;    // n_cols the LDS-resident generic kernel can hold
constexpr int kMaxRadiiPerLaunch = 8;   // radii swept per launch (more radii -> several launches)

// squared radii of one launch, passed by value (kernarg segment -> SGPRs)
struct Rad2 {
  float v[kMaxRadiiPerLaunch];
};

// ---- launchers implemented in dc_direct.hip -------------------------------------------
// pops: [n_radii_total][n_rows] radius-major; this launch fills radius rows
// r_first .. r_first+n_rad-1 for query rows [i_from, i_to).  Returns false if n_cols is unsupported.
// gate: optional device pointer to the MFMA workspace header; when given, the kernel runs only if
// gate[1] != 0 (the operand-image pass flagged data the MFMA kernels must not touch).
bool launch_pop_direct(const float* d_coords, uint32_t n_rows, uint32_t n_cols, uint32_t i_from,
                       uint32_t i_to, const Rad2& rad2, int n_rad, uint32_t* d_pops_first_row,
                       const uint32_t* gate, hipStream_t stream);

bool launch_nn_direct(const float* d_coords, uint32_t n_rows, uint32_t n_cols, const float* d_fe,
                      uint32_t i_from, uint32_t i_to, uint32_t* d_nn_idx, float* d_nn_d2,
                      uint32_t* d_hd_idx, float* d_hd_d2, const uint32_t* gate, hipStream_t stream);

// fills idx[i] = n_rows+1, d2[i] = FLT_MAX for all rows (density_clustering.cpp:242-245)
void launch_nn_init(uint32_t n_rows, uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx,
                    float* d_hd_d2, hipStream_t stream);

// fe[i] = table[pops[i]]
void launch_fe_gather(const uint32_t* d_pops, uint32_t n_rows, const float* d_table, float* d_fe,
                      hipStream_t stream);
// *d_out = max(pops)
void launch_max_u32(const uint32_t* d_pops, uint32_t n_rows, uint32_t* d_out, hipStream_t stream);
// neighbour results <-> [2][n_rows] order-preserving words (d2 bits << 32 | index)
void launch_nn_pack(const uint32_t* d_nn_idx, const float* d_nn_d2, const uint32_t* d_hd_idx,
                    const float* d_hd_d2, uint32_t n_rows, unsigned long long* d_words, hipStream_t stream);
void launch_nn_unpack(const unsigned long long* d_words, uint32_t n_rows, uint32_t* d_nn_idx, float* d_nn_d2,
                      uint32_t* d_hd_idx, float* d_hd_d2, hipStream_t stream);
// fe of every row with the device's double log; rows whose value sits within 64 ulp(double) of a float
// rounding boundary go to d_flag_list as (row, pop) pairs (d_flag_count may exceed flag_cap)
void launch_fe_log(const uint32_t* d_pops, uint32_t n_rows, uint32_t* d_state, uint32_t slot, float* d_fe,
                   uint32_t* d_flag_list, uint32_t flag_cap, double tol_rel, hipStream_t stream);
constexpr uint32_t kFeStateWords = 8;   // device state of the free-energy pass in front of its list of flagged rows

// ---- canonical squared distance ------------------------------------------------------------------------------------
// Order of operations = SURVEY.md Appendix B = what the reference binary computes (density_clustering.cpp:171-176,
// 263-268 under -O3 -ffast-math).  WHICH order that is depends on how the reference was built (CMakeLists.txt:57-83):
//   default / -DCPU_ACCELERATION=SSE*   four lane sums a_l, (a0 + a2) + (a1 + a3), then a PAIR tail and a scalar tail
//   -DCPU_ACCELERATION=AVX (-mavx)      eight lane sums a_l, b_i = a_i + a_{i+4}, (b0 + b2) + (b1 + b3); if four or more
//                                       columns remain: their squares q, s += (q0 + q2) + (q1 + q3); then up to three
//                                       SCALAR additions -- read off the code g++ 11.4 emits for the reference's loop shape
//                                       with the reference's flags and pinned against it for D = 1 .. 40, 48, 63 .. 65, 100
//                                       (the probe of the test suite, built with -mavx)
//   NATIVE_COMPILATION (-march=native) on a host with AVX2 + FMA, g++'s generic / Intel tunings (haswell ... icelake-
//                                       server, x86-64-v3: all the same code; the Zen tunings unroll differently and are NOT
//                                       covered): the AVX shape with FUSED multiply-adds in the eight-lane loop and in the
//                                       scalar tail (one rounding each), the four-column step unfused -- pinned the same way
//                                       (the probe built with -mavx2 -mfma)
// The library is built for ONE of them: `make` = the default order, `make CANON=avx` / `make CANON=fma` = the other two into
// clustering_amd/lib_avx/ and lib_fma/ (same file name, same ABI; dc_hip_canon_order() says which; DC_CANON_ORDER=avx / fma
// makes the Python host bind that build).  Every exact path of every kernel goes through the three functions below; the guard bands of
// the matrix-core classifiers bound the summation order generically ((D / 4 + 9) u d2) and cover either.
#if defined(DC_CANON_FMA) && !defined(DC_CANON_AVX)
#define DC_CANON_AVX 1   // (the FMA order has the AVX order's shape)
#endif
#ifdef DC_CANON_AVX
// DC_CANON_ACC(a, c): one more column on a lane sum or on the scalar tail.  -mavx: a + c * c, two roundings; with FMA
// (-march=native on a host with AVX2 + FMA, g++'s generic / Intel tunings: `make CANON=fma`) ONE rounding -- the eight-lane
// loop is vfmadd231ps, the scalar tail vfmadd231ss; the four-column step stays a plain multiply followed by additions.
#ifdef DC_CANON_FMA
#define DC_CANON_ORDER_NAME "fma"
#define DC_CANON_ACC(a, c) __builtin_fmaf((c), (c), (a))
#else
#define DC_CANON_ORDER_NAME "avx"
#define DC_CANON_ACC(a, c) ((a) + (c) * (c))
#endif
// the AVX order over a sequence of differences c(0) .. c(D-1) (df: k -> x_k - y_k)
template <class Df>
__device__ __forceinline__ float canon_sum_avx(Df&& df, int D) {
  auto sq = [&](int k) {
    const float c = df(k);
    return c * c;
  };
  float s = 0.0f;
  int k = 0;
  const int V8 = 8 * (D / 8);
  if (V8 != 0) {
    // lane accumulators start at +0; 0 + p == p exactly (p is a square: never -0), and fma(c, c, 0) is the rounded square
    float a0 = sq(0), a1 = sq(1), a2 = sq(2), a3 = sq(3), a4 = sq(4), a5 = sq(5), a6 = sq(6), a7 = sq(7);
    for (int k0 = 8; k0 < V8; k0 += 8) {
      a0 = DC_CANON_ACC(a0, df(k0 + 0));
      a1 = DC_CANON_ACC(a1, df(k0 + 1));
      a2 = DC_CANON_ACC(a2, df(k0 + 2));
      a3 = DC_CANON_ACC(a3, df(k0 + 3));
      a4 = DC_CANON_ACC(a4, df(k0 + 4));
      a5 = DC_CANON_ACC(a5, df(k0 + 5));
      a6 = DC_CANON_ACC(a6, df(k0 + 6));
      a7 = DC_CANON_ACC(a7, df(k0 + 7));
    }
    const float b0 = a0 + a4, b1 = a1 + a5, b2 = a2 + a6, b3 = a3 + a7;
    s = (b0 + b2) + (b1 + b3);
    k = V8;
  }
  if (D - k >= 4) {
    const float t = (sq(k) + sq(k + 2)) + (sq(k + 1) + sq(k + 3));
    s = s + t;   // (no eight-lane part: 0 + t == t exactly)
    k += 4;
  }
  for (; k < D; ++k) s = DC_CANON_ACC(s, df(k));   // (s == +0 at the first of D <= 3 columns: the rounded square either way)
  return s;
}
template <int D>
__device__ __forceinline__ float dist2_canon(const float (&q)[D], const float (&r)[D]) {
  float c[D];
#pragma unroll
  for (int k = 0; k < D; ++k) c[k] = q[k] - r[k];
  float s = 0.0f;
  constexpr int V8 = 8 * (D / 8);
  if constexpr (V8 != 0) {
    float a[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) a[l] = c[l] * c[l];
#pragma unroll
    for (int k0 = 8; k0 < V8; k0 += 8)
#pragma unroll
      for (int l = 0; l < 8; ++l) a[l] = DC_CANON_ACC(a[l], c[k0 + l]);
    const float b0 = a[0] + a[4], b1 = a[1] + a[5], b2 = a[2] + a[6], b3 = a[3] + a[7];
    s = (b0 + b2) + (b1 + b3);
  }
  constexpr int K4 = (D - V8 >= 4) ? V8 + 4 : V8;
  if constexpr (D - V8 >= 4) {
    const float t = (c[V8] * c[V8] + c[V8 + 2] * c[V8 + 2]) + (c[V8 + 1] * c[V8 + 1] + c[V8 + 3] * c[V8 + 3]);
    s = (V8 != 0) ? s + t : t;
  }
#pragma unroll
  for (int k = K4; k < D; ++k) s = (k == 0) ? c[0] * c[0] : DC_CANON_ACC(s, c[k]);
  return s;
}
__device__ __forceinline__ float dist2_canon_rt(const float* x, int sx, const float* y, int sy, int D) {
  return canon_sum_avx([&](int k) { return x[k * sx] - y[k * sy]; }, D);
}
__device__ __forceinline__ float dist2_canon_rows(const float* x, const float* y, int D) { return dist2_canon_rt(x, 1, y, 1, D); }
#else
#define DC_CANON_ORDER_NAME "sse2"
// ---- the default order: compile-time D -----------------------------------------------------------------------------
// q: this lane's query row (registers); r: reference row (registers, wave-uniform values).
// Order of operations = SURVEY.md Appendix B = what the reference binary computes
// (density_clustering.cpp:171-176, 263-268 under -O3 -ffast-math, SSE2).
template <int D>
__device__ __forceinline__ float dist2_canon(const float (&q)[D], const float (&r)[D]) {
  float p[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const float c = q[k] - r[k];
    p[k] = c * c;
  }
  if constexpr (D <= 3) {
    float s = p[0];
#pragma unroll
    for (int k = 1; k < D; ++k) s = s + p[k];
    return s;
  } else {
    constexpr int V = 4 * (D / 4);
    // lane accumulators start at +0; 0 + p == p exactly (p is a square: never -0)
    float a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
#pragma unroll
    for (int k0 = 4; k0 < V; k0 += 4) {
      a0 = a0 + p[k0 + 0];
      a1 = a1 + p[k0 + 1];
      a2 = a2 + p[k0 + 2];
      a3 = a3 + p[k0 + 3];
    }
    float s = (a0 + a2) + (a1 + a3);
    if constexpr (D - V >= 2) {
      s = s + (p[V] + p[V + 1]);
      if constexpr (D - V == 3) s = s + p[V + 2];
    } else if constexpr (D - V == 1) {
      s = s + p[V];
    }
    return s;
  }
}

// run-time D version of the same order; x, y: any addressable rows with strides sx, sy (elements)
__device__ __forceinline__ float dist2_canon_rt(const float* x, int sx, const float* y, int sy,
                                                int D) {
  auto sq = [&](int k) {
    const float c = x[k * sx] - y[k * sy];
    return c * c;
  };
  if (D <= 3) {
    float s = sq(0);
    for (int k = 1; k < D; ++k) s = s + sq(k);
    return s;
  }
  const int V = 4 * (D / 4);
  float a0 = sq(0), a1 = sq(1), a2 = sq(2), a3 = sq(3);
  for (int k0 = 4; k0 < V; k0 += 4) {
    a0 = a0 + sq(k0 + 0);
    a1 = a1 + sq(k0 + 1);
    a2 = a2 + sq(k0 + 2);
    a3 = a3 + sq(k0 + 3);
  }
  float s = (a0 + a2) + (a1 + a3);
  int k = V;
  if (D - k >= 2) {
    s = s + (sq(k) + sq(k + 1));
    k += 2;
  }
  if (D - k == 1) s = s + sq(k);
  return s;
}

// the same for two CONTIGUOUS rows in memory (the deferred exact evaluations of the matrix-core sweeps: one lane per
// pair, a row of the original coordinates each): four columns per load.  Every lane reads another row, so a load
// instruction touches 64 cache lines whatever its width -- a quarter of the instructions is a quarter of the tag
// look-ups (the flushes of the band pairs were 30 % of C5's multi-radius sweep).  Rows are only 4-byte aligned.
// Same order of operations as dist2_canon_rt, bit for bit.
typedef float f32x4_row __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float dist2_canon_rows(const float* x, const float* y, int D) {
  if (D <= 3) return dist2_canon_rt(x, 1, y, 1, D);
  const int V = 4 * (D / 4);
  f32x4_row xv = *reinterpret_cast<const f32x4_row*>(x), yv = *reinterpret_cast<const f32x4_row*>(y);
  float c0 = xv.x - yv.x, c1 = xv.y - yv.y, c2 = xv.z - yv.z, c3 = xv.w - yv.w;
  float a0 = c0 * c0, a1 = c1 * c1, a2 = c2 * c2, a3 = c3 * c3;
  for (int k0 = 4; k0 < V; k0 += 4) {
    xv = *reinterpret_cast<const f32x4_row*>(x + k0);
    yv = *reinterpret_cast<const f32x4_row*>(y + k0);
    c0 = xv.x - yv.x, c1 = xv.y - yv.y, c2 = xv.z - yv.z, c3 = xv.w - yv.w;
    a0 = a0 + c0 * c0;
    a1 = a1 + c1 * c1;
    a2 = a2 + c2 * c2;
    a3 = a3 + c3 * c3;
  }
  float s = (a0 + a2) + (a1 + a3);
  auto sq = [&](int k) {
    const float c = x[k] - y[k];
    return c * c;
  };
  int k = V;
  if (D - k >= 2) {
    s = s + (sq(k) + sq(k + 1));
    k += 2;
  }
  if (D - k == 1) s = s + sq(k);
  return s;
}

#endif   // DC_CANON_AVX

}  // namespace dc
