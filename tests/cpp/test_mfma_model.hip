// test_mfma_model.hip -- GPU self-test of the two hardware facts the guard band of the matrix-core
// sweeps rests on (dc_mfma_kernels.hpp, DESIGN.md "guard band"):
//
//  (1) accumulation model of v_mfma_f32_32x32x16_f16:  D = C + sum_k a_k b_k with every addend
//      truncated to a multiple of q = 2^(e_max - 24) (e_max: exponent of the largest |addend|,
//      C included), an exact sum and one final rounding, i.e.
//          | D_hw - D_exact |  <=  17 q + ulp(D_exact) / 2
//      checked on crafted worst cases and on random products over the exponent range of normal fp16
//      operands (the image builder flushes smaller pieces to zero: subnormal inputs are not exact);
//  (2) end to end: the accumulator of the fp16x2 Gram chain (operand images built by the product's
//      own pick_scale_* / slot_value / split2 / gram_chain) stays within the MFMA + dropped-products +
//      flush part of the band of its exact value S (|y'|^2 + c_q - 2 x'.y'), for several dimensions
//      and data scales (from 1e-3 to 1e3: the power-of-two scale S makes them all alike), under both
//      scale rules: the neighbour sweeps' (pieces at the top of the fp16 range) and the population
//      sweeps' (band <= 1: pieces in the middle of the range, scaled mid-piece products); for the
//      latter also that the whole band of the launch is at most 1, as the two-bit epilogue assumes.
//
//  (2b) the folded form of the pruned neighbour sweep (reference norm in the constant slots, C = 0, query norm
//      outside the accumulator): the same band with the folded extra (18 u M per MFMA of small products), and the
//      early-out rule: every final element >= the element after the coarse MFMAs - nn_skip_bound(M).
//
//  (3) ref_credit (the cross-lane reduction of the symmetric population sweep) against a host count on random strings.
//
// Prints a summary and exits 0 when all hold, 1 otherwise.  Built by clustering_amd/csrc/Makefile,
// run by tests/test_gpu_parity.py (pytest -m gpu).
#include "../../clustering_amd/csrc/dc_mfma_kernels.hpp"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

using namespace dc;

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));              \
      return 2;                                                                   \
    }                                                                             \
  } while (0)

// ---- (1) one MFMA on host-supplied operands ----------------------------------------------------
__global__ void one_mfma(const unsigned short* A, const unsigned short* B, const float* C, float* D) {
  // A [32][16], B [16][32] (k-major), C/D [32][32]
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  s16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (short)A[r * 16 + 8 * h + j];
    b[j] = (short)B[(8 * h + j) * 32 + r];
  }
  f32x16 c;
  for (int g = 0; g < 16; ++g) c[g] = C[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r];
  const f32x16 d = mfma16(a, b, c);
  for (int g = 0; g < 16; ++g) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = d[g];
}

// ---- (2) the product's Gram chain on 32 reference x 32 query rows ---------------------------------
template <int NM>
__global__ void gram_tile(const float* ref, const float* qry, uint32_t D, const float* ny,
                          const float* cq, ScaleExp se, float* out) {
  // ref/qry: [32][D] centred and SCALED coordinates (x'' = fl(c x'), what the image builder splits), ny / cq in
  // scaled units; se carries the piece shifts of the rule under test; out [32 ref][32 qry] in scaled units
  const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
  const Scale sc = make_scale(se);
  s16x8 a[NM], b[NM];
  for (int m = 0; m < NM; ++m)
    for (int j = 0; j < 8; ++j) {
      const uint32_t s = 16 * m + 8 * h + j;
      a[m][j] = (short)slot_value(s, D, false, sc, [&](uint32_t k) { return ref[c * D + k]; });
      b[m][j] = (short)slot_value(s, D, true, sc, [&](uint32_t k) { return qry[c * D + k]; });
    }
  const Pieces p = split2(cq[c] * sc.cinv);
  if (h == 0) {
    b[0][0] = (short)p.hi;
    b[0][1] = (short)p.mid;
  }
  f32x16 c0;
  for (int g = 0; g < 16; ++g) c0[g] = ny[(g & 3) + 8 * (g >> 2) + 4 * h];
  const f32x16 acc = gram_chain<NM>(a, b, c0);
  for (int g = 0; g < 16; ++g) out[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + c] = acc[g];
}

static unsigned short f2bf(float f) {  // to fp16: normal values with <= 11 significant bits are exact
  const _Float16 hv = (_Float16)f;
  unsigned short b;
  memcpy(&b, &hv, 2);
  return b;
}
static float bf2f(unsigned short b) {
  _Float16 hv;
  memcpy(&hv, &b, 2);
  return (float)hv;
}
static double ulp_of(long double v) {   // spacing of floats at |v|
  int e;
  frexpl(fabsl(v) > 0 ? v : 1e-300L, &e);   // |v| in [2^(e-1), 2^e)
  return ldexp(1.0, e - 24);
}
static double q_of(long double maxmag) {   // 2^(e_max - 24), 2^e_max <= maxmag
  int e;
  frexpl(maxmag > 0 ? maxmag : 1e-300L, &e);
  return ldexp(1.0, e - 1 - 24);
}

template <int NM>
static int run_gram(bool pop_rule, uint32_t D, float scale, float offset, float thr, double* worst_ratio, int trials) {
  float *d_ref, *d_qry, *d_ny, *d_cq, *d_out;
  CHECK(hipMalloc((void**)&d_ref, 32 * D * 4));
  CHECK(hipMalloc((void**)&d_qry, 32 * D * 4));
  CHECK(hipMalloc((void**)&d_ny, 128));
  CHECK(hipMalloc((void**)&d_cq, 128));
  CHECK(hipMalloc((void**)&d_out, 4096));
  std::vector<float> ref(32 * D), qry(32 * D), ny(32), cq(32), out(1024);
  const double u = ldexp(1.0, -24);
  const int nb = ((int)D + kConstSlots + 15) / 16, ns = nm_for((int)D) - nb;
  int bad = 0;
  for (int t = 0; t < trials; ++t) {
    // two nearby "frames clouds" displaced from the origin (like clusters far from the mean)
    double M = 0;
    for (int i = 0; i < 32; ++i) {
      double n1 = 0, n2 = 0;
      for (uint32_t k = 0; k < D; ++k) {
        const float base = offset * ((k % 3) - 1.0f);
        ref[i * D + k] = base + scale * (float)((rand() % 20001) - 10000) * 1e-4f;
        qry[i * D + k] = base + scale * (float)((rand() % 20001) - 10000) * 1e-4f;
        n1 += (double)ref[i * D + k] * ref[i * D + k];
        n2 += (double)qry[i * D + k] * qry[i * D + k];
      }
      M = fmax(M, fmax(n1, n2));
    }
    const float Mf = (float)M;
    const ScaleExp se = pop_rule ? pick_scale_pop(Mf, thr, (int)D) : pick_scale_nn(Mf);
    const double S = (double)se.c * (double)se.c;
    if (pop_rule) {
      const double eps = guard_eps_pop(S * (double)Mf, S * (double)thr, (int)D, se.g, se.a, se.rounded);
      if (!(eps <= 1.0) || (se.rounded && !(eps > 0.95))) {
        fprintf(stderr, "pick_scale_pop: band %.4f at the chosen scale (D = %u)\n", eps, D);
        ++bad;
      }
    }
    // the scaled data, as the image builder forms it: x'' = fl(c x'), |x''|^2 in double rounded once,
    // thresholds times fl(c c); everything below is in those units
    double Ms = 0;
    for (int i = 0; i < 32; ++i) {
      double n1 = 0, n2 = 0;
      for (uint32_t k = 0; k < D; ++k) {
        ref[i * D + k] = ref[i * D + k] * se.c;
        qry[i * D + k] = qry[i * D + k] * se.c;
        n1 += (double)ref[i * D + k] * ref[i * D + k];
        n2 += (double)qry[i * D + k] * qry[i * D + k];
      }
      ny[i] = (float)n1;
      cq[i] = (float)n2 - thr * se.s2;
      Ms = fmax(Ms, fmax(n1, n2));
    }
    const double thrs = (double)(thr * se.s2);
    CHECK(hipMemcpy(d_ref, ref.data(), 32 * D * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_qry, qry.data(), 32 * D * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_ny, ny.data(), 128, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_cq, cq.data(), 128, hipMemcpyHostToDevice));
    gram_tile<NM><<<1, 64>>>(d_ref, d_qry, D, d_ny, d_cq, se, d_out);
    CHECK(hipMemcpy(out.data(), d_out, 4096, hipMemcpyDeviceToHost));
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        long double dot = 0;
        for (uint32_t k = 0; k < D; ++k) dot += (long double)ref[i * D + k] * (long double)qry[j * D + k];
        const long double E = (long double)ny[i] + (long double)cq[j] - 2.0L * dot;
        const double absE = (double)fabsl(E);
        // MFMA + c_q pieces + dropped products + flush part of the band, in scaled units (no 1.25
        // factor, no centring / canonical / scaling-rounding terms)
        const double bound = u * (4.1 * (Ms + thrs) + 27.0 * Ms + 17.0 * (2.0 * Ms + thrs) +
                                  (nb - 1) * 18.0 * (4.02 * Ms + thrs) + ns * 18.0 * (absE + 0.004 * Ms) + absE) +
                             guard_flush(Ms, (int)D, se.g, se.a) / 1.25;
        const double err = (double)fabsl((long double)out[i * 32 + j] - E);
        if (err / bound > *worst_ratio) *worst_ratio = err / bound;
        if (err > bound) ++bad;
      }
  }
  (void)hipFree(d_ref); (void)hipFree(d_qry); (void)hipFree(d_ny); (void)hipFree(d_cq); (void)hipFree(d_out);
  return bad;
}

// ---- (2b) the folded form: acc' = |y'|^2 - 2 x'.y' from C = 0; coarse = the value after kNnCoarse<NM> MFMAs ---------
template <int NM>
__global__ void gram_tile_folded(const float* ref, const float* qry, uint32_t D, const float* ny, ScaleExp se,
                                 float* out, float* out_coarse) {
  const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
  const Scale sc = make_scale(se);
  s16x8 a[NM], b[NM];
  for (int m = 0; m < NM; ++m)
    for (int j = 0; j < 8; ++j) {
      const uint32_t s = 16 * m + 8 * h + j;
      a[m][j] = (short)slot_value(s, D, false, sc, [&](uint32_t k) { return ref[c * D + k]; });
      b[m][j] = (short)slot_value(s, D, true, sc, [&](uint32_t k) { return qry[c * D + k]; });
    }
  if (h == 0) {   // (as image_kernel mode 2 and load_query_folded)
    const Pieces p = split2(ny[c] * sc.cinv);
    a[0][0] = (short)p.hi;
    a[0][1] = (short)p.mid;
    b[0][0] = (short)const_a_bits(sc.a);
    b[0][1] = (short)const_a_bits(sc.a);
  }
  f32x16 acc;
  for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
  for (int m = 0; m < NM; ++m) {
    acc = mfma16(a[m], b[m], acc);
    if (m + 1 == kNnCoarse<NM> || (NM < kNnCoarse<NM> && m + 1 == NM))
      for (int g = 0; g < 16; ++g) out_coarse[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + c] = acc[g];
  }
  for (int g = 0; g < 16; ++g) out[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + c] = acc[g];
}

template <int NM>
static int run_gram_folded(uint32_t D, float scale, float offset, double* worst_ratio, double* worst_skip, int trials) {
  float *d_ref, *d_qry, *d_ny, *d_out, *d_outc;
  CHECK(hipMalloc((void**)&d_ref, 32 * D * 4));
  CHECK(hipMalloc((void**)&d_qry, 32 * D * 4));
  CHECK(hipMalloc((void**)&d_ny, 128));
  CHECK(hipMalloc((void**)&d_out, 4096));
  CHECK(hipMalloc((void**)&d_outc, 4096));
  std::vector<float> ref(32 * D), qry(32 * D), ny(32), out(1024), outc(1024);
  const double u = ldexp(1.0, -24);
  const int nb = ((int)D + kConstSlots + 15) / 16, ns = nm_for((int)D) - nb;
  int bad = 0;
  for (int t = 0; t < trials; ++t) {
    double M = 0;
    for (int i = 0; i < 32; ++i) {
      double n1 = 0, n2 = 0;
      for (uint32_t k = 0; k < D; ++k) {
        const float base = offset * ((k % 3) - 1.0f);
        ref[i * D + k] = base + scale * (float)((rand() % 20001) - 10000) * 1e-4f;
        // (every other trial: queries next to the references -- small d2 against a large |x'|^2)
        qry[i * D + k] = (t & 1) ? ref[((i * 7) & 31) * D + k] + 1e-3f * scale * (float)((rand() % 2001) - 1000) * 1e-3f
                                 : base + scale * (float)((rand() % 20001) - 10000) * 1e-4f;
        n1 += (double)ref[i * D + k] * ref[i * D + k];
        n2 += (double)qry[i * D + k] * qry[i * D + k];
      }
      M = fmax(M, fmax(n1, n2));
    }
    const ScaleExp se = pick_scale_nn((float)M);
    double Ms = 0;
    for (int i = 0; i < 32; ++i) {
      double n1 = 0, n2 = 0;
      for (uint32_t k = 0; k < D; ++k) {
        ref[i * D + k] = ref[i * D + k] * se.c;
        qry[i * D + k] = qry[i * D + k] * se.c;
        n1 += (double)ref[i * D + k] * ref[i * D + k];
        n2 += (double)qry[i * D + k] * qry[i * D + k];
      }
      ny[i] = (float)n1;
      Ms = fmax(Ms, fmax(n1, n2));
    }
    CHECK(hipMemcpy(d_ref, ref.data(), 32 * D * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_qry, qry.data(), 32 * D * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_ny, ny.data(), 128, hipMemcpyHostToDevice));
    gram_tile_folded<NM><<<1, 64>>>(d_ref, d_qry, D, d_ny, se, d_out, d_outc);
    CHECK(hipMemcpy(out.data(), d_out, 4096, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(outc.data(), d_outc, 4096, hipMemcpyDeviceToHost));
    const double skipb = (double)Ms * (0.001953125 * 1.01 + 1.52587890625e-05);   // nn_skip_bound
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        long double dot = 0;
        for (uint32_t k = 0; k < D; ++k) dot += (long double)ref[i * D + k] * (long double)qry[j * D + k];
        const long double E = (long double)ny[i] - 2.0L * dot;   // = d2 - |x'|^2
        const double absE = (double)fabsl(E);
        const double bound = u * (4.1 * Ms + 27.0 * Ms + 17.0 * (2.0 * Ms) + (nb - 1) * 18.0 * (4.02 * Ms) +
                                  ns * 18.0 * (absE + 0.004 * Ms) + absE) +
                             guard_flush(Ms, (int)D, se.g, se.a) / 1.25;
        const double err = (double)fabsl((long double)out[i * 32 + j] - E);
        if (err / bound > *worst_ratio) *worst_ratio = err / bound;
        if (err > bound) ++bad;
        // early-out: the final element is never more than the skip bound below the coarse one
        // (where the kernel uses it: the coarse MFMAs hold the constant and every hi x hi product)
        if ((int)D + kConstSlots <= 16 * kNnCoarse<NM>) {
          const double drop = (double)outc[i * 32 + j] - (double)out[i * 32 + j];
          if (drop / skipb > *worst_skip) *worst_skip = drop / skipb;
          if (drop > skipb) ++bad;
        }
      }
  }
  (void)hipFree(d_ref); (void)hipFree(d_qry); (void)hipFree(d_ny); (void)hipFree(d_out); (void)hipFree(d_outc);
  return bad;
}

// ---- (3) the reference-side reduction of the symmetric population sweep ---------------------------------------
// ref_credit on random strings: row i of the tile must be credited with the number of (query tile, lane of the
// row's half-wave) whose string has the sign bit of the row's element set.
template <int TQ>
__global__ void credit_probe(const uint32_t* strings /* [TQ][64] */, uint32_t* counts /* [32] */) {
  __shared__ uint32_t stage[8];
  const int lane = threadIdx.x;
  uint32_t sb[TQ];
  for (int q = 0; q < TQ; ++q) sb[q] = strings[q * 64 + lane];
  ref_credit<TQ>(sb, 0u, 32u, counts, stage, ref_credit_byte(lane), lane);
}
template <int TQ>
static int run_credit(int trials) {
  uint32_t *d_s, *d_c;
  CHECK(hipMalloc((void**)&d_s, TQ * 64 * 4));
  CHECK(hipMalloc((void**)&d_c, 128));
  std::vector<uint32_t> st(TQ * 64), got(32);
  int bad = 0;
  for (int t = 0; t < trials; ++t) {
    // dense, sparse and all-ones strings; the flag bits (even positions) are noise the reduction must ignore
    const int density = t % 4;
    for (auto& v : st) {
      uint32_t w = 0;
      for (int b = 0; b < 32; ++b) {
        const int r = rand() & 255;
        const bool on = density == 0 ? (r < 128) : density == 1 ? (r < 8) : density == 2 ? true : (r < 250);
        w |= (uint32_t)on << b;
      }
      v = w;
    }
    CHECK(hipMemcpy(d_s, st.data(), TQ * 64 * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_c, 0, 128));
    credit_probe<TQ><<<1, 64>>>(d_s, d_c);
    CHECK(hipMemcpy(got.data(), d_c, 128, hipMemcpyDeviceToHost));
    for (int row = 0; row < 32; ++row) {
      // element r of half h is row (r & 3) + 8 (r >> 2) + 4 h; its sign sits at bit 31 - 2 r
      const int hh = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
      uint32_t want = 0;
      for (int q = 0; q < TQ; ++q)
        for (int l = 32 * hh; l < 32 * hh + 32; ++l) want += (st[q * 64 + l] >> (31 - 2 * r)) & 1u;
      if (got[row] != want) ++bad;
    }
  }
  (void)hipFree(d_s); (void)hipFree(d_c);
  return bad;
}

int main() {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
    fprintf(stderr, "no HIP device\n");
    return 2;
  }
  unsigned short *dA, *dB;
  float *dC, *dD;
  CHECK(hipMalloc((void**)&dA, 1024));
  CHECK(hipMalloc((void**)&dB, 1024));
  CHECK(hipMalloc((void**)&dC, 4096));
  CHECK(hipMalloc((void**)&dD, 4096));
  std::vector<unsigned short> A(512), B(512);
  std::vector<float> C(1024), D(1024);
  int failures = 0;
  double worst_q = 0;   // worst (error - ulp/2) in units of q
  srand(20240);
  for (int E : {0, 4, 8, 12}) {   // (operands stay normal fp16: 2^-12 * [1, 2) at the least)
    for (int trial = 0; trial < 300; ++trial) {
      const bool crafted = (E == 0);
      for (int r = 0; r < 32; ++r)
        for (int k = 0; k < 16; ++k) {
          float v;
          if (crafted) {
            // every product just below one truncation unit of C = +-1 (or a few units): 255/128 * 2^-25
            // and friends; row-dependent sign patterns
            // (the products 2^-25 .. 2^-27 come from A = m 2^-12 .. 2^-14 and B = 2^-13)
            const float m = 1.0f + (float)(512 + (rand() & 511)) / 1024.0f;   // [1.5, 2)
            v = ((r + (trial & 1) * k) & 1 ? -m : m) * ldexpf(1.0f, -12 - (rand() % 3));
          } else {
            const float m = 1.0f + (rand() & 1023) / 1024.0f;
            v = ((rand() & 1) ? -m : m) * ldexpf(1.0f, -(rand() % (E + 1)));
          }
          A[r * 16 + k] = f2bf(v);
        }
      for (int k = 0; k < 16; ++k)
        for (int j = 0; j < 32; ++j) {
          float v;
          if (crafted) {
            v = ldexpf(1.0f, -13);
          } else {
            const float m = 1.0f + (rand() & 1023) / 1024.0f;
            v = ((rand() & 1) ? -m : m) * ldexpf(1.0f, -(rand() % (E + 1)));
          }
          B[k * 32 + j] = f2bf(v);
        }
      for (auto& v : C)
        v = crafted ? ((rand() & 1) ? -1.0f : 1.0f)
                    : ((rand() & 1) ? -1.0f : 1.0f) * (1.0f + (rand() & 0xffff) / 65536.0f) *
                          ldexpf(1.0f, -(rand() % (E + 1)));
      CHECK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice));
      one_mfma<<<1, 64>>>(dA, dB, dC, dD);
      CHECK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
      for (int r = 0; r < 32; ++r)
        for (int j = 0; j < 32; ++j) {
          long double ex = C[r * 32 + j], mx = fabsl(ex);
          for (int k = 0; k < 16; ++k) {
            const long double p = (long double)bf2f(A[r * 16 + k]) * (long double)bf2f(B[k * 32 + j]);
            ex += p;
            if (fabsl(p) > mx) mx = fabsl(p);
          }
          const double err = (double)fabsl((long double)D[r * 32 + j] - ex);
          const double q = q_of(mx), half_ulp = 0.5 * ulp_of(ex);
          const double in_q = (err - half_ulp) / q;
          if (in_q > worst_q) worst_q = in_q;
          if (err > 17.0 * q + half_ulp * 1.0000001) ++failures;
        }
    }
  }
  printf("mfma accumulate model: worst (error - ulp/2) = %.3f q  (bound 17 q), violations %d\n", worst_q,
         failures);

  double worst_ratio = 0;
  int bad = 0;
  for (int rule = 0; rule < 2; ++rule) {
    const bool pr = rule == 1;
    bad += run_gram<nm_for(2)>(pr, 2, 1.0f, 0.0f, 0.01f, &worst_ratio, 40);
    bad += run_gram<nm_for(3)>(pr, 3, 0.3f, 1.0f, 0.04f, &worst_ratio, 40);
    bad += run_gram<nm_for(10)>(pr, 10, 0.1f, 0.5f, 0.04f, &worst_ratio, 60);
    bad += run_gram<nm_for(10)>(pr, 10, 0.01f, 3.0f, 0.0004f, &worst_ratio, 60);
    bad += run_gram<nm_for(10)>(pr, 10, 100.0f, 1000.0f, 2500.0f, &worst_ratio, 40);
    bad += run_gram<nm_for(14)>(pr, 14, 0.1f, 0.5f, 0.04f, &worst_ratio, 40);
    bad += run_gram<nm_for(15)>(pr, 15, 0.1f, 0.5f, 0.04f, &worst_ratio, 40);
    bad += run_gram<nm_for(10)>(pr, 10, 1e-3f, 0.0f, 4e-6f, &worst_ratio, 40);
    bad += run_gram<nm_for(30)>(pr, 30, 0.1f, 0.5f, 0.09f, &worst_ratio, 40);
    bad += run_gram<nm_for(32)>(pr, 32, 1e-3f, 1e-2f, 1e-5f, &worst_ratio, 40);
    bad += run_gram<nm_for(48)>(pr, 48, 0.1f, 0.5f, 0.3f, &worst_ratio, 30);
    bad += run_gram<nm_for(64)>(pr, 64, 0.05f, 2.0f, 0.2f, &worst_ratio, 30);
    // radius far beyond / far below the extent of the data, tiny and huge coordinates
    bad += run_gram<nm_for(10)>(pr, 10, 0.1f, 0.5f, 400.0f, &worst_ratio, 20);
    bad += run_gram<nm_for(10)>(pr, 10, 0.1f, 0.5f, 1e-9f, &worst_ratio, 20);
    bad += run_gram<nm_for(5)>(pr, 5, 1e-12f, 1e-11f, 1e-24f, &worst_ratio, 20);
    bad += run_gram<nm_for(5)>(pr, 5, 1e10f, 3e10f, 1e20f, &worst_ratio, 20);
    printf("fp16x2 gram chain, %s scale: worst error / (MFMA + dropped-product + flush part of the band) = %.3f, violations %d\n",
           pr ? "population" : "neighbour", worst_ratio, bad);
  }
  {
    double wr = 0, ws = 0;
    int bf = 0;
    bf += run_gram_folded<nm_for(2)>(2, 1.0f, 0.0f, &wr, &ws, 40);
    bf += run_gram_folded<nm_for(3)>(3, 0.3f, 1.0f, &wr, &ws, 40);
    bf += run_gram_folded<nm_for(5)>(5, 1e-12f, 1e-11f, &wr, &ws, 20);
    bf += run_gram_folded<nm_for(5)>(5, 1e10f, 3e10f, &wr, &ws, 20);
    bf += run_gram_folded<nm_for(10)>(10, 0.1f, 0.5f, &wr, &ws, 60);
    bf += run_gram_folded<nm_for(10)>(10, 0.01f, 3.0f, &wr, &ws, 60);
    bf += run_gram_folded<nm_for(10)>(10, 100.0f, 1000.0f, &wr, &ws, 40);
    bf += run_gram_folded<nm_for(14)>(14, 0.1f, 0.5f, &wr, &ws, 40);
    bf += run_gram_folded<nm_for(15)>(15, 0.1f, 0.5f, &wr, &ws, 40);
    bf += run_gram_folded<nm_for(20)>(20, 0.1f, 2.0f, &wr, &ws, 40);
    bf += run_gram_folded<nm_for(30)>(30, 0.1f, 0.5f, &wr, &ws, 40);
    bf += run_gram_folded<nm_for(64)>(64, 0.05f, 2.0f, &wr, &ws, 30);
    printf("folded neighbour chain: worst error / band part = %.3f, worst (coarse - final) / skip bound = %.3f, violations %d\n",
           wr, ws, bf);
    bad += bf;
  }
  const int bad_credit = run_credit<2>(40) + run_credit<4>(40) + run_credit<6>(40);
  printf("reference-side reduction of the symmetric sweep (2, 4, 6 strings): violations %d\n", bad_credit);
  bad += bad_credit;
  // ---- (5) thresholds taken off the accumulator in place (dc_mfma_msym.hpp): `steps` MFMAs of ones x (three fp16 pieces of
  //      -delta) on accumulators whose FINAL value lies near the band; bound = guard_shift without its factor 1.25
  {
    int bad_shift = 0;
    double worst_shift = 0;
    for (int trial = 0; trial < 400; ++trial) {
      const int steps = (trial & 1) ? 7 : 3;
      const float span = ldexpf(1.0f + (rand() & 1023) / 1024.0f, 2 + rand() % 14);   // 4 ... 65 000 scaled units
      float delta[8];
      double sum = 0;
      for (int k = 0; k < steps; ++k) {
        delta[k] = span * (0.02f + (rand() & 1023) / 1024.0f) / (float)steps;
        if (trial % 5 == 0 && k == 1) delta[k] = -delta[k];   // (radii in any order)
      }
      // pieces exactly as the kernel forms them
      unsigned short pc[8][3];
      double held[8];
      for (int k = 0; k < steps; ++k) {
        const float x = -delta[k];
        auto fz = [](float v) { return fabsf(v) < 6.103515625e-5f ? 0.0f : v; };
        const float p0 = fz(bf2f(f2bf(x)));
        const float r1 = x - p0;
        const float p1 = fz(bf2f(f2bf(r1)));
        const float p2 = fz(bf2f(f2bf(r1 - p1)));
        pc[k][0] = f2bf(p0); pc[k][1] = f2bf(p1); pc[k][2] = f2bf(p2);
        held[k] = (double)p0 + (double)p1 + (double)p2;
        sum += (double)x;
      }
      for (auto& v : C) v = (float)(-sum) + 8.0f * ((rand() & 0xffff) / 65536.0f - 0.5f);   // final value in [-4, 4)
      std::vector<float> C0 = C;
      for (int r = 0; r < 32; ++r)
        for (int k = 0; k < 16; ++k) A[r * 16 + k] = f2bf(k < 3 ? 1.0f : 0.0f);
      for (int k = 0; k < steps; ++k) {
        for (int kk = 0; kk < 16; ++kk)
          for (int j = 0; j < 32; ++j) B[kk * 32 + j] = (kk < 3) ? pc[k][kk] : f2bf(0.0f);
        CHECK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice));
        one_mfma<<<1, 64>>>(dA, dB, dC, dD);
        CHECK(hipMemcpy(C.data(), dD, 4096, hipMemcpyDeviceToHost));
      }
      const double u = ldexp(1.0, -24);
      double span_abs = 0, run = 0, run_max = 0;
      for (int k = 0; k < steps; ++k) { run += fabs((double)delta[k]); run_max = run; }
      span_abs = run_max;
      const double bound = steps * (4.5 * u * (span_abs + 4.0) + 3.0 * ldexp(1.0, -14));
      for (int e = 0; e < 1024; ++e) {
        const double err = fabs((double)C[e] - ((double)C0[e] + sum));
        if (err / bound > worst_shift) worst_shift = err / bound;
        if (err > bound) ++bad_shift;
      }
      (void)held;
    }
    printf("in-place threshold shifts (3 and 7 steps): worst error / bound = %.3f, violations %d\n", worst_shift, bad_shift);
    bad += bad_shift;
  }
  const bool ok = failures == 0 && bad == 0;
  printf("%s\n", ok ? "OK" : "FAILED");
  return ok ? 0 : 1;
}
