// test_scale_host.hip -- host-side check of the population sweeps' scale rule (pick_scale_pop,
// dc_mfma_kernels.hpp): for data extents and radii over many orders of magnitude the chosen power of two is the
// LARGEST at which the guard band of the launch is <= 1 (what the two-bit epilogue assumes), the coordinates and
// the folded constant fit fp16 at that scale, and the split into two pieces reproduces a value to 2^-22 (or to
// 2^(-14-g) in absolute terms).  No device code runs: built with hipcc, run by tests/test_capi_symbols.py on CPU.
#include "../../clustering_amd/csrc/dc_mfma_kernels.hpp"

#include <stdio.h>

using namespace dc;

int main() {
  int bad = 0, cases = 0;
  const float Ms[] = {0.0f, 1e-30f, 3e-12f, 1e-4f, 0.37f, 1.3f, 5.0625f, 812.0f, 3e9f, 1e20f, 9e35f};
  const float r2s[] = {0.0f, 1e-38f, 1e-12f, 1e-4f, 0.01f, 0.04f, 0.36f, 2.25f, 1e4f, 1e30f, INFINITY};
  for (int D = 1; D <= 64; D += (D < 16 ? 1 : 7))
    for (float M : Ms)
      for (float r2 : r2s) {
        ++cases;
        const ScaleExp e = pick_scale_pop(M, r2, D);
        const double S = (double)e.c * (double)e.c;
        const double eps = guard_eps_pop(S * (double)M, S * (double)r2, D, e.g, e.a, e.rounded);
        bool ok = e.g == kMidShiftPop && e.a == kConstShiftPop && e.c > 0.0f && e.s2 == (float)S;
        if (e.rounded) {
          // the regular case: the band is at most 1 and the scale within a few percent of the largest such
          // (the sub-linear flush terms are the difference)
          ok = ok && eps <= 1.0 && eps > 0.95;
        } else {
          // the clamps: a power of two; admissible unless the radius is beyond every scale (then thr is capped)
          int ex = 0;
          ok = ok && frexpf(e.c, &ex) == 0.5f;
          if (r2 < INFINITY && M > 0.0f) ok = ok && eps <= 1.0;
        }
        if (eps <= 1.0) {
          // coordinates (A form, and -2x in the B form) and c_q / 2^a fit the fp16 range
          const double xa = sqrt(S * (double)M);
          const double cq = S * (double)M + fmin(S * (double)r2, (double)kThrCapPop) + 1.0;
          ok = ok && 2.0 * xa < 32768.0 && ldexp(cq, -e.a) < 65504.0 && ldexp(65504.0, e.a) > 4.0 * S * (double)M + 2.0;
        }
        if (!ok) {
          ++bad;
          fprintf(stderr, "D %d M %g r2 %g: c %g rounded %d eps %g\n", D, M, r2, e.c, e.rounded, eps);
        }
      }
  // pieces: v = hi + mid 2^-g + rho with |rho| <= max(2^-22 |v|, 2^(-14-g)); the hi 2^-g copy is exact or zero
  const Scale sc = make_scale(ScaleExp{181.0f, 32761.0f, kMidShiftPop, kConstShiftPop, 1});
  uint32_t seed = 12345u;
  for (int i = 0; i < 200000; ++i) {
    seed = seed * 1664525u + 1013904223u;
    const float mag = ldexpf(1.0f + (float)(seed >> 9) * (1.0f / 8388608.0f), (int)(seed % 37u) - 26);   // 2^-26 .. 2^11
    const float v = (seed & 0x100u) ? -mag : mag;
    const Pieces p = split2(v, sc.up, sc.dn);
    const double rec = (double)f16_val(p.hi) + (double)f16_val(p.mid) * (double)sc.dn;
    const double tol = fmax(ldexp(fabs((double)v), -22), ldexp(1.0, -14 - sc.g));
    const double hd = (double)f16_val(p.hi_dn), want = (double)f16_val(p.hi) * (double)sc.dn;
    if (fabs(rec - (double)v) > tol || !(hd == want || (hd == 0.0 && fabs(want) < ldexp(1.0, -14)))) {
      ++bad;
      if (bad < 20) fprintf(stderr, "split2(%g): hi %g mid %g hi_dn %g\n", v, f16_val(p.hi), f16_val(p.mid), f16_val(p.hi_dn));
    }
  }
  printf("scale rule: %d cases, pieces: 200000 values, violations %d\n%s\n", cases, bad, bad ? "FAILED" : "OK");
  return bad ? 1 : 0;
}
