/*
 * fastmath_probe.cpp -- TEST INFRASTRUCTURE (never linked into the product).
 *
 * Own code, written from scratch: the LOOP SHAPE of the reference's distance
 * reduction (density_clustering.cpp:171-176 and :263-268: `dist = 0; for k:
 * c = a[i*n+k] - a[j*n+k]; dist += c*c;` on a 32-byte-aligned row-major float
 * matrix, n_cols known only at run time) and of its free-energy line
 * (density_clustering.cpp:201-209: `(float) -1.0f * log(pops[i]/max_pop)` with
 * size_t pops and float max_pop), compiled by the same g++ with the reference's
 * own flags (CMakeLists.txt:37-45: -std=c++11 -O3 -ftree-vectorize -ffast-math
 * -fopenmp).  Under -ffast-math the summation order is the compiler's choice;
 * tests/test_oracle.py checks that dc_oracle.c's explicit "canonical" order is
 * bitwise what gcc produces for this loop shape, for every D in 1..40.
 *
 * This corroborates SURVEY.md Appendix B; it is NOT a build of the reference
 * (which needs Boost and a cmake-generated header and is unbuildable here).
 */
#include <cmath>
#include <cstddef>
#include <limits>

extern "C" {

// all-pairs squared distances of rows [0,n_rows): out[i*n_rows+j], nested like
// the reference's NN sweep (outer i, inner j, innermost k reduction).
__attribute__((visibility("default"))) void
probe_pairwise_d2(const float* coords, std::size_t n_rows, std::size_t n_cols, float* out) {
  coords = (const float*)__builtin_assume_aligned(coords, 32);
  std::size_t i, j, c;
  float dist, d;
  #pragma omp parallel for default(none) private(i, j, c, dist, d) \
      firstprivate(n_rows, n_cols) shared(coords, out) schedule(dynamic, 64)
  for (i = 0; i < n_rows; ++i) {
    for (j = 0; j < n_rows; ++j) {
      if (i != j) {
        dist = 0.0f;
        for (c = 0; c < n_cols; ++c) {
          d = coords[i * n_cols + c] - coords[j * n_cols + c];
          dist += d * d;
        }
        out[i * n_rows + j] = dist;
      } else {
        out[i * n_rows + j] = 0.0f;
      }
    }
  }
}

__attribute__((visibility("default"))) void
probe_free_energies(const std::size_t* pops, std::size_t n_frames, float* fe) {
  std::size_t mx = 0;
  for (std::size_t i = 0; i < n_frames; ++i) if (pops[i] > mx) mx = pops[i];
  const float max_pop = (float)mx;
  std::size_t i;
  #pragma omp parallel for default(none) private(i) firstprivate(max_pop, n_frames) \
      shared(fe, pops)
  for (i = 0; i < n_frames; ++i) {
    fe[i] = (float)-1.0f * log(pops[i] / max_pop);
  }
}

// box index as the reference's grid computes it: (x - min) / radius -> int
__attribute__((visibility("default"))) void
probe_box_index(const float* x, std::size_t n, float min_x, float radius, int* out) {
  for (std::size_t i = 0; i < n; ++i) {
    int b = 0;
    b = (x[i] - min_x) / radius;
    out[i] = b;
  }
}

}  // extern "C"
