/*
 * dc_oracle.c -- CPU restatement of the `clustering density` hot path of
 * moldyn/Clustering (population count -> free energy -> nearest neighbours).
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (clustering_amd/csrc, libdcdensity.so) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" in the bitwise sense.  The reference ships no
 * tests, golden vectors or fixtures for this path, and its sources cannot be
 * compiled in this image without stand-ins (density_clustering.hpp:35 needs
 * Boost.Program_options, tools.hpp needs the cmake-generated config.hpp), so no
 * reference-produced vectors exist to pin against.  What this oracle IS checked
 * against (tests/test_oracle.py):
 *   - hand-computable known-answer cases (tests/golden/kat_*.json);
 *   - oracle/fastmath_probe.c: the reference's loop SHAPE (not its source)
 *     compiled by the same g++ 11.4 with the reference's own flags
 *     (CMakeLists.txt:37-45: -O3 -ftree-vectorize -ffast-math), which reproduces
 *     the summation order gcc gives that loop -- bitwise equal to dco_dist2();
 *   - the statistics of the reference's own run recorded in BASELINE.md section 2
 *     (C1/C2: mean/max population per radius; C3: sigma^2) on the seeded
 *     generator of SURVEY.md section 8(d).
 *
 * Arithmetic follows SURVEY.md Appendix B ("canonical arithmetic"):
 *   rad2   = fl32(r*r)                                   density_clustering.cpp:137-140
 *   p_k    = fl32(fl32(x_k-y_k)^2), no FMA               density_clustering.cpp:173-176, 265-268
 *   d2     = SSE2 4-lane partial sums, (a0+a2)+(a1+a3), pair tail, scalar tail
 *   pop    = 1 + #{j != i : d2 < rad2}  (strict)         density_clustering.cpp:132-134, 177-188
 *   fe     = fl32(-log_f64(f64(fl32(fl32(pop)*fl32(1/max_pop)))))   :197-212 as compiled
 *   nn     = lexicographic min over j != i of (d2, j)    density_clustering.cpp:256-273
 *   nn_hd  = same over {j : fe[j] < fe[i]}; none -> (N+1, FLT_MAX)  :242-245, 275-279
 *   sigma2 = f64 sum of nn d2 in frame order / N         density_clustering.cpp:334-343
 *
 * Build: see oracle/Makefile.  Canonical build MUST NOT use -ffast-math or FMA
 * contraction (-ffp-contract=off); the DCO_FAST build (cpu_baseline timing only)
 * may.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DCO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* canonical squared distance                          */
/* ------------------------------------------------------------------------- */
#ifndef DCO_FAST
#if defined(__FAST_MATH__)
#error "canonical oracle build must not use -ffast-math (order of float ops is the spec)"
#endif
static inline float dco_sq(float x, float y) {
  const float c = x - y; /* separate sub and mul; build uses -ffp-contract=off */
  return c * c;
}

#if defined(DCO_CANON_FMA) && !defined(DCO_CANON_AVX)
#define DCO_CANON_AVX 1 /* the FMA order has the AVX order's shape */
#endif
#ifdef DCO_CANON_AVX
/* one more column on a lane sum or on the scalar tail: -mavx two roundings; a -march=native build on an AVX2 + FMA host
 * (CMakeLists.txt:53-56; g++'s generic / Intel tunings) ONE -- vfmadd231ps in the eight-lane loop, vfmadd231ss in the
 * tail, the four-column step unfused: read off the probe built with -mavx2 -mfma and pinned against it. */
#ifdef DCO_CANON_FMA
#define DCO_ACC(a, x, y) fmaf((x) - (y), (x) - (y), (a))
#else
#define DCO_ACC(a, x, y) ((a) + dco_sq((x), (y)))
#endif
/* The order of an AVX build of the reference (CMakeLists.txt:73-76: -DCPU_ACCELERATION=AVX adds -mavx): eight lane sums,
 * b_i = a_i + a_{i+4}, (b0 + b2) + (b1 + b3); a four-column step (q0 + q2) + (q1 + q3) added to that if four or more
 * columns remain; then up to three scalar additions.  Read off what g++ emits for the reference's loop shape under its
 * own flags plus -mavx (fastmath_probe.cpp) and pinned against that build in tests/test_oracle.py. */
static inline float dist2_canonical(const float* x, const float* y, size_t D) {
  float s = 0.0f;
  size_t k = 0;
  const size_t V8 = 8 * (D / 8);
  if (V8 != 0) {
    float a[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (size_t k0 = 0; k0 < V8; k0 += 8)
      for (int l = 0; l < 8; ++l) a[l] = DCO_ACC(a[l], x[k0 + l], y[k0 + l]);
    const float b0 = a[0] + a[4], b1 = a[1] + a[5], b2 = a[2] + a[6], b3 = a[3] + a[7];
    s = (b0 + b2) + (b1 + b3);
    k = V8;
  }
  if (D - k >= 4) {
    const float t = (dco_sq(x[k], y[k]) + dco_sq(x[k + 2], y[k + 2])) + (dco_sq(x[k + 1], y[k + 1]) + dco_sq(x[k + 3], y[k + 3]));
    s = s + t;
    k += 4;
  }
  for (; k < D; ++k) s = DCO_ACC(s, x[k], y[k]);
  return s;
}
#else
static inline float dist2_canonical(const float* x, const float* y, size_t D) {
  if (D <= 3) {
    float s = dco_sq(x[0], y[0]);
    for (size_t k = 1; k < D; ++k) s = s + dco_sq(x[k], y[k]);
    return s;
  }
  const size_t V = 4 * (D / 4);
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  for (size_t k0 = 0; k0 < V; k0 += 4) {
    a0 = a0 + dco_sq(x[k0 + 0], y[k0 + 0]);
    a1 = a1 + dco_sq(x[k0 + 1], y[k0 + 1]);
    a2 = a2 + dco_sq(x[k0 + 2], y[k0 + 2]);
    a3 = a3 + dco_sq(x[k0 + 3], y[k0 + 3]);
  }
  float s = (a0 + a2) + (a1 + a3);
  size_t k = V;
  if (D - k >= 2) {
    float t = dco_sq(x[k], y[k]) + dco_sq(x[k + 1], y[k + 1]);
    s = s + t;
    k += 2;
  }
  if (D - k == 1) s = s + dco_sq(x[k], y[k]);
  return s;
}
#endif /* DCO_CANON_AVX */
#else
/* timing-only build: the reference's loop shape, compiler picks the order */
static inline float dist2_canonical(const float* x, const float* y, size_t D) {
  float dist = 0.0f;
  for (size_t k = 0; k < D; ++k) {
    float c = x[k] - y[k];
    dist += c * c;
  }
  return dist;
}
#endif

DCO_API float dco_dist2(const float* x, const float* y, size_t n_cols) {
  return dist2_canonical(x, y, n_cols);
}

DCO_API int dco_is_fast_build(void) {
#ifdef DCO_FAST
  return 1;
#else
  return 0;
#endif
}

DCO_API int dco_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* populations, brute force: the defining formula.                            */
/* pops is [n_radii][n_rows] in the order of the radii argument; rows outside */
/* [i_from, i_to) are left 0 (the per-GPU partial of cuda.cu:45-137).         */
/* ------------------------------------------------------------------------- */
DCO_API void dco_populations_brute(const float* coords, size_t n_rows, size_t n_cols,
                                   const float* radii, size_t n_radii, size_t i_from,
                                   size_t i_to, uint64_t* pops) {
  float* rad2 = (float*)malloc(sizeof(float) * (n_radii ? n_radii : 1));
  for (size_t r = 0; r < n_radii; ++r) rad2[r] = radii[r] * radii[r];
  memset(pops, 0, sizeof(uint64_t) * n_radii * n_rows);
  if (i_to > n_rows) i_to = n_rows;
#pragma omp parallel for schedule(dynamic, 64)
  for (size_t i = i_from; i < i_to; ++i) {
    const float* xi = coords + i * n_cols;
    for (size_t r = 0; r < n_radii; ++r) pops[r * n_rows + i] = 1; /* self, :132-134 */
    for (size_t j = 0; j < n_rows; ++j) {
      if (j == i) continue;
      const float d = dist2_canonical(xi, coords + j * n_cols, n_cols);
      for (size_t r = 0; r < n_radii; ++r)
        if (d < rad2[r]) pops[r * n_rows + i] += 1;
    }
  }
  free(rad2);
}

/* ------------------------------------------------------------------------- */
/* populations, reference-shaped: 2-D box grid on columns 0/1 with cell =     */
/* largest radius, i<j symmetric counting, radii descending with early break. */
/* density_clustering.cpp:41-89 (grid), :126-195 (count).                     */
/* ------------------------------------------------------------------------- */
typedef struct {
  int nb[2];
  int* box_of;      /* [n_rows][2] */
  size_t* start;    /* [nb0*nb1 + 1] CSR offsets */
  int* members;     /* [n_rows] frame ids, ascending inside each box */
} dco_grid;

static void grid_build(dco_grid* g, const float* coords, size_t n_rows, size_t n_cols,
                       float radius) {
  float min1 = coords[0], max1 = coords[0], min2 = 0.0f, max2 = 0.0f;
  if (n_cols > 1) min2 = max2 = coords[1];
  for (size_t i = 1; i < n_rows; ++i) {
    const float a = coords[i * n_cols];
    if (a < min1) min1 = a;
    if (a > max1) max1 = a;
    if (n_cols > 1) {
      const float b = coords[i * n_cols + 1];
      if (b < min2) min2 = b;
      if (b > max2) max2 = b;
    }
  }
  g->nb[0] = (int)((max1 - min1) / radius + 1);          /* :72 */
  g->nb[1] = (n_cols > 1) ? (int)((max2 - min2) / radius + 1) : 1; /* :73-77 */
  const size_t nbox = (size_t)g->nb[0] * (size_t)g->nb[1];
  g->box_of = (int*)malloc(sizeof(int) * 2 * n_rows);
  g->start = (size_t*)calloc(nbox + 1, sizeof(size_t));
  g->members = (int*)malloc(sizeof(int) * n_rows);
  for (size_t i = 0; i < n_rows; ++i) {
    int b1 = (int)((coords[i * n_cols] - min1) / radius);               /* :81 */
    int b2 = (n_cols > 1) ? (int)((coords[i * n_cols + 1] - min2) / radius) : 0; /* :83 */
    g->box_of[2 * i] = b1;
    g->box_of[2 * i + 1] = b2;
    g->start[(size_t)b1 * g->nb[1] + b2 + 1] += 1;
  }
  for (size_t b = 0; b < nbox; ++b) g->start[b + 1] += g->start[b];
  size_t* fill = (size_t*)malloc(sizeof(size_t) * (nbox + 1));
  memcpy(fill, g->start, sizeof(size_t) * (nbox + 1));
  for (size_t i = 0; i < n_rows; ++i) {
    const size_t b = (size_t)g->box_of[2 * i] * g->nb[1] + g->box_of[2 * i + 1];
    g->members[fill[b]++] = (int)i;
  }
  free(fill);
}

static void grid_free(dco_grid* g) {
  free(g->box_of);
  free(g->start);
  free(g->members);
}

static int cmp_float_desc(const void* a, const void* b) {
  const float x = *(const float*)a, y = *(const float*)b;
  return (x < y) - (x > y);
}

DCO_API void dco_populations_boxgrid(const float* coords, size_t n_rows, size_t n_cols,
                                     const float* radii_in, size_t n_radii,
                                     uint64_t* pops) {
  /* outputs stay in the caller's radius order; the sweep uses descending order */
  float* radii = (float*)malloc(sizeof(float) * n_radii);
  size_t* slot = (size_t*)malloc(sizeof(size_t) * n_radii);
  memcpy(radii, radii_in, sizeof(float) * n_radii);
  qsort(radii, n_radii, sizeof(float), cmp_float_desc);   /* :135 */
  for (size_t l = 0; l < n_radii; ++l) {
    slot[l] = 0;
    for (size_t r = 0; r < n_radii; ++r)
      if (radii_in[r] == radii[l]) { slot[l] = r; break; } /* std::map key semantics */
  }
  float* rad2 = (float*)malloc(sizeof(float) * n_radii);
  for (size_t l = 0; l < n_radii; ++l) rad2[l] = radii[l] * radii[l]; /* :137-140 */
  for (size_t k = 0; k < n_radii * n_rows; ++k) pops[k] = 1;          /* :132-134 */
  dco_grid g;
  grid_build(&g, coords, n_rows, n_cols, radii[0]);                    /* :143 */
  static const int BOX_DIFF[9][2] = {{-1, 1}, {0, 1},  {1, 1},  {-1, 0}, {0, 0},
                                     {1, 0},  {-1, -1}, {0, -1}, {1, -1}};
#pragma omp parallel for schedule(dynamic, 1024)
  for (size_t i = 0; i < n_rows; ++i) {
    const float* xi = coords + i * n_cols;
    for (int nbi = 0; nbi < 9; ++nbi) {
      const int b1 = g.box_of[2 * i] + BOX_DIFF[nbi][0];
      const int b2 = g.box_of[2 * i + 1] + BOX_DIFF[nbi][1];
      if (b1 < 0 || b1 >= g.nb[0] || b2 < 0 || b2 >= g.nb[1]) continue; /* :97-105 */
      const size_t b = (size_t)b1 * g.nb[1] + b2;
      for (size_t m = g.start[b]; m < g.start[b + 1]; ++m) {
        const size_t j = (size_t)g.members[m];
        if (!(i < j)) continue;                                          /* :170 */
        const float d = dist2_canonical(xi, coords + j * n_cols, n_cols);
        for (size_t l = 0; l < n_radii; ++l) {
          if (d < rad2[l]) {                                             /* :178 */
#pragma omp atomic
            pops[slot[l] * n_rows + i] += 1;
#pragma omp atomic
            pops[slot[l] * n_rows + j] += 1;
          } else {
            break;                                                       /* :183-187 */
          }
        }
      }
    }
  }
  grid_free(&g);
  free(rad2);
  free(slot);
  free(radii);
}

/* ------------------------------------------------------------------------- */
/* free energies. density_clustering.cpp:197-212 as compiled with -ffast-math */
/* (reciprocal hoisted, libm double log): SURVEY.md section 8(a) a3.          */
/* ------------------------------------------------------------------------- */
DCO_API void dco_free_energies(const uint64_t* pops, size_t n_rows, float* fe) {
  uint64_t mx = 0;
  for (size_t i = 0; i < n_rows; ++i)
    if (pops[i] > mx) mx = pops[i];
  const float max_pop = (float)mx;
  const float rec = 1.0f / max_pop;
#pragma omp parallel for
  for (size_t i = 0; i < n_rows; ++i) {
    const float q = (float)pops[i] * rec;
    fe[i] = (float)(-log((double)q));
  }
}

/* ------------------------------------------------------------------------- */
/* nearest neighbour and nearest neighbour with lower free energy for rows    */
/* [i_from, i_to). density_clustering.cpp:230-288. Rows outside the range are */
/* left untouched.                                                            */
/* ------------------------------------------------------------------------- */
DCO_API void dco_nearest_neighbors(const float* coords, size_t n_rows, size_t n_cols,
                                   const float* fe, size_t i_from, size_t i_to,
                                   uint64_t* nn_idx, float* nn_d2, uint64_t* hd_idx,
                                   float* hd_d2) {
  if (i_to > n_rows) i_to = n_rows;
#pragma omp parallel for schedule(dynamic, 64)
  for (size_t i = i_from; i < i_to; ++i) {
    const float* xi = coords + i * n_cols;
    float mind = FLT_MAX, mind_hd = FLT_MAX;          /* :257-258 */
    uint64_t minj = n_rows + 1, minj_hd = n_rows + 1; /* :259-260 */
    const float fei = fe[i];
    for (size_t j = 0; j < n_rows; ++j) {
      if (i == j) continue;                            /* :262 */
      const float d = dist2_canonical(xi, coords + j * n_cols, n_cols);
      if (d < mind) { mind = d; minj = j; }            /* :270-273 */
      if (fe[j] < fei && d < mind_hd) { mind_hd = d; minj_hd = j; } /* :275-279 */
    }
    nn_idx[i] = minj;
    nn_d2[i] = mind;
    hd_idx[i] = minj_hd;
    hd_d2[i] = mind_hd;
  }
}

/* density_clustering.cpp:334-343 */
DCO_API double dco_sigma2(const float* nn_d2, size_t n_rows) {
  double s = 0.0;
  for (size_t i = 0; i < n_rows; ++i) s += (double)nn_d2[i];
  return s / (double)n_rows;
}

/* density_clustering.cpp:669-672, 728-731: radius_lump = sqrt(4*sigma2) -> float */
DCO_API float dco_lumping_radius(double sigma2) { return (float)sqrt(4.0 * sigma2); }
