// numerics probe: how does v_mfma_f32_32x32x16_f16 accumulate?  (gfx950)
// D[i][j] = C[i][j] + sum_k A[i][k] B[k][j]; products of bf16 are exact in fp32; the question is the
// rounding/truncation of the 17-term sum.  Each "case" is one (i, j) cell with its own C and 16
// products, so one MFMA evaluates 1024 crafted sums.  B is the identity-like selector: we put the
// whole product into A (value) times B (power of two), so any product pattern is expressible.
//   hipcc --offload-arch=gfx950 -O2 -o scratch/mfma_probe scratch/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__global__ void probe(const unsigned short* A, const unsigned short* B, const float* C, float* D) {
  // A [32][16], B [16][32] (k-major), C/D [32][32]
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  h16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = __builtin_bit_cast(_Float16, A[r * 16 + 8 * h + j]);
    b[j] = __builtin_bit_cast(_Float16, B[(8 * h + j) * 32 + r]);
  }
  f32x16 c;
  for (int g = 0; g < 16; ++g) c[g] = C[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r];
  f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int g = 0; g < 16; ++g) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = d[g];
}

static unsigned short f2bf(float f) {  // to fp16 (exact for <= 11 significant bits in range; subnormals kept)
  _Float16 h = (_Float16)f;
  unsigned short b;
  memcpy(&b, &h, 2);
  return b;
}
static float bf2f(unsigned short b) {
  _Float16 h;
  memcpy(&h, &b, 2);
  return (float)h;
}

int main() {
  unsigned short *dA, *dB;
  float *dC, *dD;
  hipMalloc(&dA, 32 * 16 * 2);
  hipMalloc(&dB, 16 * 32 * 2);
  hipMalloc(&dC, 4096);
  hipMalloc(&dD, 4096);
  std::vector<unsigned short> A(512), B(512);
  std::vector<float> C(1024), D(1024);
  auto run = [&]() {
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  };
  // --- crafted cases: row i uses A[i][k]; column j uses B[k][j]. Use j = 0 only (B[k][0] = 1), rows vary.
  for (auto& v : B) v = 0;
  for (int k = 0; k < 16; ++k) B[k * 32 + 0] = f2bf(ldexpf(1.0f, -14));
  for (auto& v : A) v = 0;
  for (auto& v : C) v = 0;
  const char* names[32] = {0};
  int i = 0;
  auto setrow = [&](const char* nm, float c, std::vector<float> prods) {
    names[i] = nm;
    C[i * 32] = c;
    for (size_t k = 0; k < prods.size(); ++k) A[i * 16 + k] = f2bf(ldexpf(prods[k], 14));
    ++i;
  };
  const float u = ldexpf(1.0f, -24);  // half ulp of 1.0
  setrow("1 + 16 x 2^-27 (=1 ulp total)", 1.0f, std::vector<float>(16, ldexpf(1, -27)));
  setrow("1 + 16 x 2^-28 (=half ulp)", 1.0f, std::vector<float>(16, ldexpf(1, -28)));
  setrow("1 + 2^-24 (tie -> even = 1)", 1.0f, {u});
  setrow("1 + 3*2^-25 (0.75 ulp -> up)", 1.0f, {ldexpf(3, -25)});
  setrow("1 - 2^-25 (quarter ulp below)", 1.0f, {-ldexpf(1, -25)});
  setrow("1 - 2^-26 - 2^-26", 1.0f, {-ldexpf(1, -26), -ldexpf(1, -26)});
  setrow("1 + 1 - 1 + 2^-30 x 8 (cancel)", 1.0f, {1.0f, -1.0f, ldexpf(1,-30),ldexpf(1,-30),ldexpf(1,-30),ldexpf(1,-30),ldexpf(1,-30),ldexpf(1,-30),ldexpf(1,-30),ldexpf(1,-30)});
  setrow("1 + 2^-23 x (1/16) x 16 lanes", 1.0f, std::vector<float>(16, ldexpf(1, -27)));
  setrow("C=2^-10, prods 1 and -1 and 2^-30", ldexpf(1, -10), {1.0f, -1.0f, ldexpf(1, -30)});
  const int ncases = i;
  run();
  printf("crafted cases (exact value in parentheses):\n");
  for (int c = 0; c < ncases; ++c) {
    long double ex = C[c * 32];
    for (int k = 0; k < 16; ++k) ex += (long double)bf2f(A[c * 16 + k]) * (long double)ldexpf(1.0f, -14);
    printf("  %-44s hw = %.10g  (exact %.12Lg; hw-exact = %.3Lg ulp(1)=2^-23 units)\n", names[c], D[c * 32], ex,
           ((long double)D[c * 32] - ex) / ldexpl(1, -23));
  }
  // --- random cases: products with random exponents in [-E, 0], random signs; C ~ 1
  srand(12345);
  for (int E : {4, 12, 20, 24}) {
    double worst_res = 0, worst_max = 0;
    for (int trial = 0; trial < 200; ++trial) {
      for (int r = 0; r < 32; ++r)
        for (int k = 0; k < 16; ++k) {
          const float m = 1.0f + (rand() & 1023) / 1024.0f;  // 11 significant bits
          A[r * 16 + k] = f2bf(((rand() & 1) ? -m : m) * ldexpf(1.0f, -(rand() % (E + 1))));
        }
      for (int k = 0; k < 16; ++k)
        for (int j = 0; j < 32; ++j) {
          const float m = 1.0f + (rand() & 1023) / 1024.0f;
          B[k * 32 + j] = f2bf(((rand() & 1) ? -m : m) * ldexpf(1.0f, -(rand() % (E + 1))));
        }
      for (auto& v : C) v = ((rand() & 1) ? -1.0f : 1.0f) * (1.0f + (rand() & 0xffff) / 65536.0f) * ldexpf(1.0f, -(rand() % (E + 1)));
      run();
      for (int r = 0; r < 32; ++r)
        for (int j = 0; j < 32; ++j) {
          long double ex = C[r * 32 + j], mx = fabsl(ex);
          for (int k = 0; k < 16; ++k) {
            const long double p = (long double)bf2f(A[r * 16 + k]) * (long double)bf2f(B[k * 32 + j]);
            ex += p;
            if (fabsl(p) > mx) mx = fabsl(p);
          }
          const long double err = fabsl((long double)D[r * 32 + j] - ex);
          int e1, e2;
          frexpl(fabsl(ex) > 0 ? ex : 1e-300L, &e1);
          frexpl(mx, &e2);
          const double ulp_res = ldexp(1.0, e1 - 24), ulp_max = ldexp(1.0, e2 - 24);
          if ((double)err / ulp_res > worst_res) worst_res = (double)err / ulp_res;
          if ((double)err / ulp_max > worst_max) worst_max = (double)err / ulp_max;
        }
    }
    printf("random, exponent spread 2^-%d: worst error = %.3f ulp(result), %.3f ulp(max addend)\n", E, worst_res, worst_max);
  }
  return 0;
}
