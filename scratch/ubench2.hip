// micro-benchmark 2: VALU op costs; co-issue with f32 / bf16 MFMA shapes (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define OPS(X) X(0,"v_add_f32 %0, %0, %1") X(1,"v_sub_f32 %0, %1, %0") X(2,"v_alignbit_b32 %0, %0, %1, 31") \
  X(3,"v_min3_u32 %0, %0, %1, %2") X(4,"v_min3_f32 %0, %0, %1, %2") X(5,"v_lshrrev_b32 %0, 31, %1") \
  X(6,"v_add_u32 %0, %0, %1") X(7,"v_add3_u32 %0, %0, %1, %2") X(8,"v_lshl_or_b32 %0, %0, 1, %1") \
  X(9,"v_bfe_u32 %0, %1, 31, 1") X(10,"v_min_u32 %0, %0, %1") X(11,"v_min_f32 %0, %0, %1") \
  X(12,"v_bcnt_u32_b32 %0, %1, %0") X(13,"v_and_or_b32 %0, %1, %2, %0") X(14,"v_mov_b32 %0, %1") \
  X(15,"v_max_f32 %0, %0, %1") X(16,"v_mul_f32 %0, %0, %1 clamp") X(17,"v_fma_f32 %0, %1, %2, %0") \
  X(18,"v_sad_u32 %0, %1, %2, %0") X(19,"v_cndmask_b32 %0, %0, %1, vcc") X(20, "v_xad_u32 %0, %0, %1, %2") \
  X(21,"v_med3_f32 %0, %0, %1, %2") X(22,"v_lshl_add_u32 %0, %0, 1, %1") X(23,"v_mad_u32_u24 %0, %1, %2, %0") \
  X(24,"v_ashrrev_i32 %0, 31, %1") X(25,"v_sub_u32 %0, %0, %1") X(26,"v_or_b32 %0, %0, %1") X(27, "v_perm_b32 %0, %0, %1, %2")

template <int KIND>
__device__ __forceinline__ void op(unsigned& d, unsigned a, unsigned b) {
#define X(i, s) if (KIND == i) asm volatile(s : "+v"(d) : "v"(a), "v"(b));
  OPS(X)
#undef X
}
static const char* opname(int k) {
#define X(i, s) if (k == i) return s;
  OPS(X)
#undef X
  return "?";
}

// MT: 0 none, 1 f32 32x32x2, 2 f32 16x16x4, 3 bf16 32x32x16, 4 bf16 16x16x32 ; DEP: dependent chain (1) or 2 alternating accumulators (0)
template <int NV, int KIND, int MT, int NM, int DEP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc = {0}, acc2 = {0};
  f32x4 c4 = {0}, c42 = {0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  s16x8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (short)(0x3f80 + threadIdx.x + i); bh[i] = (short)(0x3f00 + i); }
  unsigned u[8];
  for (int i = 0; i < 8; ++i) u[i] = i * 77 + threadIdx.x;
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const bool alt = (!DEP) && (m & 1);
      if (MT == 1) { if (alt) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
                     else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }
      if (MT == 2) { if (alt) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c42) : "v"(a), "v"(b));
                     else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c4) : "v"(a), "v"(b)); }
      if (MT == 3) { if (alt) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc2) : "v"(ah), "v"(bh));
                     else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah), "v"(bh)); }
      if (MT == 4) { if (alt) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c42) : "v"(ah), "v"(bh));
                     else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c4) : "v"(ah), "v"(bh)); }
#pragma unroll
      for (int v = 0; v < NV; ++v) op<KIND>(u[v & 7], ua, ub);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
  for (int i = 0; i < 4; ++i) s += c4[i] + c42[i];
  for (int i = 0; i < 8; ++i) s += u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int KIND, int MT, int NM, int DEP>
float run(int blocks_per_cu, int iters = 20000) {
  static float* out = nullptr;
  if (!out) hipMalloc(&out, 256 * 8 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * blocks_per_cu), block(256);
  hipLaunchKernelGGL((k<NV, KIND, MT, NM, DEP>), grid, block, 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, KIND, MT, NM, DEP>), grid, block, 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters;   // ns per iteration
}

template <int KIND> void valu_cost() {
  float t1 = run<32, KIND, 0, 1, 1>(1), t2 = run<32, KIND, 0, 1, 1>(2), t4 = run<32, KIND, 0, 1, 1>(4);
  printf("VALU %-34s ns/op: 1w %.2f  2w(per SIMD) %.2f  4w(per SIMD) %.2f\n", opname(KIND), t1 / 32, t2 / 64, t4 / 128);
}

template <int MT, int DEP> void mfma_mix(const char* name) {
  for (int w = 1; w <= 2; ++w) {
    float m0 = run<0, 0, MT, 2, DEP>(w) / 2, m4 = run<4, 0, MT, 2, DEP>(w) / 2, m8 = run<8, 0, MT, 2, DEP>(w) / 2,
          m16 = run<16, 0, MT, 2, DEP>(w) / 2, m8i = run<8, 6, MT, 2, DEP>(w) / 2;
    printf("%-22s dep=%d w=%d ns per (MFMA + n VALU): n=0 %.1f  n=4 %.1f  n=8 %.1f  n=16 %.1f  n=8(int add) %.1f\n", name, DEP, w, m0, m4, m8, m16, m8i);
  }
}

int main() {
  valu_cost<0>(); valu_cost<1>(); valu_cost<2>(); valu_cost<3>(); valu_cost<4>(); valu_cost<5>(); valu_cost<6>();
  valu_cost<7>(); valu_cost<8>(); valu_cost<9>(); valu_cost<10>(); valu_cost<11>(); valu_cost<12>(); valu_cost<13>();
  valu_cost<14>(); valu_cost<15>(); valu_cost<16>(); valu_cost<17>(); valu_cost<18>(); valu_cost<19>(); valu_cost<20>();
  valu_cost<21>(); valu_cost<22>(); valu_cost<23>(); valu_cost<24>(); valu_cost<25>(); valu_cost<26>(); valu_cost<27>();
  mfma_mix<1, 1>("f32 32x32x2"); mfma_mix<1, 0>("f32 32x32x2");
  mfma_mix<2, 1>("f32 16x16x4"); mfma_mix<2, 0>("f32 16x16x4");
  mfma_mix<3, 1>("bf16 32x32x16"); mfma_mix<3, 0>("bf16 32x32x16");
  mfma_mix<4, 1>("bf16 16x16x32"); mfma_mix<4, 0>("bf16 16x16x32");
  return 0;
}
