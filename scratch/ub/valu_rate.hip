// microbenchmark: issue cost of single VALU instructions on gfx950 (2 waves per SIMD, independent chains)
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS(X) X X X X X X X X X X X X X X X X
template <int K>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int iters) {
  unsigned v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 2654435761u + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (K == 0) { OPS(asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(v[1]), "v"(v[2])); asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[3]) : "v"(v[4]), "v"(v[5]));) }
      if (K == 1) { OPS(asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(v[0]) : "v"(v[1])); asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(v[3]) : "v"(v[4]));) }
      if (K == 2) { OPS(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(v[1]), "v"(v[2])); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[3]) : "v"(v[4]), "v"(v[5]));) }
      if (K == 3) { OPS(asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(v[0]) : "v"(v[1]), "v"(v[2])); asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(v[3]) : "v"(v[4]), "v"(v[5]));) }
      if (K == 4) { OPS(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(unsigned long long*)&v[0]) : "v"(*(unsigned long long*)&v[2])); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(unsigned long long*)&v[4]) : "v"(*(unsigned long long*)&v[6]));) }
      if (K == 5) { OPS(asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(v[1]), "v"(v[2])); asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[3]) : "v"(v[4]), "v"(v[5]));) }
      if (K == 6) { OPS(asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(v[1]), "v"(v[2])); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[3]) : "v"(v[4]), "v"(v[5]));) }
      if (K == 7) { OPS(asm volatile("v_bfe_u32 %0, %1, 31, 1" : "+v"(v[0]) : "v"(v[1])); asm volatile("v_bfe_u32 %0, %1, 31, 1" : "+v"(v[3]) : "v"(v[4]));) }
      if (K == 8) { OPS(asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(v[0]) : "v"(v[1]), "v"(v[2]) : "vcc"); ) }
      if (K == 9) { OPS(asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[0]) : "v"(v[1])); asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[3]) : "v"(v[4]));) }
      if (K == 10) { OPS(asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(v[0]) : "v"(v[1])); asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(v[3]) : "v"(v[4]));) }
      if (K == 11) { OPS(asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(v[0]) : "v"(v[1])); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(v[3]) : "v"(v[4]));) }
      if (K == 12) { OPS(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[0]) : "v"(v[1])); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[3]) : "v"(v[4]));) }
      if (K == 13) { OPS(asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(v[0]) : "v"(v[1]), "v"(v[2])); asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(v[3]) : "v"(v[4]), "v"(v[5]));) }
      if (K == 14) { OPS(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(v[1]), "v"(v[2])); asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[3]) : "v"(v[4]), "v"(v[5]));) }
      if (K == 15) { OPS(asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(v[0]) : "v"(v[1])); asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[3]) : "v"(v[4]));) }
    }
  }
  unsigned s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K>
void run(unsigned* d, const char* name, int per_macro) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int occ = 1; occ <= 2; ++occ) {
    hipLaunchKernelGGL((k<K>), dim3(256 * occ), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<K>), dim3(256 * occ), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double n = (double)iters * 4 * 16 * per_macro;   // instructions per wave
    printf("%-28s waves/SIMD=%d: %.3f ns per instruction per SIMD\n", name, occ, best * 1e6 / n / occ);
  }
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 256 * 1024);
  run<0>(d, "v_min3_f32", 2); run<1>(d, "v_alignbit_b32", 2); run<2>(d, "v_perm_b32", 2); run<3>(d, "v_dot4_u32_u8", 2);
  run<4>(d, "v_pk_add_f32", 2); run<5>(d, "v_and_or_b32", 2); run<6>(d, "v_add3_u32", 2); run<7>(d, "v_bfe_u32", 2);
  run<8>(d, "v_cmp+v_addc (pair)", 2); run<9>(d, "v_add_f32", 2); run<10>(d, "v_lshl_or_b32", 2); run<11>(d, "v_bcnt_u32_b32", 2);
  run<12>(d, "v_mul_hi_u32", 2); run<13>(d, "v_sad_u32", 2); run<14>(d, "v_max3/med3_f32", 2); run<15>(d, "v_pk_min_i16/u16", 2);
  return 0;
}
