// rocPRIM Onesweep configurations for 10^6 (key, value) pairs of 32 bits, 16 and 24 key bits (the two orderings of a step)
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
template <class OS>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, OS, 32768>;
template <class C>
void run(const char* name, uint32_t* kin, uint32_t* kout, uint32_t* vin, uint32_t* vout, size_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs<C>(nullptr, bytes, kin, kout, vin, vout, n, 0u, 32u);
  void* tmp; (void)hipMalloc(&tmp, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (unsigned bits : {16u, 24u}) {
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
      (void)hipEventRecord(e0, 0);
      (void)rocprim::radix_sort_pairs<C>(tmp, bytes, kin, kout, vin, vout, n, 0u, bits, 0);
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep > 1 && ms < best) best = ms;
    }
    printf("%-44s %2u bits: %7.1f us\n", name, bits, best * 1e3f);
  }
  (void)hipFree(tmp);
}
int main() {
  const size_t n = 1000000;
  std::vector<uint32_t> h(n); std::mt19937 g(2); for (auto& x : h) x = g() & 0xFFFFFFu;
  uint32_t *kin, *kout, *vin, *vout;
  (void)hipMalloc(&kin, n * 4); (void)hipMalloc(&kout, n * 4); (void)hipMalloc(&vin, n * 4); (void)hipMalloc(&vout, n * 4);
  (void)hipMemcpy(kin, h.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(vin, h.data(), n * 4, hipMemcpyHostToDevice);
  using namespace rocprim;
  run<Cfg<default_config>>("default", kin, kout, vin, vout, n);
  run<Cfg<radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<256, 12>, 8>>>("256x12 / 256x12, 8 bits", kin, kout, vin, vout, n);
  run<Cfg<radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<256, 6>, 8>>>("256x12 / 256x6, 8 bits", kin, kout, vin, vout, n);
  run<Cfg<radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<256, 4>, 8>>>("256x12 / 256x4, 8 bits", kin, kout, vin, vout, n);
  run<Cfg<radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<256, 18>, 8>>>("256x12 / 256x18, 8 bits", kin, kout, vin, vout, n);
  run<Cfg<radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<256, 8>, 8>>>("256x12 / 256x8, 8 bits", kin, kout, vin, vout, n);
  run<Cfg<radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<128, 12>, 8>>>("256x12 / 128x12, 8 bits", kin, kout, vin, vout, n);
  run<Cfg<radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<256, 8>, 6>>>("256x12 / 256x8, 6 bits", kin, kout, vin, vout, n);
  return 0;
}
