// dc_sort.hip -- key/value radix sort used to order reference frames by free energy for the
// neighbour sweep.  A library primitive (hipCUB / rocPRIM device radix sort), kept in its own
// translation unit because the header-only sort is slow to compile.
#include "dc_mfma.hpp"

#include <hipcub/hipcub.hpp>

namespace dc {

size_t sort_temp_bytes(size_t n) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                     (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n);
  return (bytes + 255) & ~(size_t)255;
}

int sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                   uint32_t* vals_out, size_t n, void* temp, size_t temp_bytes,
                   hipStream_t stream) {
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in,
                                                    vals_out, (int)n, 0, 32, stream);
  return e == hipSuccess ? 0 : -1;
}

}  // namespace dc
