// dc_sort.hip -- key/value radix sort used to order the frames (by free energy, by grid cell) for the
// matrix-core sweeps.  A library primitive (rocPRIM device radix sort), kept in its own translation
// unit because the header-only sort is slow to compile.
//
// rocPRIM's default configuration switches to a merge sort up to 2^20 items -- 21 kernel launches and
// 0.35 ms per sort at C3's 10^6 frames (measured, rocprofv3 kernel trace), three sorts per step.  The
// Onesweep radix sort it uses above that limit needs 2 + ceil(bits / 8) launches; the limit is lowered
// here so that every frame count of interest takes it, and callers pass the number of key bits.
#include "dc_mfma.hpp"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace dc {

namespace {
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                              rocprim::default_config, 32768>;
}

size_t sort_temp_bytes(size_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                              (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0u, 32u);
  return (bytes + 255) & ~(size_t)255;
}

int sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                   uint32_t* vals_out, size_t n, void* temp, size_t temp_bytes,
                   hipStream_t stream, unsigned key_bits) {
  hipError_t e = rocprim::radix_sort_pairs<SortConfig>(temp, temp_bytes, keys_in, keys_out, vals_in,
                                                       vals_out, n, 0u, key_bits, stream);
  return e == hipSuccess ? 0 : -1;
}

}  // namespace dc
