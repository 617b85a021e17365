// placeholder until the MFMA variants land: reports "unsupported" for every n_cols
#include "dc_mfma.hpp"
namespace dc {
bool mfma_supports(size_t) { return false; }
size_t mfma_workspace_bytes(size_t, size_t) { return 0; }
int mfma_prepare(const float*, uint32_t, uint32_t, void*, hipStream_t) { return -1; }
void launch_pop_mfma(const float*, uint32_t, uint32_t, uint32_t, uint32_t, const Rad2&, int,
                     uint32_t*, void*, hipStream_t) {}
void launch_nn_mfma(const float*, uint32_t, uint32_t, const float*, uint32_t, uint32_t, uint32_t*,
                    float*, uint32_t*, float*, void*, hipStream_t) {}
}  // namespace dc
