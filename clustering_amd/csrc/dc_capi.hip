// dc_capi.hip -- the C ABI of include/dc_density.h on top of the HIP kernels (gfx950).
// Host-side orchestration only: argument checks, memsets, launches, the host-libm free-energy
// referee and the host-pointer wrappers that mirror the reference's per-GPU functions
// (density_clustering_cuda.cu:45-137, :184-284).  The resident multi-GPU path (sessions, RCCL) is
// dc_session.hip.  No CPU implementation of the sweeps exists
// here: without a HIP device every compute entry point fails with DC_ERR_NO_DEVICE / DC_ERR_HIP.
#include "../../include/dc_density.h"
#include "dc_common.hpp"
#include "dc_mfma.hpp"

#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define DC_HIP_TRY(expr)                                                                   \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return fail(DC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                  __LINE__);                                                               \
  } while (0)

}  // namespace

namespace dc {
// the calling thread's last error (also used by dc_session.hip)
int set_error(int code, const char* msg) {
  g_last_error = msg ? msg : "";
  return code;
}
}  // namespace dc

namespace {

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(DC_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return DC_OK;
}

int check_sizes(size_t n_rows, size_t n_cols, size_t i_from, size_t i_to) {
  if (n_cols == 0) return fail(DC_ERR_INVALID_ARGUMENT, "n_cols must be >= 1");
  if (n_rows + 1 > (size_t)UINT32_MAX)
    return fail(DC_ERR_TOO_LARGE, "n_rows=%zu: frame ids must fit uint32", n_rows);
  if (n_rows * n_cols > (size_t)UINT32_MAX * 4ull)
    return fail(DC_ERR_TOO_LARGE, "n_rows*n_cols=%zu too large", n_rows * n_cols);
  if (i_from > i_to || i_to > n_rows)
    return fail(DC_ERR_INVALID_ARGUMENT, "row range [%zu, %zu) outside [0, %zu)", i_from, i_to,
                n_rows);
  if (n_cols > (size_t)dc::kMaxColsGeneric)
    return fail(DC_ERR_INVALID_ARGUMENT, "n_cols=%zu not supported (max %d)", n_cols,
                dc::kMaxColsGeneric);
  return DC_OK;
}

// fe value of one population, exactly as the reference binary computes it
// (density_clustering.cpp:201-209 under -ffast-math: reciprocal hoisted, double libm log).
inline float fe_of_pop(uint32_t pop, float rec) {
  const float q = (float)pop * rec;
  return (float)(-log((double)q));
}

void fill_fe_table(std::vector<float>& table, uint32_t max_pop) {
  const float rec = 1.0f / (float)max_pop;
  const size_t n = table.size();
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  // about 16k logs per thread (spawning a thread costs as much as a few thousand logs)
  const unsigned nt = (unsigned)std::min<size_t>(std::min(hw, 16u), std::max<size_t>(1, n >> 14));
  auto work = [&](size_t lo, size_t hi) {
    for (size_t p = lo; p < hi; ++p) table[p] = fe_of_pop((uint32_t)p, rec);
  };
  if (nt == 1) {
    work(0, n);
    return;
  }
  std::vector<std::thread> th;
  const size_t chunk = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t) {
    const size_t lo = t * chunk, hi = std::min(n, lo + chunk);
    if (lo < hi) th.emplace_back(work, lo, hi);
  }
  for (auto& x : th) x.join();
}

bool want_mfma(int variant, size_t n_cols) {
  if (variant == DC_VARIANT_DIRECT) return false;
  return dc::mfma_supports(n_cols);
}

}  // namespace

extern "C" {

const char* dc_hip_last_error(void) { return g_last_error.c_str(); }

int dc_hip_abi_version(void) { return DC_HIP_ABI_VERSION; }

int dc_hip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e == hipErrorNoDevice) {
    (void)hipGetLastError();
    return 0;
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(DC_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  return n;
}

size_t dc_hip_workspace_bytes(size_t n_rows, size_t n_cols, size_t n_radii) {
  (void)n_radii;
  return dc::mfma_workspace_bytes(n_rows, n_cols);
}

int dc_hip_workspace_components_dev(const void* d_workspace, size_t n_rows, size_t n_cols, uint32_t* n_components,
                                    float* extent2_global, float* extent2_local, float* scale, void* stream) {
  if (!d_workspace || !n_components || !extent2_global || !extent2_local || !scale)
    return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  if (!dc::mfma_supports(n_cols)) return fail(DC_ERR_INVALID_ARGUMENT, "no matrix-core sweep for n_cols=%zu", n_cols);
  if (dc::components_info(d_workspace, n_rows, n_cols, n_components, extent2_global, extent2_local, scale,
                          (hipStream_t)stream) != 0)
    return fail(DC_ERR_HIP, "reading the workspace header failed");
  return DC_OK;
}

int dc_hip_sweep_timing(int enable) {
  dc::sweep_timer_enable(enable != 0);
  return DC_OK;
}

int dc_hip_last_sweep_ms(int kind, float* ms) {
  if (!ms || (kind != 0 && kind != 1)) return fail(DC_ERR_INVALID_ARGUMENT, "kind 0 (population) or 1 (neighbour), ms != NULL");
  if (dc::sweep_timer_read(kind, ms) != 0) return fail(DC_ERR_INVALID_ARGUMENT, "no timed sweep of kind %d on this device", kind);
  return DC_OK;
}

int dc_hip_workspace_counters_dev(const void* d_workspace, uint64_t* pop_tiles, uint64_t* nn_tiles,
                                  void* stream) {
  if (!d_workspace) return fail(DC_ERR_INVALID_ARGUMENT, "null workspace");
  uint64_t c[2] = {0, 0};   // header bytes 8..23 (see dc_mfma_step.hip)
  hipStream_t s = (hipStream_t)stream;
  DC_HIP_TRY(hipMemcpyAsync(c, (const char*)d_workspace + 8, sizeof(c), hipMemcpyDeviceToHost, s));
  DC_HIP_TRY(hipStreamSynchronize(s));
  if (pop_tiles) *pop_tiles = c[0];
  if (nn_tiles) *nn_tiles = c[1];
  return DC_OK;
}

int dc_hip_workspace_layout_status_dev(const void* d_workspace, int* mismatch, void* stream) {
  if (!d_workspace || !mismatch) return fail(DC_ERR_INVALID_ARGUMENT, "null argument");
  uint32_t w = 0;   // header word 18 (dc_mfma_kernels.hpp kHdrLayoutBad)
  hipStream_t s = (hipStream_t)stream;
  DC_HIP_TRY(hipMemcpyAsync(&w, (const char*)d_workspace + 4 * 18, sizeof(w), hipMemcpyDeviceToHost, s));
  DC_HIP_TRY(hipStreamSynchronize(s));
  *mismatch = w != 0u ? 1 : 0;
  return DC_OK;
}

int dc_hip_workspace_mfma_counters_dev(const void* d_workspace, uint64_t* pop_mfma, uint64_t* nn_mfma, void* stream) {
  if (!d_workspace) return fail(DC_ERR_INVALID_ARGUMENT, "null workspace");
  uint32_t h[32];   // header words 6..7 (population sweeps), 26..27 (neighbour sweeps): dc_mfma_kernels.hpp kHdrMfma*
  hipStream_t s = (hipStream_t)stream;
  DC_HIP_TRY(hipMemcpyAsync(h, d_workspace, sizeof(h), hipMemcpyDeviceToHost, s));
  DC_HIP_TRY(hipStreamSynchronize(s));
  if (pop_mfma) *pop_mfma = ((uint64_t)h[7] << 32) | h[6];
  if (nn_mfma) *nn_mfma = ((uint64_t)h[27] << 32) | h[26];
  return DC_OK;
}

}  // extern "C"

namespace {
// rows of one shard exactly as density_clustering_cuda.cu:149,165-169 (the last one takes the rest)
void shard_rows(size_t n_rows, size_t n_shards, size_t shard, size_t* lo, size_t* hi) {
  const size_t rng = n_rows / n_shards;
  *lo = shard * rng;
  *hi = (shard == n_shards - 1) ? n_rows : (shard + 1) * rng;
}

// n_segments == 0: the row range [i_from, i_to).  n_segments > 0: segment `segment` of the spatial
// order for the pruned matrix-core sweep; whenever that sweep does not run (other variant, n_cols it
// does not handle, flagged data) the same call answers for row block `segment` instead -- any
// partition of the rows serves a sharded run, as long as every rank uses the same rule.
int populations_impl(const float* d_coords, size_t n_rows, size_t n_cols, const float* radii,
                     size_t n_radii, size_t i_from, size_t i_to, size_t segment, size_t n_segments,
                     uint32_t* d_pops, void* d_workspace, size_t workspace_bytes, int variant,
                     void* stream) {
  const bool stats_valid = (variant & DC_FLAG_STATS_VALID) != 0;
  variant &= DC_VARIANT_MASK;
  if (n_segments > 0) {
    if (segment >= n_segments) return fail(DC_ERR_INVALID_ARGUMENT, "segment %zu of %zu", segment, n_segments);
    shard_rows(n_rows, n_segments, segment, &i_from, &i_to);
  }
  if (int rc = check_sizes(n_rows, n_cols, i_from, i_to)) return rc;
  if (n_radii == 0 || n_rows == 0) return DC_OK;
  if (!d_coords || !radii || !d_pops) return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  DC_HIP_TRY(hipMemsetAsync(d_pops, 0, sizeof(uint32_t) * n_radii * n_rows, s));
  if (i_from == i_to && n_segments == 0) return DC_OK;
  const bool mfma = want_mfma(variant, n_cols);
  const bool pruned = mfma && variant != DC_VARIANT_MFMA && variant != DC_VARIANT_MFMA32;   // (the pruned sweep will run)
  if ((variant == DC_VARIANT_MFMA || variant == DC_VARIANT_MFMA_PRUNED) && !mfma)
    return fail(DC_ERR_INVALID_ARGUMENT, "MFMA variant does not support n_cols=%zu", n_cols);
  if (variant == DC_VARIANT_MFMA32 && (!mfma || !dc::mfma32_supports(n_cols) || n_segments > 0))
    return fail(DC_ERR_INVALID_ARGUMENT, "the fp32-MFMA variant handles n_cols 9..10 and row ranges only (n_cols=%zu)", n_cols);
  if (mfma) {
    if (!d_workspace || workspace_bytes < dc::mfma_workspace_bytes(n_rows, n_cols))
      return fail(DC_ERR_WORKSPACE, "workspace of %zu bytes needed, got %zu",
                  dc::mfma_workspace_bytes(n_rows, n_cols), d_workspace ? workspace_bytes : 0);
    if (int rc = dc::mfma_prepare(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, d_workspace,
                                  variant == DC_VARIANT_MFMA, s, stats_valid, pruned))
      return fail(DC_ERR_HIP, "mfma_prepare failed (%d)", rc);
  }
  for (size_t r0 = 0; r0 < n_radii; r0 += dc::kMaxRadiiPerLaunch) {
    const bool comp_clean = pruned && !stats_valid && r0 == 0;   // (the component region: zero-filled with the header)
    const int n_rad = (int)std::min((size_t)dc::kMaxRadiiPerLaunch, n_radii - r0);
    dc::Rad2 rad2;
    for (int r = 0; r < dc::kMaxRadiiPerLaunch; ++r)
      rad2.v[r] = (r < n_rad) ? radii[r0 + r] * radii[r0 + r] : -1.0f;  // fl32(r*r), :137-140
    uint32_t* out = d_pops + r0 * n_rows;
    // MFMA variant: the MFMA kernel runs unless the operand-image pass flagged the data
    // (non-finite / overflow-prone rows), in which case the gated direct kernel does the work;
    // both are enqueued, the choice is made on the device (no host synchronisation).
    if (mfma && variant == DC_VARIANT_MFMA32)
      dc::launch_pop_mfma32(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, (uint32_t)i_from, (uint32_t)i_to, rad2, n_rad,
                            out, d_workspace, s);
    else if (mfma && variant == DC_VARIANT_MFMA)
      dc::launch_pop_mfma(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, (uint32_t)i_from,
                          (uint32_t)i_to, rad2, n_rad, out, d_workspace, s);
    else if (mfma && n_segments > 0)
      dc::launch_pop_pruned_segment(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, (uint32_t)segment,
                                    (uint32_t)n_segments, rad2, n_rad, out, d_workspace, s, comp_clean);
    else if (mfma)
      dc::launch_pop_pruned(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, (uint32_t)i_from,
                            (uint32_t)i_to, rad2, n_rad, out, d_workspace, s, comp_clean);
    if (i_from != i_to && !dc::launch_pop_direct(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, (uint32_t)i_from,
                               (uint32_t)i_to, rad2, n_rad, out,
                               mfma ? (const uint32_t*)d_workspace : nullptr, s))
      return fail(DC_ERR_INVALID_ARGUMENT, "n_cols=%zu not supported", n_cols);
    if (int rc = check_launch("population sweep launch")) return rc;
  }
  return DC_OK;
}
}  // namespace

extern "C" {

int dc_hip_populations_dev(const float* d_coords, size_t n_rows, size_t n_cols, const float* radii,
                           size_t n_radii, size_t i_from, size_t i_to, uint32_t* d_pops,
                           void* d_workspace, size_t workspace_bytes, int variant, void* stream) {
  return populations_impl(d_coords, n_rows, n_cols, radii, n_radii, i_from, i_to, 0, 0, d_pops,
                          d_workspace, workspace_bytes, variant, stream);
}

int dc_hip_populations_segment_dev(const float* d_coords, size_t n_rows, size_t n_cols,
                                   const float* radii, size_t n_radii, size_t segment,
                                   size_t n_segments, uint32_t* d_pops, void* d_workspace,
                                   size_t workspace_bytes, int variant, void* stream) {
  if (n_segments == 0) return fail(DC_ERR_INVALID_ARGUMENT, "n_segments must be positive");
  return populations_impl(d_coords, n_rows, n_cols, radii, n_radii, 0, 0, segment, n_segments, d_pops,
                          d_workspace, workspace_bytes, variant, stream);
}

int dc_hip_free_energies_dev(const uint32_t* d_pops, size_t n_rows, float* d_fe,
                             uint32_t* max_pop_out, void* stream) {
  if (n_rows == 0) {
    if (max_pop_out) *max_pop_out = 0;
    return DC_OK;
  }
  if (!d_pops || !d_fe) return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  if (n_rows + 1 > (size_t)UINT32_MAX) return fail(DC_ERR_TOO_LARGE, "n_rows too large");
  hipStream_t s = (hipStream_t)stream;
  // per-device scratch that lives across calls (no allocation, and no hipFree with its
  // device-wide synchronisation, on the per-step path): max word + table on the device, table on
  // the host
  struct Scratch {
    int device = -1;
    uint32_t* d_max = nullptr;     // kFeStateWords of state (fe_log_kernel), then the list of flagged (row, pop) pairs
    uint32_t* h_head = nullptr;    // pinned host memory (max, count) are copied to
    uint32_t seq = 0;
    bool dirty = false;            // a call did not run to its end: its slots of the state may not be zero
    float* d_table = nullptr;
    size_t table_cap = 0;
    std::vector<float> table;
  };
  static Scratch scratch[16];
  static std::mutex scratch_mutex[16];
  int dev = 0;
  DC_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(scratch_mutex[dev & 15]);   // (one call at a time per device)
  Scratch& S = scratch[dev & 15];
  constexpr uint32_t kFlagCap = 4096;
  if (S.device != dev) {   // first use on this device (or a slot shared by devices 16 apart)
    if (S.d_max) (void)hipFree(S.d_max);
    if (S.d_table) (void)hipFree(S.d_table);
    if (S.h_head) (void)hipHostFree(S.h_head);
    S = Scratch();
    S.device = dev;
    DC_HIP_TRY(hipMalloc((void**)&S.d_max, sizeof(uint32_t) * (dc::kFeStateWords + 2 * kFlagCap)));
    DC_HIP_TRY(hipMemset(S.d_max, 0, sizeof(uint32_t) * dc::kFeStateWords));   // (the kernels keep words 0..2 zero between calls)
    if (hipHostMalloc((void**)&S.h_head, 64, hipHostMallocDefault) != hipSuccess) {   // (then a pageable destination)
      (void)hipGetLastError();
      S.h_head = nullptr;
    }
  }
  // Default: every row's free energy from the device's double log, the rows it cannot vouch for
  // (value within 64 ulp(double) of a float rounding boundary: one in 2^22) recomputed by the host libm
  // -- one hand-off to the host (max_pop and the number of listed rows), no table.  DC_FE_HOST_TABLE=1 (and more
  // flagged rows than the list holds) takes the table path below: one host log per distinct population.
  static const bool host_table = [] {
    const char* v = getenv("DC_FE_HOST_TABLE");
    return v && v[0] == '1';
  }();
  // margin around the float rounding boundaries, relative to the value: 64 ulp(double) unless the test
  // suite widens it (DC_FE_REFEREE_TOL) to drive rows through the referee and the overflow path
  static const double tol_rel = [] {
    const char* v = getenv("DC_FE_REFEREE_TOL");
    const double t = v ? atof(v) : 0.0;
    return t > 1.5e-14 ? t : 1.5e-14;
  }();
  uint32_t max_pop = 0;
  hipError_t e = hipSuccess;
  if (!host_table) {
    const uint32_t slot = (S.seq++) & 1u;   // (alternate calls use alternate slots of the state; each clears the other's)
    if (S.dirty) DC_HIP_TRY(hipMemsetAsync(S.d_max, 0, sizeof(uint32_t) * dc::kFeStateWords, s));
    S.dirty = true;   // (until this call has run to its end)
    dc::launch_fe_log(d_pops, (uint32_t)n_rows, S.d_max, slot, d_fe, S.d_max + dc::kFeStateWords, kFlagCap, tol_rel, s);
    // (max, count) into pinned host memory where there is some: a true asynchronous copy, no staging
    uint32_t head_local[2] = {0, 0};
    uint32_t* head = S.h_head ? S.h_head : head_local;   // max_pop, number of flagged rows
    e = hipMemcpyAsync(head, S.d_max + 2u * slot, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(DC_ERR_HIP, "free energies: %s", hipGetErrorString(e));
    S.dirty = false;
    max_pop = head[0];
    if (max_pop_out) *max_pop_out = max_pop;
    if (head[1] == 0) return DC_OK;
    if (head[1] <= kFlagCap) {
      std::vector<uint32_t> list(2 * (size_t)head[1]);
      e = hipMemcpyAsync(list.data(), S.d_max + dc::kFeStateWords, sizeof(uint32_t) * list.size(), hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      const float rec = 1.0f / (float)max_pop;
      std::vector<float> fixed(head[1]);
      for (uint32_t k = 0; k < head[1] && e == hipSuccess; ++k) {
        fixed[k] = fe_of_pop(list[2 * k + 1], rec);
        e = hipMemcpyAsync(d_fe + list[2 * k], &fixed[k], sizeof(float), hipMemcpyHostToDevice, s);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) return fail(DC_ERR_HIP, "free energies (host referee): %s", hipGetErrorString(e));
      return DC_OK;
    }
  } else {
    dc::launch_max_u32(d_pops, (uint32_t)n_rows, S.d_max + 4, s);   // (its own word, filled first)
    e = hipMemcpyAsync(&max_pop, S.d_max + 4, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(DC_ERR_HIP, "max population: %s", hipGetErrorString(e));
    if (max_pop_out) *max_pop_out = max_pop;
  }
  // one double log per DISTINCT population value, evaluated by the host libm like the
  // reference (which computes every FE on the host, density_clustering.cpp:687)
  S.table.resize((size_t)max_pop + 1);
  fill_fe_table(S.table, max_pop);
  if (S.table_cap < S.table.size()) {
    if (S.d_table) (void)hipFree(S.d_table);
    S.d_table = nullptr;
    S.table_cap = 0;
    DC_HIP_TRY(hipMalloc((void**)&S.d_table, sizeof(float) * S.table.size()));
    S.table_cap = S.table.size();
  }
  e = hipMemcpyAsync(S.d_table, S.table.data(), sizeof(float) * S.table.size(), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    dc::launch_fe_gather(d_pops, (uint32_t)n_rows, S.d_table, d_fe, s);
    e = hipGetLastError();
  }
  // (the host table must stay untouched until the copy has been consumed: one more sync)
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return fail(DC_ERR_HIP, "free-energy gather: %s", hipGetErrorString(e));
  return DC_OK;
}

}  // extern "C"

namespace {
// (segments: see populations_impl)
int nearest_neighbors_impl(const float* d_coords, size_t n_rows, size_t n_cols, const float* d_fe,
                           size_t i_from, size_t i_to, size_t segment, size_t n_segments,
                           uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx, float* d_hd_d2,
                           void* d_workspace, size_t workspace_bytes, int variant, void* stream) {
  const bool stats_valid = (variant & DC_FLAG_STATS_VALID) != 0;
  variant &= DC_VARIANT_MASK;
  if (n_segments > 0) {
    if (segment >= n_segments) return fail(DC_ERR_INVALID_ARGUMENT, "segment %zu of %zu", segment, n_segments);
    shard_rows(n_rows, n_segments, segment, &i_from, &i_to);
  }
  if (int rc = check_sizes(n_rows, n_cols, i_from, i_to)) return rc;
  if (n_rows == 0) return DC_OK;
  if (!d_coords || !d_fe || !d_nn_idx || !d_nn_d2 || !d_hd_idx || !d_hd_d2)
    return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (i_from != 0 || i_to != n_rows || n_segments > 1)
    dc::launch_nn_init((uint32_t)n_rows, d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2, s);
  if (i_from == i_to && n_segments == 0) return check_launch("nn init");
  const bool mfma = want_mfma(variant, n_cols);
  if ((variant == DC_VARIANT_MFMA || variant == DC_VARIANT_MFMA_PRUNED) && !mfma)
    return fail(DC_ERR_INVALID_ARGUMENT, "MFMA variant does not support n_cols=%zu", n_cols);
  if (variant == DC_VARIANT_MFMA32 && (!mfma || !dc::mfma32_supports(n_cols) || n_segments > 0))
    return fail(DC_ERR_INVALID_ARGUMENT, "the fp32-MFMA variant handles n_cols 9..10 and row ranges only (n_cols=%zu)", n_cols);
  if (mfma) {
    if (!d_workspace || workspace_bytes < dc::mfma_workspace_bytes(n_rows, n_cols))
      return fail(DC_ERR_WORKSPACE, "workspace of %zu bytes needed, got %zu",
                  dc::mfma_workspace_bytes(n_rows, n_cols), d_workspace ? workspace_bytes : 0);
    // (the pruned sweep packs reference POSITIONS of the padded order into 30 bits: kQueuePosMask)
    const bool full_sweep = variant == DC_VARIANT_MFMA || variant == DC_VARIANT_MFMA32 || n_rows + dc::kOrderPadRows > ((size_t)1 << 30);
    if (int rc = dc::mfma_prepare(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, d_workspace,
                                  full_sweep, s, stats_valid, !full_sweep, (stats_valid && !full_sweep) ? d_fe : nullptr))
      return fail(DC_ERR_HIP, "mfma_prepare failed (%d)", rc);
    if (variant == DC_VARIANT_MFMA32)
      dc::launch_nn_mfma32(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, d_fe, (uint32_t)i_from, (uint32_t)i_to, d_nn_idx,
                           d_nn_d2, d_hd_idx, d_hd_d2, d_workspace, s);
    else if (full_sweep)
      dc::launch_nn_mfma(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, d_fe, (uint32_t)i_from,
                         (uint32_t)i_to, d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2, d_workspace, s);
    else if (n_segments > 0)
      dc::launch_nn_pruned_segment(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, d_fe,
                                   (uint32_t)segment, (uint32_t)n_segments, d_nn_idx, d_nn_d2,
                                   d_hd_idx, d_hd_d2, d_workspace, s, stats_valid);
    else
      dc::launch_nn_pruned(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, d_fe, (uint32_t)i_from,
                           (uint32_t)i_to, d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2, d_workspace, s, stats_valid);
  }
  if (i_from != i_to && !dc::launch_nn_direct(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, d_fe, (uint32_t)i_from,
                            (uint32_t)i_to, d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2,
                            mfma ? (const uint32_t*)d_workspace : nullptr, s))
    return fail(DC_ERR_INVALID_ARGUMENT, "n_cols=%zu not supported", n_cols);
  return check_launch("nearest-neighbour sweep launch");
}
}  // namespace

extern "C" {

int dc_hip_nearest_neighbors_dev(const float* d_coords, size_t n_rows, size_t n_cols,
                                 const float* d_fe, size_t i_from, size_t i_to, uint32_t* d_nn_idx,
                                 float* d_nn_d2, uint32_t* d_hd_idx, float* d_hd_d2,
                                 void* d_workspace, size_t workspace_bytes, int variant,
                                 void* stream) {
  return nearest_neighbors_impl(d_coords, n_rows, n_cols, d_fe, i_from, i_to, 0, 0, d_nn_idx, d_nn_d2,
                                d_hd_idx, d_hd_d2, d_workspace, workspace_bytes, variant, stream);
}

int dc_hip_nearest_neighbors_segment_dev(const float* d_coords, size_t n_rows, size_t n_cols,
                                         const float* d_fe, size_t segment, size_t n_segments,
                                         uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx,
                                         float* d_hd_d2, void* d_workspace, size_t workspace_bytes,
                                         int variant, void* stream) {
  if (n_segments == 0) return fail(DC_ERR_INVALID_ARGUMENT, "n_segments must be positive");
  return nearest_neighbors_impl(d_coords, n_rows, n_cols, d_fe, 0, 0, segment, n_segments, d_nn_idx,
                                d_nn_d2, d_hd_idx, d_hd_d2, d_workspace, workspace_bytes, variant,
                                stream);
}

int dc_hip_neighbors_pack_dev(const uint32_t* d_nn_idx, const float* d_nn_d2, const uint32_t* d_hd_idx,
                              const float* d_hd_d2, size_t n_rows, unsigned long long* d_words,
                              void* stream) {
  if (n_rows == 0) return DC_OK;
  if (!d_nn_idx || !d_nn_d2 || !d_hd_idx || !d_hd_d2 || !d_words)
    return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  if (n_rows + 1 > (size_t)UINT32_MAX) return fail(DC_ERR_TOO_LARGE, "n_rows too large");
  dc::launch_nn_pack(d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2, (uint32_t)n_rows, d_words, (hipStream_t)stream);
  return check_launch("neighbour pack launch");
}

int dc_hip_neighbors_unpack_dev(const unsigned long long* d_words, size_t n_rows, uint32_t* d_nn_idx,
                                float* d_nn_d2, uint32_t* d_hd_idx, float* d_hd_d2, void* stream) {
  if (n_rows == 0) return DC_OK;
  if (!d_nn_idx || !d_nn_d2 || !d_hd_idx || !d_hd_d2 || !d_words)
    return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  if (n_rows + 1 > (size_t)UINT32_MAX) return fail(DC_ERR_TOO_LARGE, "n_rows too large");
  dc::launch_nn_unpack(d_words, (uint32_t)n_rows, d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2, (hipStream_t)stream);
  return check_launch("neighbour unpack launch");
}

size_t dc_hip_neighbors_block_rows(size_t n_rows, size_t n_cols, size_t n_segments) {
  return dc::nn_block_rows(n_rows, n_cols, n_segments);
}

namespace {
// did dc_hip_nearest_neighbors_segment_dev(variant) run the pruned matrix-core sweep (segments of the spatial order)?
bool segment_sweep_is_pruned(int variant, size_t n_rows, size_t n_cols) {
  variant &= DC_VARIANT_MASK;
  return want_mfma(variant, n_cols) && variant != DC_VARIANT_MFMA && n_rows + dc::kOrderPadRows <= ((size_t)1 << 30);
}
}  // namespace

int dc_hip_neighbors_block_pack_dev(const uint32_t* d_nn_idx, const float* d_nn_d2, const uint32_t* d_hd_idx,
                                    const float* d_hd_d2, size_t n_rows, size_t n_cols, size_t segment,
                                    size_t n_segments, const void* d_workspace, size_t workspace_bytes, int variant,
                                    uint32_t* d_block, void* stream) {
  if (n_segments == 0 || segment >= n_segments) return fail(DC_ERR_INVALID_ARGUMENT, "segment %zu of %zu", segment, n_segments);
  if (int rc = check_sizes(n_rows, n_cols, 0, n_rows)) return rc;
  if (n_rows == 0) return DC_OK;
  if (!d_nn_idx || !d_nn_d2 || !d_hd_idx || !d_hd_d2 || !d_block) return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  const bool pruned = segment_sweep_is_pruned(variant, n_rows, n_cols);
  if (pruned && (!d_workspace || workspace_bytes < dc::mfma_workspace_bytes(n_rows, n_cols)))
    return fail(DC_ERR_WORKSPACE, "the workspace of the segment sweep is needed (its ordering)");
  dc::launch_nn_block_pack(d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2, (uint32_t)n_rows, (uint32_t)n_cols, (uint32_t)segment,
                           (uint32_t)n_segments, pruned, d_workspace, d_block, (hipStream_t)stream);
  return check_launch("neighbour block pack launch");
}

int dc_hip_neighbors_block_unpack_dev(const uint32_t* d_blocks, size_t n_rows, size_t n_cols, size_t n_segments,
                                      const void* d_workspace, size_t workspace_bytes, int variant,
                                      uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx, float* d_hd_d2,
                                      void* stream) {
  if (n_segments == 0) return fail(DC_ERR_INVALID_ARGUMENT, "n_segments must be positive");
  if (int rc = check_sizes(n_rows, n_cols, 0, n_rows)) return rc;
  if (n_rows == 0) return DC_OK;
  if (!d_nn_idx || !d_nn_d2 || !d_hd_idx || !d_hd_d2 || !d_blocks) return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  const bool pruned = segment_sweep_is_pruned(variant, n_rows, n_cols);
  if (pruned && (!d_workspace || workspace_bytes < dc::mfma_workspace_bytes(n_rows, n_cols)))
    return fail(DC_ERR_WORKSPACE, "the workspace of the segment sweep is needed (its ordering)");
  dc::launch_nn_block_unpack(d_blocks, (uint32_t)n_rows, (uint32_t)n_cols, (uint32_t)n_segments, pruned, d_workspace,
                             d_nn_idx, d_nn_d2, d_hd_idx, d_hd_d2, (hipStream_t)stream);
  return check_launch("neighbour block unpack launch");
}

int dc_hip_sigma2_dev(const float* d_nn_d2, size_t n_rows, double* sigma2_out, void* stream) {
  if (!sigma2_out) return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  if (n_rows == 0) {
    *sigma2_out = 0.0 / 0.0;  // the reference divides by nh.size() == 0
    return DC_OK;
  }
  std::vector<float> h(n_rows);
  hipStream_t s = (hipStream_t)stream;
  DC_HIP_TRY(hipMemcpyAsync(h.data(), d_nn_d2, sizeof(float) * n_rows, hipMemcpyDeviceToHost, s));
  DC_HIP_TRY(hipStreamSynchronize(s));
  double acc = 0.0;  // frame order, double: density_clustering.cpp:334-343
  for (size_t i = 0; i < n_rows; ++i) acc += (double)h[i];
  *sigma2_out = acc / (double)n_rows;
  return DC_OK;
}

int dc_hip_radius_pairs_dev(const float* d_coords, size_t n_rows, size_t n_cols, float r2,
                            uint32_t* d_pops, uint32_t* d_pairs, size_t capacity,
                            unsigned long long* d_count, void* d_workspace, size_t workspace_bytes,
                            void* stream) {
  if (int rc = check_sizes(n_rows, n_cols, 0, n_rows)) return rc;
  if (!d_count) return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (n_rows == 0) {
    DC_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s));
    return DC_OK;
  }
  if (!d_coords || !d_pops) return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  if (!dc::mfma_supports(n_cols))
    return fail(DC_ERR_INVALID_ARGUMENT, "radius pairs need n_cols <= 64 (got %zu)", n_cols);
  if (!d_workspace || workspace_bytes < dc::mfma_workspace_bytes(n_rows, n_cols))
    return fail(DC_ERR_WORKSPACE, "workspace of %zu bytes needed, got %zu",
                dc::mfma_workspace_bytes(n_rows, n_cols), d_workspace ? workspace_bytes : 0);
  DC_HIP_TRY(hipMemsetAsync(d_pops, 0, sizeof(uint32_t) * n_rows, s));
  if (int rc = dc::mfma_prepare(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, d_workspace, false, s, false, true))
    return fail(DC_ERR_HIP, "mfma_prepare failed (%d)", rc);
  dc::launch_radius_pairs(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, r2, d_pops, (uint2*)d_pairs,
                          (unsigned long long)capacity, d_count, d_workspace, s);
  return check_launch("radius pair sweep launch");
}

int dc_hip_radius_min_edge_dev(const float* d_coords, size_t n_rows, size_t n_cols, float r2,
                               const uint32_t* d_comp, const uint32_t* d_rank,
                               unsigned long long* d_best, uint32_t* d_pops, void* d_workspace,
                               size_t workspace_bytes, void* stream) {
  return dc_hip_radius_min_edge_segment_dev(d_coords, n_rows, n_cols, r2, d_comp, d_rank, 0, 0, d_best,
                                            d_pops, d_workspace, workspace_bytes, stream);
}

int dc_hip_radius_min_edge_segment_dev(const float* d_coords, size_t n_rows, size_t n_cols, float r2,
                                       const uint32_t* d_comp, const uint32_t* d_rank, size_t segment,
                                       size_t n_segments, unsigned long long* d_best, uint32_t* d_pops,
                                       void* d_workspace, size_t workspace_bytes, void* stream) {
  if (n_segments > 0 && segment >= n_segments)
    return fail(DC_ERR_INVALID_ARGUMENT, "segment %zu of %zu", segment, n_segments);
  if (int rc = check_sizes(n_rows, n_cols, 0, n_rows)) return rc;
  if (n_rows == 0) return DC_OK;
  if (!d_coords || !d_comp || !d_rank || !d_best || !d_pops)
    return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  if (!dc::mfma_supports(n_cols))
    return fail(DC_ERR_INVALID_ARGUMENT, "the radius graph needs n_cols <= 64 (got %zu)", n_cols);
  if (n_rows > dc::kMinEdgeMaxRows)
    return fail(DC_ERR_INVALID_ARGUMENT, "min-edge sweeps need n_rows <= %zu (got %zu)",
                (size_t)dc::kMinEdgeMaxRows, n_rows);
  if (!d_workspace || workspace_bytes < dc::mfma_workspace_bytes(n_rows, n_cols))
    return fail(DC_ERR_WORKSPACE, "workspace of %zu bytes needed, got %zu",
                dc::mfma_workspace_bytes(n_rows, n_cols), d_workspace ? workspace_bytes : 0);
  hipStream_t s = (hipStream_t)stream;
  DC_HIP_TRY(hipMemsetAsync(d_pops, 0, sizeof(uint32_t) * n_rows, s));
  if (int rc = dc::mfma_prepare(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, d_workspace, false, s, false, true))
    return fail(DC_ERR_HIP, "mfma_prepare failed (%d)", rc);
  dc::launch_radius_min_edge(d_coords, (uint32_t)n_rows, (uint32_t)n_cols, r2, d_comp, d_rank, d_best,
                             d_pops, d_workspace, s, (uint32_t)segment, (uint32_t)n_segments);
  return check_launch("min-edge sweep launch");
}

// ------------------------------------------------------------------------------------------
// host-pointer wrappers
// ------------------------------------------------------------------------------------------
namespace {

// the host-pointer entry points select their device themselves and leave the caller's current device as it was
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() {
    if (hipGetDevice(&prev) != hipSuccess) {
      (void)hipGetLastError();
      prev = -1;
    }
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

struct DeviceJob {
  int device = 0;
  hipStream_t stream = nullptr;
  float* d_coords = nullptr;
  float* d_fe = nullptr;
  uint32_t* d_pops = nullptr;
  uint32_t* d_idx = nullptr;  // [2][n_rows]
  float* d_d2 = nullptr;      // [2][n_rows]
  void* d_ws = nullptr;
  size_t ws_bytes = 0;
  void release() {
    (void)hipSetDevice(device);
    if (d_coords) (void)hipFree(d_coords);
    if (d_fe) (void)hipFree(d_fe);
    if (d_pops) (void)hipFree(d_pops);
    if (d_idx) (void)hipFree(d_idx);
    if (d_d2) (void)hipFree(d_d2);
    if (d_ws) (void)hipFree(d_ws);
    if (stream) (void)hipStreamDestroy(stream);
    *this = DeviceJob();
  }
};

int job_open(DeviceJob& j, int device, const float* coords, size_t n_rows, size_t n_cols) {
  int n = dc_hip_device_count();
  if (n < 0) return n;
  if (n == 0) return fail(DC_ERR_NO_DEVICE, "no HIP device found");
  if (device < 0 || device >= n)
    return fail(DC_ERR_INVALID_ARGUMENT, "device %d out of range [0,%d)", device, n);
  j.device = device;
  DC_HIP_TRY(hipSetDevice(device));
  DC_HIP_TRY(hipStreamCreate(&j.stream));
  DC_HIP_TRY(hipMalloc((void**)&j.d_coords, sizeof(float) * std::max<size_t>(1, n_rows * n_cols)));
  DC_HIP_TRY(hipMemcpyAsync(j.d_coords, coords, sizeof(float) * n_rows * n_cols,
                            hipMemcpyHostToDevice, j.stream));
  j.ws_bytes = dc_hip_workspace_bytes(n_rows, n_cols, 1);
  if (j.ws_bytes) DC_HIP_TRY(hipMalloc(&j.d_ws, j.ws_bytes));
  return DC_OK;
}

}  // namespace

int dc_hip_populations(const float* coords, size_t n_rows, size_t n_cols, const float* radii,
                       size_t n_radii, size_t i_from, size_t i_to, int device, uint32_t* pops) {
  if (int rc = check_sizes(n_rows, n_cols, i_from, i_to)) return rc;
  if (n_rows == 0 || n_radii == 0) return DC_OK;
  if (!coords || !radii || !pops) return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  DeviceGuard guard;
  DeviceJob j;
  int rc = job_open(j, device, coords, n_rows, n_cols);
  if (rc == DC_OK) {
    hipError_t e = hipMalloc((void**)&j.d_pops, sizeof(uint32_t) * n_radii * n_rows);
    if (e != hipSuccess) rc = fail(DC_ERR_HIP, "hipMalloc pops: %s", hipGetErrorString(e));
  }
  if (rc == DC_OK)
    rc = dc_hip_populations_dev(j.d_coords, n_rows, n_cols, radii, n_radii, i_from, i_to, j.d_pops,
                                j.d_ws, j.ws_bytes, DC_VARIANT_AUTO, j.stream);
  if (rc == DC_OK) {
    hipError_t e = hipMemcpyAsync(pops, j.d_pops, sizeof(uint32_t) * n_radii * n_rows,
                                  hipMemcpyDeviceToHost, j.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(j.stream);
    if (e != hipSuccess) rc = fail(DC_ERR_HIP, "population sweep: %s", hipGetErrorString(e));
  }
  j.release();
  return rc;
}

int dc_hip_nearest_neighbors(const float* coords, size_t n_rows, size_t n_cols, const float* fe,
                             size_t i_from, size_t i_to, int device, uint32_t* nn_idx, float* nn_d2,
                             uint32_t* hd_idx, float* hd_d2) {
  if (int rc = check_sizes(n_rows, n_cols, i_from, i_to)) return rc;
  if (n_rows == 0) return DC_OK;
  if (!coords || !fe || !nn_idx || !nn_d2 || !hd_idx || !hd_d2)
    return fail(DC_ERR_INVALID_ARGUMENT, "null pointer");
  DeviceGuard guard;
  DeviceJob j;
  int rc = job_open(j, device, coords, n_rows, n_cols);
  hipError_t e = hipSuccess;
  if (rc == DC_OK) {
    e = hipMalloc((void**)&j.d_fe, sizeof(float) * n_rows);
    if (e == hipSuccess) e = hipMalloc((void**)&j.d_idx, sizeof(uint32_t) * 2 * n_rows);
    if (e == hipSuccess) e = hipMalloc((void**)&j.d_d2, sizeof(float) * 2 * n_rows);
    if (e == hipSuccess)
      e = hipMemcpyAsync(j.d_fe, fe, sizeof(float) * n_rows, hipMemcpyHostToDevice, j.stream);
    if (e != hipSuccess) rc = fail(DC_ERR_HIP, "nn setup: %s", hipGetErrorString(e));
  }
  if (rc == DC_OK)
    rc = dc_hip_nearest_neighbors_dev(j.d_coords, n_rows, n_cols, j.d_fe, i_from, i_to, j.d_idx,
                                      j.d_d2, j.d_idx + n_rows, j.d_d2 + n_rows, j.d_ws,
                                      j.ws_bytes, DC_VARIANT_AUTO, j.stream);
  if (rc == DC_OK) {
    e = hipMemcpyAsync(nn_idx, j.d_idx, sizeof(uint32_t) * n_rows, hipMemcpyDeviceToHost, j.stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(hd_idx, j.d_idx + n_rows, sizeof(uint32_t) * n_rows, hipMemcpyDeviceToHost,
                         j.stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(nn_d2, j.d_d2, sizeof(float) * n_rows, hipMemcpyDeviceToHost, j.stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(hd_d2, j.d_d2 + n_rows, sizeof(float) * n_rows, hipMemcpyDeviceToHost,
                         j.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(j.stream);
    if (e != hipSuccess) rc = fail(DC_ERR_HIP, "nn sweep: %s", hipGetErrorString(e));
  }
  j.release();
  return rc;
}

}  // extern "C"
