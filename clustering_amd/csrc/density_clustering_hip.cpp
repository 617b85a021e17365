// density_clustering_hip.cpp -- implementation of density_clustering_hip.hpp over the C ABI.
// Plain C++ (no HIP headers): everything device-side lives behind include/dc_density.h.
#include "density_clustering_hip.hpp"
#include "screening_host.hpp"

#include "../../include/dc_density.h"

#include <algorithm>
#include <cstring>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <mutex>

namespace Clustering {
namespace Density {
namespace CUDA {

namespace {
int g_last_status = 0;

void fail_now(const std::string& msg) {
  // reference convention: message on stderr, then exit (density_clustering_cuda.cu:21-30)
  std::cerr << "HIP error: " << msg << "\n" << dc_hip_last_error() << std::endl;
  exit(EXIT_FAILURE);
}

void must(int status, const char* what) {
  g_last_status = status;
  if (status != DC_OK) fail_now(what);
}

// The reference's call sequence is calculate_populations(coords, ...) -> host free energies ->
// nearest_neighbors(coords, ..., fe) -> screening(..., coords, ...) on ONE trajectory
// (density_clustering.cpp:616-621, 659-663, 716-720, 746-748), and its GPU code uploads that trajectory
// again inside every call (density_clustering_cuda.cu:65-81, 201-225).  Here the first call opens a
// session on all GPUs (dc_hip_session_open) and the later ones find the coordinates -- and whatever the
// previous phase left in HBM -- still resident.  The session is keyed on the pointer, the shape and a
// fingerprint of the WHOLE buffer, so a caller that rewrites the buffer in place gets a fresh upload;
// HIP::release_resident() drops the session and frees the device memory (screening does that itself once
// a scan is over -- see release_resident's note).
struct Resident {
  dc_hip_session* session = nullptr;
  const float* coords = nullptr;
  std::size_t n_rows = 0, n_cols = 0;
  std::uint64_t fingerprint = 0;
};
Resident* g_resident = nullptr;   // (heap object, never destroyed: no HIP calls during static destruction)
bool g_hdn_invalidate = false;    // HIP::invalidate_neighborhood_cache(): high_density_neighborhood starts from scratch

// fingerprint of the WHOLE buffer (every word enters; four interleaved multiply-xor lanes, one pass at memory
// bandwidth: ~10 ms for the 40 MB of C3, against the sweeps' 30 ms and the upload's 5): an in-place change of
// any element, or another trajectory at the same address, is seen and the stale device copy is dropped
std::uint64_t full_fingerprint(const float* coords, std::size_t n) {
  const std::uint32_t* w = reinterpret_cast<const std::uint32_t*>(coords);
  std::uint64_t h[4] = {1469598103934665603ull ^ n, 0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull};
  std::size_t k = 0;
  for (; k + 4 <= n; k += 4)
    for (int l = 0; l < 4; ++l) {
      h[l] = (h[l] ^ w[k + l]) * 0x100000001B3ull;
      h[l] ^= h[l] >> 29;
    }
  for (; k < n; ++k) h[k & 3] = (h[k & 3] ^ w[k]) * 0x100000001B3ull;
  std::uint64_t out = h[0];
  for (int l = 1; l < 4; ++l) out = (out ^ (h[l] + 0x9E3779B97F4A7C15ull + (out << 6) + (out >> 2))) * 0x100000001B3ull;
  return out;
}

void check_abi_once() {
  static const bool ok = [] {
    if (dc_hip_abi_version() != DC_HIP_ABI_VERSION) {
      std::cerr << "HIP error: libdcdensity.so has ABI version " << dc_hip_abi_version() << ", this host was built "
                << "against " << DC_HIP_ABI_VERSION << std::endl;
      exit(EXIT_FAILURE);
    }
    return true;
  }();
  (void)ok;
}

dc_hip_session* resident_session(const float* coords, std::size_t n_rows, std::size_t n_cols) {
  check_abi_once();
  const std::uint64_t fp = full_fingerprint(coords, n_rows * n_cols);
  if (!g_resident) g_resident = new Resident();
  Resident& r = *g_resident;
  if (r.session && r.coords == coords && r.n_rows == n_rows && r.n_cols == n_cols && r.fingerprint == fp)
    return r.session;
  if (r.session) dc_hip_session_close(r.session);
  r = Resident();
  must(dc_hip_session_open(coords, n_rows, n_cols, nullptr, 0, &r.session), "uploading the coordinates");
  // (the reference's merge, density_clustering_cuda.cu:152-180, always does what it says; a session that could not get
  //  RCCL on several devices merges through the host -- correct and slow, so say it)
  if (dc_hip_session_merge_mode(r.session) == 2) std::cerr << "warning: " << dc_hip_session_merge_note(r.session) << std::endl;
  r.coords = coords;
  r.n_rows = n_rows;
  r.n_cols = n_cols;
  r.fingerprint = fp;
  return r.session;
}
}  // namespace

void check_error(std::string msg) {
  if (g_last_status != DC_OK) fail_now(msg);
}

int get_num_gpus() {
  check_abi_once();
  int n_gpus = dc_hip_device_count();
  if (n_gpus < 0) fail_now("trying to get number of available GPUs");
  if (n_gpus == 0) {
    std::cerr << "error: no HIP-compatible GPUs found" << std::endl;
    exit(EXIT_FAILURE);
  }
  return n_gpus;
}

Pops calculate_populations_per_gpu(const float* coords, std::size_t n_rows, std::size_t n_cols,
                                   std::vector<float> radii, std::size_t i_from, std::size_t i_to,
                                   int i_gpu) {
  std::sort(radii.begin(), radii.end());   // (results are keyed by radius: see calculate_populations)
  std::vector<std::uint32_t> partial(n_rows * radii.size());
  must(dc_hip_populations(coords, n_rows, n_cols, radii.data(), radii.size(), i_from, i_to, i_gpu,
                          partial.data()),
       "population sweep");
  Pops pops;
  for (std::size_t r = 0; r < radii.size(); ++r) {
    std::vector<std::size_t>& dst = pops[radii[r]];
    dst.assign(n_rows, 0);
    for (std::size_t i = i_from; i < i_to; ++i) dst[i] = partial[r * n_rows + i];
  }
  return pops;
}

Pops calculate_populations_partial(const float* coords, std::size_t n_rows, std::size_t n_cols,
                                   std::vector<float> radii, std::size_t i_from, std::size_t i_to,
                                   int i_gpu) {
  return calculate_populations_per_gpu(coords, n_rows, n_cols, radii, i_from, i_to, i_gpu);
}

Pops calculate_populations_partial(const float* coords, const std::vector<float>& /*sorted_coords*/,
                                   const std::vector<float>& /*blimits*/, std::size_t n_rows,
                                   std::size_t n_cols, std::vector<float> radii, std::size_t i_from,
                                   std::size_t i_to, int i_gpu) {
  return calculate_populations_per_gpu(coords, n_rows, n_cols, radii, i_from, i_to, i_gpu);
}

Pops calculate_populations(const float* coords, const std::size_t n_rows, const std::size_t n_cols,
                           std::vector<float> radii) {
  // (density_clustering_cuda.cu:147 sorts descending for its early break; the result is a map keyed by radius, so the
  //  order is the library's to choose: ascending lets the multi-radius sweep leave out the small radii a chain holds
  //  nothing of -- dc_mfma_msym.hpp mr_chain_k)
  std::sort(radii.begin(), radii.end());
  get_num_gpus();
  // all GPUs, one segment each, partials summed on the devices (density_clustering_cuda.cu:149-180 shards
  // row blocks over OpenMP threads and sums on the host)
  dc_hip_session* session = resident_session(coords, n_rows, n_cols);
  std::vector<std::uint32_t> flat(n_rows * radii.size());
  must(dc_hip_session_populations(session, radii.data(), radii.size(), flat.data()), "population sweep");
  Pops pops;
  for (std::size_t r = 0; r < radii.size(); ++r) {
    std::vector<std::size_t>& dst = pops[radii[r]];
    dst.assign(flat.begin() + r * n_rows, flat.begin() + (r + 1) * n_rows);
  }
  return pops;
}

std::tuple<Neighborhood, Neighborhood> nearest_neighbors_per_gpu(
    const float* coords, const std::size_t n_rows, const std::size_t n_cols,
    const std::vector<float>& free_energy, std::size_t i_from, std::size_t i_to, int i_gpu) {
  std::vector<std::uint32_t> nn_idx(n_rows), hd_idx(n_rows);
  std::vector<float> nn_d2(n_rows), hd_d2(n_rows);
  must(dc_hip_nearest_neighbors(coords, n_rows, n_cols, free_energy.data(), i_from, i_to, i_gpu,
                                nn_idx.data(), nn_d2.data(), hd_idx.data(), hd_d2.data()),
       "nearest-neighbour sweep");
  Neighborhood nh, nhhd;
  for (std::size_t i = 0; i < n_rows; ++i) {
    nh.emplace_hint(nh.end(), i, Clustering::Tools::Neighbor(nn_idx[i], nn_d2[i]));
    nhhd.emplace_hint(nhhd.end(), i, Clustering::Tools::Neighbor(hd_idx[i], hd_d2[i]));
  }
  return std::make_tuple(nh, nhhd);
}

std::tuple<Neighborhood, Neighborhood> nearest_neighbors(const float* coords,
                                                         const std::size_t n_rows,
                                                         const std::size_t n_cols,
                                                         const std::vector<float>& free_energy) {
  // density_clustering_cuda.cu:286-328 runs one host thread per GPU on a row block each and stitches the
  // blocks on the host; the session does the same with one segment per device, concurrently, and merges
  // on the devices.  The free energies are the caller's (the reference computes them on the host).
  get_num_gpus();
  if (free_energy.size() != n_rows) fail_now("nearest_neighbors: free_energy does not match n_rows");
  dc_hip_session* session = resident_session(coords, n_rows, n_cols);
  must(dc_hip_session_set_free_energies(session, free_energy.data()), "uploading the free energies");
  std::vector<std::uint32_t> nn_idx(n_rows), hd_idx(n_rows);
  std::vector<float> nn_d2(n_rows), hd_d2(n_rows);
  must(dc_hip_session_nearest_neighbors(session, nn_idx.data(), nn_d2.data(), hd_idx.data(), hd_d2.data(), nullptr),
       "nearest-neighbour sweep");
  Neighborhood nh, nhhd;
  for (std::size_t i = 0; i < n_rows; ++i) {
    nh.emplace_hint(nh.end(), i, Clustering::Tools::Neighbor(nn_idx[i], nn_d2[i]));
    nhhd.emplace_hint(nhhd.end(), i, Clustering::Tools::Neighbor(hd_idx[i], hd_d2[i]));
  }
  return std::make_tuple(nh, nhhd);
}

std::vector<std::size_t> screening(const std::vector<float>& free_energy, const Neighborhood& nh,
                                   const float free_energy_threshold, const float* coords,
                                   const std::size_t n_rows, const std::size_t n_cols,
                                   const std::vector<std::size_t> initial_clusters) {
  namespace H = Clustering::Density::HIP;
  // compute_sigma2 (density_clustering.cpp:334-343): double sum in map (frame) order
  double sigma2 = 0.0;
  for (const auto& match : nh) sigma2 += match.second.second;
  sigma2 /= nh.size();
  const float max_dist = (float)(4 * sigma2);   // the reference passes 4*sigma2 into a float parameter
  // one free-energy order and one graph per trajectory, shared by the thresholds of a scan.  A scan
  // that starts from an empty clustering and feeds every result into the next, higher threshold (the
  // reference's only use, density_clustering.cpp:786-811) is served by the spanning forest
  // (screening_host.hpp); any other initial clustering gets the full pair list.
  struct Cache {
    const float* coords = nullptr;
    std::size_t n_rows = 0, n_cols = 0;
    float max_dist = 0.0f;
    std::vector<float> fe;
    std::vector<H::FreeEnergy> fe_sorted;
    H::RadiusGraph forest, graph;
    bool have_forest = false, have_graph = false;
    std::vector<std::size_t> last_result;
    float last_threshold = 0.0f;
  };
  static Cache cache;
  if (cache.coords != coords || cache.n_rows != n_rows || cache.n_cols != n_cols ||
      cache.max_dist != max_dist || cache.fe != free_energy) {
    cache = Cache();
    cache.coords = coords;
    cache.n_rows = n_rows;
    cache.n_cols = n_cols;
    cache.max_dist = max_dist;
    cache.fe = free_energy;
    cache.fe_sorted = H::sorted_free_energies(free_energy);
  }
  bool empty_start = initial_clusters.size() != n_rows;
  if (!empty_start) {
    empty_start = true;
    for (std::size_t c : initial_clusters) empty_start = empty_start && (c == 0);
  }
  const bool continues_scan = !empty_start && !cache.last_result.empty() &&
                              !(free_energy_threshold < cache.last_threshold) &&
                              initial_clusters == cache.last_result;
  const bool use_forest = (empty_start || continues_scan) && n_rows <= ((std::size_t)1 << 24);
  std::string err;
  if (use_forest && !cache.have_forest) {
    if (!H::build_radius_forest(resident_session(coords, n_rows, n_cols), n_rows, max_dist, cache.fe_sorted,
                                &cache.forest, &err)) {
      std::cerr << "error during screening (radius forest)\n" << err << std::endl;
      exit(EXIT_FAILURE);
    }
    cache.have_forest = true;
  }
  if (!use_forest && !cache.have_graph) {
    if (!H::build_radius_graph(resident_session(coords, n_rows, n_cols), n_rows, max_dist, &cache.graph, &err)) {
      std::cerr << "error during screening (radius graph)\n" << err << std::endl;
      exit(EXIT_FAILURE);
    }
    cache.have_graph = true;
  }
  std::vector<std::size_t> result =
      H::screening_with_graph(free_energy, cache.fe_sorted, use_forest ? cache.forest : cache.graph,
                              free_energy_threshold, initial_clusters);
  cache.last_result = result;
  cache.last_threshold = free_energy_threshold;
  return result;
}

std::vector<std::size_t> sanitize_state_names(std::vector<std::size_t> clustering) {
  // Declared in density_clustering_cuda.hpp:44-45 and defined nowhere in the reference.  What the name and
  // the screening code around it (normalized_cluster_names, density_clustering.cpp:437-456) suggest: states
  // renumbered 1..K in ascending order of their old names, 0 (= no state) kept.
  std::vector<std::size_t> names(clustering);
  std::sort(names.begin(), names.end());
  names.erase(std::unique(names.begin(), names.end()), names.end());
  std::size_t shift = (!names.empty() && names[0] == 0) ? 0 : 1;
  for (std::size_t& c : clustering)
    c = (std::size_t)(std::lower_bound(names.begin(), names.end(), c) - names.begin()) + shift;
  return clustering;
}

std::set<std::size_t> high_density_neighborhood(const float* coords, const std::size_t n_cols,
                                                const std::vector<FreeEnergy>& sorted_fe,
                                                const std::size_t i_frame, const std::size_t limit,
                                                const float max_dist) {
  // CPU semantics: density_clustering.cpp:292-332 -- the positions j < limit (in order of free energy) of
  // the frames whose squared distance to frame sorted_fe[i_frame] is < max_dist, plus i_frame itself.  The
  // partner lists of ALL frames come from one GPU sweep (radius graph), cached per (coords, max_dist).
  namespace H = Clustering::Density::HIP;
  const std::size_t n_rows = sorted_fe.size();
  // The reference calls this once per FRAME of a screening pass, i_frame ascending (density_clustering_common.cpp:37-134),
  // so a cache hit must cost O(1) INSIDE a pass: there the key is the identity of both arrays (address, size) plus a
  // fingerprint of 64 samples of each.  Whenever a new pass begins -- i_frame does not continue upwards from the previous
  // call -- or invalidate_neighborhood_cache() was called, EVERY word of both arrays is fingerprinted again (one pass at
  // memory bandwidth, ~10 ms at C3) and compared with what the cached graph was built from: an in-place edit of a few rows
  // or a re-sorted order between passes is caught, not "probably" caught.  Precondition that remains
  // (density_clustering_hip.hpp): coords and sorted_fe are not modified BETWEEN the calls of one ascending pass.
  // Cost of any OTHER access pattern: a caller that repeats a frame, walks downwards or interleaves two passes pays the
  // full fingerprint (O(N D), ~10 ms at C3) on every such call -- safe, not fast; only the reference's ascending loop is
  // O(1) per call.  One caller at a time: the cache is process-wide, calls are serialised by a mutex.
  static std::mutex cache_mutex;
  std::lock_guard<std::mutex> cache_lock(cache_mutex);
  struct Cache {
    const float* coords = nullptr;
    const FreeEnergy* order = nullptr;
    std::size_t n_rows = 0, n_cols = 0;
    float max_dist = 0.0f;
    std::uint64_t sample_fp = 0, full_fp = 0;
    std::size_t last_i_frame = 0;
    bool valid = false;
    H::RadiusGraph graph;
    std::vector<std::uint32_t> pos_of;   // frame -> position in sorted_fe
  };
  static Cache cache;
  if (g_hdn_invalidate) {
    cache = Cache();
    g_hdn_invalidate = false;
  }
  std::uint64_t sample_fp = 1469598103934665603ull;
  {
    const std::uint32_t* w = reinterpret_cast<const std::uint32_t*>(coords);
    const std::size_t n_words = n_rows * n_cols;
    for (std::size_t k = 0; k < 64 && n_words > 0; ++k)
      sample_fp = (sample_fp ^ w[(k * 0x9E3779B97F4A7C15ull) % n_words]) * 1099511628211ull;
    for (std::size_t k = 0; k < 64 && n_rows > 0; ++k) {
      const FreeEnergy& e = sorted_fe[(k * 0xC2B2AE3D27D4EB4Full) % n_rows];
      std::uint32_t fb;
      std::memcpy(&fb, &e.second, sizeof fb);
      sample_fp = (sample_fp ^ e.first ^ ((std::uint64_t)fb << 32)) * 1099511628211ull;
    }
  }
  auto full_fp_now = [&]() {
    std::uint64_t h = full_fingerprint(coords, n_rows * n_cols);
    for (std::size_t p = 0; p < n_rows; ++p) {
      std::uint32_t fb;
      std::memcpy(&fb, &sorted_fe[p].second, sizeof fb);
      h = (h ^ (std::uint64_t)sorted_fe[p].first ^ ((std::uint64_t)fb << 32)) * 0x100000001B3ull;
      h ^= h >> 29;
    }
    return h;
  };
  const bool same_key = cache.valid && cache.coords == coords && cache.order == sorted_fe.data() && cache.n_rows == n_rows &&
                        cache.n_cols == n_cols && cache.max_dist == max_dist && cache.sample_fp == sample_fp;
  const bool new_pass = !same_key || i_frame <= cache.last_i_frame;
  std::uint64_t full_fp = 0;
  bool hit = same_key;
  if (new_pass) {
    full_fp = full_fp_now();
    hit = same_key && full_fp == cache.full_fp;
  }
  if (!hit) {
    cache = Cache();
    std::string err;
    if (!H::build_radius_graph(resident_session(coords, n_rows, n_cols), n_rows, max_dist, &cache.graph, &err)) {
      std::cerr << "error in high_density_neighborhood (radius graph)\n" << err << std::endl;
      exit(EXIT_FAILURE);
    }
    cache.coords = coords;
    cache.order = sorted_fe.data();
    cache.n_rows = n_rows;
    cache.n_cols = n_cols;
    cache.max_dist = max_dist;
    cache.sample_fp = sample_fp;
    cache.full_fp = full_fp;
    cache.valid = true;
    cache.pos_of.assign(n_rows, 0xFFFFFFFFu);
    for (std::size_t p = 0; p < n_rows; ++p) {
      if (sorted_fe[p].first >= n_rows) {
        std::cerr << "error in high_density_neighborhood: frame id " << sorted_fe[p].first << " out of range" << std::endl;
        exit(EXIT_FAILURE);
      }
      cache.pos_of[sorted_fe[p].first] = (std::uint32_t)p;
    }
  }
  cache.last_i_frame = i_frame;
  if (i_frame >= n_rows) {
    std::cerr << "error in high_density_neighborhood: i_frame " << i_frame << " out of range" << std::endl;
    exit(EXIT_FAILURE);
  }
  std::set<std::size_t> nh;
  const std::size_t frame = sorted_fe[i_frame].first;
  for (std::uint64_t k = cache.graph.offset[frame]; k < cache.graph.offset[frame + 1]; ++k) {
    const std::size_t j = cache.pos_of[cache.graph.neighbor[k]];
    if (j < limit) nh.insert(j);
  }
  nh.insert(i_frame);
  return nh;
}

}  // namespace CUDA

namespace HIP {

void invalidate_neighborhood_cache() { CUDA::g_hdn_invalidate = true; }

void release_resident() {
  if (CUDA::g_resident && CUDA::g_resident->session) dc_hip_session_close(CUDA::g_resident->session);
  if (CUDA::g_resident) *CUDA::g_resident = CUDA::Resident();
}

DensityResult density_all(const float* coords, std::size_t n_rows, std::size_t n_cols,
                          const std::vector<float>& radii, std::size_t fe_radius_index,
                          bool want_neighbors, int n_gpus) {
  DensityResult out;
  std::vector<std::uint32_t> pops(n_rows * radii.size()), nn_idx, hd_idx;
  out.free_energy.assign(n_rows, 0.0f);
  if (want_neighbors) {
    nn_idx.resize(n_rows);
    hd_idx.resize(n_rows);
    out.nn_d2.resize(n_rows);
    out.hd_d2.resize(n_rows);
  }
  const int rc = dc_hip_density_all(coords, n_rows, n_cols, radii.data(), radii.size(),
                                    fe_radius_index, n_gpus, pops.data(), out.free_energy.data(),
                                    want_neighbors ? nn_idx.data() : nullptr,
                                    want_neighbors ? out.nn_d2.data() : nullptr,
                                    want_neighbors ? hd_idx.data() : nullptr,
                                    want_neighbors ? out.hd_d2.data() : nullptr);
  if (rc != DC_OK) {
    std::cerr << "HIP error: density sweep\n" << dc_hip_last_error() << std::endl;
    exit(EXIT_FAILURE);
  }
  out.pops.resize(radii.size());
  for (std::size_t r = 0; r < radii.size(); ++r)
    out.pops[r].assign(pops.begin() + r * n_rows, pops.begin() + (r + 1) * n_rows);
  if (want_neighbors) {
    out.nn_idx.assign(nn_idx.begin(), nn_idx.end());
    out.hd_idx.assign(hd_idx.begin(), hd_idx.end());
    double s = 0.0;   // frame order, double: density_clustering.cpp:334-343
    for (std::size_t i = 0; i < n_rows; ++i) s += (double)out.nn_d2[i];
    out.sigma2 = s / (double)n_rows;
  }
  return out;
}

}  // namespace HIP
}  // namespace Density
}  // namespace Clustering
