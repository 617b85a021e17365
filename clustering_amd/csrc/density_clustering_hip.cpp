// density_clustering_hip.cpp -- implementation of density_clustering_hip.hpp over the C ABI.
// Plain C++ (no HIP headers): everything device-side lives behind include/dc_density.h.
#include "density_clustering_hip.hpp"
#include "screening_host.hpp"

#include "../../include/dc_density.h"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <iostream>

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
}  // namespace

void check_error(std::string msg) {
  if (g_last_status != DC_OK) fail_now(msg);
}

int get_num_gpus() {
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
  std::sort(radii.begin(), radii.end(), std::greater<float>());   // density_clustering_cuda.cu:147
  const int n_gpus = get_num_gpus();
  std::vector<std::uint32_t> flat(n_rows * radii.size());
  must(dc_hip_density_all(coords, n_rows, n_cols, radii.data(), radii.size(), 0, n_gpus,
                          flat.data(), nullptr, nullptr, nullptr, nullptr, nullptr),
       "population sweep");
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
  // row blocks per device and the row-ownership merge are density_clustering_cuda.cu:293-326;
  // here every device handles its block through the per-GPU entry point and the blocks are
  // stitched in frame order.
  const int n_gpus = get_num_gpus();
  const std::size_t gpu_range = n_rows / n_gpus;
  std::vector<std::uint32_t> nn_idx(n_rows), hd_idx(n_rows), pi(n_rows), ph(n_rows);
  std::vector<float> nn_d2(n_rows), hd_d2(n_rows), pd(n_rows), pdh(n_rows);
  for (int g = 0; g < n_gpus; ++g) {
    const std::size_t lo = g * gpu_range;
    const std::size_t hi = (g == n_gpus - 1) ? n_rows : (g + 1) * gpu_range;
    must(dc_hip_nearest_neighbors(coords, n_rows, n_cols, free_energy.data(), lo, hi, g, pi.data(),
                                  pd.data(), ph.data(), pdh.data()),
         "nearest-neighbour sweep");
    for (std::size_t i = lo; i < hi; ++i) {
      nn_idx[i] = pi[i];
      nn_d2[i] = pd[i];
      hd_idx[i] = ph[i];
      hd_d2[i] = pdh[i];
    }
  }
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
    if (!H::build_radius_forest(coords, n_rows, n_cols, max_dist, cache.fe_sorted, 0, &cache.forest, &err)) {
      std::cerr << "error during screening (radius forest)\n" << err << std::endl;
      exit(EXIT_FAILURE);
    }
    cache.have_forest = true;
  }
  if (!use_forest && !cache.have_graph) {
    if (!H::build_radius_graph(coords, n_rows, n_cols, max_dist, 0, &cache.graph, &err)) {
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

}  // namespace CUDA

namespace HIP {

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
