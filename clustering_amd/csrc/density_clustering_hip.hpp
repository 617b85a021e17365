// density_clustering_hip.hpp -- C++ host mirror of the reference's GPU plug-in surface
// (density_clustering_cuda.hpp:13-54) on top of the C ABI in include/dc_density.h.
//
// Drop-in: a maintainer replaces `#include "density_clustering_cuda.hpp"`
// (density_clustering.cpp:31-32) by this header and links libdcdensity.so + density_clustering_hip.cpp
// instead of density_clustering_cuda.cu / ..._cuda_kernels.cu.  Namespace,
// names, argument order/meaning, return types and the error convention ("message on stderr, then
// exit(EXIT_FAILURE)", density_clustering_cuda.cu:21-30) are the reference's.
#pragma once

#include <cstddef>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

namespace Clustering {
namespace Tools {
//! matches neighbor's frame id to distance            (tools.hpp:64)
using Neighbor = std::pair<std::size_t, float>;
//! map frame id to neighbors                          (tools.hpp:66)
using Neighborhood = std::map<std::size_t, Clustering::Tools::Neighbor>;
}  // namespace Tools
namespace Density {
//! radius -> populations                              (density_clustering_common.hpp:39)
typedef std::map<float, std::vector<std::size_t>> Pops;
//! (frame id, free energy)                            (density_clustering.hpp:54)   
using FreeEnergy = std::pair<std::size_t, float>;

namespace CUDA {   // the reference's namespace name is kept so that call sites compile unchanged

using Neighborhood = Clustering::Tools::Neighborhood;

//! prints msg + the library's last error and exits if the last C-ABI call failed
//! (density_clustering_cuda.hpp:13-14, density_clustering_cuda.cu:21-30)
void check_error(std::string msg = "");

//! number of usable GPUs; exits if there is none (density_clustering_cuda.cu:32-43)
int get_num_gpus();

//! populations of rows [i_from, i_to) on device i_gpu, all other rows 0
//! (calculate_populations_per_gpu, density_clustering_cuda.cu:45-137; the header declares the
//! same job as calculate_populations_partial, density_clustering_cuda.hpp:21-30)
Pops calculate_populations_partial(const float* coords, std::size_t n_rows, std::size_t n_cols,
                                   std::vector<float> radii, std::size_t i_from, std::size_t i_to,
                                   int i_gpu);
Pops calculate_populations_per_gpu(const float* coords, std::size_t n_rows, std::size_t n_cols,
                                   std::vector<float> radii, std::size_t i_from, std::size_t i_to,
                                   int i_gpu);
//! the signature the reference's header declares (density_clustering_cuda.hpp:21-30; never defined
//! there).  sorted_coords / blimits belong to a column-0-sorted pruning scheme (tools.hxx:120-204) that
//! the reference's kernels never used; the sweeps here build their own spatial ordering on the device,
//! so both are ignored (result-neutral, like the CPU path's box grid).
Pops calculate_populations_partial(const float* coords, const std::vector<float>& sorted_coords,
                                   const std::vector<float>& blimits, std::size_t n_rows,
                                   std::size_t n_cols, std::vector<float> radii, std::size_t i_from,
                                   std::size_t i_to, int i_gpu);

//! populations for several radii in one sweep, rows sharded over all GPUs
//! (density_clustering_cuda.hpp:32-36, density_clustering_cuda.cu:139-182)
Pops calculate_populations(const float* coords, const std::size_t n_rows, const std::size_t n_cols,
                           std::vector<float> radii);

//! nearest neighbour / nearest neighbour with lower free energy of rows [i_from, i_to) on device
//! i_gpu (density_clustering_cuda.cu:184-284).  Rows outside the range hold (n_rows+1, FLT_MAX).
std::tuple<Neighborhood, Neighborhood> nearest_neighbors_per_gpu(
    const float* coords, const std::size_t n_rows, const std::size_t n_cols,
    const std::vector<float>& free_energy, std::size_t i_from, std::size_t i_to, int i_gpu);

//! (density_clustering_cuda.hpp:38-42, density_clustering_cuda.cu:286-328)
std::tuple<Neighborhood, Neighborhood> nearest_neighbors(const float* coords,
                                                         const std::size_t n_rows,
                                                         const std::size_t n_cols,
                                                         const std::vector<float>& free_energy);

//! free-energy screening for one threshold (density_clustering_cuda.hpp:47-54; CPU semantics
//! density_clustering_common.cpp:37-134, which the results follow).  The partner lists of ALL frames
//! for max_dist = 4*sigma2 come from one GPU sweep (dc_hip_radius_pairs) that is cached across the
//! calls of a -T scan (same coords / n_rows / n_cols / sigma2); see screening_host.hpp.
std::vector<std::size_t> screening(const std::vector<float>& free_energy, const Neighborhood& nh,
                                   const float free_energy_threshold, const float* coords,
                                   const std::size_t n_rows, const std::size_t n_cols,
                                   const std::vector<std::size_t> initial_clusters);

//! declared in density_clustering_cuda.hpp:44-45, defined nowhere in the reference: states renumbered
//! 1..K in ascending order of their old names, 0 kept (normalized_cluster_names' numbering rule)
std::vector<std::size_t> sanitize_state_names(std::vector<std::size_t> clustering);

//! declared in density_clustering_cuda.hpp:56-62, defined nowhere in the reference's CUDA sources; CPU
//! semantics density_clustering.cpp:292-332.  All ids are positions in sorted_fe.  Served from the GPU
//! radius graph of the resident trajectory (n_rows = sorted_fe.size()), cached per (coords, sorted_fe, max_dist).
//! The reference calls it once per frame of a screening pass with i_frame ASCENDING: inside such a pass a call costs
//! O(1) (the arrays are recognised by address, size and 64 sampled words); whenever i_frame does not continue upwards
//! -- a new pass -- both arrays are fingerprinted in full and a change of any element rebuilds the graph.
//! PRECONDITION: coords and sorted_fe are not modified between the calls of one ascending pass (the reference's
//! screening never does); HIP::invalidate_neighborhood_cache() forces the rebuild where a caller must.
//! Any other access pattern (a repeated or descending i_frame, two interleaved passes) is correct but pays the full
//! fingerprint, O(N D), on every such call.  The cache is process-wide; calls are serialised by a mutex.
std::set<std::size_t> high_density_neighborhood(const float* coords, const std::size_t n_cols,
                                                const std::vector<FreeEnergy>& sorted_fe,
                                                const std::size_t i_frame, const std::size_t limit,
                                                const float max_dist);

}  // namespace CUDA

// ---- flat-array helpers for hosts that do not want the node-based containers -------------------
namespace HIP {
struct DensityResult {
  std::vector<std::vector<std::size_t>> pops;   // [radius index][frame], in the caller's radius order
  std::vector<float> free_energy;               // from radius fe_radius_index
  std::vector<std::size_t> nn_idx, hd_idx;      // empty if neighbours were not requested
  std::vector<float> nn_d2, hd_d2;
  double sigma2 = 0.0;                          // mean nn d2 (compute_sigma2, density_clustering.cpp:334-343)
};
//! The CUDA:: entry points above keep the trajectory they were last called with resident on the GPUs
//! (one upload for populations, neighbours and screening; keyed on pointer, shape and a fingerprint of
//! the whole buffer, so re-using a host buffer for other coordinates is detected).  This drops it and frees
//! the device memory and the RCCL communicators of all GPUs; call it when the density phase is over.
void release_resident();
//! the next CUDA::high_density_neighborhood call rebuilds its radius graph and position table from scratch
void invalidate_neighborhood_cache();
//! whole path (pop -> FE -> NN) with coordinates kept resident on the devices between the phases
DensityResult density_all(const float* coords, std::size_t n_rows, std::size_t n_cols,
                          const std::vector<float>& radii, std::size_t fe_radius_index,
                          bool want_neighbors, int n_gpus = 0);
}  // namespace HIP

}  // namespace Density
}  // namespace Clustering
