// screening_host.hpp -- free-energy screening on top of the GPU radius graph (host side, plain C++).
//
// The reference's screening (density_clustering_common.cpp:37-134) visits the frames below a free-
// energy threshold in order of free energy and, for each one not yet assigned, scans ALL of them for
// partners closer than 4*sigma2 (high_density_neighborhood, density_clustering.cpp:292-332: O(M) per
// frame, O(M^2) per threshold) before merging cluster names (lump_initial_clusters, :506-555) and
// renumbering them (normalized_cluster_names, :437-456).  Here the partner lists come from ONE pruned
// GPU sweep (dc_hip_radius_pairs: every frame pair with canonical d2 < 4*sigma2) that is shared by all
// thresholds of a -T scan; the name bookkeeping is the reference's, with the "rename every frame that
// carries one of these names" loops replaced by a union-find over names.  Same results, same
// std::sort call for the free-energy order (so ties fall the way the reference's do).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

struct dc_hip_session;   // include/dc_density.h: a trajectory resident on the GPUs

namespace Clustering {
namespace Density {
namespace HIP {

//! (frame id, free energy), sorted lowest to highest exactly like sorted_free_energies
//! (density_clustering.cpp:214-228: std::sort on the free energy only)
using FreeEnergy = std::pair<std::size_t, float>;
std::vector<FreeEnergy> sorted_free_energies(const std::vector<float>& fe);

//! radius graph of a trajectory: every unordered frame pair with canonical d2 < max_dist, as
//! adjacency lists (CSR over frame ids).  Built once per (coords, max_dist) by one GPU sweep.
struct RadiusGraph {
  std::vector<std::uint64_t> offset;      // [n_rows + 1]
  std::vector<std::uint32_t> neighbor;    // [2 * n_pairs]
  std::size_t n_pairs = 0;
};
//! returns false and sets *error on failure (no exit here; the shim adds the reference's convention)
bool build_radius_graph(const float* coords, std::size_t n_rows, std::size_t n_cols, float max_dist,
                        int device, RadiusGraph* out, std::string* error);
//! the same on the coordinates an open session already holds on the device (no second upload)
bool build_radius_graph(dc_hip_session* session, std::size_t n_rows, float max_dist, RadiusGraph* out,
                        std::string* error);

//! the same for screenings that START FROM AN EMPTY CLUSTERING (a -T scan): a spanning forest of the
//! radius graph that, for every threshold, connects the frames below it exactly as the whole graph
//! does (dc_hip_radius_forest with rank = position in fe_sorted).  At most n_rows-1 pairs, a dozen
//! pruned sweeps and no pair list -- the whole graph of C3 has 9e8 pairs.  screening_with_graph gives
//! identical results on it as long as every initial clustering handed to it is the result of a lower
//! threshold on the same data (frames that come with a state are not expanded by the reference either,
//! and the names that matter -- the smallest of each component -- are allocated in the same order).
bool build_radius_forest(const float* coords, std::size_t n_rows, std::size_t n_cols, float max_dist,
                         const std::vector<FreeEnergy>& fe_sorted, int device, RadiusGraph* out,
                         std::string* error);
//! the same on an open session: all its devices take part (one segment each, candidates merged over RCCL)
bool build_radius_forest(dc_hip_session* session, std::size_t n_rows, float max_dist,
                         const std::vector<FreeEnergy>& fe_sorted, RadiusGraph* out, std::string* error);

//! screening for one threshold (density_clustering_common.cpp:37-134) given the radius graph for
//! max_dist = 4*sigma2: cluster id per frame, 0 = not assigned (above the threshold)
std::vector<std::size_t> screening_with_graph(const std::vector<float>& free_energy,
                                              const std::vector<FreeEnergy>& fe_sorted,
                                              const RadiusGraph& graph, float free_energy_threshold,
                                              const std::vector<std::size_t>& initial_clusters);

//! assign_low_density_frames (density_clustering.cpp:345-360): frames without a state take the state
//! of their nearest neighbour with lower free energy, in order of free energy
std::vector<std::size_t> assign_low_density_frames(const std::vector<std::size_t>& initial_clustering,
                                                   const std::vector<std::uint32_t>& hd_idx,
                                                   const std::vector<float>& free_energy);

//! sorted_cluster_names (density_clustering.cpp:458-493): states renamed by population
std::vector<std::size_t> sorted_cluster_names(const std::vector<std::size_t>& clustering);

}  // namespace HIP
}  // namespace Density
}  // namespace Clustering
