// screening_host.cpp -- see screening_host.hpp
#include "screening_host.hpp"

#include <algorithm>
#include <map>
#include <set>
#include <string>

#include "dc_density.h"

namespace Clustering {
namespace Density {
namespace HIP {

std::vector<FreeEnergy> sorted_free_energies(const std::vector<float>& fe) {
  std::vector<FreeEnergy> fe_sorted;
  for (std::size_t i = 0; i < fe.size(); ++i) fe_sorted.push_back(FreeEnergy(i, fe[i]));
  // the SAME call as the reference (density_clustering.cpp:222-226): std::sort is not stable, frames
  // of equal free energy end up where this library implementation puts them -- in both programs
  std::sort(fe_sorted.begin(), fe_sorted.end(),
            [](const FreeEnergy& d1, const FreeEnergy& d2) -> bool { return d1.second < d2.second; });
  return fe_sorted;
}

namespace {
void pairs_to_graph(const std::vector<std::uint32_t>& pairs, std::size_t n_pairs, std::size_t n_rows,
                    RadiusGraph* out);
}

namespace {
// a session for the duration of one call, for the callers that hold none
struct ScopedSession {
  dc_hip_session* s = nullptr;
  bool open(const float* coords, std::size_t n_rows, std::size_t n_cols, int device, std::string* error) {
    if (dc_hip_session_open(coords, n_rows, n_cols, &device, 1, &s) == DC_OK) return true;
    if (error) *error = dc_hip_last_error();
    return false;
  }
  ~ScopedSession() { dc_hip_session_close(s); }
};
}  // namespace

bool build_radius_graph(dc_hip_session* session, std::size_t n_rows, float max_dist, RadiusGraph* out,
                        std::string* error) {
  unsigned long long count = 0;
  // counting sweep, then the listing sweep with a buffer of exactly that size
  int rc = dc_hip_session_radius_pairs(session, max_dist, nullptr, 0, &count);
  std::vector<std::uint32_t> pairs;
  if (rc == DC_OK && count > 0) {
    pairs.resize(2 * (std::size_t)count);
    unsigned long long again = 0;
    rc = dc_hip_session_radius_pairs(session, max_dist, pairs.data(), (std::size_t)count, &again);
    if (rc == DC_OK && again != count) {
      if (error) *error = "radius pair sweeps disagree";
      return false;
    }
  }
  if (rc != DC_OK) {
    if (error) *error = dc_hip_last_error();
    return false;
  }
  pairs_to_graph(pairs, (std::size_t)count, n_rows, out);
  return true;
}

bool build_radius_graph(const float* coords, std::size_t n_rows, std::size_t n_cols, float max_dist,
                        int device, RadiusGraph* out, std::string* error) {
  ScopedSession tmp;
  return tmp.open(coords, n_rows, n_cols, device, error) && build_radius_graph(tmp.s, n_rows, max_dist, out, error);
}

namespace {
void pairs_to_graph(const std::vector<std::uint32_t>& pairs, std::size_t n_pairs, std::size_t n_rows,
                    RadiusGraph* out) {
  out->n_pairs = n_pairs;
  out->offset.assign(n_rows + 1, 0);
  for (std::size_t k = 0; k < 2 * n_pairs; ++k) ++out->offset[pairs[k] + 1];
  for (std::size_t i = 0; i < n_rows; ++i) out->offset[i + 1] += out->offset[i];
  out->neighbor.resize(2 * n_pairs);
  std::vector<std::uint64_t> fill(out->offset.begin(), out->offset.end() - 1);
  for (std::size_t k = 0; k < n_pairs; ++k) {
    const std::uint32_t a = pairs[2 * k], b = pairs[2 * k + 1];
    out->neighbor[fill[a]++] = b;
    out->neighbor[fill[b]++] = a;
  }
}
}  // namespace

bool build_radius_forest(dc_hip_session* session, std::size_t n_rows, float max_dist,
                         const std::vector<FreeEnergy>& fe_sorted, RadiusGraph* out, std::string* error) {
  std::vector<std::uint32_t> rank(n_rows);
  for (std::size_t i = 0; i < n_rows; ++i) rank[fe_sorted[i].first] = (std::uint32_t)i;
  std::vector<std::uint32_t> pairs(2 * (n_rows ? n_rows - 1 : 0) + 2);
  std::size_t n_pairs = 0;
  const int rc = dc_hip_session_radius_forest(session, max_dist, rank.data(), pairs.data(), &n_pairs, nullptr);
  if (rc != DC_OK) {
    if (error) *error = dc_hip_last_error();
    return false;
  }
  pairs_to_graph(pairs, n_pairs, n_rows, out);
  return true;
}

bool build_radius_forest(const float* coords, std::size_t n_rows, std::size_t n_cols, float max_dist,
                         const std::vector<FreeEnergy>& fe_sorted, int device, RadiusGraph* out,
                         std::string* error) {
  ScopedSession tmp;
  return tmp.open(coords, n_rows, n_cols, device, error) &&
         build_radius_forest(tmp.s, n_rows, max_dist, fe_sorted, out, error);
}

namespace {
// cluster names with "merge under the smallest name" (lump_initial_clusters picks *names.begin())
struct Names {
  std::map<std::size_t, std::size_t> parent;   // only names that were merged away have an entry
  std::size_t find(std::size_t name) {
    std::size_t root = name;
    for (auto it = parent.find(root); it != parent.end(); it = parent.find(root)) root = it->second;
    while (name != root) {   // path compression
      auto it = parent.find(name);
      const std::size_t next = it->second;
      it->second = root;
      name = next;
    }
    return root;
  }
};
}  // namespace

std::vector<std::size_t> screening_with_graph(const std::vector<float>& free_energy,
                                              const std::vector<FreeEnergy>& fe_sorted,
                                              const RadiusGraph& graph, float free_energy_threshold,
                                              const std::vector<std::size_t>& initial_clusters) {
  const std::size_t n_rows = free_energy.size();
  // prepare_initial_clustering (density_clustering.cpp:382-435)
  const bool have_initial_clusters = (initial_clusters.size() == n_rows);
  std::vector<std::size_t> clustering = have_initial_clusters ? initial_clusters : std::vector<std::size_t>(n_rows);
  auto lb = std::upper_bound(fe_sorted.begin(), fe_sorted.end(), FreeEnergy(0, free_energy_threshold),
                             [](const FreeEnergy& d1, const FreeEnergy& d2) -> bool { return d1.second < d2.second; });
  const std::size_t first_frame_above_threshold = (std::size_t)(lb - fe_sorted.begin());
  std::size_t distinct_name = n_rows ? *std::max_element(clustering.begin(), clustering.end()) : 0;
  std::vector<std::size_t> pos_of(n_rows);       // frame id -> index in order of free energy
  for (std::size_t i = 0; i < n_rows; ++i) pos_of[fe_sorted[i].first] = i;

  // the frames below the threshold, in order of free energy; frames that come with a state are
  // "visited" (their neighbourhoods are not expanded, :417-427)
  Names names;
  std::set<std::size_t> cluster_names;
  std::vector<std::size_t> local_nh;   // frame ids
  for (std::size_t i = 0; i < first_frame_above_threshold; ++i) {
    const std::size_t frame = fe_sorted[i].first;
    if (have_initial_clusters && initial_clusters[frame] != 0) continue;
    // local neighbourhood: partners below the threshold, and the frame itself (:292-332)
    local_nh.clear();
    for (std::uint64_t k = graph.offset[frame]; k < graph.offset[frame + 1]; ++k) {
      const std::size_t j = graph.neighbor[k];
      if (pos_of[j] < first_frame_above_threshold) local_nh.push_back(j);
    }
    local_nh.push_back(frame);
    // lump_initial_clusters (:506-555)
    cluster_names.clear();
    std::size_t last_raw = ~(std::size_t)0;   // partners mostly share one name: skip repeated look-ups
    for (std::size_t j : local_nh) {
      const std::size_t raw = clustering[j];
      if (raw == last_raw) continue;
      last_raw = raw;
      cluster_names.insert(raw == 0 ? 0 : names.find(raw));
    }
    if (!(cluster_names.size() == 1 && cluster_names.count(0) != 1)) {
      cluster_names.erase(0);
      std::size_t common_name;
      if (!cluster_names.empty())
        common_name = *cluster_names.begin();   // smallest name wins
      else
        common_name = ++distinct_name;
      for (std::size_t j : local_nh) clustering[j] = common_name;
      // "every frame below the threshold that carries one of these names gets the common name":
      // one union per name instead of a pass over the frames
      for (std::size_t name : cluster_names)
        if (name != common_name) names.parent[name] = common_name;
    }
  }
  // frames below the threshold read their name through the merges; frames above it keep what
  // they had (the reference's renaming loop only runs over the frames below the threshold)
  for (std::size_t i = 0; i < first_frame_above_threshold; ++i) {
    std::size_t& c = clustering[fe_sorted[i].first];
    if (c != 0) c = names.find(c);
  }
  // normalized_cluster_names (:437-456)
  std::set<std::size_t> final_names;
  for (std::size_t i = 0; i < first_frame_above_threshold; ++i) final_names.insert(clustering[fe_sorted[i].first]);
  std::map<std::size_t, std::size_t> old_to_new;
  old_to_new[0] = 0;
  std::size_t new_name = 0;
  for (auto name : final_names) old_to_new[name] = ++new_name;
  for (auto& elem : clustering) elem = old_to_new[elem];   // unknown names become 0, like operator[]
  return clustering;
}

std::vector<std::size_t> assign_low_density_frames(const std::vector<std::size_t>& initial_clustering,
                                                   const std::vector<std::uint32_t>& hd_idx,
                                                   const std::vector<float>& free_energy) {
  std::vector<FreeEnergy> fe_sorted = sorted_free_energies(free_energy);
  std::vector<std::size_t> clustering(initial_clustering);
  for (const auto& fe : fe_sorted) {
    const std::size_t id = fe.first;
    if (clustering[id] == 0) {
      const std::size_t neighbor_id = hd_idx[id];
      // (the frame of lowest free energy has no such neighbour: the reference reads out of bounds
      //  there unless it already has a state; it keeps 0 here)
      if (neighbor_id < clustering.size()) clustering[id] = clustering[neighbor_id];
    }
  }
  return clustering;
}

std::vector<std::size_t> sorted_cluster_names(const std::vector<std::size_t>& clustering) {
  std::map<std::size_t, std::size_t> counts;
  for (std::size_t c : clustering) ++counts[c];
  std::vector<std::pair<std::size_t, std::size_t>> counts_vec(counts.begin(), counts.end());
  // the SAME call as the reference (:478, comparator compare2DVector: by population only)
  std::sort(counts_vec.begin(), counts_vec.end(),
            [](const std::pair<std::size_t, std::size_t>& p1, const std::pair<std::size_t, std::size_t>& p2) {
              return p1.second < p2.second;
            });
  std::map<std::size_t, std::size_t> map_names;
  for (std::size_t i = 0; i < counts_vec.size(); ++i) map_names[counts_vec[i].first] = counts_vec.size() - i;
  std::vector<std::size_t> out(clustering.size());
  for (std::size_t i = 0; i < clustering.size(); ++i) out[i] = map_names[clustering[i]];
  return out;
}

}  // namespace HIP
}  // namespace Density
}  // namespace Clustering
