// screening_oracle.cpp -- CPU restatement of the reference's free-energy screening and microstate
// assignment.  TEST INFRASTRUCTURE ONLY (tests/ and smoke checks): the product never links this.
// PARITY UNPINNED like the rest of oracle/: the reference cannot be built in this image (DESIGN.md
// section 2); the functions below follow the reference's control flow line by line, with the
// quadratic scans it really does:
//   dso_screening           density_clustering_common.cpp:37-134 (screening), with
//                           density_clustering.cpp:382-435 (prepare_initial_clustering),
//                           :292-332 (high_density_neighborhood), :506-555 (lump_initial_clusters),
//                           :437-456 (normalized_cluster_names), :214-228 (sorted_free_energies),
//                           :334-343 (compute_sigma2)
//   dso_assign_low_density  density_clustering.cpp:345-360
//   dso_sorted_names        density_clustering.cpp:458-493
// The pair distance is the canonical d2 of dc_oracle.c (the loop of high_density_neighborhood has
// the shape of the population / neighbour loops).
#include <algorithm>
#include <cstddef>
#include <map>
#include <set>
#include <utility>
#include <vector>

extern "C" float dco_dist2(const float* x, const float* y, size_t n_cols);

namespace {
typedef std::pair<std::size_t, float> FreeEnergy;

std::vector<FreeEnergy> sorted_free_energies(const float* fe, std::size_t n) {
  std::vector<FreeEnergy> fe_sorted;
  for (std::size_t i = 0; i < n; ++i) fe_sorted.push_back(FreeEnergy(i, fe[i]));
  std::sort(fe_sorted.begin(), fe_sorted.end(),
            [](const FreeEnergy& d1, const FreeEnergy& d2) -> bool { return d1.second < d2.second; });
  return fe_sorted;
}
}  // namespace

extern "C" __attribute__((visibility("default"))) void dso_screening(
    const float* fe, const float* nn_d2, float threshold, const float* coords, std::size_t n_rows,
    std::size_t n_cols, const std::size_t* initial /* may be NULL */, std::size_t* out) {
  // prepare_initial_clustering
  const bool have_initial = initial != nullptr;
  std::vector<std::size_t> clustering(n_rows, 0);
  if (have_initial) clustering.assign(initial, initial + n_rows);
  std::vector<FreeEnergy> fe_sorted = sorted_free_energies(fe, n_rows);
  auto lb = std::upper_bound(fe_sorted.begin(), fe_sorted.end(), FreeEnergy(0, threshold),
                             [](const FreeEnergy& d1, const FreeEnergy& d2) -> bool { return d1.second < d2.second; });
  const std::size_t first_above = (std::size_t)(lb - fe_sorted.begin());
  double sigma2 = 0.0;   // compute_sigma2: double, frame order
  for (std::size_t i = 0; i < n_rows; ++i) sigma2 += nn_d2[i];
  sigma2 /= n_rows;
  std::size_t distinct_name = *std::max_element(clustering.begin(), clustering.end());
  std::set<std::size_t> visited;
  if (have_initial)
    for (std::size_t i = 0; i < first_above; ++i)
      if (initial[fe_sorted[i].first] != 0) visited.insert(i);
  const float max_dist = 4 * sigma2;   // double product, converted at the call like the reference

  bool merged = false;
  while (!merged) {
    merged = true;
    for (std::size_t i = 0; i < first_above; ++i) {
      if (visited.count(i) != 0) continue;
      visited.insert(i);
      // high_density_neighborhood: scan every frame below the threshold
      std::set<std::size_t> nh;
      const float* xi = coords + fe_sorted[i].first * n_cols;
      for (std::size_t j = 0; j < first_above; ++j) {
        if (i == j) continue;
        const float d2 = dco_dist2(xi, coords + fe_sorted[j].first * n_cols, n_cols);
        if (d2 < max_dist) nh.insert(j);
      }
      nh.insert(i);
      // lump_initial_clusters
      std::set<std::size_t> names;
      for (auto j : nh) names.insert(clustering[fe_sorted[j].first]);
      if (!(names.size() == 1 && names.count(0) != 1)) {
        merged = false;
        if (names.count(0) == 1) names.erase(0);
        std::size_t common;
        if (names.size() > 0)
          common = *names.begin();
        else
          common = ++distinct_name;
        for (auto j : nh) clustering[fe_sorted[j].first] = common;
        for (std::size_t j = 0; j < first_above; ++j) {
          const std::size_t ndx = fe_sorted[j].first;
          if (names.count(clustering[ndx]) == 1) clustering[ndx] = common;
        }
      }
    }
  }
  // normalized_cluster_names
  std::set<std::size_t> final_names;
  for (std::size_t i = 0; i < first_above; ++i) final_names.insert(clustering[fe_sorted[i].first]);
  std::map<std::size_t, std::size_t> old_to_new;
  old_to_new[0] = 0;
  std::size_t new_name = 0;
  for (auto name : final_names) old_to_new[name] = ++new_name;
  for (std::size_t i = 0; i < n_rows; ++i) out[i] = old_to_new[clustering[i]];
}

extern "C" __attribute__((visibility("default"))) void dso_assign_low_density(
    const std::size_t* initial, const std::size_t* hd_idx, const float* fe, std::size_t n_rows,
    std::size_t* out) {
  std::vector<FreeEnergy> fe_sorted = sorted_free_energies(fe, n_rows);
  std::vector<std::size_t> clustering(initial, initial + n_rows);
  for (const auto& f : fe_sorted) {
    const std::size_t id = f.first;
    if (clustering[id] == 0 && hd_idx[id] < n_rows) clustering[id] = clustering[hd_idx[id]];
  }
  for (std::size_t i = 0; i < n_rows; ++i) out[i] = clustering[i];
}

extern "C" __attribute__((visibility("default"))) void dso_sorted_names(const std::size_t* clustering,
                                                                         std::size_t n_rows,
                                                                         std::size_t* out) {
  std::map<std::size_t, std::size_t> counts;
  for (std::size_t i = 0; i < n_rows; ++i) ++counts[clustering[i]];
  std::vector<std::pair<std::size_t, std::size_t>> v(counts.begin(), counts.end());
  std::sort(v.begin(), v.end(), [](const std::pair<std::size_t, std::size_t>& a,
                                   const std::pair<std::size_t, std::size_t>& b) { return a.second < b.second; });
  std::map<std::size_t, std::size_t> names;
  for (std::size_t i = 0; i < v.size(); ++i) names[v[i].first] = v.size() - i;
  for (std::size_t i = 0; i < n_rows; ++i) out[i] = names[clustering[i]];
}
