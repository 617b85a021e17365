// test driver for the C++ shim (density_clustering_hip.hpp): calls the reference-shaped entry
// points Clustering::Density::CUDA::{get_num_gpus, calculate_populations, nearest_neighbors, screening} with
// the reference's container types and dumps the results as text for tests/test_gpu_cli.py.
//   test_shim coords.f32 n_rows n_cols fe.f32 r1 [r2 ...]
//   test_shim coords.f32 n_rows n_cols fe.f32 hdn max_dist     CUDA::high_density_neighborhood across in-place edits
#include "../../clustering_amd/csrc/density_clustering_hip.hpp"

#include "../../clustering_amd/csrc/screening_host.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const std::size_t n_rows = std::strtoull(argv[2], nullptr, 10), n_cols = std::strtoull(argv[3], nullptr, 10);
  std::vector<float> coords(n_rows * n_cols), fe(n_rows);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(coords.data(), sizeof(float), coords.size(), f) != coords.size()) return 3;
  std::fclose(f);
  f = std::fopen(argv[4], "rb");
  if (!f || std::fread(fe.data(), sizeof(float), fe.size(), f) != fe.size()) return 3;
  std::fclose(f);
  namespace G = Clustering::Density::CUDA;
  if (std::strcmp(argv[5], "hdn") == 0 && argc >= 7) {
    // passes of ascending i_frame like the reference's screening (density_clustering_common.cpp:37-134); between the
    // passes ONE row of the coordinates, then two entries of the order are changed IN PLACE (same addresses, same sizes)
    const float max_dist = std::strtof(argv[6], nullptr);
    std::vector<Clustering::Density::FreeEnergy> order = Clustering::Density::HIP::sorted_free_energies(fe);
    auto pass = [&](const char* tag) {
      std::printf("order_%s", tag);   // (std::sort leaves frames of equal free energy in an order of its own)
      for (const auto& e : order) std::printf(" %zu", e.first);
      std::printf("\n");
      for (std::size_t i = 0; i < n_rows; ++i) {
        const std::set<std::size_t> nh = G::high_density_neighborhood(coords.data(), n_cols, order, i, n_rows, max_dist);
        std::printf("%s %zu", tag, i);
        for (std::size_t j : nh) std::printf(" %zu", j);
        std::printf("\n");
      }
    };
    pass("hdn1");
    const std::size_t moved = order[3].first;
    std::printf("moved %zu\n", moved);
    for (std::size_t k = 0; k < n_cols; ++k) coords[moved * n_cols + k] += 100.0f;
    pass("hdn2");
    std::swap(order[5], order[9]);
    pass("hdn3");
    return 0;
  }
  std::vector<float> radii;
  for (int i = 5; i < argc; ++i) radii.push_back(std::strtof(argv[i], nullptr));

  std::printf("gpus %d\n", G::get_num_gpus());
  Clustering::Density::Pops pops = G::calculate_populations(coords.data(), n_rows, n_cols, radii);
  for (const auto& kv : pops) {            // std::map: ascending radius
    std::printf("pops %.9g", kv.first);
    for (std::size_t p : kv.second) std::printf(" %zu", p);
    std::printf("\n");
  }
  Clustering::Density::Pops part = G::calculate_populations_partial(coords.data(), n_rows, n_cols, radii,
                                                                    n_rows / 3, n_rows / 2, 0);
  for (const auto& kv : part) {
    std::printf("part %.9g", kv.first);
    for (std::size_t p : kv.second) std::printf(" %zu", p);
    std::printf("\n");
  }
  auto nh = G::nearest_neighbors(coords.data(), n_rows, n_cols, fe);
  const auto& nn = std::get<0>(nh);
  const auto& hd = std::get<1>(nh);
  for (std::size_t i = 0; i < n_rows; ++i)
    std::printf("nn %zu %zu %.9g %zu %.9g\n", i, nn.at(i).first, nn.at(i).second, hd.at(i).first, hd.at(i).second);
  // screening, chained over three thresholds like Density::main does (density_clustering.cpp:801-812)
  std::vector<std::size_t> clustering;
  for (float t : {0.5f, 1.5f, 3.0f}) {
    clustering = G::screening(fe, nn, t, coords.data(), n_rows, n_cols, clustering);
    std::printf("screen %.9g", t);
    for (std::size_t v : clustering) std::printf(" %zu", v);
    std::printf("\n");
  }
  // an initial clustering that is NOT the result of a lower threshold (neighbouring frames may carry
  // different names): the shim must walk the whole radius graph, not the spanning forest
  std::vector<std::size_t> foreign(n_rows, 0);
  for (std::size_t i = 0; i < n_rows; ++i)
    if (fe[i] < 1.0f) foreign[i] = 1 + i % 3;
  clustering = G::screening(fe, nn, 2.0f, coords.data(), n_rows, n_cols, foreign);
  std::printf("foreign 2");
  for (std::size_t v : clustering) std::printf(" %zu", v);
  std::printf("\n");
  return 0;
}
