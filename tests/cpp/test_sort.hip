// GPU self-test of the library's radix sort (clustering_amd/csrc/dc_sort.hip): against std::stable_sort for sizes around
// the tile and wave boundaries, 1 .. 32 key bits (only the low key_bits bits order the items), heavy ties and all-equal
// keys -- the orderings of the sweeps rely on it being a STABLE sort (deterministic: every rank of a sharded run must
// derive the same order).  Built from the library's source file; run by tests/test_gpu_parity.py.
#include "../../clustering_amd/csrc/dc_sort.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CHECK(e)                                                                      \
  do {                                                                                \
    hipError_t _e = (e);                                                              \
    if (_e != hipSuccess) {                                                           \
      std::printf("HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); \
      return 2;                                                                       \
    }                                                                                 \
  } while (0)

int main() {
  std::mt19937 rng(12345);
  const size_t sizes[] = {1, 2, 63, 64, 65, 255, 256, 1023, 1024, 1025, 4095, 4096, 4097, 8192, 12289, 100000, 1032768, 3000001};
  const unsigned bits_list[] = {1, 7, 8, 9, 15, 16, 17, 24, 25, 32};
  size_t cases = 0;
  for (size_t n : sizes) {
    uint32_t *d_ki, *d_ko, *d_vi, *d_vo;
    void* d_tmp;
    const size_t tmp_bytes = dc::sort_temp_bytes(n);
    CHECK(hipMalloc((void**)&d_ki, 4 * n));
    CHECK(hipMalloc((void**)&d_ko, 4 * n));
    CHECK(hipMalloc((void**)&d_vi, 4 * n));
    CHECK(hipMalloc((void**)&d_vo, 4 * n));
    CHECK(hipMalloc(&d_tmp, tmp_bytes));
    for (unsigned bits : bits_list) {
      if (n > 200000 && bits != 16 && bits != 24 && bits != 32) continue;
      for (int kind = 0; kind < 3; ++kind) {
        std::vector<uint32_t> k(n), v(n);
        for (size_t i = 0; i < n; ++i) {
          const uint32_t r = rng();
          k[i] = kind == 0 ? r : (kind == 1 ? (r % 37u) * 0x01010101u + (r >> 28) : 0xDEADBEEFu);   // random / few values / all equal
          v[i] = (uint32_t)(n - i) * 2654435761u;
        }
        CHECK(hipMemcpy(d_ki, k.data(), 4 * n, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_vi, v.data(), 4 * n, hipMemcpyHostToDevice));
        CHECK(hipMemset(d_ko, 0xFF, 4 * n));
        if (dc::sort_pairs_u32(d_ki, d_ko, d_vi, d_vo, n, d_tmp, tmp_bytes, nullptr, bits) != 0) {
          std::printf("sort_pairs_u32 failed: n=%zu bits=%u\n", n, bits);
          return 1;
        }
        CHECK(hipDeviceSynchronize());
        std::vector<uint32_t> ko(n), vo(n);
        CHECK(hipMemcpy(ko.data(), d_ko, 4 * n, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(vo.data(), d_vo, 4 * n, hipMemcpyDeviceToHost));
        const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
        std::vector<uint32_t> idx(n);
        std::iota(idx.begin(), idx.end(), 0u);
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return (k[a] & mask) < (k[b] & mask); });
        for (size_t i = 0; i < n; ++i)
          if (ko[i] != k[idx[i]] || vo[i] != v[idx[i]]) {
            std::printf("MISMATCH n=%zu bits=%u kind=%d at %zu: got (%08x, %08x) want (%08x, %08x)\n", n, bits, kind, i, ko[i], vo[i],
                        k[idx[i]], v[idx[i]]);
            return 1;
          }
        ++cases;
      }
    }
    CHECK(hipFree(d_ki));
    CHECK(hipFree(d_ko));
    CHECK(hipFree(d_vi));
    CHECK(hipFree(d_vo));
    CHECK(hipFree(d_tmp));
  }
  // the last pass into a PADDED order (SortRemap): segments = the top bits of the keys, every segment moved to a multiple
  // of 192 positions; untouched positions keep their preset, a tile's tag is the segment of its first position; three
  // and four passes (the passes in between must not touch the output)
  for (size_t n : {(size_t)37, (size_t)5000, (size_t)70001, (size_t)1000003}) {
    for (unsigned bits : {16u, 24u, 32u}) {
      const uint32_t n_seg = 64, group = 192;
      std::vector<uint32_t> k(n), v(n), cnt(n_seg, 0), start(n_seg + 1, 0), base(n_seg + 1, 0);
      const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
      for (size_t i = 0; i < n; ++i) {
        const uint32_t seg = (rng() % 7u) * 9u;   // (most segments empty)
        k[i] = ((seg << (bits - 6)) | (rng() & ((1u << (bits - 6)) - 1u))) & mask;
        v[i] = (uint32_t)i;
        ++cnt[seg];
      }
      uint32_t run = 0, pos = 0;
      for (uint32_t s2 = 0; s2 < n_seg; ++s2) {
        start[s2] = run;
        base[s2] = pos;
        run += cnt[s2];
        pos += (cnt[s2] + group - 1) / group * group;
      }
      start[n_seg] = (uint32_t)n;
      base[n_seg] = pos;
      const size_t n_pos = pos + 32, n_tiles = (n_pos + 31) / 32;
      uint32_t *d_ki, *d_vi, *d_ko, *d_out, *d_tags, *d_start, *d_base;
      void* d_tmp;
      const size_t tmp_bytes = dc::sort_temp_bytes(n);
      CHECK(hipMalloc((void**)&d_ki, 4 * n));
      CHECK(hipMalloc((void**)&d_vi, 4 * n));
      CHECK(hipMalloc((void**)&d_ko, 4 * n));
      CHECK(hipMalloc((void**)&d_out, 4 * n_pos));
      CHECK(hipMalloc((void**)&d_tags, 4 * n_tiles));
      CHECK(hipMalloc((void**)&d_start, 4 * (n_seg + 1)));
      CHECK(hipMalloc((void**)&d_base, 4 * (n_seg + 1)));
      CHECK(hipMalloc(&d_tmp, tmp_bytes));
      CHECK(hipMemcpy(d_ki, k.data(), 4 * n, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(d_vi, v.data(), 4 * n, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(d_start, start.data(), 4 * (n_seg + 1), hipMemcpyHostToDevice));
      CHECK(hipMemcpy(d_base, base.data(), 4 * (n_seg + 1), hipMemcpyHostToDevice));
      CHECK(hipMemset(d_out, 0xFF, 4 * n_pos));
      CHECK(hipMemset(d_tags, 0xEE, 4 * n_tiles));
      const dc::SortRemap remap{d_start, d_base, n_seg, d_tags};
      if (dc::sort_pairs_u32(d_ki, d_ko, d_vi, d_out, n, d_tmp, tmp_bytes, nullptr, bits, &remap) != 0) return 1;
      CHECK(hipDeviceSynchronize());
      std::vector<uint32_t> out(n_pos), tags(n_tiles), want(n_pos, 0xFFFFFFFFu), want_tags(n_tiles, 0xEEEEEEEEu);
      CHECK(hipMemcpy(out.data(), d_out, 4 * n_pos, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(tags.data(), d_tags, 4 * n_tiles, hipMemcpyDeviceToHost));
      std::vector<uint32_t> idx(n);
      std::iota(idx.begin(), idx.end(), 0u);
      std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
      for (size_t i = 0; i < n; ++i) {
        const uint32_t seg = k[idx[i]] >> (bits - 6);
        const size_t p2 = base[seg] + (i - start[seg]);
        want[p2] = v[idx[i]];
        if (p2 % 32 == 0) want_tags[p2 / 32] = seg;
      }
      if (out != want || tags != want_tags) {
        std::printf("REMAP MISMATCH n=%zu bits=%u\n", n, bits);
        return 1;
      }
      ++cases;
      CHECK(hipFree(d_ki)); CHECK(hipFree(d_vi)); CHECK(hipFree(d_ko)); CHECK(hipFree(d_out)); CHECK(hipFree(d_tags));
      CHECK(hipFree(d_start)); CHECK(hipFree(d_base)); CHECK(hipFree(d_tmp));
    }
  }
  std::printf("%zu cases OK\n", cases);
  return 0;
}
