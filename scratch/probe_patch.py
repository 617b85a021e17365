"""Patches dc_mfma_kernels.hpp (in place) with the slow-wave probe counters read by scratch/slowwave.py:
time, chains, candidate-path entries, special tiles, triggers, flush slots and rings of the slowest wave of
a neighbour sweep (two 64-bit atomicMax words in the workspace header).  Build with scratch/build_variant.sh,
then restore the header."""
p = 'clustering_amd/csrc/dc_mfma_kernels.hpp'
s = open(p).read()
def rep(old, new):
    global s
    assert s.count(old) == 1, old[:60]
    s = s.replace(old, new)
rep("  // evaluate and empty the candidate queue of query tile qi (all lanes in parallel per slot)\n  auto flush = [&](int qi) {\n    if (__builtin_amdgcn_ballot_w64(qcount[qi] != 0) == 0) return;\n    NnPQ& Q = q[qi];",
'''  uint32_t n_rare = 0, n_special = 0, n_trig = 0, n_slots = 0, n_rings = 0;
  const unsigned long long t_start = wall_clock64();
  // evaluate and empty the candidate queue of query tile qi (all lanes in parallel per slot)
  auto flush = [&](int qi) {
    if (__builtin_amdgcn_ballot_w64(qcount[qi] != 0) == 0) return;
    { uint32_t mx = qcount[qi];
      for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
      n_slots += mx; }
    NnPQ& Q = q[qi];''')
rep('''        if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare) != 0, 0)) {
          const bool all_lower = fr.y < Q.feq;''', '''        if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare) != 0, 0)) {
          ++n_rare;
          const bool all_lower = fr.y < Q.feq;''')
rep('''          if (any_special) {
            // masked per-element minima''', '''          if (any_special) {
            ++n_special;
            // masked per-element minima''')
rep('''          if (__builtin_amdgcn_ballot_w64(trig) != 0) {
            // park this tile's candidates''', '''          if (__builtin_amdgcn_ballot_w64(trig) != 0) {
            ++n_trig;
            // park this tile's candidates''')
rep("    const bool any_open = __builtin_amdgcn_ballot_w64(need > 0.0f) != 0;", "    ++n_rings;\n    const bool any_open = __builtin_amdgcn_ballot_w64(need > 0.0f) != 0;")
rep("  if (lane == 0 && chain_counter) atomicAdd(chain_counter, (unsigned long long)chains);\n\n#pragma unroll\n  for (int qt = 0; qt < TQ; ++qt) {\n    NnPQ& Q = q[qt];",
'''  { const unsigned long long dt = (wall_clock64() - t_start) & 0xFFFFFull;   // 100 MHz ticks
    if (lane == 0 && chain_counter) {
      atomicMax(chain_counter, (dt << 44) | ((unsigned long long)(chains & 0xFFFF) << 28) | ((unsigned long long)(n_rare & 0x3FFF) << 14) | (unsigned long long)(n_special & 0x3FFF));
      atomicMax(chain_counter - 1, (dt << 44) | ((unsigned long long)(n_trig & 0x3FFF) << 30) | ((unsigned long long)(n_slots & 0xFFFF) << 14) | ((unsigned long long)(n_rings & 0xFF) << 6) | (unsigned long long)(chunk & 0x3F));
    } }

#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    NnPQ& Q = q[qt];''')
open(p, 'w').write(s)
