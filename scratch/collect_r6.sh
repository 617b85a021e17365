#!/bin/bash
# profiles/r6_* from the gpurun_out/ of scratch/profile_r6.sh (run in this container, from the repo root)
set -e
O=gpurun_out
for w in c3 c2 c5; do
  cp $O/r6_${w}_bench.json profiles/r6_${w}_bench.json
  f=$(find $O/r6_${w}_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" profiles/r6_${w}_kernel_stats.csv
done
python3 scratch/make_pmc_profile.py r6_c3 $O/r6_c3_bench.json '{"n_rows": 1000000, "n_cols": 10, "radii": [0.2], "what": "bench.py default (C3): pop + FE + nn"}' $O/r6_c3_sq1 $O/r6_c3_sq2 -- $O/r6_c3_tcc
python3 scratch/make_pmc_profile.py r6_c2 $O/r6_c2_bench.json '{"n_rows": 100000, "n_cols": 10, "radii": [0.1, 0.2, 0.3], "what": "C2: pop + FE, three one-radius sweeps per call", "dispatches_per_call": {"pop": 3}}' $O/r6_c2_sq1 $O/r6_c2_sq2 -- $O/r6_c2_tcc
python3 scratch/make_pmc_profile.py r6_c5 $O/r6_c5_bench.json '{"n_rows": 5000000, "n_cols": 30, "radii": [0.3, 0.35, 0.4, 0.45, 0.5, 0.55, 0.6, 0.65], "what": "C5: segment 3 of 8 (one rank): eight radii in ONE symmetric sweep (pop_msym_kernel), a full one-radius sweep for the free energies, nn segment"}' $O/r6_c5_sq1 $O/r6_c5_sq2 -- $O/r6_c5_tcc
cp $O/r6_c5_onesided_pop.json profiles/r6_c5_onesided_pop.json
cp $O/r6_spread10_bench.json profiles/r6_spread10_bench.json
cp $O/r6_unfav_oneblob.json profiles/r6_unfav_oneblob.json
cp $O/r6_unfav_uniform.json profiles/r6_unfav_uniform.json
grep SEG $O/r6_seg.txt > profiles/r6_segments.txt
f=$(find $O/r6_seg8_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" profiles/r6_seg8_kernel_stats.csv
cp $O/r6_c3_mfma32_bench.json profiles/r6_c3_mfma32_bench.json
f=$(find $O/r6_c3_mfma32_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" profiles/r6_c3_mfma32_kernel_stats.csv
python3 - <<'PY'
# profiles/r6_c3_mfma32_pmc.json: the counters of the fp32-input MFMA instance + what bench.py's fp32_mfma_instance quotes
import json, subprocess
raw = json.loads(subprocess.check_output(['python3', 'scratch/pmc_summary.py', 'gpurun_out/r6_c3_mfma32_sq1']))
busy, valu = {}, {}
t_all = ((1000000 + 31) // 32) ** 2   # every tile pair is evaluated
for k, e in raw.items():
    if 'GRBM_GUI_ACTIVE' in e and 'SQ_VALU_MFMA_BUSY_CYCLES' in e:
        name = 'population_count' if 'pop_' in k else 'nearest_neighbor_search'
        busy[name] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * e['GRBM_GUI_ACTIVE'] / 8)
        valu[name] = e['SQ_INSTS_VALU'] / t_all
        e['tile_pairs_per_dispatch'] = t_all
        e['valu_insts_per_32x32_tile_pair'] = valu[name]
out = {"note": "rocprofv3 --kernel-trace --pmc of `bench.py --variant mfma32` (C3, every pair on v_mfma_f32_32x32x2_f32): "
               "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); valu_insts_per_32x32_tile_pair = "
               "SQ_INSTS_VALU (the five MFMAs of a chain included) / ceil(N/32)^2",
       "n_rows": 1000000, "n_cols": 10, "csrc_digest": open('gpurun_out/r6_csrc_digest.txt').read().strip(),
       "mfma_busy": busy, "valu_insts_per_32x32_tile_pair": valu, "kernels": raw}
json.dump(out, open('profiles/r6_c3_mfma32_pmc.json', 'w'), indent=1)
print(busy, valu)
PY
ls -la profiles | grep r6_
