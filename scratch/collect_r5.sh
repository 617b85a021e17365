#!/bin/bash
# profiles/r5_* from the gpurun_out/ of scratch/profile_r5.sh (run in this container, from the repo root)
set -e
O=gpurun_out
for w in c3 c2 c5; do
  cp $O/r5_${w}_bench.json profiles/r5_${w}_bench.json
  f=$(find $O/r5_${w}_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" profiles/r5_${w}_kernel_stats.csv
done
python3 scratch/make_pmc_profile.py r5_c3 $O/r5_c3_bench.json '{"n_rows": 1000000, "n_cols": 10, "radii": [0.2], "what": "bench.py default (C3): pop + FE + nn"}' $O/r5_c3_sq1 $O/r5_c3_sq2 -- $O/r5_c3_tcc
python3 scratch/make_pmc_profile.py r5_c2 $O/r5_c2_bench.json '{"n_rows": 100000, "n_cols": 10, "radii": [0.1, 0.2, 0.3], "what": "C2: pop + FE, three one-radius sweeps per call", "dispatches_per_call": {"pop": 3}}' $O/r5_c2_sq1 $O/r5_c2_sq2 -- $O/r5_c2_tcc
python3 scratch/make_pmc_profile.py r5_c5 $O/r5_c5_bench.json '{"n_rows": 5000000, "n_cols": 30, "radii": [0.3, 0.35, 0.4, 0.45, 0.5, 0.55, 0.6, 0.65], "what": "C5: segment 3 of 8 (one rank): eight radii in ONE symmetric sweep (pop_msym_kernel), a full one-radius sweep for the free energies, nn segment"}' $O/r5_c5_sq1 $O/r5_c5_sq2 -- $O/r5_c5_tcc
cp $O/r5_c5_onesided_pop.json profiles/r5_c5_onesided_pop.json
cp $O/r5_spread10_bench.json profiles/r5_spread10_bench.json
cp $O/r5_unfav_oneblob.json profiles/r5_unfav_oneblob.json
cp $O/r5_unfav_uniform.json profiles/r5_unfav_uniform.json
grep SEG $O/r5_seg.txt > profiles/r5_segments.txt
f=$(find $O/r5_seg8_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" profiles/r5_seg8_kernel_stats.csv
cp $O/r5_c3_mfma32_bench.json profiles/r5_c3_mfma32_bench.json
f=$(find $O/r5_c3_mfma32_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" profiles/r5_c3_mfma32_kernel_stats.csv
python3 - <<'PY'
# profiles/r5_c3_mfma32_pmc.json: the counters of the fp32-input MFMA instance + what bench.py's fp32_mfma_instance quotes
import json, subprocess
raw = json.loads(subprocess.check_output(['python3', 'scratch/pmc_summary.py', 'gpurun_out/r5_c3_mfma32_sq1']))
busy = {}
for k, e in raw.items():
    if 'GRBM_GUI_ACTIVE' in e and 'SQ_VALU_MFMA_BUSY_CYCLES' in e:
        busy['population_count' if 'pop_' in k else 'nearest_neighbor_search'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * e['GRBM_GUI_ACTIVE'] / 8)
out = {"note": "rocprofv3 --kernel-trace --pmc of `bench.py --variant mfma32` (C3, every pair on v_mfma_f32_32x32x2_f32): "
               "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)",
       "n_rows": 1000000, "n_cols": 10, "csrc_digest": open('gpurun_out/r5_csrc_digest.txt').read().strip(),
       "mfma_busy": busy, "kernels": raw}
json.dump(out, open('profiles/r5_c3_mfma32_pmc.json', 'w'), indent=1)
print(busy)
PY
ls -la profiles | grep r5_
