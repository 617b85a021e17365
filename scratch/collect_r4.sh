#!/bin/bash
# profiles/r4_* from the gpurun_out/ of scratch/profile_r4.sh (run in this container, from the repo root)
set -e
O=gpurun_out
for w in c3 c2 c5; do
  cp $O/r4_${w}_bench.json profiles/r4_${w}_bench.json
  f=$(find $O/r4_${w}_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" profiles/r4_${w}_kernel_stats.csv
done
python3 scratch/make_pmc_profile.py r4_c3 $O/r4_c3_bench.json '{"n_rows": 1000000, "n_cols": 10, "radii": [0.2], "what": "bench.py default (C3): pop + FE + nn"}' $O/r4_c3_sq1 $O/r4_c3_sq2 -- $O/r4_c3_tcc
python3 scratch/make_pmc_profile.py r4_c2 $O/r4_c2_bench.json '{"n_rows": 100000, "n_cols": 10, "radii": [0.1, 0.2, 0.3], "what": "C2: pop + FE, three one-radius sweeps per call", "dispatches_per_call": {"pop": 3}}' $O/r4_c2_sq1 $O/r4_c2_sq2 -- $O/r4_c2_tcc
python3 scratch/make_pmc_profile.py r4_c5 $O/r4_c5_bench.json '{"n_rows": 5000000, "n_cols": 30, "radii": [0.3, 0.35, 0.4, 0.45, 0.5, 0.55, 0.6, 0.65], "what": "C5: segment 3 of 8 (one rank): eight radii in ONE symmetric sweep (pop_msym_kernel), a full one-radius sweep for the free energies, nn segment"}' $O/r4_c5_sq1 $O/r4_c5_sq2 -- $O/r4_c5_tcc
cp $O/r4_c5_onesided_pop.json profiles/r4_c5_onesided_pop.json
cp $O/r4_spread10_bench.json profiles/r4_spread10_bench.json
cp $O/r4_unfav_oneblob.json profiles/r4_unfav_oneblob.json
cp $O/r4_unfav_uniform.json profiles/r4_unfav_uniform.json
grep SEG $O/r4_seg.txt > profiles/r4_segments.txt
f=$(find $O/r4_seg8_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" profiles/r4_seg8_kernel_stats.csv
cp $O/r4_c3_mfma32_bench.json profiles/r4_c3_mfma32_bench.json
f=$(find $O/r4_c3_mfma32_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" profiles/r4_c3_mfma32_kernel_stats.csv
python3 scratch/pmc_summary.py $O/r4_c3_mfma32_sq1 > profiles/r4_c3_mfma32_pmc.json
ls -la profiles | grep r4_
