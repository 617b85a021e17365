#!/bin/bash
# A/B on one GPU box: for each variant library (clustering_amd/lib/variants/NAME.so) run kbench, twice, interleaved
cd $GRAFT_REPO_ROOT
ARGS="${KBENCH_ARGS:---n 1000000 --d 10 --variant pruned --reps 3}"
cp clustering_amd/lib/libdcdensity.so /tmp/lib_saved.so
for round in 1 2; do
  for v in "$@"; do
    cp clustering_amd/lib/variants/$v.so clustering_amd/lib/libdcdensity.so
    echo "== $v (round $round)"; timeout 200 python3 scratch/kbench.py $ARGS 2>&1 | grep -v "^ \|amdgpu.ids"
  done
done
cp /tmp/lib_saved.so clustering_amd/lib/libdcdensity.so
