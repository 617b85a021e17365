#!/bin/bash
# A/B of variant libraries on the neighbour sweep (C3, 300k x 26, 1M x 3), alternating twice; parity of the last variant first
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
cp clustering_amd/lib/libdcdensity.so /tmp/lib_saved.so
last="${@: -1}"
cp clustering_amd/lib/variants/$last.so clustering_amd/lib/libdcdensity.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for v in "$@"; do
  cp clustering_amd/lib/variants/$v.so clustering_amd/lib/libdcdensity.so
  echo "== $v (round $rep)"
  timeout 300 python3 scratch/kbench.py --n 1000000 --d 10 --variant pruned --reps 3 --what nn 2>&1 | grep "pruned n="
  [ "$rep" = 1 ] && timeout 300 python3 scratch/kbench.py --n 300000 --d 26 --radii 0.5 --variant pruned --reps 3 --what nn 2>&1 | grep "pruned n="
done; done
cp /tmp/lib_saved.so clustering_amd/lib/libdcdensity.so
