#!/bin/bash
# parity of the last variant, then A/B of variant libraries: C3 (pop, nn), one rank of C5, 300k x 26
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
cp clustering_amd/lib/libdcdensity.so /tmp/lib_saved.so
last="${@: -1}"
cp clustering_amd/lib/variants/$last.so clustering_amd/lib/libdcdensity.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for v in "$@"; do
  cp clustering_amd/lib/variants/$v.so clustering_amd/lib/libdcdensity.so
  echo "== $v (round $rep)"
  timeout 300 python3 scratch/kbench.py --n 1000000 --d 10 --variant pruned --reps 3 2>&1 | grep "pruned n="
  if [ "$rep" = 1 ]; then
    timeout 300 python3 scratch/kbench.py --n 300000 --d 26 --radii 0.5 --variant pruned --reps 3 2>&1 | grep "pruned n="
    timeout 900 python3 scratch/c5_bench.py --reps 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(d[k],1) for k in ('pop_8_radii_ms','nn_ms','full_single_radius_sweep_all_rows_ms')})"
  fi
done; done
cp /tmp/lib_saved.so clustering_amd/lib/libdcdensity.so
