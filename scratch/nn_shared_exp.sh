#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R
DC_NN_SHARED=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -4
for v in 0 1; do
  echo "== DC_NN_SHARED=$v"
  DC_NN_SHARED=$v python3 scratch/kbench.py --n 1000000 --d 30 --radii 0.5 --variant pruned --reps 3 --what nn 2>&1 | grep "pruned n="
  DC_NN_SHARED=$v python3 scratch/kbench.py --n 1000000 --d 40 --radii 0.6 --variant pruned --reps 2 --what nn 2>&1 | grep "pruned n="
  DC_NN_SHARED=$v python3 scratch/kbench.py --n 300000 --d 26 --radii 0.5 --variant pruned --reps 3 --what nn 2>&1 | grep "pruned n="
  DC_NN_SHARED=$v timeout 600 python3 scratch/c5_bench.py --reps 2 --radii 0.5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('nn_ms',)}, d['evaluated_fraction']['nn'])"
done
