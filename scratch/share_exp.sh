#!/bin/bash
# experiment: L2-sized reference shares (DC_SHARE_KB) at C5 (segment) and C3
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R
for kb in 0 8192 3072 1536 768; do
  echo "== DC_SHARE_KB=$kb (C5 segment, 2 radii)"
  DC_SHARE_KB=$kb timeout 600 python3 scratch/c5_bench.py --reps 1 --radii 0.35 0.6 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('pop_per_radius_ms','nn_ms','full_single_radius_sweep_all_rows_ms')})"
done
for kb in 0 3072 1536 768; do
  echo "== DC_SHARE_KB=$kb (C3)"
  DC_SHARE_KB=$kb timeout 300 python3 scratch/kbench.py --n 1000000 --d 10 --variant pruned --reps 3 2>/dev/null | grep "pruned n="
done
