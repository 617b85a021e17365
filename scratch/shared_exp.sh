#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R
DC_POP_SHARED=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -4
for v in 0 1; do echo "== DC_POP_SHARED=$v"; DC_POP_SHARED=$v timeout 600 python3 scratch/c5_bench.py --reps 2 --radii 0.35 0.6 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('pop_per_radius_ms','nn_ms','full_single_radius_sweep_all_rows_ms')})"; done
for v in 0 1; do echo "== DC_POP_SHARED=$v kbench"; DC_POP_SHARED=$v python3 scratch/kbench.py --n 400000 --d 30 --radii 0.5 --variant pruned --reps 3 --what pop 2>&1 | grep "pruned n="; DC_POP_SHARED=$v python3 scratch/kbench.py --n 1000000 --d 10 --variant pruned --reps 3 --what pop 2>&1 | grep "pruned n="; DC_POP_SHARED=$v python3 scratch/kbench.py --n 2000000 --d 20 --radii 0.4 --variant pruned --reps 2 --what pop 2>&1 | grep "pruned n="; done
