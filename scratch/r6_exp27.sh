#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R
for v in hd6 ent6 hd6 ent6; do
  echo -n "$v: "; DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so timeout 900 python3 scratch/c5_bench.py --reps 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pop %.1f nn %.2f full1 %.1f' % (d['pop_8_radii_ms'], d['nn_ms'], d['full_single_radius_sweep_all_rows_ms']))"
done
