#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R
for v in hd6 ck6 hd6 ck6; do
  echo -n "$v: "; DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so timeout 900 python3 scratch/c5_bench.py --pop-only --reps 2 2>/dev/null | tail -1
done
