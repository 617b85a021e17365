#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R
for v in head_full red head_full red; do
  echo -n "$v: "; DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so timeout 900 python3 scratch/c5_bench.py --pop-only --reps 2 2>/dev/null | tail -1
done
DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_red.so timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_components.py -x -q -k "multi_radius or shared_operand or adjacent" 2>&1 | tail -2
