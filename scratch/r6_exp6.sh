#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R; O=$R/gpurun_out
for v in pubst nopubst; do
  DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so timeout 300 python3 bench.py --variant mfma32 --steps 1 --warmup 0 --cpu-sample 0 --no-full-sweep > $O/r6_exp6_$v.json 2> $O/r6_exp6_$v.err
  echo "== $v"; grep nn32 $O/r6_exp6_$v.json | sort | uniq | head -40
done
