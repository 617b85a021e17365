#!/bin/bash
# several radii in one symmetric sweep, other shapes than C5: the library before / after the second half of round 6
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R
for shape in "1000000 30 0.30 0.35 0.40 0.45 0.50 0.55 0.60 0.65" "1000000 16 0.24 0.28 0.32 0.36" "300000 26 0.4 0.44 0.48 0.52 0.56 0.6 0.64 0.68" "2000000 20 0.30 0.33 0.36 0.39 0.42 0.45 0.48 0.51" "600000 12 0.18 0.20 0.22 0.24 0.26 0.28 0.30 0.32"; do
  set -- $shape; n=$1; d=$2; shift 2
  for v in old456 new456 old456 new456; do
    printf "n=%s d=%s radii=%s %s: " $n $d "$#" $v
    DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so timeout 600 python3 scratch/kbench.py --n $n --d $d --radii "$@" --variant pruned --reps 3 --what pop 2>&1 | grep "pruned n=" | sed 's/.*radii=[0-9]*: //'
  done
done
