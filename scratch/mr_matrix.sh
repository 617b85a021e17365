#!/bin/bash
# several radii per call: one sweep per radius (per-wave streams) against one shared-operand sweep for all radii
cd $GRAFT_REPO_ROOT
for shape in "1000000 16 0.22 0.24 0.26 0.28 0.30 0.32 0.34 0.36" "1000000 16 0.24 0.28 0.32 0.36" "600000 12 0.18 0.20 0.22 0.24 0.26 0.28 0.30 0.32" "2000000 20 0.30 0.33 0.36 0.39 0.42 0.45 0.48 0.51" "1000000 10 0.1 0.15 0.2 0.25 0.3 0.35 0.4 0.45" "300000 26 0.4 0.44 0.48 0.52 0.56 0.6 0.64 0.68"; do
  set -- $shape; n=$1; d=$2; shift 2
  for mr in 1000 3; do
    printf "n=%s d=%s radii=%s DC_MR_MIN_RADII=%s: " $n $d "$#" $mr
    DC_MR_MIN_RADII=$mr timeout 600 python3 scratch/kbench.py --n $n --d $d --radii "$@" --variant pruned --reps 2 --what pop 2>&1 | grep "pruned n=" | sed 's/.*radii=[0-9]*: //'
  done
done
