#!/bin/bash
# population sweep, per-wave operand streams (DC_POP_SHARED=0) against LDS-shared operands (=1), over shapes
cd $GRAFT_REPO_ROOT
for shape in "300000 26 0.5" "1000000 16 0.3" "2000000 20 0.4" "1000000 30 0.5" "3000000 24 0.45" "1000000 40 0.6" "600000 12 0.25" "4000000 12 0.25"; do
  set -- $shape
  for v in 0 1; do
    printf "n=%s d=%s r=%s shared=%s: " $1 $2 $3 $v
    DC_POP_SHARED=$v timeout 300 python3 scratch/kbench.py --n $1 --d $2 --radii $3 --variant pruned --reps 2 --what pop 2>&1 | grep "pruned n=" | sed 's/.*radii=1: //'
  done
done
