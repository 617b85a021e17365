#!/bin/bash
# DC_NN_BLOCK_KEY (2 x 2 cell blocks, coarse free energy first) on other shapes than C3
cd $GRAFT_REPO_ROOT
for shape in "1000000 3" "300000 26" "600000 12" "1000000 16" "100000 10" "1000000 30"; do
  set -- $shape
  for cb in 0 2 3; do
    echo -n "n=$1 d=$2 block key $cb: "
    DC_NN_BLOCK_KEY=$cb python3 scratch/kbench.py --n $1 --d $2 --variant pruned --what nn --reps 3 --radii 0.3 | tail -1 | cut -c1-60
  done
done
for cb in 0 2; do echo -n "C5 block key $cb: "; DC_NN_BLOCK_KEY=$cb python3 scratch/c5_bench.py --reps 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nn', round(d['nn_ms'],1), 'pop8', round(d['pop_8_radii_ms'],1))"; done
