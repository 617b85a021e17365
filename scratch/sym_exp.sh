#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -8
for rep in 1 2; do for v in 0 1; do
  echo "== DC_POP_SYM=$v"
  DC_POP_SYM=$v timeout 300 python3 scratch/kbench.py --n 1000000 --d 10 --variant pruned --reps 3 --what pop 2>&1 | grep "pruned n=\|raw counter"
done; done
for v in 0 1; do
  echo "== DC_POP_SYM=$v"
  DC_POP_SYM=$v timeout 300 python3 scratch/kbench.py --n 100000 --d 10 --radii 0.1 0.2 0.3 --variant pruned --reps 5 --what pop 2>&1 | grep "pruned n="
  DC_POP_SYM=$v timeout 300 python3 scratch/kbench.py --n 300000 --d 26 --radii 0.5 --variant pruned --reps 3 --what pop 2>&1 | grep "pruned n="
  DC_POP_SYM=$v timeout 300 python3 scratch/kbench.py --n 1000000 --d 3 --radii 0.05 --variant pruned --reps 3 --what pop 2>&1 | grep "pruned n="
done
