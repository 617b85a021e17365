#!/bin/bash
# block-cyclic deal of the query groups: per-rank times of an eighth of C3 against the block size
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for b in 1 4 16 32 64 128 256 1024; do
  echo -n "rep $rep block $b: "
  DC_SEG_BLOCK=$b python3 scratch/seg_bench.py 1000000 10 8 | tail -1 | cut -c5- | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print('pop kernel mean/max', round(d['pop_kernel_ms']['mean'],3), round(d['pop_kernel_ms']['max'],3), ' nn kernel mean/max', round(d['nn_kernel_ms']['mean'],3), round(d['nn_kernel_ms']['max'],3), ' per-rank', round(d['per_rank_step_ms_before_collectives'],3))"
done; done
