#!/bin/bash
cd $GRAFT_REPO_ROOT
for l in 0 1 4 8 16 0; do
  echo -n "DC_NN_LEAVE=$l: "
  DC_NN_LEAVE=$l python3 scratch/spread_bench.py 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nn call', round(d['nn_call_ms'],3), 'kernel', round(d['nn_kernel_ms'],3), 'tiles', d['nn_tiles'], 'sigma2', d['sigma2'])"
done
