#!/bin/bash
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd $R
for cfg in "0 0" "98304 1024" "98304 2048" "49152 512" "49152 1024" "196608 512" "32768 1024"; do
  set -- $cfg
  echo "== target $1 floor $2"
  DC_WAVE_TARGET=$1 DC_SHARE_FLOOR=$2 timeout 300 python scratch/seg_bench.py 1000000 10 8 2>&1 | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip()[4:]); print('  G8 pop k %.3f call %.3f | nn k %.3f call %.3f | per-rank %.3f'%(l['pop_kernel_ms']['mean'], l['pop_call_ms']['mean'], l['nn_kernel_ms']['mean'], l['nn_call_ms']['mean'], l['per_rank_pop_nn_prep_ms']))"
  DC_WAVE_TARGET=$1 DC_SHARE_FLOOR=$2 timeout 300 python scratch/seg_bench.py 1000000 10 1 2>&1 | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip()[4:]); print('  G1 pop k %.3f call %.3f | nn k %.3f call %.3f'%(l['pop_kernel_ms']['mean'], l['pop_call_ms']['mean'], l['nn_kernel_ms']['mean'], l['nn_call_ms']['mean']))"
done
