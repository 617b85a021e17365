#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R; O=$R/gpurun_out
for rep in 1 2; do for w in 4 8 2; do for g in 8 4; do
  DC_NN_COOP_WAVES=$w timeout 300 python3 scratch/seg_bench.py 1000000 10 $g | tail -1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('SEG'):
        d=json.loads(l[4:]); print('coop waves $w G=$g: nn_kernel %.3f nn_call %.3f step %.3f' % (d['nn_kernel_ms']['mean'], d['nn_call_ms']['max'], d['per_rank_step_ms_before_collectives']))"
done; done; done
