#!/bin/bash
# COOP neighbour sweep at G = 8: shares per group (wave target x share floor)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R; O=$R/gpurun_out
for wt in 16000 24000 40960 65536; do for fl in 450 900 1800; do
  DC_WAVE_TARGET=$wt DC_SHARE_FLOOR=$fl timeout 300 python3 scratch/seg_bench.py 1000000 10 8 | tail -1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('SEG'):
        d=json.loads(l[4:]); print('target $wt floor $fl: nn_kernel %.3f nn_call %.3f pop_kernel %.3f pop_call %.3f' % (d['nn_kernel_ms']['mean'], d['nn_call_ms']['max'], d['pop_kernel_ms']['mean'], d['pop_call_ms']['max']))"
done; done
