#!/bin/bash
# round 6, experiment 1: the neighbour sweep's shares as waves of one workgroup (DC_NN_COOP) at G = 8 / 4 / 1; the
# fp32-input MFMA instance with clumped chains and the two-bit epilogue
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R; O=$R/gpurun_out
{
for rep in 1 2; do
for coop in 0 1; do
  for g in 8 4 1; do echo "== DC_NN_COOP=$coop G=$g"; DC_NN_COOP=$coop timeout 300 python3 scratch/seg_bench.py 1000000 10 $g | tail -1; done
done; done
} > $O/r6_exp1_seg.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "fp32_mfma or segment or neighb" > $O/r6_exp1_tests.txt 2>&1
timeout 600 python3 bench.py --variant mfma32 --steps 3 --warmup 1 --cpu-sample 0 > $O/r6_exp1_mfma32.json 2> $O/r6_exp1_mfma32.err
tail -3 $O/r6_exp1_tests.txt; cat $O/r6_exp1_seg.txt | grep -E "==|SEG" | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('=='): print(l.strip(), end='  ')
    elif l.startswith('SEG'):
        d=json.loads(l[4:]); print('nn_kernel %.3f nn_call %.3f pop_call %.3f step %.3f' % (d['nn_kernel_ms']['mean'], d['nn_call_ms']['max'], d['pop_call_ms']['max'], d['per_rank_step_ms_before_collectives']))
"
python3 -c "
import json;d=json.loads(open('$O/r6_exp1_mfma32.json').read().strip().split('\n')[-1]);print(json.dumps(d.get('roofline_by_kernel'),indent=0)[:1500]); print(d.get('check'))"
