#!/bin/bash
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_session.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python scratch/comp_diag.py 1 10 100 2>&1 | tail -3
timeout 300 python scratch/spread_exp.py 2>&1 | tail -4
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-full-sweep > gpurun_out/r3_comp_b1.json 2> gpurun_out/r3_comp_b1.err
python3 -c "
import json
l=json.loads(open('gpurun_out/r3_comp_b1.json').read().strip().splitlines()[-1])
print('ms/step', l['ms_per_step'], {k:round(v,3) for k,v in l['phases_ms'].items()}, l['check'])
"
timeout 300 python scratch/seg_bench.py 1000000 10 8 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r3_comp_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_comp_trace -o s -- python3 $R/scratch/comp_diag.py 1 > /dev/null 2>&1
find $R/gpurun_out/r3_comp_trace -name '*kernel_trace.csv' -size +20M -delete
