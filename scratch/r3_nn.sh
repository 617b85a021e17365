#!/bin/bash
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_components.py tests/test_gpu_fuzz.py tests/test_gpu_session.py -x -q -m gpu 2>&1 | tail -12
timeout 300 python scratch/spread_exp.py 2>&1 | tail -4
timeout 300 python scratch/seg_bench.py 1000000 10 8 | tail -1
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-full-sweep 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step', l['ms_per_step'], {k:round(v,3) for k,v in l['phases_ms'].items()}, l['check'])
"
