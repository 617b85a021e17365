#!/bin/bash
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd $R
timeout 1500 python -m pytest tests/test_gpu_components.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python scratch/comp_diag.py 1 10 2>&1 | tail -2
timeout 900 python scratch/c5_bench.py > gpurun_out/r3_c5_bench.json 2> gpurun_out/r3_c5_bench.err
python3 -c "
import json
l=json.loads(open('gpurun_out/r3_c5_bench.json').read().strip().splitlines()[-1])
print('C5 pop8', l['pop_8_radii_ms'], 'nn', l['nn_ms'], 'full1', l['full_single_radius_sweep_all_rows_ms'], l['components'])
"
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-full-sweep 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step', l['ms_per_step'], {k:round(v,3) for k,v in l['phases_ms'].items()})
"
