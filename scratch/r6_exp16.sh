#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R; O=$R/gpurun_out
timeout 900 python3 scratch/c5_bench.py --pop-only --reps 3 2>/dev/null | tail -1
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_components.py -x -q -k "multi_radius or shared_operand or sweep_forms or adjacent" 2>&1 | tail -3
timeout 900 python3 scratch/c5_bench.py --reps 2 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pop %.1f nn %.1f' % (d['pop_8_radii_ms'], d['nn_ms']))"
