#!/bin/bash
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_components.py tests/test_gpu_session.py -x -q -m gpu 2>&1 | tail -6
timeout 300 python scratch/nn_diag.py 2>&1 | tail -2
timeout 300 python scratch/seg_bench.py 1000000 10 8 | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r3_nn_trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3_nn_trace -o s -- python3 $R/scratch/nn_diag.py > /dev/null 2>&1
