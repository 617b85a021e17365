#!/bin/bash
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd $R
echo "== components on"; timeout 300 python scratch/comp_diag.py 1 10 2>&1 | tail -2
echo "== components off"; DC_POP_COMPONENTS=0 timeout 300 python scratch/comp_diag.py 1 10 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r3_comp_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_comp_trace -o s -- python3 $R/scratch/comp_diag.py 1 > /dev/null 2>&1
find $R/gpurun_out/r3_comp_trace -name '*kernel_trace.csv' -size +20M -delete
ls $R/gpurun_out/r3_comp_trace
