#!/bin/bash
# kernel trace of one pruned pop + nn call at C3 (for the per-kernel timeline of the preparation passes)
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace1 -o s -- python3 $R/scratch/kbench.py --n 1000000 --d 10 --variant pruned --reps 2 > /dev/null 2>&1
ls $R/gpurun_out/trace1
