#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
PYTHONPATH=$R/scratch/oldpkg python3 $R/scratch/ab_pop.py 2>&1 | tail -1
PYTHONPATH=$R python3 $R/scratch/ab_pop.py 2>&1 | tail -1
DC_POP_COMPONENTS=0 PYTHONPATH=$R python3 $R/scratch/ab_pop.py 2>&1 | tail -1
done
rm -rf $R/gpurun_out/ab_old $R/gpurun_out/ab_new
PYTHONPATH=$R/scratch/oldpkg timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab_old -o s -- python3 $R/scratch/ab_pop.py > /dev/null 2>&1
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab_new -o s -- python3 $R/scratch/ab_pop.py > /dev/null 2>&1
