#!/bin/bash
# average resident waves per SIMD of the pruned sweeps at C3 (SQ_WAVE_CYCLES against GRBM_GUI_ACTIVE)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/occ
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/occ -o s -- python3 $R/scratch/kbench.py --n 1000000 --d 10 --variant pruned --reps 2 > /dev/null 2>&1
cd $R; python3 scratch/pmc_summary.py gpurun_out/occ | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,e in d.items():
    if 'pruned' in k:
        cyc=e['GRBM_GUI_ACTIVE']/8
        print(k[30:70], 'waves/SIMD', round(e['SQ_WAVE_CYCLES']*4/(1024*cyc),3), 'valu issue util', round(e['SQ_INSTS_VALU']*4/(1024*cyc),3), 'mfma busy', round(e['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*cyc),3), 'waves', e['SQ_WAVES'])"
