#!/bin/bash
# resident waves per SIMD of the pruned sweeps, unsharded and as one eighth (SQ_WAVE_CYCLES against GRBM_GUI_ACTIVE)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
for G in 1 8; do
  rm -rf $R/gpurun_out/occ_g$G
  timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/occ_g$G -o s -- python3 $R/scratch/seg_bench.py 1000000 10 $G > /dev/null 2>&1
  (cd $R; python3 scratch/pmc_summary.py gpurun_out/occ_g$G | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,e in d.items():
    if 'pruned' in k:
        cyc=e['GRBM_GUI_ACTIVE']/8
        print('G=$G', k[30:62], 'dispatches', e.get('dispatches'), 'waves/SIMD', round(e['SQ_WAVE_CYCLES']*4/(1024*cyc),3), 'busy/active', round(e['SQ_BUSY_CYCLES']/ (e['GRBM_GUI_ACTIVE']*4),3) , 'valu issue util', round(e['SQ_INSTS_VALU']*4/(1024*cyc),3), 'mfma busy', round(e['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*cyc),3), 'waves', e['SQ_WAVES'])")
done
