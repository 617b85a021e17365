#!/bin/bash
# clock and matrix-pipe duty of the fp32 sweeps, with and without the published bounds (why is the variant with FEWER rare
# chains slower?)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SQ1="GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM"
for v in pub nopub; do
  rm -rf $O/r6_exp8_${v}_sq1 $O/r6_exp8_${v}_stats
  DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r6_exp8_${v}_stats -o s -- python3 $R/bench.py --variant mfma32 --steps 2 --warmup 1 --cpu-sample 0 --no-full-sweep > /dev/null 2>&1
  DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/r6_exp8_${v}_sq1 -o s -- python3 $R/bench.py --variant mfma32 --steps 2 --warmup 1 --cpu-sample 0 --no-full-sweep > /dev/null 2>&1
  echo "== $v"
  python3 $R/scratch/pmc_summary.py $O/r6_exp8_${v}_sq1 | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print(k[:60], {a:('%.4g'%b) for a,b in v.items()})"
  grep -h "mfma32_kernel" $(find $O/r6_exp8_${v}_stats -name '*kernel_stats.csv') | cut -c1-200
  find $O/r6_exp8_${v}_stats $O/r6_exp8_${v}_sq1 -name '*.csv' -size +5M -delete
done
