#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ3="SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
TCC="TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run() { # tag env...
  local tag=$1; shift
  for set in SQ2 SQ3 TCC; do
    rm -rf $O/r6_exp11_${tag}_$set
    env "$@" timeout 600 rocprofv3 --kernel-trace --pmc ${!set} --output-format csv -d $O/r6_exp11_${tag}_$set -o s -- python3 $R/bench.py --variant mfma32 --steps 1 --warmup 1 --cpu-sample 0 --no-full-sweep > /dev/null 2>&1
  done
  echo "== $tag"
  python3 $R/scratch/pmc_summary.py $O/r6_exp11_${tag}_SQ2 $O/r6_exp11_${tag}_SQ3 $O/r6_exp11_${tag}_TCC | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print(k[30:58], {a:('%.4g'%b) for a,b in v.items()})"
  find $O/r6_exp11_${tag}_* -name '*.csv' -size +5M -delete
}
run pub DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_pub.so
run nopub DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_nopub.so
