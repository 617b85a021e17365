#!/bin/bash
# rocprofv3 kernel stats + three PMC passes around an arbitrary command:  prof_cmd.sh <tag> <program> [args...]
# (run through gpurun from the repo root; summaries land in gpurun_out/<tag>_*)
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
tag=$1; shift
SQ1="GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"
TCC="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
rm -rf $O/${tag}_stats $O/${tag}_sq1 $O/${tag}_sq2 $O/${tag}_tcc
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o s -- "$@" > $O/${tag}_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/${tag}_sq1 -o s -- "$@" > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/${tag}_sq2 -o s -- "$@" > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $TCC --output-format csv -d $O/${tag}_tcc -o s -- "$@" > /dev/null 2>&1
find $O/${tag}_stats -name '*kernel_trace.csv' -size +20M -delete
python3 $R/scratch/pmc_summary.py $O/${tag}_sq1 $O/${tag}_sq2 $O/${tag}_tcc > $O/${tag}_pmc.json
f=$(find $O/${tag}_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200 > $O/${tag}_kernel_stats_head.csv
cat $O/${tag}_pmc.json | head -80
