#!/bin/bash
# Round-2 measurements (run through gpurun from the repo root):  gpurun --timeout 2400 -- 'bash scratch/profile_r2.sh [c3] [c2] [c5]'
# Per workload: the bench-style JSON line, rocprofv3 --kernel-trace --stats, and PMC passes in their own runs.
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
WHAT="${@:-c3 c2 c5}"
SQ1="GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
TCC="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
prof() {  # tag, program args...
  local tag=$1; shift
  rm -rf $O/${tag}_stats $O/${tag}_sq1 $O/${tag}_sq2 $O/${tag}_tcc
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o s -- "$@" > $O/${tag}_stats.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/${tag}_sq1 -o s -- "$@" > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/${tag}_sq2 -o s -- "$@" > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $TCC --output-format csv -d $O/${tag}_tcc -o s -- "$@" > /dev/null 2>&1
  # keep what is needed for the summaries only (the traces are large)
  find $O/${tag}_stats -name '*kernel_trace.csv' -size +20M -delete
}
for w in $WHAT; do
  case $w in
    c3)
      (cd $R && timeout 600 python3 bench.py --steps 10 --warmup 3 > $O/c3_bench.json 2> $O/c3_bench.err)
      prof c3 python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 ;;
    c2)
      (cd $R && timeout 600 python3 bench.py --n-rows 100000 --radii 0.1 0.2 0.3 --no-nn --steps 20 --warmup 3 --cpu-sample 100000 > $O/c2_bench.json 2> $O/c2_bench.err)
      prof c2 python3 $R/bench.py --n-rows 100000 --radii 0.1 0.2 0.3 --no-nn --steps 5 --warmup 1 --cpu-sample 0 ;;
    c5)
      (cd $R && timeout 900 python3 scratch/c5_bench.py > $O/c5_bench.json 2> $O/c5_bench.err)
      prof c5 python3 $R/scratch/c5_bench.py --reps 1 ;;
  esac
done
ls $O
for w in $WHAT; do tail -c 1500 $O/${w}_bench.json; echo; tail -c 300 $O/${w}_bench.err; done
