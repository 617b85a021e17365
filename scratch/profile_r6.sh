#!/bin/bash
# Round-6 measurements (run through gpurun from the repo root):
#   gpurun --timeout 2700 -- 'bash scratch/profile_r6.sh [c3] [c2] [c5] [spread] [seg] [unfav] [mfma32]'
# Per workload: the bench-style JSON line, rocprofv3 --kernel-trace --stats, and PMC passes in their own runs; the digest
# of the kernel sources the numbers belong to (bench.py refuses a counter profile of other sources).
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
WHAT="${@:-c3 c2 c5 spread seg unfav mfma32}"
SQ1="GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
TCC="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
(cd $R && python3 -c "import bench; print(bench.library_digest())" > $O/r6_csrc_digest.txt)   # the digest the LOADED library carries (dc_hip_build_digest)
prof() {  # tag, program args...
  local tag=$1; shift
  rm -rf $O/${tag}_stats $O/${tag}_sq1 $O/${tag}_sq2 $O/${tag}_tcc
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o s -- "$@" > $O/${tag}_stats.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/${tag}_sq1 -o s -- "$@" > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/${tag}_sq2 -o s -- "$@" > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $TCC --output-format csv -d $O/${tag}_tcc -o s -- "$@" > /dev/null 2>&1
  find $O/${tag}_stats -name '*kernel_trace.csv' -size +20M -delete
}
for w in $WHAT; do
  case $w in
    c3)
      (cd $R && timeout 600 python3 bench.py --steps 10 --warmup 3 > $O/r6_c3_bench.json 2> $O/r6_c3_bench.err)
      prof r6_c3 python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 ;;
    c2)
      (cd $R && timeout 600 python3 bench.py --n-rows 100000 --radii 0.1 0.2 0.3 --no-nn --steps 20 --warmup 3 --cpu-sample 100000 > $O/r6_c2_bench.json 2> $O/r6_c2_bench.err)
      prof r6_c2 python3 $R/bench.py --n-rows 100000 --radii 0.1 0.2 0.3 --no-nn --steps 5 --warmup 1 --cpu-sample 0 ;;
    c5)
      (cd $R && timeout 900 python3 scratch/c5_bench.py > $O/r6_c5_bench.json 2> $O/r6_c5_bench.err)
      (cd $R && DC_POP_MSYM=0 timeout 900 python3 scratch/c5_bench.py --pop-only > $O/r6_c5_onesided_pop.json 2>/dev/null)
      prof r6_c5 python3 $R/scratch/c5_bench.py --reps 1 ;;
    spread)
      (cd $R && timeout 600 python3 scratch/spread_bench.py 10 > $O/r6_spread10_bench.json 2> $O/r6_spread10_bench.err) ;;
    seg)
      (cd $R && for g in 1 2 4 8; do timeout 300 python3 scratch/seg_bench.py 1000000 10 $g | tail -1; done > $O/r6_seg.txt 2>&1)
      rm -rf $O/r6_seg8_stats
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r6_seg8_stats -o s -- python3 $R/scratch/seg_bench.py 1000000 10 8 > /dev/null 2>&1
      find $O/r6_seg8_stats -name '*kernel_trace.csv' -size +20M -delete ;;
    unfav)
      (cd $R && timeout 600 python3 scratch/unfav_bench.py oneblob > $O/r6_unfav_oneblob.json 2> $O/r6_unfav.err)
      (cd $R && timeout 600 python3 scratch/unfav_bench.py uniform > $O/r6_unfav_uniform.json 2>> $O/r6_unfav.err) ;;
    mfma32)
      (cd $R && timeout 600 python3 bench.py --variant mfma32 --steps 3 --warmup 1 --cpu-sample 0 > $O/r6_c3_mfma32_bench.json 2> $O/r6_c3_mfma32_bench.err)
      rm -rf $O/r6_c3_mfma32_stats $O/r6_c3_mfma32_sq1
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r6_c3_mfma32_stats -o s -- python3 $R/bench.py --variant mfma32 --steps 1 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
      timeout 900 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/r6_c3_mfma32_sq1 -o s -- python3 $R/bench.py --variant mfma32 --steps 1 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
      find $O/r6_c3_mfma32_stats -name '*kernel_trace.csv' -size +20M -delete ;;
  esac
done
ls $O | grep r6_
