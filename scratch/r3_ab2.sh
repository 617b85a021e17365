#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SQ1="GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
for tag in old new; do
  if [ $tag = old ]; then export PYTHONPATH=$R/scratch/oldpkg; else export PYTHONPATH=$R; fi
  rm -rf $R/gpurun_out/ab2_${tag}_sq1 $R/gpurun_out/ab2_${tag}_sq2 $R/gpurun_out/ab2_${tag}_tcc
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $R/gpurun_out/ab2_${tag}_sq1 -o s -- python3 $R/scratch/ab_pop.py > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $R/gpurun_out/ab2_${tag}_sq2 -o s -- python3 $R/scratch/ab_pop.py > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/ab2_${tag}_tcc -o s -- python3 $R/scratch/ab_pop.py > /dev/null 2>&1
  find $R/gpurun_out/ab2_${tag}_* -name '*kernel_trace.csv' -delete
done
ls $R/gpurun_out | grep ab2
