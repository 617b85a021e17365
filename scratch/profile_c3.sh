#!/bin/bash
# Bench line, kernel-trace stats and PMC passes of the C3 workload (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash scratch/profile_c3.sh'
# Outputs under gpurun_out/ (bench.json, stats/, pmc_f1..3/); scratch/make_pmc_profile.py turns the PMC
# passes into profiles/r1_pruned_pmc.json.
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/stats $R/gpurun_out/pmc_f1 $R/gpurun_out/pmc_f2 $R/gpurun_out/pmc_f3
(cd $R && timeout 600 python3 bench.py --steps 5 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err)
tail -c 600 $R/gpurun_out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats -o s -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
KB="python3 $R/scratch/kbench.py --n 1000000 --d 10 --variant pruned --reps 1"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_f1 -o s -- $KB > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_f2 -o s -- $KB > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $R/gpurun_out/pmc_f3 -o s -- $KB > /dev/null 2>&1
ls $R/gpurun_out/stats $R/gpurun_out/pmc_f1 $R/gpurun_out/pmc_f2 $R/gpurun_out/pmc_f3
cat $R/gpurun_out/bench.json | cut -c1-400
