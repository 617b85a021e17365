#!/bin/bash
# Round-3 baseline on this round's boxes: bench line, segment timings + kernel trace of a G=8 segment run,
# the spread experiment, and C3 with the shared-operand sweeps forced.
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
(cd $R && timeout 600 python3 bench.py --steps 10 --warmup 3 > $O/r3_base_bench.json 2> $O/r3_base_bench.err)
(cd $R && timeout 300 python3 scratch/seg_bench.py 1000000 10 8 > $O/r3_base_seg8.txt 2>&1)
(cd $R && timeout 300 python3 scratch/seg_bench.py 1000000 10 1 >> $O/r3_base_seg8.txt 2>&1)
rm -rf $O/r3_base_segtrace
(cd $R && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/r3_base_segtrace -o s -- python3 scratch/seg_bench.py 1000000 10 8 > $O/r3_base_segtrace.log 2>&1)
(cd $R && timeout 600 python3 scratch/spread_exp.py > $O/r3_base_spread.txt 2>&1)
(cd $R && DC_NN_SHARED=1 DC_POP_SHARED=1 timeout 600 python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 > $O/r3_base_shared_bench.json 2> $O/r3_base_shared_bench.err)
cat $O/r3_base_seg8.txt $O/r3_base_spread.txt
tail -c 600 $O/r3_base_bench.json
