#!/bin/bash
# fuzz of the final libraries (all three summation orders; the COOP neighbour sweep forced on small shapes)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R; O=$R/gpurun_out/r6_fuzz_final.txt; : > $O
run() { echo "== $*" >> $O; ( "$@" 2>&1 | tail -3 ) >> $O; }
run timeout 900 python3 scratch/fuzz.py 6101 400
run timeout 600 python3 scratch/fuzz.py 6102 40 big
run env DC_NN_COOP=1 DC_SHARE_FLOOR=16 DC_NN_COOP_WAVES=4 timeout 600 python3 scratch/fuzz.py 6103 150
run env DC_NN_COOP=1 DC_SHARE_FLOOR=8 DC_NN_COOP_WAVES=2 timeout 600 python3 scratch/fuzz.py 6104 100
run env DC_CANON_ORDER=avx timeout 600 python3 scratch/fuzz.py 6105 120
run env DC_CANON_ORDER=fma timeout 600 python3 scratch/fuzz.py 6106 120
for ch in 0 3; do run env DC_MFMA32_CHUNKS=$ch timeout 600 python3 scratch/fuzz32.py $((ch+6110)) 150; done
cat $O
