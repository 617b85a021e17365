#!/bin/bash
# fuzz with the shared-operand sweeps forced onto every shape they have an instance for (the multi-radius symmetric sweep with its
# two threshold paths, pop_shared_kernel, nn_shared_kernel)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R; O=$R/gpurun_out/r6_fuzz_shared.txt; : > $O
run() { echo "== $*" >> $O; ( "$@" 2>&1 | grep -v amdgpu.ids | tail -3 ) >> $O; }
run env DC_POP_SHARED=1 DC_NN_SHARED=1 timeout 1200 python3 scratch/fuzz.py 6201 700
run env DC_POP_SHARED=1 DC_NN_SHARED=1 timeout 900 python3 scratch/fuzz.py 6202 40 big
run env DC_POP_SHARED=1 DC_POP_MSYM=0 timeout 900 python3 scratch/fuzz.py 6203 200
run env DC_POP_SHARED=1 DC_NN_SHARED=1 DC_CANON_ORDER=fma timeout 900 python3 scratch/fuzz.py 6204 150
cat $O
