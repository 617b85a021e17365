#!/bin/bash
# experiment builds of the neighbour sweep in /tmp: query tiles per wave and register budget (waves per SIMD)
cd $GRAFT_REPO_ROOT
run() {  # tag, flags
  rm -rf /tmp/exp && mkdir -p /tmp/exp && cp -r clustering_amd include scratch /tmp/exp/
  (cd /tmp/exp/clustering_amd/csrc && touch dc_mfma_kernels.hpp && make -j16 MFMA_STEPS="2" CXXFLAGS_EXTRA="$2" ../lib/libdcdensity.so > /tmp/exp/build.log 2>&1) || { tail -5 /tmp/exp/build.log; return; }
  echo "== $1 ($2)"
  (cd /tmp/exp && python3 scratch/spread_bench.py 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pop', round(d['pop_kernel_ms'],3), 'nn', round(d['nn_kernel_ms'],3), 'nn tiles', d['nn_tiles'])")
}
run base ""
run tq2 "-DDC_EXP_NN_TQ2"
run tq2_wg3 "-DDC_EXP_NN_TQ2 -DDC_NN_MIN_WG=3"
run tq2_wg4 "-DDC_EXP_NN_TQ2 -DDC_NN_MIN_WG=4"
run tq4_wg3 "-DDC_NN_MIN_WG=3"
run base ""
