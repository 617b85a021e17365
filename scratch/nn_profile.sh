#!/bin/bash
# builds a profiling copy of the library in /tmp and runs scratch/nn_profile.py against it
set -e
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof && mkdir -p /tmp/prof && cp -r clustering_amd include scratch /tmp/prof/
cd /tmp/prof/clustering_amd/csrc && touch dc_mfma_kernels.hpp && make -j16 MFMA_STEPS="2" CXXFLAGS_EXTRA=-DDC_NN_PROFILE ../lib/libdcdensity.so > /tmp/prof/build.log 2>&1 || { tail -20 /tmp/prof/build.log; exit 1; }
cd /tmp/prof && python3 scratch/nn_profile.py
