#!/bin/bash
# run kbench with one variant library on the GPU box: run_one.sh NAME [kbench args]
cd $GRAFT_REPO_ROOT
v=$1; shift
cp clustering_amd/lib/libdcdensity.so /tmp/lib_saved.so
cp clustering_amd/lib/variants/$v.so clustering_amd/lib/libdcdensity.so
timeout 200 python3 scratch/kbench.py "$@" 2>&1 | grep -v "amdgpu.ids"
cp /tmp/lib_saved.so clustering_amd/lib/libdcdensity.so
