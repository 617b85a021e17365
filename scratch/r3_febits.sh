#!/bin/bash
# What the free-energy order inside the cells of the neighbour sweep's ordering is worth (C3 data, one box).
cd $GRAFT_REPO_ROOT
for cfg in "" "DC_NN_FE_BITS=0" "DC_NN_FE_BITS=4" "DC_NN_FE_BITS=0 DC_NN_CELL_FRAMES=64" "DC_NN_CELL_FRAMES=64" "DC_NN_FE_BITS=0 DC_NN_CELL_FRAMES=32"; do
  echo "== $cfg"
  env $cfg timeout 300 python3 scratch/spread_bench.py 1 2>&1 | tail -1 | cut -c1-330
done
