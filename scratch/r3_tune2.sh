#!/bin/bash
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd $R
for v in 128 64 96 192 256 384; do echo "NN_CELL_FRAMES $v: $(DC_NN_CELL_FRAMES=$v python scratch/nn_diag.py | tail -1)"; done
for v in 64 32 48 96 128 192; do echo "POP_CELL_FRAMES $v: $(DC_POP_CELL_FRAMES=$v python scratch/nn_diag.py | tail -1)"; done
