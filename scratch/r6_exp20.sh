#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_session.py -x -q -k "sweep_forms or device_list or duplicate_devices" 2>&1 | tail -3
for cfg in "1 16 4" "1 8 8" "1 64 2"; do set -- $cfg
  echo "fuzz COOP=$1 floor=$2 waves=$3"; DC_NN_COOP=$1 DC_SHARE_FLOOR=$2 DC_NN_COOP_WAVES=$3 timeout 900 python3 scratch/fuzz.py $(( $2 + 100 )) 120 2>&1 | tail -2
done
DC_NN_COOP=1 DC_SHARE_FLOOR=64 timeout 900 python3 scratch/fuzz.py 77 25 big 2>&1 | tail -2
