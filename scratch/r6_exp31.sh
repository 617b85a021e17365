#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R
DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_nnp.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "shared_operand_neighbour" 2>&1 | tail -2
for v in hd6 nnp hd6 nnp; do
  echo -n "$v: "; DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so timeout 900 python3 scratch/c5_bench.py --reps 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pop %.1f nn %.2f' % (d['pop_8_radii_ms'], d['nn_ms']))"
done
