#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R; O=$R/gpurun_out
for rep in 1 2 3; do for v in nnA nnB; do
  DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so timeout 300 python3 bench.py --variant mfma32 --steps 2 --warmup 1 --cpu-sample 0 --no-full-sweep > $O/r6_exp22.json 2> $O/r6_exp22.err
  python3 -c "
import json;d=json.loads(open('$O/r6_exp22.json').read().strip().split('\n')[-1]);r=d['roofline_by_kernel'];print('$v pop %.1f ms %.4f' % (r['population_count']['launch_ms'], r['population_count']['frac']), 'nn %.1f ms %.4f' % (r['nearest_neighbor_search']['launch_ms'], r['nearest_neighbor_search']['frac']))"
done; done
