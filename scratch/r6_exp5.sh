#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R; O=$R/gpurun_out
for v in pub nopub pubnn pub nopub; do
  DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so timeout 300 python3 bench.py --variant mfma32 --steps 2 --warmup 1 --cpu-sample 0 --no-full-sweep > $O/r6_exp5_$v.json 2> $O/r6_exp5_$v.err
  python3 -c "
import json;d=json.loads(open('$O/r6_exp5_$v.json').read().strip().split('\n')[-1]);r=d['roofline_by_kernel'];print('$v', 'pop %.1f ms %.4f' % (r['population_count']['launch_ms'], r['population_count']['frac']), 'nn %.1f ms %.4f' % (r['nearest_neighbor_search']['launch_ms'], r['nearest_neighbor_search']['frac']), d['check']['mean_pop_r0'], d['check']['sigma2'])"
done
