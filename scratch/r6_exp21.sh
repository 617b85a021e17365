#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R; O=$R/gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "fp32_mfma" 2>&1 | tail -2
for rep in 1 2; do
  timeout 300 python3 bench.py --variant mfma32 --steps 2 --warmup 1 --cpu-sample 0 --no-full-sweep > $O/r6_exp21.json 2> $O/r6_exp21.err
  python3 -c "
import json;d=json.loads(open('$O/r6_exp21.json').read().strip().split('\n')[-1]);r=d['roofline_by_kernel'];p=d['phases_ms'];print('pop %.1f ms %.4f' % (r['population_count']['launch_ms'], r['population_count']['frac']), 'nn %.1f ms %.4f' % (r['nearest_neighbor_search']['launch_ms'], r['nearest_neighbor_search']['frac']), 'prep %.2f %.2f' % (p['pop_prep'], p['nn_prep']), d['check']['mean_pop_r0'], d['check']['sigma2'])"
done
for ch in 0 1 3 16; do DC_MFMA32_CHUNKS=$ch timeout 600 python3 scratch/fuzz32.py $((ch+40)) 150 2>&1 | tail -1; done
