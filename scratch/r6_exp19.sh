#!/bin/bash
# row norms of the survivor tiles through a wave-private LDS ring (NormRing) in pop_pruned_kernel: A/B against HEAD
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R; O=$R/gpurun_out
for rep in 1 2; do for v in head ring; do
  DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so timeout 300 python3 bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-full-sweep --no-fp32-instance > $O/r6_exp19.json 2> $O/r6_exp19.err
  python3 -c "
import json;d=json.loads(open('$O/r6_exp19.json').read().strip().split('\n')[-1]);p=d['phases_ms'];print('$v', 'step %.2f' % d['ms_per_step'], 'pop_kernel %.2f nn_kernel %.2f pop_prep %.2f nn_prep %.2f' % (p['pop_kernel'],p['nn_kernel'],p['pop_prep'],p['nn_prep']), d['check']['mean_pop_r0'], d['check']['sigma2'])"
  DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so timeout 300 python3 bench.py --n-rows 100000 --radii 0.1 0.2 0.3 --no-nn --steps 20 --warmup 3 --cpu-sample 0 --no-full-sweep > $O/r6_exp19.json 2> $O/r6_exp19.err
  python3 -c "
import json;d=json.loads(open('$O/r6_exp19.json').read().strip().split('\n')[-1]);print('   C2 $v step %.3f' % d['ms_per_step'])"
done; done
DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_ring.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "pruned or segment" 2>&1 | tail -2
