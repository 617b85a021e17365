#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R
for v in "$@"; do
  echo -n "$v: "; DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so timeout 600 python3 bench.py --steps 12 --warmup 3 --cpu-sample 0 --no-full-sweep 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phases_ms']; print('step %.3f pop_k %.3f nn_k %.3f prep %.3f+%.3f' % (d['ms_per_step'], p['pop_kernel'], p['nn_kernel'], p['pop_prep'], p['nn_prep']), d['check']['mean_pop_r0'], d['check']['sigma2'])"
done
