#!/bin/bash
# A/B of variant libraries on what one rank of C5 runs (8-radius pop, single-radius full sweep, nn), alternating twice
cd $GRAFT_REPO_ROOT
cp clustering_amd/lib/libdcdensity.so /tmp/lib_saved.so
for rep in 1 2; do
for v in "$@"; do
  cp clustering_amd/lib/variants/$v.so clustering_amd/lib/libdcdensity.so
  echo "== $v (round $rep)"
  timeout 900 python3 scratch/c5_bench.py --reps 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(d[k],1) for k in ('pop_8_radii_ms','nn_ms','full_single_radius_sweep_all_rows_ms')})"
done
done
cp /tmp/lib_saved.so clustering_amd/lib/libdcdensity.so
