#!/bin/bash
# A/B on one GPU box with an arbitrary command: run_variants_cmd.sh "cmd" V1 V2 ...  (two interleaved rounds)
cd $GRAFT_REPO_ROOT
CMD="$1"; shift
cp clustering_amd/lib/libdcdensity.so /tmp/lib_saved.so
for round in 1 2; do
  for v in "$@"; do
    cp clustering_amd/lib/variants/$v.so clustering_amd/lib/libdcdensity.so
    echo "== $v (round $round)"; timeout 300 bash -c "$CMD" 2>&1 | grep -v "amdgpu.ids\|raw counter\|evaluated fraction"
  done
done
cp /tmp/lib_saved.so clustering_amd/lib/libdcdensity.so
