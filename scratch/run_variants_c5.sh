#!/bin/bash
# A/B of variant libraries on the C5 segment sweep (pop only) and 400k x 30
cd $GRAFT_REPO_ROOT
cp clustering_amd/lib/libdcdensity.so /tmp/lib_saved.so
for v in "$@"; do
  cp clustering_amd/lib/variants/$v.so clustering_amd/lib/libdcdensity.so
  for sh in 1 0; do
  echo "== $v DC_POP_SHARED=$sh"
  DC_POP_SHARED=$sh timeout 600 python3 - <<'PY' 2>&1 | grep -v amdgpu
import sys, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
c = torch.from_numpy(gaussian_blobs(5_000_000, 30)).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for r in (0.35, 0.6):
    ts = []
    for rep in range(3):
        ev[0].record(); dens.calculate_populations_segment(c, [r], 3, 8); ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    print(f"  C5 segment 3/8 r={r}: {min(ts):.1f} ms   tiles {dens.evaluated_tiles(c.device)[0]}")
PY
  done
done
cp /tmp/lib_saved.so clustering_amd/lib/libdcdensity.so
