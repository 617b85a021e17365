#!/bin/bash
# CLI with the reference's ASCII input at C3 size: where the end-to-end time goes
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/e2e && cd /tmp/e2e
python3 - <<PY
import sys, time, numpy as np
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from clustering_amd.synth import gaussian_blobs
c = gaussian_blobs(1000000, 10)
t0 = time.time(); np.savetxt("coords.txt", c, fmt="%.9g"); print(f"numpy savetxt {time.time()-t0:.1f} s")
PY
ls -la coords.txt
CLI=$GRAFT_REPO_ROOT/clustering_amd/bin/clustering
T0=$(date +%s%N); $CLI density -f coords.txt -r 0.2 -p pop -d fe -b nn > /dev/null 2>&1; T1=$(date +%s%N); echo "ASCII in, pop + fe + nn out: $(( (T1 - T0) / 1000000 )) ms"
