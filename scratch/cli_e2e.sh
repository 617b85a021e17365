#!/bin/bash
# end-to-end CLI run at C3 size: .npy in, populations / free energies / neighbours / -T screening out
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/e2e && cd /tmp/e2e
python3 - <<PY
import sys, numpy as np
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from clustering_amd.synth import gaussian_blobs
np.save("coords.npy", gaussian_blobs(1000000, 10))
PY
CLI=$GRAFT_REPO_ROOT/clustering_amd/bin/clustering
T0=$(date +%s%N); $CLI density -f coords.npy -r 0.2 -p pop -d fe -b nn > /dev/null 2>&1; T1=$(date +%s%N); echo "pop + fe + nn incl. file IO: $(( (T1 - T0) / 1000000 )) ms"
T0=$(date +%s%N); $CLI density -f coords.npy -r 0.2 -D fe -B nn -T 0.5 1.0 6.0 -o clust -v 2>&1 | tail -10; T1=$(date +%s%N); echo "screening scan (reads fe / nn back) incl. file IO: $(( (T1 - T0) / 1000000 )) ms"
ls -la clust.* | head -8
T0=$(date +%s%N); $CLI density -f coords.npy -p pop2 -d fe2 -b nn2 > /dev/null 2>&1; T1=$(date +%s%N); echo "no -r (pop(1.0) + nn for sigma2, pop(lumping radius) + nn; ONE upload): $(( (T1 - T0) / 1000000 )) ms"
T0=$(date +%s%N); $CLI density -f coords.npy -r 0.2 -p pop3 -d fe3 -b nn3 -T 0.5 1.0 6.0 -o clust3 > /dev/null 2>&1; T1=$(date +%s%N); echo "everything in one run (pop, fe, nn, forest, 6 thresholds): $(( (T1 - T0) / 1000000 )) ms"
