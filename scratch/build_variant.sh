#!/bin/bash
# developer build (n_cols 5..10 only) of the working tree, kept as clustering_amd/lib/variants/$1.so
set -e
cd "$(dirname "$0")/../clustering_amd/csrc"
make -j8 MFMA_STEPS="${MFMA_STEPS:-2}" > /tmp/make_variant.log 2>&1 || { grep -E "error" -A6 /tmp/make_variant.log | head -30; exit 1; }
mkdir -p ../lib/variants
cp ../lib/libdcdensity.so ../lib/variants/$1.so
echo "built variant $1"
