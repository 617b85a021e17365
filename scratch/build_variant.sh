#!/bin/bash
# developer build of the working tree (MFMA_STEPS, default "2": n_cols 5..10 only; EXTRA: extra compiler flags
# for the step translation units), kept as clustering_amd/lib/variants/$1.so
set -e
cd "$(dirname "$0")/../clustering_amd/csrc"
touch dc_mfma_step.hip dc_mfma.hip
make -j8 MFMA_STEPS="${MFMA_STEPS:-2}" CXXFLAGS_EXTRA="${EXTRA:-}" > /tmp/make_variant.log 2>&1 || { grep -E "error" -A6 /tmp/make_variant.log | head -30; exit 1; }
mkdir -p ../lib/variants
cp ../lib/libdcdensity.so ../lib/variants/$1.so
echo "built variant $1"
