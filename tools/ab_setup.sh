#!/bin/bash
# A/B measurement on one GPU box: exports commit $1 (default HEAD) into ab_old/ and builds it there, so that
# `python ab_old/bench.py` and `python bench.py` can run back to back in the same gpurun call (boxes differ by
# several percent, so runs from different calls do not compare).
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}
rm -rf ab_old && mkdir ab_old
git archive "$rev" | tar -x -C ab_old
(cd ab_old && python -m unet_amd.build > build.log 2>&1) || { tail -20 ab_old/build.log; exit 1; }
rm -rf ab_old/tests/golden ab_old/profiles ab_old/build
echo "ab_old/ = $(git rev-parse --short "$rev")"
