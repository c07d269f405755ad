#!/bin/bash
# Counter passes of the hot kernels at HEAD (after the two-link Schur chain, the per-point inner cost and the Gram-matrix cache).
set -u
O=gpurun_out/r4p
mkdir -p $O
COMMIT=${1:-unknown}
timeout 1500 tools/pmc_collect.sh $O/pmc $COMMIT
python tools/pmc_merge.py $O/pmc $O/hot_kernels_pmc.json $COMMIT
cat $O/pmc/passes.log | tail -8
