#!/bin/bash
set -u
O=gpurun_out/r4f
mkdir -p $O
export TMPDIR=/tmp
for mode in default det; do
  if [ $mode = det ]; then export PXR_DETERMINISTIC=1; else unset PXR_DETERMINISTIC; fi
  ( cd /tmp && rm -rf /tmp/ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --no-telemetry > $GRAFT_REPO_ROOT/$O/bench_$mode.json 2> $GRAFT_REPO_ROOT/$O/traced_$mode.err ); find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$mode.csv \;
done
unset PXR_DETERMINISTIC
