#!/bin/bash
set -u
O=gpurun_out/r4t
mkdir -p $O
export TMPDIR=/tmp
for v in base gpad2 gpad4 gpad6; do
  if [ $v = base ]; then unset PXR_HIP_LIB; else export PXR_HIP_LIB=$GRAFT_REPO_ROOT/tools/debug/libpixsfm_hip_$v.so; fi
  ( cd /tmp && rm -rf /tmp/ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --no-telemetry > /dev/null 2> $GRAFT_REPO_ROOT/$O/traced.err ); find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$v.csv \;
  echo "$v $(grep k_inner_gram_packed $O/kernel_stats_$v.csv | awk -F'",' '{print $2}')"
done
