#!/bin/bash
set -u
O=gpurun_out/r4g
mkdir -p $O
export TMPDIR=/tmp
for v in base s164 s168 s176; do
  if [ $v = base ]; then unset PXR_HIP_LIB; else export PXR_HIP_LIB=$GRAFT_REPO_ROOT/tools/debug/libpixsfm_hip_gram_$v.so; fi
  ( cd /tmp && rm -rf /tmp/ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --no-telemetry --lm-iters 4 > /dev/null 2> $GRAFT_REPO_ROOT/$O/traced.err ); find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$v.csv \;
  echo "$v $(grep k_inner_gram $O/kernel_stats_$v.csv | cut -d, -f2-4)" >> $O/summary.txt
done
unset PXR_HIP_LIB
