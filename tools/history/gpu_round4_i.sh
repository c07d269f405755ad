#!/bin/bash
# Gram-matrix cache of the LM loop: kernel stats + bench lines.
set -u
O=gpurun_out/r4i
mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --no-telemetry > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/traced.err )
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout 600 python bench.py --steps 5 --warmup 2 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e > $O/bench.json 2> $O/bench.err
grep -E "gram|ba_eval_kernel|k_inner" $O/kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2}'
tail -c 1500 $O/bench.json
