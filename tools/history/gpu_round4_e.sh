#!/bin/bash
set -u
O=gpurun_out/r4e
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_api_gpu.py tests/test_deterministic_gpu.py tests/test_pixsfm_shim.py -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
timeout 900 python tools/bench_api_e2e.py > $O/api_e2e.json 2> $O/api_e2e.err; echo "e2e rc=$?" >> $O/steps.log
( cd /tmp && rm -rf /tmp/ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --no-telemetry > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/traced.err ); find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
