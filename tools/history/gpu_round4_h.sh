#!/bin/bash
# Schur kernel with the restructured load chain: solver parity tests, then kernel stats of the LM loop.
set -u
O=gpurun_out/r4h
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ba_solve_gpu.py tests/test_full_size_gpu.py tests/test_deterministic_gpu.py tests/test_zz_multi_rank_gpu.py tests/test_ba_inner_gpu.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
( cd /tmp && rm -rf /tmp/ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --no-telemetry > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/traced.err )
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout 600 python bench.py --steps 5 --warmup 2 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest.log; head -12 $O/kernel_stats.csv
