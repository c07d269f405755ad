#!/bin/bash
# Full GPU suite + the bench line + kernel stats of the same command (round 4, after the Gram-matrix cache).
set -u
O=gpurun_out/r4o
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench exit $?" >> $O/bench_n1.err
( cd /tmp && rm -rf /tmp/ks && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-api-e2e --no-telemetry > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/traced.err )
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
tail -4 $O/pytest.log; tail -2 $O/bench_n1.err
