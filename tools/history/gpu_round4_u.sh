#!/bin/bash
set -u
O=gpurun_out/r4u
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ba_solve_gpu.py tests/test_deterministic_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
( cd /tmp && rm -rf /tmp/ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --no-telemetry > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/traced.err ); find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
tail -2 $O/pytest.log
grep -E "k_img|k_jac|k_point\(|k_schur|k_chol_step|k_backsub" $O/kernel_stats.csv | awk -F'",' '{print substr($1,1,40), $2}'
