#!/bin/bash
# round 5: the drop-in calls after the index fast path (arena slots from the prefetched stacks' layout) -- API / cost-map / multi-rank
# tests, then the end-to-end timing with 8 (default) and 12 upload threads
set -u
O=gpurun_out/r5t
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_api_gpu.py tests/test_costmap_gpu.py tests/test_zz_multi_rank_gpu.py tests/test_localization_golden.py -m gpu -q --maxfail=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
timeout 300 python tools/bench_api_e2e.py --skip-ka > $O/e2e_default.json 2> $O/e2e_default.err; echo "e2e rc=$?" >> $O/steps.log
PXR_UPLOAD_THREADS=12 timeout 300 python tools/bench_api_e2e.py --skip-ka > $O/e2e_12.json 2> $O/e2e_12.err; echo "e2e12 rc=$?" >> $O/steps.log
tail -4 $O/pytest.log; cat $O/steps.log; tail -1 $O/e2e_default.json; tail -1 $O/e2e_12.json
