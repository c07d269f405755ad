#!/bin/bash
set -u
O=gpurun_out/r4d
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_deterministic_gpu.py tests/test_zz_multi_rank_gpu.py tests/test_ba_solve_gpu.py tests/test_ka_gpu.py tests/test_full_size_gpu.py -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
timeout 900 python bench.py --no-api-e2e --no-costmap > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/steps.log
