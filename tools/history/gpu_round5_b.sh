#!/bin/bash
# round 5, second GPU call: the tests that failed in r5a after the fixes (U kept in fixed point, adaptive KA grid), KA timing
set -u
O=gpurun_out/r5b
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_deterministic_gpu.py tests/test_zz_multi_rank_gpu.py tests/test_ka_gpu.py tests/test_ka_unary_gpu.py tests/test_full_size_gpu.py tests/test_api_gpu.py tests/test_ba_solve_gpu.py -m gpu -q --maxfail=25 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
timeout 300 python tools/bench_ka.py > $O/bench_ka.json 2> $O/bench_ka.err; echo "bench_ka rc=$?" >> $O/steps.log
PXR_DETERMINISTIC=0 timeout 300 python tools/bench_ka.py > $O/bench_ka_nondet.json 2> $O/bench_ka_nondet.err; echo "bench_ka nondet rc=$?" >> $O/steps.log
tail -5 $O/pytest.log; cat $O/steps.log
