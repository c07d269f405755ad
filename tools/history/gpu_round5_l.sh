#!/bin/bash
# round 5: KA after the register-Cholesky rework (shipped) and the streamed line-search probe (experiment build, -DPXR_KA_STREAM_PROBE)
set -u
O=gpurun_out/r5l
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ka_gpu.py tests/test_ka_unary_gpu.py tests/test_deterministic_gpu.py tests/test_full_size_gpu.py tests/test_zz_multi_rank_gpu.py tests/test_api_gpu.py tests/test_edge_cases_gpu.py -m gpu -q --maxfail=20 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
PXR_HIP_LIB=$PWD/tools/debug/libpixsfm_hip_stream.so timeout 900 python -m pytest tests/test_ka_gpu.py tests/test_deterministic_gpu.py tests/test_full_size_gpu.py -m gpu -q --maxfail=20 > $O/pytest_stream.log 2>&1; echo "pytest stream rc=$?" >> $O/steps.log
timeout 300 python tools/bench_ka.py > $O/bench_ka.json 2> $O/bench_ka.err
PXR_DETERMINISTIC=0 timeout 300 python tools/bench_ka.py > $O/bench_ka_nondet.json 2> $O/bench_ka_nondet.err
PXR_HIP_LIB=$PWD/tools/debug/libpixsfm_hip_stream.so timeout 300 python tools/bench_ka.py > $O/bench_ka_stream.json 2> $O/bench_ka_stream.err
tail -3 $O/pytest.log; tail -3 $O/pytest_stream.log; cat $O/steps.log
python -c "
import json
for f in ('bench_ka','bench_ka_nondet','bench_ka_stream'):
    d=json.load(open('$O/%s.json'%f)); s=d['solve']; print(f, s['kernel_ms'], s['kernel_ms_min'], s['successful_steps'], s['lm_iterations_max'], repr(s['final_cost']))"
