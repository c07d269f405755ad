#!/bin/bash
# round 5: KA with the streamed line-search probe and the reworked register Cholesky -- parity tests, timing against PXR_KA_STREAM=0
set -u
O=gpurun_out/r5k
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ka_gpu.py tests/test_ka_unary_gpu.py tests/test_deterministic_gpu.py tests/test_full_size_gpu.py tests/test_zz_multi_rank_gpu.py tests/test_api_gpu.py tests/test_edge_cases_gpu.py -m gpu -q --maxfail=20 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
timeout 300 python tools/bench_ka.py > $O/bench_ka.json 2> $O/bench_ka.err
PXR_KA_STREAM=0 timeout 300 python tools/bench_ka.py > $O/bench_ka_nostream.json 2> $O/bench_ka_nostream.err
PXR_DETERMINISTIC=0 timeout 300 python tools/bench_ka.py > $O/bench_ka_nondet.json 2> $O/bench_ka_nondet.err
tail -15 $O/pytest.log; cat $O/steps.log
python -c "
import json
for f in ('bench_ka','bench_ka_nostream','bench_ka_nondet'):
    d=json.load(open('$O/%s.json'%f)); s=d['solve']; print(f, s['kernel_ms'], s['kernel_ms_min'], s['successful_steps'], s['lm_iterations_max'], repr(s['final_cost']), d['accuracy_px'])"
