#!/bin/bash
# round 5, seventh GPU call: KA with its guard state in LDS, polled vs blocking synchronisation, then the evidence files:
# per-kernel statistics of the bench command and the counter passes of the hot kernels
set -u
O=gpurun_out/r5g
mkdir -p $O
export TMPDIR=/tmp
COMMIT=$(git rev-parse --short HEAD 2>/dev/null || cat .commit_id 2>/dev/null || echo unknown)
timeout 900 python -m pytest tests/test_ka_gpu.py tests/test_ka_unary_gpu.py tests/test_deterministic_gpu.py tests/test_zz_multi_rank_gpu.py tests/test_api_gpu.py tests/test_full_size_gpu.py -m gpu -q --maxfail=20 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
timeout 300 python tools/bench_ka.py > $O/bench_ka.json 2> $O/bench_ka.err
PXR_DETERMINISTIC=0 timeout 300 python tools/bench_ka.py > $O/bench_ka_nondet.json 2> $O/bench_ka_nondet.err
timeout 300 python tools/_lm_solve_once.py > $O/lm_spin.json 2> $O/lm_spin.err
PXR_BLOCKING_WAIT=1 timeout 300 python tools/_lm_solve_once.py > $O/lm_blocking.json 2> $O/lm_blocking.err
( cd /tmp && rm -rf /tmp/kstats && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-api-e2e --detail-out $GRAFT_REPO_ROOT/$O/bench_traced_detail.json > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/bench_traced.err ); echo "kstats rc=$?" >> $O/steps.log
find /tmp/kstats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout 1500 tools/pmc_collect.sh $O/pmc $COMMIT; echo "pmc rc=$?" >> $O/steps.log
tail -3 $O/pytest.log; cat $O/steps.log; cat $O/lm_spin.json $O/lm_blocking.json; python -c "
import json
for f in ('bench_ka','bench_ka_nondet'):
    d=json.load(open('$O/%s.json'%f)); print(f, d['solve']['kernel_ms'], d['solve']['kernel_ms_min'], d['solve']['successful_steps'])"
