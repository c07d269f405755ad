#!/bin/bash
# round 5, eighth GPU call (after the container was re-created and the earlier calls' outputs were lost): full suite, KA timing,
# LM loop timing, full bench, per-kernel statistics of the bench command, counter passes of the hot kernels
set -u
O=gpurun_out/r5h
mkdir -p $O
export TMPDIR=/tmp
COMMIT=$(cat .commit_id 2>/dev/null || echo unknown)
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
timeout 300 python tools/bench_ka.py > $O/bench_ka.json 2> $O/bench_ka.err
PXR_DETERMINISTIC=0 timeout 300 python tools/bench_ka.py > $O/bench_ka_nondet.json 2> $O/bench_ka_nondet.err
timeout 300 python tools/_lm_solve_once.py > $O/lm_spin.json 2> $O/lm_spin.err
timeout 900 python bench.py --detail-out $O/bench_detail.json > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/steps.log
( cd /tmp && rm -rf /tmp/kstats && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-api-e2e --detail-out $GRAFT_REPO_ROOT/$O/bench_traced_detail.json > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/bench_traced.err ); echo "kstats rc=$?" >> $O/steps.log
find /tmp/kstats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout 1200 tools/pmc_collect.sh $O/pmc $COMMIT; echo "pmc rc=$?" >> $O/steps.log
tail -3 $O/pytest.log; cat $O/steps.log; cat $O/lm_spin.json; wc -c $O/bench_n1.json
python -c "
import json
for f in ('bench_ka','bench_ka_nondet'):
    d=json.load(open('$O/%s.json'%f)); print(f, d['solve']['kernel_ms'], d['solve']['kernel_ms_min'], d['solve']['successful_steps'])"
