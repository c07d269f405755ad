#!/bin/bash
# round 5, sixth GPU call: full suite, polled vs blocking host synchronisation of the LM loop, KA timing, full bench
set -u
O=gpurun_out/r5f
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
timeout 300 python tools/_lm_solve_once.py > $O/lm_spin.json 2> $O/lm_spin.err
PXR_BLOCKING_WAIT=1 timeout 300 python tools/_lm_solve_once.py > $O/lm_blocking.json 2> $O/lm_blocking.err
timeout 300 python tools/bench_ka.py > $O/bench_ka.json 2> $O/bench_ka.err
timeout 900 python bench.py --detail-out $O/bench_detail.json > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/steps.log
tail -5 $O/pytest.log; cat $O/steps.log; cat $O/lm_spin.json $O/lm_blocking.json; wc -c $O/bench_n1.json
