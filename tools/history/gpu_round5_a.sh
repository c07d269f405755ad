#!/bin/bash
# round 5, first GPU call: suite (deterministic default, integer all-reduce, overflow guards), bench with the compact line
set -u
O=gpurun_out/r5a
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
timeout 900 python bench.py --detail-out $O/bench_detail.json > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/steps.log
tail -5 $O/pytest.log; cat $O/steps.log; wc -c $O/bench_n1.json
