#!/bin/bash
# round 5, fourth GPU call: suite (both arithmetics of the oracle comparisons), the LM loop traced per kernel in deterministic
# and floating-point-atomics mode, bench
set -u
O=gpurun_out/r5d
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
for MODE in det nondet; do
  D=1; [ $MODE = nondet ] && D=0
  ( cd /tmp && rm -rf /tmp/k_$MODE && PXR_DETERMINISTIC=$D timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k_$MODE -- python $GRAFT_REPO_ROOT/tools/_lm_solve_once.py > $GRAFT_REPO_ROOT/$O/lm_$MODE.json 2> $GRAFT_REPO_ROOT/$O/lm_$MODE.err ); echo "trace $MODE rc=$?" >> $O/steps.log
  find /tmp/k_$MODE -name "*kernel_stats.csv" -exec cp {} $O/lm_${MODE}_kernel_stats.csv \;
done
timeout 900 python bench.py --detail-out $O/bench_detail.json > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/steps.log
tail -5 $O/pytest.log; cat $O/steps.log; wc -c $O/bench_n1.json
