#!/bin/bash
# round 5: what an LM iteration is made of at HEAD (per-kernel statistics of two lm + two lm_no_inner solves of 10 iterations)
set -u
O=gpurun_out/r5n
mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/k_lm && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k_lm -- python $GRAFT_REPO_ROOT/tools/_lm_solve_once.py > $GRAFT_REPO_ROOT/$O/lm.json 2> $GRAFT_REPO_ROOT/$O/lm.err )
find /tmp/k_lm -name "*kernel_stats.csv" -exec cp {} $O/lm_kernel_stats.csv \;
find /tmp/k_lm -name "*kernel_trace.csv" -exec cp {} $O/lm_kernel_trace.csv \;
cat $O/lm.json
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5n/lm_kernel_stats.csv')))
tot=0
for r in rows[:40]:
    n=r['Name'][:70]; print('%-70s %6s %10.1f us avg %8.3f ms total' % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
