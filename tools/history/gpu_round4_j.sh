#!/bin/bash
set -u
O=gpurun_out/r4j
mkdir -p $O
export TMPDIR=/tmp
python tools/_probe_gram.py 2>&1 | tail -12 > $O/probe.txt
timeout 900 python -m pytest tests/test_ba_inner_gpu.py tests/test_ba_solve_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
( cd /tmp && rm -rf /tmp/ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --no-telemetry > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/traced.err )
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout 600 python bench.py --steps 5 --warmup 2 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e > $O/bench.json 2> $O/bench.err
cat $O/probe.txt; tail -3 $O/pytest.log
grep -E "gram|ba_eval_kernel|k_inner" $O/kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2}'
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4j/bench.json').read().strip().splitlines()[-1])
print('lm', d['lm']['ms_per_iter'], 'no_inner', d['lm_no_inner']['ms_per_iter'])
print(json.dumps(d['lm']['gram_cache'], indent=0))
PY
