#!/bin/bash
set -u
O=gpurun_out/r4n
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ba_inner_gpu.py tests/test_ba_solve_gpu.py tests/test_deterministic_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
( cd /tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --no-telemetry > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/traced.err )
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = []
for r in rows:
    n = r['Kernel_Name']
    if 'k_inner_gram' in n or 'k_gram_build' in n or 'k_gram_eval' in n:
        out.append('%-22s %8.1f us' % (n.split('(')[0].split('<')[0][-22:], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
open('gpurun_out/r4n/gram_calls.txt', 'w').write('\n'.join(out) + '\n')
inner = [l for l in out if 'gram_eval' not in l and 'GramArgs' not in l]
print('\n'.join(inner))
PY
timeout 600 python bench.py --steps 5 --warmup 2 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4n/bench.json').read().strip().splitlines()[-1])
print('lm', d['lm']['ms_per_iter'], 'no_inner', d['lm_no_inner']['ms_per_iter'], 'det', d['lm']['deterministic']['ms_per_iter'])
print({k: (v['ms_per_iter'], v['successful']) for k, v in d['lm']['gram_cache'].items()})
PY
