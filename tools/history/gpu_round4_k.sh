#!/bin/bash
set -u
O=gpurun_out/r4k
mkdir -p $O
timeout 600 python bench.py --steps 5 --warmup 2 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e > $O/bench.json 2> $O/bench.err
tail -5 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4k/bench.json').read().strip().splitlines()[-1])
print(json.dumps(d['gram_evaluation'], indent=0))
print('telemetry', d.get('telemetry',{}).get('sclk_mhz'), d.get('telemetry',{}).get('power_w'))
print('lm', d['lm']['ms_per_iter'], 'no_inner', d['lm_no_inner']['ms_per_iter'])
print(json.dumps(d['lm']['gram_cache'], indent=0))
PY
