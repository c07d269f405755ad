ulimit -c 0
timeout 200 python -m pytest tests/test_ba_inner_gpu.py tests/test_costmap_gpu.py -q -x 2>&1 | tail -3
timeout 200 python bench.py --no-api-e2e 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d['lm'], d.get('lm_no_inner'))
"
