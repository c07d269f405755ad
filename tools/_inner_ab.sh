ulimit -c 0
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 2 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --lm-iters 10"
(cd $R && timeout 200 python -m pytest tests/test_ba_inner_gpu.py -q -x 2>&1 | tail -2)
for rep in 1 2; do
rm -rf /tmp/st; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/st --output-format csv -- $B > /tmp/bench.json 2>/dev/null
python - <<PY
import csv, glob, json
for f in glob.glob('/tmp/st/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'inner' in r['Name']: print(r['Name'][:40], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
d = json.loads(open('/tmp/bench.json').read().strip().splitlines()[-1])
print(d['lm']['ms_per_iter'], d['lm']['final_cost'], d['lm']['successful'], d['lm_no_inner']['ms_per_iter'])
PY
done
