ulimit -c 0
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 2 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --lm-iters 10"
(cd $R && timeout 400 python -m pytest tests/test_ba_solve_gpu.py tests/test_full_size_gpu.py tests/test_ba_inner_gpu.py tests/test_costmap_gpu.py -q -x 2>&1 | tail -2)
for old in 0 1; do
rm -rf /tmp/st; if [ $old = 1 ]; then export PXR_SCHUR_LDS=1; fi
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/st --output-format csv -- $B > /tmp/bench.json 2>/dev/null
python - <<PY
import csv, glob, json
for f in glob.glob('/tmp/st/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('k_schur', 'k_pair', 'k_scan')): print(r['Name'][:50], r['Calls'], r['AverageNs'])
d = json.loads(open('/tmp/bench.json').read().strip().splitlines()[-1])
print('old=$old lm', d['lm']['ms_per_iter'], d['lm']['final_cost'], 'no_inner', d['lm_no_inner']['ms_per_iter'], d['lm_no_inner']['final_cost'], 'setup', d['lm_no_inner']['setup_ms'])
PY
done
