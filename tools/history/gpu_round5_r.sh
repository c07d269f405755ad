#!/bin/bash
# round 5: the iterative solver in deterministic mode (ordered partial sums) -- its tests, the multi-rank tests, the Aachen-shaped bench
set -u
O=gpurun_out/r5r
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_deterministic_gpu.py tests/test_ba_pcg_gpu.py tests/test_zz_multi_rank_gpu.py tests/test_ba_solve_gpu.py tests/test_ba_inner_gpu.py -m gpu -q --maxfail=20 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
timeout 900 python bench.py --preset aachen --no-ka --no-cpu-baseline --no-api-e2e --no-costmap --detail-out $O/aachen_detail.json > $O/aachen.json 2> $O/aachen.err; echo "aachen rc=$?" >> $O/steps.log
PXR_DETERMINISTIC=0 timeout 900 python bench.py --preset aachen --no-ka --no-cpu-baseline --no-api-e2e --no-costmap --detail-out $O/aachen_fp_detail.json > $O/aachen_fp.json 2> $O/aachen_fp.err; echo "aachen fp rc=$?" >> $O/steps.log
tail -25 $O/pytest.log; cat $O/steps.log
python -c "
import json
for f in ('aachen','aachen_fp'):
    d=json.load(open('$O/%s.json'%f)); print(f, d['value'], {k: (d[k].get('ms_per_iter'), d[k].get('linear_iterations'), d[k].get('final_cost')) for k in ('lm','lm_no_inner') if k in d})"
