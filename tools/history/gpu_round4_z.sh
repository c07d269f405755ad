#!/bin/bash
# round 4, first GPU call: suite, bench, two-rank artefact, kernel statistics, phase times, counters
set -u
O=gpurun_out/r4z
mkdir -p $O
export TMPDIR=/tmp
COMMIT=$(cat .commit_id 2>/dev/null || echo unknown)
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/steps.log
PXR_BENCH_ONE_DEVICE=1 PXR_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "bench2 rc=$?" >> $O/steps.log
PXR_PHASE_TIMING=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ka --no-costmap --no-api-e2e --no-telemetry > /dev/null 2> $O/lm_phases.err; echo "phases rc=$?" >> $O/steps.log
( cd /tmp && rm -rf /tmp/kstats && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-api-e2e > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/bench_traced.err ); echo "kstats rc=$?" >> $O/steps.log
find /tmp/kstats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout 1500 tools/pmc_collect.sh $O/pmc $COMMIT; echo "pmc rc=$?" >> $O/steps.log
ls /sys/class/drm/ > $O/sysfs.txt 2>&1; for c in /sys/class/drm/card*/device; do echo $c; ls $c | head -80; cat $c/pp_dpm_sclk 2>&1 | head; ls $c/hwmon/*/ 2>&1; done >> $O/sysfs.txt 2>&1
nproc > $O/host.txt; lscpu | head -30 >> $O/host.txt; free -g >> $O/host.txt
