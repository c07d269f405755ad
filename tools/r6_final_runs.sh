#!/bin/bash
# round 6, the runs behind profiles/r6_*: full GPU suite, the bench line, the bench with the forced one-rank collective,
# the counter passes (tools/pmc_collect.sh), rocprofv3 --kernel-trace --stats of the bench command and of LM solves.
#   bash tools/r6_final_runs.sh <out dir> <commit>
OUT=${1:-gpurun_out/r6_final2}
COMMIT=${2:-unknown}          # the commit the snapshot was taken at (the GPU box has no .git): stamped into the counter files
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$R/$OUT"
cd "$R"
timeout 2400 python -m pytest tests -m gpu -q > "$OUT/gpu_tests.txt" 2>&1; tail -3 "$OUT/gpu_tests.txt"
timeout 900 bash tools/pmc_collect.sh "$OUT/pmc" "$COMMIT"; tail -4 "$OUT/pmc/passes.log"
timeout 900 python bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"; cp bench_detail.json "$OUT/bench_detail.json" 2>/dev/null
PXR_BENCH_FORCE_DIST=1 timeout 900 python bench.py --no-cpu-baseline --no-api-e2e --no-costmap --no-ka-points > "$OUT/bench_forced_collective.json" 2> "$OUT/bench_forced.err"
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/ks && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python "$R/bench.py" --no-cpu-baseline --no-api-e2e --no-telemetry --no-ka-points > "$R/$OUT/bench_under_rocprof.json" 2> /dev/null )
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) "$OUT/bench_kernel_stats.csv"
( cd /tmp && rm -rf /tmp/tr && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python "$R/tools/_lm_setup_probe.py" > /dev/null 2>&1 )
python tools/lm_timeline.py /tmp/tr > "$OUT/lm_timeline.txt" 2>&1
python tools/_lm_setup_timeline.py /tmp/tr > "$OUT/lm_setup_timeline.txt" 2>&1
tail -c 400 "$OUT/bench_n1.json"
