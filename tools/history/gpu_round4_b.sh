#!/bin/bash
set -u
O=gpurun_out/r4b
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ba_inner_gpu.py tests/test_camera_models_ext.py tests/test_ba_solve_gpu.py -q > $O/pytest_inner.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
timeout 600 python tools/_time_inner.py > $O/time_inner.txt 2>&1
( cd /tmp && rm -rf /tmp/ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --no-telemetry > /dev/null 2> $GRAFT_REPO_ROOT/$O/traced.err ); find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
PXR_HIP_LIB=$GRAFT_REPO_ROOT/tools/debug/libpixsfm_hip_innerprof.so python bench.py --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --no-telemetry --steps 2 --warmup 1 --lm-iters 2 2>&1 | grep "inner gram" > $O/gram_profile.txt
