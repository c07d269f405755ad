#!/bin/bash
set -u
O=gpurun_out/r4c
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_real_images_gpu.py tests/test_ba_inner_gpu.py tests/test_edge_cases_gpu.py -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
timeout 600 python tools/bench_real_images.py > $O/real_images.json 2> $O/real_images.err; echo "real rc=$?" >> $O/steps.log
