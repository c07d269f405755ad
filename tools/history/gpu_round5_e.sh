#!/bin/bash
# round 5, fifth GPU call: tests after going back to 62-bit grids, half cost maps with fp32 partial sums, what the deterministic
# KA instantiation pays for (probe builds), cost-map extraction timing
set -u
O=gpurun_out/r5e
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ka_gpu.py tests/test_ka_unary_gpu.py tests/test_camera_models_ext.py tests/test_ba_solve_gpu.py tests/test_api_gpu.py tests/test_deterministic_gpu.py tests/test_costmap_gpu.py tests/test_gram_cache_gpu.py tests/test_ba_inner_gpu.py tests/test_zz_multi_rank_gpu.py -m gpu -q --maxfail=40 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/steps.log
for V in default ka_noguard ka_fpatom ka_both; do
  L=""; [ $V != default ] && L=$PWD/tools/debug/libpixsfm_hip_$V.so
  PXR_HIP_LIB=$L timeout 300 python tools/bench_ka.py > $O/bench_ka_$V.json 2> $O/bench_ka_$V.err; echo "bench_ka $V rc=$?" >> $O/steps.log
done
PXR_DETERMINISTIC=0 timeout 300 python tools/bench_ka.py > $O/bench_ka_nondet.json 2> $O/bench_ka_nondet.err
timeout 600 python bench.py --lm-iters 0 --no-ka --no-cpu-baseline --no-api-e2e --detail-out $O/bench_costmap_detail.json > $O/bench_costmap.json 2> $O/bench_costmap.err; echo "bench costmap rc=$?" >> $O/steps.log
timeout 600 python tools/fuzz_costmap_vs_reference.py > $O/fuzz_costmap.txt 2>&1; echo "fuzz costmap rc=$?" >> $O/steps.log
tail -5 $O/pytest.log; cat $O/steps.log
