#!/bin/bash
set -u
O=gpurun_out/r4l
mkdir -p $O
for v in base nomfma noload; do
  if [ $v = base ]; then unset PXR_HIP_LIB; else export PXR_HIP_LIB=$GRAFT_REPO_ROOT/tools/debug/libpixsfm_hip_$v.so; fi
  timeout 600 python bench.py --steps 2 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --no-telemetry --lm-iters 0 > $O/bench_$v.json 2> $O/bench_$v.err
  python - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r4l/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['gram_evaluation'])
PY
done
