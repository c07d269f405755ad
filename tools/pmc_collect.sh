#!/bin/bash
# Counter passes of the hot kernels: separate rocprofv3 --pmc runs of the SAME bench command (gpurun
# refuses mixing trace domains with --pmc; FETCH_SIZE and WRITE_SIZE do not fit one pass), summarised per kernel into
# $OUT/*.json by tools/pmc_summary.py / tools/pmc_traffic.py.  Usage: tools/pmc_collect.sh <out dir> [commit]
set -u
OUT=${1:-gpurun_out/pmc}
COMMIT=${2:-unknown}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CMD="python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-api-e2e --no-costmap --no-telemetry --no-ka-points"
CS=$ROOT/pixel-perfect-sfm_amd/csrc
KERNELS="ba_eval_kernel ka_solve_kernel k_schur_lds k_inner_gram k_gram_build k_gram_eval k_jac k_img k_point k_chol_step"
pass() {   # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  ( cd /tmp && rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -- $CMD > "$ROOT/$OUT/$name.stdout" 2> "$ROOT/$OUT/$name.stderr" )
  echo "pass $name rc=$?" >> "$ROOT/$OUT/passes.log"
  for k in $KERNELS; do
    python "$ROOT/tools/pmc_summary.py" /tmp/pmc_$name $k "$ROOT/$OUT/${name}_$k.json" > /dev/null 2>> "$ROOT/$OUT/passes.log" || true
  done
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU TCC_HIT_sum TCC_MISS_sum
# roofline.traffic of the dominant kernel (with-Jacobian instantiation), stamped with the commit it was measured at
python "$ROOT/tools/pmc_traffic.py" /tmp/pmc_fetch /tmp/pmc_write ba_eval_kernelIDF16_Li128ELb1ELb0 "$ROOT/$OUT/ba_eval_pmc.json" "$CMD" "$COMMIT" \
  $CS/pxr_ba_eval.hip $CS/pxr_interp.h $CS/pxr_device.h > /dev/null 2>> "$ROOT/$OUT/passes.log" || true
PMC_PER_KERNEL=ka_order_kernel python "$ROOT/tools/pmc_traffic.py" /tmp/pmc_fetch /tmp/pmc_write ka_solve_kernel "$ROOT/$OUT/ka_solve_kernel_traffic.json" "$CMD" "$COMMIT" \
  $CS/pxr_ka.hip $CS/pxr_interp.h $CS/pxr_device.h > /dev/null 2>> "$ROOT/$OUT/passes.log" || true
for k in k_schur_lds k_inner_gram k_gram_build k_gram_eval; do
  python "$ROOT/tools/pmc_traffic.py" /tmp/pmc_fetch /tmp/pmc_write $k "$ROOT/$OUT/${k}_traffic.json" "$CMD" "$COMMIT" > /dev/null 2>> "$ROOT/$OUT/passes.log" || true
done
rm -f "$ROOT/$OUT"/*.stdout
