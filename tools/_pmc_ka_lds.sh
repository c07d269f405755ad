set -u
O=gpurun_out/r5j; mkdir -p $O; export TMPDIR=/tmp; ROOT=$PWD
for S in 1 0; do
rm -rf /tmp/pmc_lds$S
( cd /tmp && PXR_KA_STREAM=$S PXR_HIP_LIB=$ROOT/tools/debug/libpixsfm_hip_v2.so timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/pmc_lds$S -- python $ROOT/tools/_ka_iter_hist.py > /dev/null 2> $ROOT/$O/lds$S.stderr )
python tools/pmc_summary.py /tmp/pmc_lds$S ka_solve_kernel $O/lds_stream${S}_ka_solve_kernel.json > /dev/null 2>> $O/passes.log
python -c "
import json; d=json.load(open('$O/lds_stream${S}_ka_solve_kernel.json'))
for k,v in d.items(): print('stream=$S', {a: round(b) for a,b in v['mean'].items()})"
done
