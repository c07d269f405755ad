ulimit -c 0
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 2 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --lm-iters 10"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/st --output-format csv -- $B > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/st/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'inner' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d /tmp/pm --output-format csv -- $B > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pm k_inner $R/gpurun_out/inner_pmc_packed.json | grep -v "^ *\"launch\|Grid\|Workgroup_Size" 
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_TRANS --kernel-trace -d /tmp/pm2 --output-format csv -- $B > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pm2 k_inner $R/gpurun_out/inner_pmc_packed2.json | grep "SQ_"
