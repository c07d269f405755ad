ulimit -c 0
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 2 --warmup 1 --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --lm-iters 10"
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d /tmp/pm --output-format csv -- $B > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pm k_inner $R/gpurun_out/inner_pmc_tcc.json | grep "TCC\|FETCH"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pm2 --output-format csv -- $B > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pm2 k_inner $R/gpurun_out/inner_pmc_fetch.json | grep "TCC\|FETCH"
timeout 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace -d /tmp/pm3 --output-format csv -- $B > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pm3 k_inner $R/gpurun_out/inner_pmc_wait.json | grep "SQ_"
