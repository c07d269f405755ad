#!/bin/bash
# round 5: instruction-cache and wait counters of the KA solve kernel (196 KB of code against a 64 KB instruction cache per CU pair)
set -u
O=gpurun_out/r5j
mkdir -p $O
export TMPDIR=/tmp
ROOT=$PWD
pass() { name=$1; shift
  rm -rf /tmp/pmc_$name
  ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -- python $ROOT/tools/_ka_iter_hist.py > $ROOT/$O/$name.stdout 2> $ROOT/$O/$name.stderr )
  python tools/pmc_summary.py /tmp/pmc_$name ka_solve_kernel $O/${name}_ka_solve_kernel.json > /dev/null 2>> $O/passes.log
}
pass icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass waits SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU
cat $O/icache_ka_solve_kernel.json $O/waits_ka_solve_kernel.json; cat $O/passes.log; tail -2 $O/icache.stderr
