#!/bin/bash
# Profiles bench.py on the GPU box: one rocprofv3 --kernel-trace --stats run, then one --pmc run per counter group
# (never mixed with trace domains).  usage: tools/profile_round.sh <out-dir-under-gpurun_out> [bench args...]
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-recall --no-concurrent --no-boundary --no-traffic $*"
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- $B > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum -d $O/pmc_fetch --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT -d $O/pmc_inst --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $O/pmc_sq --output-format csv -- $B > /dev/null 2>&1
find $O -name "*.csv" | head -20
