#!/bin/bash
# SQ / GRBM counters of the default bench (own pass, kernel-trace only): issue-slot accounting per kernel -> gpurun_out/pmc_sq/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps ${STEPS:-3} --warmup 1 --lanes ${LANES:-512} ${EXTRA:-} --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o sq -- python $ROOT/bench.py $ARGS > /dev/null 2>$OUT/err1.txt
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT -o sq2 -- python $ROOT/bench.py $ARGS > /dev/null 2>$OUT/err2.txt
rm -f $OUT/*agent_info.csv
python $ROOT/tools/summarize_prof.py $OUT
ls -la $OUT
