#!/bin/bash
# kernel-trace only (no PMC passes): python bench.py under rocprofv3, condensed per-kernel CSV -> gpurun_out/quick/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps ${STEPS:-6} --warmup ${WARMUP:-2} --lanes ${LANES:-512} ${EXTRA:-} --no-cpu-baseline"
python $ROOT/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python $ROOT/bench.py $ARGS > $OUT/bench_under_rocprof.json 2>/dev/null
rm -f $OUT/*agent_info.csv
python $ROOT/tools/summarize_prof.py $OUT
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/bench.json
