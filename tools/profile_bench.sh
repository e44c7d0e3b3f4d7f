#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace + stats of the default bench command, then the PMC passes the
# roofline "traffic" figure needs (FETCH_SIZE and WRITE_SIZE in separate passes; --pmc is never combined with
# --sys-trace / --hip-trace).  Summaries land in gpurun_out/profiles_r01/ ; copy what matters into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profiles_r01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${EXTRA:-}"   # default: exactly `python bench.py` (the command the driver runs)
python $ROOT/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $ROOT/bench.py $ARGS --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o pmc_fetch -- python $ROOT/bench.py $ARGS --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o pmc_write -- python $ROOT/bench.py $ARGS --no-cpu-baseline > /dev/null 2>&1
rm -f $OUT/*agent_info.csv
python $ROOT/tools/summarize_prof.py $OUT
python $ROOT/tools/make_pmc_traffic.py $OUT $OUT/pmc_traffic.json ${LANES:-512}
ls -la $OUT
