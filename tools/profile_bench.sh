#!/bin/bash
# Runs on the GPU box (through gpurun): the default bench line, then kernel-trace + stats of the same workload, then the PMC passes the
# roofline "traffic" figure needs (FETCH_SIZE and WRITE_SIZE in separate passes; --pmc is never combined with --sys-trace / --hip-trace),
# the SQ issue-slot pass, and a kernel trace of BASELINE config 5 (1280x960, 4 levels).  Every profiler pass is bounded by `timeout`.
# Summaries land in gpurun_out/profiles_${ROUND}/ ; copy what matters into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r04}
OUT=$ROOT/gpurun_out/profiles_$ROUND
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LIGHT="--reps 1 --no-extras --check-streams 0 --no-cpu-baseline"     # what the profiled passes drop: repetitions, the extra configurations, the oracle check
python $ROOT/bench.py ${EXTRA:-} > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $ROOT/bench.py ${EXTRA:-} $LIGHT > $OUT/bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o pmc_fetch -- python $ROOT/bench.py ${EXTRA:-} $LIGHT --steps 3 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o pmc_write -- python $ROOT/bench.py ${EXTRA:-} $LIGHT --steps 3 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o sq -- python $ROOT/bench.py ${EXTRA:-} $LIGHT --steps 3 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace1280 -- python $ROOT/bench.py --rows 960 --cols 1280 --levels 4 --lanes 128 --streams 8 $LIGHT > $OUT/bench1280_under_rocprof.json 2>/dev/null
rm -f $OUT/*agent_info.csv
python $ROOT/tools/summarize_prof.py $OUT
python $ROOT/tools/make_pmc_traffic.py $OUT $OUT/pmc_traffic.json ${LANES:-2048} || true
python $ROOT/tools/make_pmc_traffic_all.py $OUT $OUT/pmc_traffic_all.json ${LANES:-2048} || true
python $ROOT/tools/sq_table.py $OUT > $OUT/sq_table.md || true
find $OUT -name "*kernel_trace.csv" -size +4M -delete
ls -la $OUT
