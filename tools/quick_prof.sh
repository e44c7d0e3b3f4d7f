#!/bin/bash
# kernel-trace only (no PMC passes): python bench.py under rocprofv3, condensed per-kernel CSV -> gpurun_out/quick${TAG}/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/quick${TAG:-}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps ${STEPS:-6} --warmup ${WARMUP:-2} --reps 1 --lanes ${LANES:-2048} ${EXTRA:-} --no-cpu-baseline --no-extras --check-streams 0"
rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python $ROOT/bench.py $ARGS > $OUT/bench_under_rocprof.json 2>/dev/null
rm -f $OUT/*agent_info.csv
python $ROOT/tools/summarize_prof.py $OUT
find $OUT -name "*kernel_trace.csv" -delete
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/trace_kernels_rgbid.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {r["Calls"]:>5} act {r["CallsActive"]:>5} avgAct {float(r["AverageActiveNs"])/1e3:9.1f} us  max {float(r["MaxNs"])/1e3:9.1f}  vgpr {r["VGPRs"]:>3}  {r["Name"][:95]}')
PY
