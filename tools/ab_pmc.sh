#!/bin/bash
# PMC A/B of kernel variants selected by environment variables (own --pmc passes, kernel-trace only).
# usage: COUNTERS="FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum" FILTER=warp_pair bash tools/ab_pmc.sh "ENV=a" "ENV=b"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps ${STEPS:-2} --warmup 1 --reps 1 --lanes ${LANES:-2048} ${EXTRA:-} --no-cpu-baseline --no-extras --check-streams 0"
i=0
for cfg in "$@"; do
  i=$((i+1))
  IFS='|' read -ra PASSES <<< "${COUNTERS:-FETCH_SIZE|WRITE_SIZE}"
  for pass in "${PASSES[@]}"; do
    OUT=$ROOT/gpurun_out/pmc_ab$i; rm -rf $OUT; mkdir -p $OUT
    env $cfg rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT -o p -- python $ROOT/bench.py $ARGS > /dev/null 2>&1
    rm -f $OUT/*agent_info.csv
    python $ROOT/tools/summarize_prof.py $OUT > /dev/null
    echo "=== [$i] $cfg  pass: $pass"
    python - <<PY
import csv
px = ${LANES:-2048} * 480 * 640
for r in csv.DictReader(open("$OUT/p_pmc_rgbid.csv")):
    if "${FILTER:-rgbid::}" in r["Name"]:
        v = float(r["Max"]); unit = 1024.0 if "SIZE" in r["Counter"] else 1.0
        print(f'  {r["Counter"]:14s} max {v:14.1f}  -> {v*unit/px:8.3f} per px (raw; FETCH_SIZE x2 for wide loads)   {r["Name"][:70]}')
PY
  done
done
