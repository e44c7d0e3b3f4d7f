#!/bin/bash
# A/B of kernel variants selected by environment variables: each argument is one "VAR=val VAR2=val" set; prints the per-kernel table of a
# short bench run under rocprofv3 --kernel-trace for each.   usage: bash tools/ab_env.sh "RGBID_X=0" "RGBID_X=1"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for cfg in "$@"; do
  i=$((i+1))
  echo "=== [$i] $cfg"
  env $cfg TAG=_ab$i STEPS=${STEPS:-3} WARMUP=1 EXTRA="${EXTRA:-}" bash $ROOT/tools/quick_prof.sh 2>&1 | grep -E "${FILTER:-rgbid::}" | head -${HEAD:-8}
  grep -o '"value": [0-9.]*' $ROOT/gpurun_out/quick_ab$i/bench_under_rocprof.json | head -1
done
