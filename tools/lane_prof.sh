#!/bin/bash
# per-kernel time PER LANE at several lane counts (kernel trace of a short bench run each): which kernels do not scale linearly?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for L in ${LANES_LIST:-512 1024 2048}; do
  TAG=_l$L LANES=$L STEPS=4 WARMUP=1 EXTRA="--streams 32" bash $ROOT/tools/quick_prof.sh > /dev/null 2>&1
  echo "=== lanes $L  $(grep -o '"value": [0-9.]*' $ROOT/gpurun_out/quick_l$L/bench_under_rocprof.json | head -1)"
  python - <<PY
import csv, re
rows = [r for r in csv.DictReader(open("$ROOT/gpurun_out/quick_l$L/trace_kernels_rgbid.csv")) if ("rgbid" in r["Name"] or "anonymous" in r["Name"]) and "at::native" not in r["Name"]]
steps = 6.0
for r in rows[:14]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"].replace("void rgbid::", "").replace("void ", "")).split("(")[0][:58]
    print(f'  {float(r["TotalDurationNs"])/1e3/steps/$L:8.3f} us/lane/step  calls {r["Calls"]:>4}  avgAct {float(r["AverageActiveNs"])/1e3:9.1f}  {n}')
print(f'  {sum(float(r["TotalDurationNs"]) for r in rows)/1e3/steps/$L:8.3f} us/lane/step  TOTAL')
PY
done
