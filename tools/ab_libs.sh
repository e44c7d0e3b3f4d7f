#!/bin/bash
# A/B on ONE box: kernel trace of the default bench with lib/librgbid_hip.so (new) and lib/librgbid_hip_old.so (old), alternating
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
L=$ROOT/rgbid-slam_amd/lib
cp $L/librgbid_hip.so /tmp/new.so; cp $L/librgbid_hip_old.so /tmp/old.so
for rep in 1 2; do for v in old new; do
  cp /tmp/$v.so $L/librgbid_hip.so
  STEPS=${STEPS:-8} bash $ROOT/tools/quick_prof.sh > /dev/null 2>&1
  echo "== $v rep $rep: $(python -c "import json;d=json.loads(open('$ROOT/gpurun_out/quick/bench.json').read().strip().splitlines()[-1]);print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_us'],1))")"
  python - <<PY
import csv
for r in list(csv.DictReader(open("$ROOT/gpurun_out/quick/trace_kernels_rgbid.csv")))[:16]:
    print("  ", r["Name"][:58].ljust(58), r["Calls"].rjust(5), str(round(float(r["TotalDurationNs"])/1e6,2)).rjust(8), str(round(float(r["AverageActiveNs"])/1e3,1)).rjust(8), str(round(float(r["MaxNs"])/1e3,1)).rjust(8))
PY
done; done
cp /tmp/new.so $L/librgbid_hip.so
