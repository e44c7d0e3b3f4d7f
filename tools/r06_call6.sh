#!/bin/bash
# round 6, call 6: the profile set on the final tree (second box of the round), the driver's command, eager vs hipGraph at the headline lane count
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c6; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
ROUND=r06b bash tools/profile_bench.sh > $O/profile.log 2>&1
python tools/kernel_bench.py --lanes 1024 --json gpurun_out/profiles_r06b/kernel_bench_1024.json > $O/kernel_bench.log 2>&1
for G in 0 1 0 1; do
  python bench.py --steps 20 --warmup 5 --reps 3 --graph $G --no-extras --no-cpu-baseline --check-streams 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('graph $G', 'frames/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))" >> $O/graph.txt
done
cat $O/graph.txt
python - <<PY
import json
for f in ("$O/bench_driver_cmd.json", "gpurun_out/profiles_r06b/bench.json", "gpurun_out/profiles_r06b/bench_under_rocprof.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"],2), round(d["roofline"]["frac"],4), round(d["roofline"]["avg_launch_us"],1))
PY
cat gpurun_out/profiles_r06b/sq_table.md | head -6
