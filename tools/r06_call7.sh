#!/bin/bash
# round 6, call 7: the final tree (warp unit really built without SLP): whole GPU suite, smoke, the profile set (ROUND=r06), per-kernel bench, few-lane latencies
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c7; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "all rc $?" > $O/rc.txt
tail -3 $O/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/rc.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
ROUND=r06 bash tools/profile_bench.sh > $O/profile.log 2>&1
python tools/kernel_bench.py --lanes 1024 --json gpurun_out/profiles_r06/kernel_bench_1024.json > $O/kernel_bench.log 2>&1
LANES_LIST="1 8 64" GRAPHS="0 1" bash tools/lane_latency.sh > $O/lanes.txt 2>&1
cat $O/rc.txt $O/lanes.txt
python - <<PY
import json
for f in ("$O/bench_driver_cmd.json", "gpurun_out/profiles_r06/bench.json", "gpurun_out/profiles_r06/bench_under_rocprof.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"],2), round(d["roofline"]["frac"],4), round(d["roofline"]["avg_launch_us"],1))
PY
cat gpurun_out/profiles_r06/sq_table.md | head -8
grep -E "us/lane" $O/kernel_bench.log
