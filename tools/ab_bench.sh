#!/bin/bash
# same-box A/B of two product libraries IN THE BENCH PROTOCOL (no profiler): lib/librgbid_hip.so against lib/librgbid_hip_<VARIANT>.so, alternating.
# The sustained protocol (20 timed steps) is the regime in which the level-0 kernel's clock is power-limited; tools/ab_kernels.sh times kernels alone.
#   usage (through gpurun): VARIANT=old STEPS=20 WARMUP=5 REPS=3 PAIRS=3 bash tools/ab_bench.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
L=$ROOT/rgbid-slam_amd/lib
V=${VARIANT:-old}
for rep in $(seq 1 ${PAIRS:-3}); do for v in base $V; do
  lib=$L/librgbid_hip.so; [ $v = $V ] && lib=$L/librgbid_hip_$V.so
  RGBID_HIP_LIB=$lib python $ROOT/bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --reps ${REPS:-3} --lanes ${LANES:-2048} --no-extras --no-cpu-baseline --check-streams 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v'.ljust(8), 'frames/s', round(d['value']), ' ms/step', round(d['ms_per_step'],3), ' level-0 kernel us', round(d['roofline']['avg_launch_us'],1), ' frac', round(d['roofline']['frac'],4))"
done; done
