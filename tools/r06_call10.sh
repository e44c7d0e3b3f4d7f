#!/bin/bash
# round 6, call 10: the round's run-time-switchable changes in the driver protocol, alternating on one box (old = two-sided bilateral, no pyrDown fast path, no map skew)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c10; mkdir -p $O
for rep in 1 2 3; do for v in old new; do
  if [ $v = old ]; then export RGBID_BILATERAL_TWO_SIDED=1 RGBID_PYRDOWN_NO_FASTPATH=1 RGBID_ENGINE_MAP_SKEW=0; else unset RGBID_BILATERAL_TWO_SIDED RGBID_PYRDOWN_NO_FASTPATH RGBID_ENGINE_MAP_SKEW; fi
  python bench.py --steps 20 --warmup 5 --reps 3 --no-extras --no-cpu-baseline --check-streams 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'frames/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'level-0 kernel us', round(d['roofline']['avg_launch_us'],1), 'frac', round(d['roofline']['frac'],4))" >> $O/ab_round.txt
done; done
cat $O/ab_round.txt
