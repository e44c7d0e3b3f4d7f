#!/bin/bash
# round 6, call 2: the new GPU tests, then same-box A/Bs (kernel bench: bilateral / pyrDown old vs new by environment switch; bench protocol: packed Jacobian rows)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c2; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_batched.py -x -q -m gpu -k "bilateral or pyr or natively" > $O/t_kernels.log 2>&1; echo "kernels rc $?" > $O/rc.txt
timeout 1800 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "chi or configurations" > $O/t_chi.log 2>&1; echo "chi rc $?" >> $O/rc.txt
timeout 1200 python -m pytest tests/test_gpu_tracker_cpp.py -x -q -m gpu -k "chi or keyframe_align or round5" > $O/t_cpp.log 2>&1; echo "cpp rc $?" >> $O/rc.txt
timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "bilateral" > $O/t_fuzz.log 2>&1; echo "fuzz rc $?" >> $O/rc.txt
for rep in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export RGBID_BILATERAL_TWO_SIDED=1 RGBID_PYRDOWN_NO_FASTPATH=1; else unset RGBID_BILATERAL_TWO_SIDED RGBID_PYRDOWN_NO_FASTPATH; fi
    echo "== $v rep $rep lanes 1024" >> $O/ab_kernels.txt
    python tools/kernel_bench.py --lanes 1024 --only bilateral,pyr 2>&1 | grep -E "us/lane" >> $O/ab_kernels.txt
    echo "== $v rep $rep lanes 8" >> $O/ab_kernels.txt
    python tools/kernel_bench.py --lanes 8 --only bilateral,pyr 2>&1 | grep -E "us/lane" >> $O/ab_kernels.txt
  done
done
unset RGBID_BILATERAL_TWO_SIDED RGBID_PYRDOWN_NO_FASTPATH
VARIANT=pkj STEPS=20 WARMUP=5 REPS=3 PAIRS=2 bash tools/ab_bench.sh > $O/ab_pkj.txt 2>&1
for v in old new; do
  if [ $v = old ]; then export RGBID_BILATERAL_TWO_SIDED=1 RGBID_PYRDOWN_NO_FASTPATH=1; else unset RGBID_BILATERAL_TWO_SIDED RGBID_PYRDOWN_NO_FASTPATH; fi
  python bench.py --steps 20 --warmup 5 --reps 3 --no-extras --no-cpu-baseline --check-streams 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'frames/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))" >> $O/ab_prep.txt
done
cat $O/rc.txt; tail -3 $O/t_*.log; cat $O/ab_kernels.txt $O/ab_pkj.txt $O/ab_prep.txt
