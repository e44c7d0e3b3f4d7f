# does the runtime's kernarg placement move the few-lane step latency? (HIP_FORCE_DEV_KERNARG: kernel arguments in device memory)
for L in 1 64; do for E in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; do
 env $E timeout 300 python bench.py --lanes $L --streams $L --steps 20 --warmup 4 --reps 3 --graph 1 --persistent 0 --no-extras --no-cpu-baseline --check-streams 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
print('lanes',$L,'$E','ms/step',round(d['ms_per_step'],3))"
done; done
