for L in 1 4 16 64; do for G in 0 1; do
 timeout 300 python bench.py --lanes $L --streams $L --steps 20 --warmup 4 --reps 3 --graph $G --no-extras --no-cpu-baseline --check-streams 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
print('lanes',$L,'graph',$G,'ms/step',round(d['ms_per_step'],3),'fps',round(d['value'],1))"
done; done
