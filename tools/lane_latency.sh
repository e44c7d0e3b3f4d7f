# ms per step (= latency of one frame) at small lane counts, eager and graph replay
for L in ${LANES_LIST:-1 4 8 16 32 64 128}; do for G in ${GRAPHS:-0 1}; do
 timeout 300 python bench.py --lanes $L --streams $L --steps 20 --warmup 4 --reps 3 --graph $G --no-extras --no-cpu-baseline --check-streams 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
print('lanes',$L,'graph',$G,'ms/step',round(d['ms_per_step'],3),'fps',round(d['value'],1),'launches',d['config']['launches_per_step'])"
done; done
