for L in ${LANES_LIST:-256 512 768 1024 2048}; do
 timeout 600 python bench.py --lanes $L --steps 8 --warmup 2 --reps 2 --no-extras --no-cpu-baseline --check-streams 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
print('lanes',$L,'fps',round(d['value'],1),'ms/step',round(d['ms_per_step'],2),'u1 us/lane',round(d['roofline']['avg_launch_us']/$L,3), 'GB', round(d['config']['engine_hbm_bytes']/1e9,1))"
done
