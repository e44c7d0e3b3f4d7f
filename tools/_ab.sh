cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/nt
V=$GRAFT_REPO_ROOT/rgbid-slam_amd/lib/librgbid_hip_base.so
B="timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > gpurun_out/nt/$name.json 2> gpurun_out/nt/$name.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/nt/$name.json") if l.startswith("{")][-1])
    p=d.get("parity",{})
    print("$name", round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), p.get("lanes_bit_identical"), p.get("within_1e-4"), p.get("worst_trans_m"))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/nt/$name.err").read()[-1500:])
PY
}
run base1 RGBID_HIP_LIB=$V
run new1 X=1
run base2 RGBID_HIP_LIB=$V
run new2 X=1
run base3 RGBID_HIP_LIB=$V
run new3 X=1
timeout 600 python tools/kernel_bench.py --lanes 1024 --only fuse,prep,sobel,maps
RGBID_HIP_LIB=$V timeout 600 python tools/kernel_bench.py --lanes 1024 --only fuse
timeout 1200 python -m pytest tests/test_gpu_batched.py tests/test_gpu_engine.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3
