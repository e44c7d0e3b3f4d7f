"""Where a bench step's wall time goes at different lane counts: GPU time of K steps (sync) vs the record read-back, per step enqueue time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rgbid-slam_amd"))
import numpy as np, torch
import bench
from rgbid import device, engine as E, synth

dev = torch.device("cuda:0")
ctx = device.Context(0)
K = synth.TUM_K
for B in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048]:
    T = 11
    seqs, depth, rgb = bench.make_inputs(B, T, 480, 640, K, dev, 32)
    eng = E.Engine(ctx, E.default_config(rows=480, cols=640, levels=3, lanes=B, K=K, iters=[10, 5, 3], record_capacity=T, keyframe_capacity=2))
    for mode in ("sync", "async", "async+prof"):
        ctx.set_async(0 if mode == "sync" else 1)
        for rep in range(3):
            eng.reset()
            for k in range(3): eng.step(depth[k], rgb[k])
            ctx.sync(); torch.cuda.synchronize(dev)
            if mode == "async+prof": eng.profile_begin(11 * 8)
            t0 = time.perf_counter(); enq = []
            for k in range(3, 11):
                a = time.perf_counter(); eng.step(depth[k], rgb[k]); enq.append(time.perf_counter() - a)
            t1 = time.perf_counter()
            ctx.sync(); torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            rec = eng.records(3, 8)
            t3 = time.perf_counter()
            if mode == "async+prof": eng.profile_end()
        print(f"lanes {B} {mode}: enqueue of 8 steps {1e3*(t1-t0):.2f} ms (per step {1e3*np.mean(enq):.2f}), GPU done after {1e3*(t2-t0):.2f} ms ({1e3*(t2-t0)/8:.2f} per step, {1e6*(t2-t0)/8/B:.2f} us/lane), records() {1e3*(t3-t2):.2f} ms, total fps {B*8/(t3-t0):.0f}")
    eng.close(); del depth, rgb, seqs
    torch.cuda.empty_cache()
