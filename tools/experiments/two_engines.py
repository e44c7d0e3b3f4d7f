"""Experiment (measured, not adopted): the 2 048-lane workload as TWO engines of 1 024 lanes on two HIP streams, in bench.py's protocol, next to the one-engine run.
23 700 - 25 200 frames/s with the keyframe export ring, 25 800 - 26 000 without, against 26 000 - 26 400 for one engine (tools/multi_engine.py's looser loop had shown
+2.5 %): the overlap of launch tails does not pay for two half-size launches of every kernel.  python tools/experiments/two_engines.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rgbid-slam_amd"))
import numpy as np, torch
import time
import bench
from rgbid import device

def two_engine_config(dev, depth, rgb, rows, cols, levels, iters, B, Kst, W, K, keyframes, graph, fused, fast, ref_rec, defer_maps=0, reps=3):
    """The headline workload as TWO engines of B/2 lanes, each on its own context (HIP stream): the launch tails and the small latency-bound kernels of one group
    run under the bandwidth-bound kernels of the other ('overlap ... on separate HIP streams').  Same frames, same per-lane results."""
    from rgbid import device, engine as E
    T = 1 + W + Kst
    half = B // 2
    ctxs = [device.Context(dev.index or 0, use_torch_stream=False) for _ in range(2)]
    engs = [E.Engine(c, E.default_config(rows=rows, cols=cols, levels=levels, lanes=half, K=K, iters=iters, use_graph=graph, fused_gn=fused, record_capacity=T,
                                         keyframe_capacity=keyframes, fast_numerics=fast, defer_keyframe_maps=defer_maps)) for c in ctxs]
    d = [depth[:, g * half:(g + 1) * half].contiguous() for g in range(2)]
    r = [rgb[:, g * half:(g + 1) * half].contiguous() for g in range(2)]
    times, recs = [], None
    for rep in range(reps):
        if rep:
            for e in engs: e.reset()
        for k in range(0, 1 + W):
            for g in range(2): engs[g].step(d[g][k], r[g][k])
        for c in ctxs: c.sync()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(1 + W, T):
            for g in range(2): engs[g].step(d[g][k], r[g][k])
        recs = [e.records(1 + W, Kst) for e in engs]     # synchronises each engine's stream
        torch.cuda.synchronize(dev)
        times.append(time.perf_counter() - t0)
    el = float(np.median(times))
    allr = np.concatenate(recs, axis=1)
    same = ref_rec is not None and allr.tobytes() == ref_rec.tobytes()
    dR = float(np.abs(allr["R"] - ref_rec["R"]).max()) if ref_rec is not None else None
    dt = float(np.abs(allr["t"] - ref_rec["t"]).max()) if ref_rec is not None else None
    status_same = bool(np.array_equal(allr["status"], ref_rec["status"])) if ref_rec is not None else None
    for e in engs: e.close()
    for c in ctxs: c.close()
    return {"value": B * Kst / el, "unit": "frames/s", "ms_per_step": el / Kst * 1e3, "lanes": B, "engines": 2, "lanes_per_engine": half, "steps": Kst, "warmup": W,
            "repetitions": reps, "records_identical_to_headline_run": bool(same), "status_words_identical_to_headline_run": status_same,
            "max_abs_pose_entry_difference_vs_headline_run": {"R": dR, "t_m": dt},
            "note": "a 1 024-lane engine sums a lane's normal equations over a different workgroup partition than the 2 048-lane engine (the launch plan depends on the lane count): poses agree to rounding, not bit for bit"}


dev = torch.device("cuda", 0)
K = (525.0, 525.0, 319.5, 239.5)
B, Kst, W = 2048, 8, 2
seqs, depth, rgb = bench.make_inputs(B, 1 + W + Kst, 480, 640, K, dev)
ctx = device.Context(0)
work = torch.cuda.current_stream(dev)
for it in range(2):
    res, keep = bench.run_config(ctx, dev, work, 480, 640, 3, [10, 5, 3], B, Kst, W, 3, 32, 0, 1, 2, K, {"use_dist": False, "world": 1}, check_streams=0, fast_numerics=1, inputs=(seqs, depth, rgb))
    rec = keep[4]; keep[3].close(); del keep
    print("one engine  :", round(res["value"]), round(res["ms_per_step"], 2), flush=True)
    for kf in (2, 0):
        r2 = two_engine_config(dev, depth, rgb, 480, 640, 3, [10, 5, 3], B, Kst, W, K, kf, 0, 1, 1, rec, defer_maps=0, reps=3)
        print("two engines, keyframe ring", kf, ":", round(r2["value"]), round(r2["ms_per_step"], 2), flush=True)
