"""Experiment: G engines of lanes/G lanes each on their own HIP streams vs one engine -- do latency-bound kernels (sigma/nu, solve) of one group
overlap with the bandwidth-bound kernels of the others?  python tools/multi_engine.py [lanes] [groups...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rgbid-slam_amd"))
import numpy as np, torch
from rgbid import device, engine as E
import bench

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 128
groups = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
K = (525.0, 525.0, 319.5, 239.5)
T, W = 11, 2
dev = torch.device("cuda", 0)
seqs, depth, rgb = bench.make_inputs(lanes, T, 480, 640, K, dev)
for G in groups:
    per = lanes // G
    ctxs = [device.Context(0) for _ in range(G)]
    for c in ctxs: c.set_async(1)
    engs = [E.Engine(c, E.default_config(rows=480, cols=640, levels=3, lanes=per, K=K, iters=[10, 5, 3], record_capacity=T, use_graph=int(os.environ.get("RG_GRAPH", "0")))) for c in ctxs]
    d = [depth[:, g * per:(g + 1) * per].contiguous() for g in range(G)]
    r = [rgb[:, g * per:(g + 1) * per].contiguous() for g in range(G)]
    for k in range(0, 1 + W):
        for g in range(G): engs[g].step(d[g][k], r[g][k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1 + W, T):
        for g in range(G): engs[g].step(d[g][k], r[g][k])
    recs = [e.records(1 + W, T - 1 - W) for e in engs]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n = T - 1 - W
    print(f"groups={G} lanes/group={per}: {el / n * 1e3:.3f} ms/step  {lanes * n / el:.0f} frames/s", flush=True)
    for e in engs: e.close()
    for c in ctxs: c.close()
