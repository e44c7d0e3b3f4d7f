"""Keyframe-switch rate of the bench workload: fraction of tracked frames that re-key the odometry / integration keyframe."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rgbid-slam_amd"))
import numpy as np, torch
from rgbid import device, engine as E
import bench
K = (525.0, 525.0, 319.5, 239.5)
T, B = 41, 16
dev = torch.device("cuda", 0)
seqs, depth, rgb = bench.make_inputs(B, T, 480, 640, K, dev, n_unique=16)
ctx = device.Context(0); ctx.set_async(1)
eng = E.Engine(ctx, E.default_config(rows=480, cols=640, levels=3, lanes=B, K=K, iters=[10, 5, 3], record_capacity=T, use_graph=0))
for k in range(T): eng.step(depth[k], rgb[k])
rec = eng.records()
st = rec["status"][1:]
print("frames", st.size, "odo KF switches", float(((st & E.ST_ODO_KF) != 0).mean()), "integr KF switches", float(((st & E.ST_INTEGR_KF) != 0).mean()),
      "tracked", float(((st & E.ST_TRACKED) != 0).mean()))
print("per-step odo switch fraction:", np.round(((st & E.ST_ODO_KF) != 0).mean(axis=1), 2))
