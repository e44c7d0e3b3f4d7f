"""Diagnostic: the GEOM_ONLY engine configuration against the oracle, per frame, exact and fast gather numerics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rgbid-slam_amd"))
import numpy as np, torch
from rgbid import device, engine as E
from oracle import oracle as O
from rgbid import synth
from tests.test_gpu_engine import make_lanes, rot_angle

ctx = device.Context(0)
rows, cols = 120, 160
s = cols / 640.0
K = (525.0 * s, 525.0 * s, (319.5 + 0.5) * s - 0.5, (239.5 + 0.5) * rows / 480.0 - 0.5)
n_lanes, n_frames = 2, 5
seqs, depth, rgb = make_lanes(n_lanes, n_frames, rows, cols, K, trans_step=(0.003, 0.01), rot_step_deg=(0.1, 0.6))
for wname, w in [("GEOM_ONLY", O.GEOM_ONLY), ("INDEPENDENT", 0)]:
    refs = []
    for l in range(n_lanes):
        trk = O.Tracker(O.default_config(rows=rows, cols=cols, fx=K[0], fy=K[1], cx=K[2], cy=K[3], weighting=w))
        d = depth[:, l].cpu().numpy().view(np.uint16); c = rgb[:, l].cpu().numpy()
        infos = []
        for k in range(n_frames):
            trk.track(d[k], c[k]); infos.append(trk.last_info().sigma_int)
        Rs, ts = trk.poses(); refs.append((Rs, ts, infos)); trk.close()
    for fast in (0, 1):
        eng = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=n_lanes, K=K, record_capacity=n_frames, weighting=w, fast_numerics=fast))
        for k in range(n_frames):
            eng.step(depth[k], rgb[k])
        rec = eng.records()
        for l in range(n_lanes):
            Rs, ts, infos = refs[l]
            print(wname, "fast", fast, "lane", l, " ".join(
                f"[{rot_angle(Rs[k], rec[k, l]['R']):.1e} {np.linalg.norm(ts[k] - rec[k, l]['t']):.1e} s {abs(rec[k, l]['sigma_int'] - infos[k]) / infos[k]:.1e}]"
                for k in range(1, n_frames)))
        eng.close()
