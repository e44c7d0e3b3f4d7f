"""diagnostic: engine fuzz seed 284 in both numerics classes against the oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rgbid-slam_amd"))
import numpy as np, torch
from tests import util
from tests.test_gpu_engine import make_lanes, rot_angle
from oracle import oracle as O
from rgbid import device, engine as E
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 284
r = util.rng(3000 + seed)
levels = int(r.integers(1, 5))
lo_r, lo_c = max(40, 30 << (levels - 1)), max(56, 40 << (levels - 1))
rows = int(r.integers(lo_r, lo_r + 40)); cols = int(r.integers(lo_c, lo_c + 60))
s = cols / 640.0
K = (525.0 * s, 525.0 * s * float(r.uniform(0.9, 1.1)), cols / 2.0 - 0.5 + float(r.uniform(-3, 3)), rows / 2.0 - 0.5 + float(r.uniform(-3, 3)))
iters = [int(r.integers(1, 7)) for _ in range(levels)]
print("levels", levels, "rows", rows, "cols", cols, "iters", iters, "K", K)
ctx = device.Context(0)
n_lanes, n_frames = 2, 3
seqs, depth, rgb = make_lanes(n_lanes, n_frames, rows, cols, K, trans_step=(0.002, 0.008), rot_step_deg=(0.1, 0.5))
recs = {}
for fast in (0, 1):
    eng = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=n_lanes, K=K, use_graph=0, record_capacity=n_frames, levels=levels, iters=iters, fast_numerics=fast))
    for k in range(n_frames): eng.step(depth[k], rgb[k])
    recs[fast] = eng.records().copy(); eng.close()
for l in range(n_lanes):
    trk = O.Tracker(O.default_config(rows=rows, cols=cols, fx=K[0], fy=K[1], cx=K[2], cy=K[3], levels=levels, iters=iters))
    d = depth[:, l].cpu().numpy().view(np.uint16); c = rgb[:, l].cpu().numpy()
    margins = []
    for k in range(n_frames):
        trk.track(d[k], c[k])
        if k: margins.append((k, trk.last_info().sigma_stop_margin_frame, trk.last_info().sigma_int, trk.last_info().sigma_depthinv))
    print('lane', l, 'oracle stop margins / sigmas', margins)
    Rs, ts = trk.poses()
    for k in range(1, n_frames):
        for fast in (0, 1):
            rec = recs[fast]
            print("lane", l, "frame", k, "fast" if fast else "exact", "dR %.2e dt %.2e" % (rot_angle(Rs[k], rec[k, l]["R"]), np.linalg.norm(ts[k] - rec[k, l]["t"])), "sigma", rec[k, l]["sigma_int"], rec[k, l]["sigma_depthinv"], "status", rec[k, l]["status"])
        print("   exact vs fast dR %.2e dt %.2e" % (rot_angle(recs[0][k, l]["R"], recs[1][k, l]["R"]), np.linalg.norm(recs[0][k, l]["t"] - recs[1][k, l]["t"])))
    oR, ot, ocov = trk.odometry()
    print("   oracle cov diag frame 2:", np.diag(ocov[2]) if len(ocov) > 2 else None)
