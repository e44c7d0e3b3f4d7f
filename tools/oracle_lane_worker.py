"""bench.py's parity checker for ONE lane (a subprocess so that several lanes are checked on the host cores in parallel):
runs the CPU oracle tracker over the lane's frames and prints the global pose of every frame as one JSON line.
TEST INFRASTRUCTURE: the oracle is the checker, never the product path.  argv: <frames.npz> with depth [T,rows,cols] u16, rgb
[T,rows,cols,3] u8, K, levels, iters, status [T] (the engine's status bits: its keyframe decisions are imposed where the oracle's
own covisibility ratio sits within 5e-4 of the threshold; anywhere else a different decision is an error)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from oracle import oracle as O

ST_TRACKED, ST_ODO_KF, ST_INTEGR_KF = 1, 4, 8

z = np.load(sys.argv[1])
d, c, K, st = z["depth"], z["rgb"], z["K"], z["status"]
levels = int(z["levels"]); iters = [int(v) for v in z["iters"]]
T, rows, cols = d.shape
O.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "1")))
cfg = O.default_config(rows=rows, cols=cols, fx=float(K[0]), fy=float(K[1]), cx=float(K[2]), cy=float(K[3]), levels=levels, iters=iters)
trk = O.Tracker(cfg)
imposed = 0
R, t = [np.eye(3).tolist()], [[0.0, 0.0, 0.0]]
for k in range(T):
    s = int(st[k])
    if k:
        trk.force_kf_decisions(bool(s & ST_ODO_KF), bool(s & ST_INTEGR_KF))
    ok = trk.track(d[k], c[k])
    if k == 0:
        continue
    assert ok == bool(s & ST_TRACKED), (k, ok, s)
    if ok:
        info = trk.last_info()
        if bool(info.odo_kf_natural) != bool(s & ST_ODO_KF):
            assert abs(info.visratio_odo - cfg.visratio_odo) < 5e-4, (k, info.visratio_odo)
            imposed += 1
        if bool(info.integr_kf_natural) != bool(s & ST_INTEGR_KF):
            assert abs(info.visratio_integr - cfg.visratio_integr) < 5e-4, (k, info.visratio_integr)
            imposed += 1
    Rs, ts = trk.poses()
    R.append(Rs[-1].tolist()); t.append(ts[-1].tolist())
trk.close()
print(json.dumps({"R": R, "t": t, "imposed": imposed}))
