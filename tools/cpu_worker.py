"""One single-threaded oracle tracker looping over the frames of an .npz for a time budget; prints the number of aligned frames.
Spawned by bench.py's cpu_baseline leg, one instance per host core (independent sequences are the CPU's natural parallelism too)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O

z = np.load(sys.argv[1]); budget = float(sys.argv[2])
d, c, K = z["depth"], z["rgb"], z["K"]
O.set_num_threads(1)
cfg = O.default_config(rows=d.shape[1], cols=d.shape[2], fx=float(K[0]), fy=float(K[1]), cx=float(K[2]), cy=float(K[3]))
frames = 0
t0 = time.perf_counter()
while time.perf_counter() - t0 < budget:
    trk = O.Tracker(cfg)
    trk.track(d[0], c[0])
    for k in range(1, d.shape[0]):
        trk.track(d[k], c[k]); frames += 1
        if time.perf_counter() - t0 >= budget: break
    trk.close()
print(frames, time.perf_counter() - t0)
