import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/rgbid-slam_amd")
import numpy as np, torch
from rgbid import host, synth
seq = synth.make_sequence(40, device="cuda")
d = seq["depth"].cpu().numpy().astype(np.uint16); c = seq["rgb"].cpu().numpy()
trk = host.Tracker(host.default_config())
for k in range(5): trk.track(d[k], c[k])
t0 = time.perf_counter()
for k in range(5, 40): trk.track(d[k], c[k])
dt = time.perf_counter() - t0
print("compat path (C++ VisodoTracker, host frames, synchronous bridge calls): %.2f ms/frame = %.0f frames/s" % (dt / 35 * 1e3, 35 / dt))
