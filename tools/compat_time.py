import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/rgbid-slam_amd")
import numpy as np
from rgbid import host, synth
seq = synth.make_sequence(40, device="cuda")
d = seq["depth"].cpu().numpy().astype(np.uint16); c = seq["rgb"].cpu().numpy()
for on, what in ((0, "every bridge call synchronous and timed, as the reference"), (1, "ScopedAsyncBridge (default)")):
    trk = host.Tracker(host.default_config()); trk.set_async_bridge(on)
    for k in range(5): trk.track(d[k], c[k])
    t0 = time.perf_counter()
    for k in range(5, 40): trk.track(d[k], c[k])
    dt = time.perf_counter() - t0
    print("compat path (C++ VisodoTracker, host frames), %s: %.2f ms/frame = %.0f frames/s" % (what, dt / 35 * 1e3, 35 / dt))
    trk.close()
