"""Per-frame cost of the compat VisodoTracker (C++, librgbid_host.so) on host frames at 640x480, in its three modes:
  - every bridge call synchronous and timed (the reference's bridge contract),
  - ScopedAsyncBridge (default): same call sequence, no per-call events / synchronisation,
  - setEngineBacked(true): the frame is ONE step of a one-lane device-resident engine (bit-exact numerics class).
All three produce identical results (tests/test_gpu_tracker_cpp.py).  Run on the GPU box: python tools/compat_time.py"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "rgbid-slam_amd"))
import numpy as np
from rgbid import host, synth
N, W = 60, 8
seq = synth.make_sequence(N, device="cuda")
d = seq["depth"].cpu().numpy().astype(np.uint16); c = seq["rgb"].cpu().numpy()
modes = ((0, False, "every bridge call synchronous and timed, as the reference"), (1, False, "ScopedAsyncBridge (default)"),
         (1, True, "engine-backed (setEngineBacked(true), one engine step per frame)"))
ref = None
for on, eb, what in modes:
    trk = host.Tracker(host.default_config(), engine_backed=eb); trk.set_async_bridge(on)
    for k in range(W): trk.track(d[k], c[k])
    t0 = time.perf_counter()
    for k in range(W, N): trk.track(d[k], c[k])
    dt = time.perf_counter() - t0
    R, t = trk.poses()
    if ref is None: ref = (R, t)
    same = np.array_equal(R, ref[0]) and np.array_equal(t, ref[1])
    print("compat path (C++ VisodoTracker, host frames), %s: %.2f ms/frame = %.0f frames/s%s" % (what, dt / (N - W) * 1e3, (N - W) / dt, "" if same else "  [POSES DIFFER]"))
    trk.close()
