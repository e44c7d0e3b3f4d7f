"""Measurements for the result table of BASELINE.md (run on the GPU box): python tools/baseline_table.py > gpurun_out/baseline_table.md"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rgbid-slam_amd"))
import numpy as np, torch
from oracle import oracle as O
from rgbid import device, synth
from tests import util

K = synth.TUM_K
rows, cols = 480, 640
r = util.rng(1)
maps = [util.rand_invdepth(r, rows, cols) for _ in range(8)]
A = b = None


def cpu_u1(threads):
    O.set_num_threads(threads)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        A, b = O.build_system(*maps, K)
        x = O.llt_solve6(A, b)
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts[1:]))


ncpu = os.cpu_count()
t1, tall = cpu_u1(1), cpu_u1(min(ncpu, 64))
ctx = device.Context(0)
dm = [torch.from_numpy(m).cuda() for m in maps]
ms = []
for _ in range(40):
    A, b, m = ctx.buildSystemStudentNuGridStride(*dm, 3, 0, 0.0025, 5.0, 0.0, 0.0, 5.0, 5.0, K, return_ms=True)
    ms.append(m)
g1 = float(np.median(ms[5:])) * 1e-3
u1 = 32.0 * rows * cols
print(f"| 1. single 640x480 pair, U1 + 6x6 solve | {t1*1e3:.2f} ms ({u1/t1/1e9:.2f} GB/s) | {tall*1e3:.2f} ms on {min(ncpu,64)} threads ({u1/tall/1e9:.1f} GB/s) | "
      f"{g1*1e6:.1f} us device time, cache-resident ({u1/g1/1e9:.0f} GB/s); batched x512: see row 2 | - | - | - | A,b rel 2e-5 |")


def bench(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True)
    line = [l for l in out.stdout.split("\n") if l.startswith("{")][-1]
    return json.loads(line)


d = bench([])
cb = d["cpu_baseline"]
print(f"| 2./3. synthetic TUM-like streams (stand-in; tracking + iD fusion) | - | {cb['value']:.1f} frames/s on {cb['cores']} threads | "
      f"{d['value']:.0f} frames/s ({d['config']['lanes_per_gpu']} lanes; U1 kernel {d['roofline']['achieved']:.0f} GB/s = {d['roofline']['frac']:.2f} of peak) | driver | driver | driver | <= 2e-6 / 4e-6 |")
d = bench(["--rows", "960", "--cols", "1280", "--levels", "4", "--lanes", "128", "--no-cpu-baseline"])
print(f"| 5. 1280x960 synthetic, 4 levels | - | - | {d['value']:.0f} frames/s (128 lanes; U1 kernel {d['roofline']['achieved']:.0f} GB/s = {d['roofline']['frac']:.2f} of peak, "
      f"{d['roofline']['avg_launch_us']:.0f} us/launch) | - | - | - | < 1e-4 (tests) |")
