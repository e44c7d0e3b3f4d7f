"""Quick per-kernel timings through the C-ABI (async mode, hipEvent brackets over N back-to-back launches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rgbid-slam_amd"))
import numpy as np, torch
from rgbid import device
from tests import util

ctx = device.Context(0)
ctx.set_async(1)
r = util.rng(0)
for rows, cols in [(480, 640), (240, 320), (120, 160)]:
    K = (525.0 * cols / 640, 525.0 * cols / 640, 319.5 * cols / 640, 239.5 * cols / 640)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    W0 = d(util.rand_invdepth(r, rows, cols)); I0 = d(util.rand_intensity(r, rows, cols))
    Wc = d(util.rand_invdepth(r, rows, cols)); Ic = d(util.rand_intensity(r, rows, cols))
    gWx, gWy, gIx, gIy, W1, I1 = [torch.empty((rows, cols), device="cuda") for _ in range(6)]
    R, t = util.small_motion(r, K, 0.01, 0.5)
    Rp, tp = util.project(K, *util.inv_pose(R, t))
    ctx.computeGradient(W0, gWx, gWy); ctx.computeGradient(I0, gIx, gIy)
    def timeit(name, fn, nbytes, n=200):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"{rows}x{cols} {name:28s} {us:8.2f} us  {nbytes/us/1e6:8.3f} TB/s (algorithmic)")
    N = rows * cols
    timeit("warp_invdepth", lambda: ctx.warpInvDepthWithTrafo3D(Wc, W1, W0, Rp, tp), 12 * N)
    timeit("warp_intensity", lambda: ctx.warpIntensityWithTrafo3DInvDepth(Ic, I1, W1, Rp, tp), 12 * N)
    timeit("gradient", lambda: ctx.computeGradient(W0, gWx, gWy), 12 * N)
    dst = torch.empty((rows // 2, cols // 2), device="cuda")
    timeit("pyr_down", lambda: ctx.pyrDown(W0, dst), 5 * N)
    timeit("bilateral", lambda: ctx.bilateralFilter(W0, gWx, 0.005), 8 * N)
    ctx.computeGradient(W0, gWx, gWy)
    err = torch.empty(N, device="cuda")
    timeit("compute_error", lambda: ctx.computeErrorGridStride(W1, W0, err, 10000), 12 * 19200)
ctx.set_async(0)
for rows, cols in [(480, 640), (240, 320), (120, 160)]:
    K = (525.0 * cols / 640, 525.0 * cols / 640, 319.5 * cols / 640, 239.5 * cols / 640)
    maps = [torch.from_numpy(util.rand_invdepth(r, rows, cols)).cuda() for _ in range(8)]
    ms = []
    for _ in range(30):
        A, b, m = ctx.buildSystemStudentNuGridStride(*maps, 3, 0, 0.0025, 5.0, 0.0, 0.0, 5.0, 5.0, K, return_ms=True)
        ms.append(m)
    ms = np.array(ms[5:]) * 1e3
    print(f"{rows}x{cols} build_system(+reduce) device time: median {np.median(ms):.2f} us  min {ms.min():.2f} us -> {32*rows*cols/np.median(ms)/1e6:.3f} TB/s")
    e = torch.from_numpy((0.003 * r.standard_t(5, 19200)).astype(np.float32)).cuda()
    t0 = time.perf_counter()
    for _ in range(50): ctx.computeSigmaAndNuStudent(e, 19200, 0.0, 0.0025, 5.0, 3)
    print(f"sigma_nu_student wall (sync incl. D2H): {(time.perf_counter()-t0)/50*1e6:.1f} us")
