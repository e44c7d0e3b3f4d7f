#!/usr/bin/env python
"""Per-kernel timing of the engine's hot-path kernels through the batched C-ABI (include/rgbid_batched.h) on realistic inputs: `lanes`
frame pairs of the synthetic streams (keyframe = frame 0, current = frame 1, the ground-truth relative pose), every kernel launched alone and
timed with the call's own hipEvent pair.  The tool for A/B work on one kernel without running the whole engine:

    python tools/kernel_bench.py [--lanes 512] [--rows 480 --cols 640] [--reps 20] [--only gn,lattice,sigma,pyr,...] [--json out.json]
    RGBID_HIP_LIB=/path/to/librgbid_hip_variant.so python tools/kernel_bench.py ...      (a library built with other -D flags)

Prints one line per kernel: median us per launch, us per lane, algorithmic bytes, TB/s and the fraction of 8 TB/s."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rgbid-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=512)
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--only", default="")
    ap.add_argument("--skew", type=int, default=0, help="experiment: byte offset added per map allocation (i * skew), to move the maps' relative HBM channel alignment")
    ap.add_argument("--json", default="")
    ap.add_argument("--identity-pose", action="store_true", help="experiment: every lane warps with the identity (gathers land on the sampling pixel's own row)")
    args = ap.parse_args()
    from rgbid import batched as BT, device, synth
    B, rows, cols = args.lanes, args.rows, args.cols
    s = cols / 640.0
    K = (525.0 * s, 525.0 * s, (319.5 + 0.5) * s - 0.5, (239.5 + 0.5) * s - 0.5)
    dev = torch.device("cuda", 0)
    ctx = device.Context(0)
    bt = BT.Batched(ctx)
    only = set(x for x in args.only.split(",") if x)
    want = lambda k: not only or k in only

    # ---- inputs: frame pairs of `streams` synthetic streams dealt onto the lanes
    n = min(B, args.streams)
    seqs = [synth.make_sequence(2, seed=synth.SEED + 17 * i, K=K, rows=rows, cols=cols, device=dev) for i in range(n)]
    idx = torch.arange(B, device=dev) % n
    d16 = torch.stack([q["depth"].to(torch.int16) for q in seqs], 1)[:, idx].contiguous()       # [2, B, rows, cols]
    rgb = torch.stack([q["rgb"] for q in seqs], 1)[:, idx].contiguous()
    Rp, tp = [], []
    for i in range(n):
        R, t = synth.relative_pose(seqs[i]["R_wc"][0], seqs[i]["t_wc"][0], seqs[i]["R_wc"][1], seqs[i]["t_wc"][1])
        Ri = R.T.numpy(); ti = -(R.T @ t).numpy()
        Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]])
        Rp.append((Km @ Ri @ np.linalg.inv(Km)).astype(np.float32).reshape(9)); tp.append((Km @ ti).astype(np.float32))
    Rs = [Rp[l % n] for l in range(B)]; ts = [tp[l % n] for l in range(B)]
    if args.identity_pose:
        Rs = [np.eye(3, dtype=np.float32).reshape(9) for _ in range(B)]; ts = [np.zeros(3, dtype=np.float32) for _ in range(B)]
    nalloc = [0]

    def f32(*shape):
        if not args.skew:
            return torch.empty(shape, device=dev)
        nalloc[0] += 1
        off = (nalloc[0] % 16) * args.skew // 4
        n_el = int(np.prod(shape))
        return torch.empty(n_el + 16 * args.skew // 4, device=dev)[off:off + n_el].view(*shape)
    W = [f32(B, rows, cols) for _ in range(2)]; I = [f32(B, rows, cols) for _ in range(2)]
    ch = [f32(B, rows, cols) for _ in range(3)]
    results = {}

    def timed(name, fn, bytes_per_lane, reps=None):
        ms = []
        for _ in range(reps or args.reps):
            ms.append(fn())
        us = float(np.median(ms[2:] if len(ms) > 4 else ms)) * 1e3
        tbs = bytes_per_lane * B / us / 1e6
        results[name] = dict(us=us, us_per_lane=us / B, bytes_per_lane=bytes_per_lane, tbs=tbs, frac=tbs / 8.0)
        print(f"{name:34s} {us:10.1f} us  {us / B:7.3f} us/lane  {bytes_per_lane / 1e6:8.3f} MB/lane  {tbs:6.2f} TB/s  {tbs / 8.0:5.2f} of peak", flush=True)

    N0 = rows * cols
    for k in (0, 1):
        fn = lambda k=k: bt.prep_frame(d16[k], rgb[k], W[k], I[k], *ch, 1.0)
        if k == 0 or not want("prep"):
            fn()
        else:
            timed("prep_frame", fn, 25 * N0)
    gWx, gWy, gIx, gIy = [f32(B, rows, cols) for _ in range(4)]
    bt.gradient(W[0], gWx, gWy)
    if want("sobel"):
        timed("gradient (Sobel pair)", lambda: bt.gradient(I[0], gIx, gIy), 12 * N0)
    else:
        bt.gradient(I[0], gIx, gIy)
    if want("pyr"):
        half = f32(B, rows // 2, cols // 2)
        timed("pyr_down L0->L1", lambda: bt.pyr_down(I[1], half), 5 * N0)
        del half
    if want("bilateral"):
        tmp = f32(B, rows, cols)
        timed("bilateral FAST (intensity)", lambda: bt.bilateral(I[0], tmp, 3.0, fast=True), 8 * N0, reps=6)
        del tmp
    sp = BT.sys_params(B, student_nu=1)
    spc = BT.sys_params(B, student_nu=0)
    if want("gn"):
        timed("gn_fused FAST WM1 (GN iteration)", lambda: bt.gn_fused(W[0], I[0], gWx, gWy, gIx, gIy, W[1], I[1], Rs, ts, K, sp, fast=True, return_ms=True)[2], 32 * N0)
        timed("gn_fused FAST WM2 (covariance)", lambda: bt.gn_fused(W[0], I[0], gWx, gWy, gIx, gIy, W[1], I[1], Rs, ts, K, spc, fast=True, return_ms=True)[2], 32 * N0, reps=8)
    if want("gn_exact"):
        timed("gn_fused EXACT", lambda: bt.gn_fused(W[0], I[0], gWx, gWy, gIx, gIy, W[1], I[1], Rs, ts, K, sp, fast=False, return_ms=True)[2], 32 * N0, reps=6)
    if want("unfused"):
        W1, I1 = f32(B, rows, cols), f32(B, rows, cols)
        timed("warp_pair FAST", lambda: bt.warp_pair(W[1], I[1], W[0], W1, I1, Rs, ts, fast=True), 20 * N0, reps=8)
        timed("build_system (stored W1/I1)", lambda: bt.build_system(W[0], I[0], gWx, gWy, gIx, gIy, W1, I1, K, sp, return_ms=True)[2], 32 * N0, reps=8)
        del W1, I1
    if want("lattice") or want("sigma"):
        ns = device.error_lattice_size(rows, cols, 10000)[0]
        kf_lat = f32(B, 2 * ns); res = f32(B, 2 * ns)
        bt.lattice_pack(W[0], I[0], 10000, kf_lat)
        timed("lattice_residuals FAST (packed KF)", lambda: bt.lattice_residuals(W[1], W[0], I[1], I[0], Rs, ts, 10000, res, fast=True, kf_lat=kf_lat), 36 * ns)
        if want("sigma"):
            L = bt.L; ms = __import__("ctypes").c_float()

            def sig():
                out = (BT.ScalePair * B)()
                BT.check(L.rgbid_sigma_pair_batched(bt._h, B, __import__("ctypes").c_void_p(res.data_ptr()), __import__("ctypes").c_size_t(res.stride(0)), ns, 3, out, __import__("ctypes").byref(ms)))
                return ms.value
            timed("sigma_pair (both channels)", sig, 8 * ns)
    if want("fuse"):
        kf = W[0].clone(); kfw = torch.ones_like(kf); ww = torch.zeros_like(kf)
        timed("fuse_frame FAST", lambda: bt.fuse_frame(W[1], kf, kfw, ww, Rs, ts, fast=True), 20 * N0, reps=8)
        timed("fuse_frame EXACT", lambda: bt.fuse_frame(W[1], kf, kfw, ww, Rs, ts, fast=False), 24 * N0, reps=8)
        del kf, kfw, ww
    if want("maps"):
        vm, nm = f32(B, 3 * rows, cols), f32(B, 3 * rows, cols)
        timed("kf_maps (vertices + normals)", lambda: bt.kf_maps(K, W[0], vm, nm), 28 * N0, reps=8)
        del vm, nm
    if want("vis"):
        import ctypes as C
        msv = C.c_float()

        def vis():
            k1, p1 = BT._f32(Rs, B, 9); k2, p2 = BT._f32(ts, B, 3)
            counts = np.zeros((B, 4), np.uint32)
            BT.check(bt.L.rgbid_visibility_pair_batched(bt._h, B, C.byref(BT.imgb(W[1])), C.byref(BT.imgb(W[0])), p1, p2, p1, p2, 1, counts.ctypes.data_as(C.c_void_p), C.byref(msv)))
            return msv.value
        timed("visibility_pair FAST", vis, 16 * N0, reps=8)
    if args.json:
        json.dump(dict(lanes=B, rows=rows, cols=cols, lib=os.environ.get("RGBID_HIP_LIB", "default"), kernels=results), open(args.json, "w"), indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
