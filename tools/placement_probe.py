#!/usr/bin/env python
"""Placement sensitivity of BASELINE config 5 (1280x960, 4 levels, 128 lanes): the level-0 Gauss-Newton kernel's fraction of the HBM peak in ONE fresh process,
optionally after the process has created, stepped and destroyed the 2 048-lane headline engine first (the state bench.py's extra configuration 5 runs in).
The engine's placement switches come from the environment (RGBID_ENGINE_LANE_PAD / RGBID_ENGINE_MAP_SKEW, bytes); RGBID_ENGINE_DEBUG_ALLOC=1 prints the map bases.

    python tools/placement_probe.py [--after-big] [--steps 3]      -> one JSON line"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rgbid-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--after-big", action="store_true")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--rows", type=int, default=960)
    ap.add_argument("--cols", type=int, default=1280)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--lanes", type=int, default=128)
    a = ap.parse_args()
    from rgbid import device
    dev = torch.device("cuda", 0)
    work = torch.cuda.Stream(dev)
    with torch.cuda.stream(work):
        ctx = device.Context(0)
    ctx.set_async(1)
    env = {"use_dist": False, "world": 1}
    if a.after_big:
        K = (525.0, 525.0, 319.5, 239.5)
        r, keep = bench.run_config(ctx, dev, work, 480, 640, 3, [10, 5, 3], 2048, 2, 1, 1, 32, 0, 1, 2, K, env)
        big = r["u1"]["achieved"] / bench.HBM_PEAK_GBS
        keep[3].close(); del keep, r
        torch.cuda.empty_cache()
    s = a.cols / 1280.0
    K5 = (1050.0 * s, 1050.0 * s, 639.5 * s, 479.5 * s)
    it = [10, 5, 3, 3][:a.levels]
    r5, keep5 = bench.run_config(ctx, dev, work, a.rows, a.cols, a.levels, it, a.lanes, a.steps, 1, 1, 8, 0, 1, 2, K5, env)
    out = {"after_big": bool(a.after_big), "lane_pad": os.environ.get("RGBID_ENGINE_LANE_PAD", "default"), "map_skew": os.environ.get("RGBID_ENGINE_MAP_SKEW", "default"),
           "frames_per_s": round(r5["value"], 1), "u1_frac": round(r5["u1"]["achieved"] / bench.HBM_PEAK_GBS, 4), "u1_us": round(r5["u1"]["avg_launch_us"], 1),
           "tracked": r5["tracked"], "expected": r5["expected"]}
    if a.after_big:
        out["headline_u1_frac"] = round(big, 4)
    keep5[3].close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
