#!/usr/bin/env python
"""Absolute trajectory error (ATE) and relative pose error (RPE) of a trajectory file against a ground-truth file, both in the TUM RGB-D
benchmark format the reference's evaluation mode writes (`stamp tx ty tz qx qy qz qw` per line, tools/evaluation.cpp:380-439).

Own implementation of the benchmark's two metrics (Sturm et al., IROS 2012): poses are associated by time stamp (nearest neighbour within
`max_dt`), ATE = RMSE of the translational residual after the least-squares rigid alignment (Horn / Umeyama without scale) of the
estimate onto the ground truth, RPE = error of the relative motion over a fixed frame or time delta.  This is the EXTERNAL pin the oracle
lacks: the moment a TUM / ICL-NUIM sequence with ground truth is mounted (RGBID_TUM_DIR), tests/test_gpu_datasets.py tracks it with the
product and scores it here -- independent of the oracle and of the reference's code.

    python tools/ate.py groundtruth.txt estimate.txt [--max-dt 0.02] [--delta 1 --delta-unit f|s] [--json]
"""
import argparse
import json
import sys

import numpy as np


def read_trajectory(path):
    """-> (stamps [n], t [n,3], q [n,4] as qx qy qz qw); comment lines (#) and blank lines skipped"""
    st, t, q = [], [], []
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        v = line.replace(",", " ").split()
        if len(v) < 8:
            continue
        f = [float(x) for x in v[:8]]
        if any(np.isnan(f)):
            continue
        st.append(f[0]); t.append(f[1:4]); q.append(f[4:8])
    return np.array(st), np.array(t).reshape(-1, 3), np.array(q).reshape(-1, 4)


def quat_to_rot(q):
    """[n,4] (qx qy qz qw) -> [n,3,3]"""
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def associate(st_a, st_b, max_dt=0.02, offset=0.0):
    """greedy best-first matching of time stamps (the benchmark's associate.py semantics): pairs (i, j) with |a_i - (b_j + offset)| < max_dt,
    every stamp used at most once, closest pairs first"""
    cand = []
    jb = np.searchsorted(st_b + offset, st_a)
    for i, a in enumerate(st_a):
        for j in (jb[i] - 1, jb[i]):
            if 0 <= j < len(st_b):
                d = abs(a - (st_b[j] + offset))
                if d < max_dt:
                    cand.append((d, i, j))
    cand.sort()
    used_a, used_b, out = set(), set(), []
    for d, i, j in cand:
        if i not in used_a and j not in used_b:
            used_a.add(i); used_b.add(j); out.append((i, j))
    out.sort()
    return out


def align_rigid(model, data):
    """least-squares rigid transform (R, t) with R model_k + t ~ data_k (Horn 1987 via SVD, no scale); model / data: [n,3]"""
    mu_m, mu_d = model.mean(0), data.mean(0)
    H = (model - mu_m).T @ (data - mu_d)
    U, _, Vt = np.linalg.svd(H)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = Vt.T @ S @ U.T
    return R, mu_d - R @ mu_m


def ate(gt, est, max_dt=0.02, offset=0.0):
    """gt / est: (stamps, t, q).  -> dict(rmse, mean, median, std, min, max, pairs, R, t)"""
    pairs = associate(gt[0], est[0], max_dt, offset)
    if len(pairs) < 3:
        raise ValueError(f"only {len(pairs)} associated poses (max_dt = {max_dt})")
    g = gt[1][[i for i, _ in pairs]]; e = est[1][[j for _, j in pairs]]
    R, t = align_rigid(e, g)
    err = np.linalg.norm((e @ R.T + t) - g, axis=1)
    return dict(rmse=float(np.sqrt(np.mean(err ** 2))), mean=float(err.mean()), median=float(np.median(err)), std=float(err.std()),
                min=float(err.min()), max=float(err.max()), pairs=len(pairs), R=R, t=t)


def _T(R, t):
    T = np.tile(np.eye(4), (len(t), 1, 1))
    T[:, :3, :3] = R; T[:, :3, 3] = t
    return T


def rpe(gt, est, delta=1.0, unit="f", max_dt=0.02, offset=0.0):
    """relative pose error over `delta` frames (unit 'f', in associated estimate frames) or seconds (unit 's').
    -> dict(trans_rmse [m], rot_rmse [rad], trans_mean, rot_mean, pairs)"""
    pairs = associate(gt[0], est[0], max_dt, offset)
    if len(pairs) < 2:
        raise ValueError("too few associated poses")
    gi = [i for i, _ in pairs]; ej = [j for _, j in pairs]
    Tg = _T(quat_to_rot(gt[2][gi]), gt[1][gi]); Te = _T(quat_to_rot(est[2][ej]), est[1][ej])
    st = est[0][ej]
    n = len(pairs)
    et, er = [], []
    for k in range(n):
        if unit == "f":
            k2 = k + int(delta)
            if k2 >= n:
                break
        else:
            k2 = int(np.searchsorted(st, st[k] + delta))
            if k2 >= n or abs(st[k2] - st[k] - delta) > max(0.5 * delta, max_dt):
                continue
        dg = np.linalg.inv(Tg[k]) @ Tg[k2]; de = np.linalg.inv(Te[k]) @ Te[k2]
        E = np.linalg.inv(dg) @ de
        et.append(np.linalg.norm(E[:3, 3]))
        er.append(np.arccos(np.clip((np.trace(E[:3, :3]) - 1) / 2, -1, 1)))
    if not et:
        raise ValueError("no pose pairs at this delta")
    et, er = np.array(et), np.array(er)
    return dict(trans_rmse=float(np.sqrt(np.mean(et ** 2))), rot_rmse=float(np.sqrt(np.mean(er ** 2))), trans_mean=float(et.mean()),
                rot_mean=float(er.mean()), pairs=len(et))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("groundtruth"); ap.add_argument("estimate")
    ap.add_argument("--max-dt", type=float, default=0.02)
    ap.add_argument("--offset", type=float, default=0.0, help="time offset added to the estimate's stamps")
    ap.add_argument("--delta", type=float, default=1.0); ap.add_argument("--delta-unit", default="f", choices=["f", "s"])
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    gt, est = read_trajectory(args.groundtruth), read_trajectory(args.estimate)
    a = ate(gt, est, args.max_dt, args.offset)
    r = rpe(gt, est, args.delta, args.delta_unit, args.max_dt, args.offset)
    out = {"ate": {k: v for k, v in a.items() if k not in ("R", "t")}, "rpe": r}
    if args.json:
        print(json.dumps(out))
    else:
        print(f"ATE rmse {a['rmse']:.6f} m (mean {a['mean']:.6f}, median {a['median']:.6f}, max {a['max']:.6f}) over {a['pairs']} pose pairs")
        print(f"RPE (delta {args.delta:g} {args.delta_unit}) translation rmse {r['trans_rmse']:.6f} m, rotation rmse {np.degrees(r['rot_rmse']):.4f} deg over {r['pairs']} pairs")
    return 0


if __name__ == "__main__":
    sys.exit(main())
