"""Generates tests/golden/golden_v2.npz + tests/golden/tum_mini/: fixtures for the "next" rows of SURVEY 8 --
f-1 KeyframeAlign, f-4 TUM-layout dataset I/O, f-5 custom-calibration front-end.  As for golden_v1 the vectors come from the CPU
oracle (the reference cannot be built in this image); the PNG files of tum_mini are written by an independent pure-Python
encoder (tests/test_cpu_tum_io._py_png, all five filter types) so the product's decoder is checked against files it did not write.

    python tests/golden/make_golden_v2.py      (from the repo root)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rgbid-slam_amd"))
from oracle import oracle as O  # noqa: E402
from tests import util  # noqa: E402
from tests.golden import make_golden as G1  # noqa: E402

KD = (0.12, -0.25, 0.0015, -0.0008, 0.09)
KRGB = G1.K + KD
KDEPTH = (G1.K[0] * 1.1, G1.K[1] * 1.1, G1.K[2] - 0.6, G1.K[3] + 0.4) + tuple(-v * 0.5 for v in KD)
DIST = dict(c1=1.03, c0=-0.004, q0=(0.002, -0.004, 0.003, -0.001, 0.0007, -0.0005, 0.0011, -0.0009, 0.0004),
            q1=(0.01, 0.02, -0.015, 0.004, -0.003, 0.002, 0.006, -0.005, 0.001))


def calib_case():
    d = G1.inputs()
    r = util.rng(4343)
    Kc = np.array([[G1.K[0], 0, G1.K[2]], [0, G1.K[1], G1.K[3]], [0, 0, 1]], np.float32)
    Kd = np.array([[KDEPTH[0], 0, KDEPTH[2]], [0, KDEPTH[1], KDEPTH[3]], [0, 0, 1]], np.float32)
    R, _ = util.small_motion(r, G1.K, 0.0, 1.5)
    dRc_proj = ((Kd @ R.astype(np.float32)) @ np.linalg.inv(Kc).astype(np.float32)).astype(np.float32)
    cRd_proj = np.linalg.inv(dRc_proj.astype(np.float64)).astype(np.float32)
    t_proj = (Kd @ np.array([0.025, -0.003, 0.004], np.float32)).astype(np.float32)
    return d, dRc_proj, t_proj, cRd_proj


def calib_outputs():
    d, dRc_proj, t_proj, cRd_proj = calib_case()
    o = dict(dRc_proj=dRc_proj, t_proj=t_proj, cRd_proj=cRd_proj)
    o["und_I"] = O.undistort_intensity(d["I0"], KRGB)
    o["corr_W"], o["und_W"] = O.undistort_depthinv(d["W0"], KDEPTH, O.depth_dist(**DIST))
    o["reg_inter"], o["reg_W"] = O.register_depthinv(d["W0"], dRc_proj, t_proj, cRd_proj)
    return o


def kfalign_case():
    from rgbid import synth
    Ks = (131.25, 131.25, 79.5, 59.5)
    seq = synth.make_sequence(4, K=Ks, rows=120, cols=160, trans_step=(0.01, 0.02), rot_step_deg=(0.5, 1.0))
    d = seq["depth"].numpy().astype(np.uint16); c = seq["rgb"].numpy()
    iD = [O.depth2invdepth(d[k]) for k in (0, 3)]
    grey = [np.clip(np.rint(O.intensity(c[k])), 0, 255).astype(np.uint8) for k in (0, 3)]
    return Ks, iD, grey


def kfalign_outputs():
    Ks, iD, grey = kfalign_case()
    R, t, cov = O.keyframe_align(iD[0], grey[0], iD[1], grey[1], Ks)
    return dict(ka_iD0=iD[0], ka_iD1=iD[1], ka_grey0=grey[0], ka_grey1=grey[1], ka_R=R, ka_t=t, ka_cov=cov)


def tum_mini(root):
    from tests.test_cpu_tum_io import _py_png
    r = util.rng(4444)
    rows, cols, n = 24, 32, 3
    os.makedirs(os.path.join(root, "depth"), exist_ok=True); os.makedirs(os.path.join(root, "rgb"), exist_ok=True)
    dl, cl, depth, rgb, stamps = [], [], [], [], []
    for k in range(n):
        d = r.integers(0, 50001, (rows, cols)).astype(np.uint16); d[0, :4] = [0, 2, 3, 65535]
        c = r.integers(0, 256, (rows, cols, 3)).astype(np.uint8)
        st = 1305031102.175304 + k / 30.0
        open(os.path.join(root, "depth", f"{st:.6f}.png"), "wb").write(_py_png(d[:, :, None], 16, 0, filters=[0, 1, 2, 3, 4], idat_split=2))
        open(os.path.join(root, "rgb", f"{st:.6f}.png"), "wb").write(_py_png(c, 8, 2, filters=[4, 3, 2, 1, 0]))
        dl.append(f"{st:.6f} depth/{st:.6f}.png"); cl.append(f"{st:.6f} rgb/{st:.6f}.png")
        depth.append(d); rgb.append(c); stamps.append(float("%.6f" % st))
    hdr = "# depth maps\n# file: 'tum_mini'\n# timestamp filename\n"
    open(os.path.join(root, "depth_associated.txt"), "w").write(hdr + "\n".join(dl) + "\n")
    open(os.path.join(root, "rgb_associated.txt"), "w").write(hdr.replace("depth maps", "color images") + "\n".join(cl) + "\n")
    # expected decode: depth x0.2 rounded to nearest (mm), rgb as stored
    exp_mm = np.rint(np.stack(depth).astype(np.float64) * 0.2).astype(np.uint16)
    # expected trajectory lines for three known poses (Eigen::Quaternionf convention x y z w, fixed 6 decimals)
    from scipy.spatial.transform import Rotation
    Rs = [Rotation.from_euler("xyz", [0.1 * k, -0.05 * k, 0.02 * k]).as_matrix() for k in range(n)]
    ts = [np.array([0.01 * k, 0.2, -0.3 * k]) for k in range(n)]
    lines = []
    for k in range(n):
        q = Rotation.from_matrix(Rs[k]).as_quat()
        if q[3] < 0: q = -q
        lines.append(" ".join(["%.6f" % stamps[k]] + ["%.6f" % np.float32(v) for v in ts[k]] + ["%.6f" % np.float32(v) for v in q]))
    return dict(tum_depth_mm=exp_mm, tum_rgb=np.stack(rgb), tum_stamps=np.array(stamps), tum_R=np.stack(Rs), tum_t=np.stack(ts),
                tum_lines=np.array(lines))


if __name__ == "__main__":
    out = {"cal_" + k: v for k, v in calib_outputs().items()}
    out.update(kfalign_outputs())
    out.update(tum_mini(os.path.join(ROOT, "tests", "golden", "tum_mini")))
    path = os.path.join(ROOT, "tests", "golden", "golden_v2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
