"""Generates tests/golden/golden_v1.npz: small seeded input/output vectors of every kernel on the path plus a short
tracked trajectory.  The reference itself cannot be built or run in this image (CUDA + Eigen + Boost + PCL), so the
vectors come from the CPU oracle (oracle/), which is pinned by tests/test_oracle_kat.py; they freeze the oracle's
behaviour (any later change to it shows up as a diff) and give the GPU tests a fixed target that needs no oracle run.

    python tests/golden/make_golden.py      (run from the repo root; inputs are regenerated from seeds by the tests)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rgbid-slam_amd"))
from oracle import oracle as O  # noqa: E402
from tests import util  # noqa: E402

ROWS, COLS = 48, 64
K = tuple(v * COLS / 640 for v in (525.0, 525.0, 319.5, 239.5))


def inputs():
    r = util.rng(4242)
    d = dict()
    d["depth_u16"] = r.integers(400, 6000, (ROWS, COLS)).astype(np.uint16); d["depth_u16"][r.random((ROWS, COLS)) < 0.1] = 0
    d["rgb"] = r.integers(0, 256, (ROWS, COLS, 3)).astype(np.uint8)
    d["W0"] = util.rand_invdepth(r, ROWS, COLS, 0.06); d["Wc"] = util.rand_invdepth(r, ROWS, COLS, 0.06)
    d["I0"] = util.rand_intensity(r, ROWS, COLS); d["Ic"] = util.rand_intensity(r, ROWS, COLS)
    R, t = util.small_motion(r, K, 0.02, 1.0)
    d["Rp"], d["tp"] = util.project(K, *util.inv_pose(R, t))
    d["err"] = (0.003 * r.standard_t(4, 19200)).astype(np.float32); d["err"][::97] = np.nan
    return d


def outputs(d):
    o = dict()
    o["invdepth"] = O.depth2invdepth(d["depth_u16"], 1.0)
    o["intensity"] = O.intensity(d["rgb"])
    o["gx"], o["gy"] = O.gradient(d["W0"])
    o["pyr"] = O.pyr_down(d["W0"])
    o["bilat"] = O.bilateral(d["W0"], 0.005)
    o["W1"] = O.warp_invdepth(d["Wc"], d["W0"], d["Rp"], d["tp"])
    o["I1_tex8"] = O.warp_intensity(d["Ic"], o["W1"], d["Rp"], d["tp"], O.INTERP_TEX8)
    o["I1_exact"] = O.warp_intensity(d["Ic"], o["W1"], d["Rp"], d["tp"], O.INTERP_EXACT)
    o["Ww"], o["Wwt"] = O.warp_invdepth_weighted(d["Wc"], d["W0"], d["Rp"], d["tp"])
    o["fused"], o["fused_w"] = O.integrate_warped(o["Ww"], o["Wwt"], d["W0"], np.ones_like(d["W0"]))
    ratio, nvis, nval, mask = O.visibility_ratio(d["Wc"], o["W1"], d["Rp"], d["tp"], with_mask=True)
    o["vis"] = np.array([ratio, nvis, nval], np.float64); o["mask"] = mask
    o["vmap"] = O.vmap(d["W0"], K); o["nmap"] = O.nmap_gradients(d["W0"], o["gx"], o["gy"], K)
    gix, giy = O.gradient(d["I0"])
    A, b = O.build_system(d["W0"], d["I0"], o["gx"], o["gy"], gix, giy, o["W1"], o["I1_tex8"], K, sigma_depthinv=0.003, sigma_int=6.0,
                          bias_depthinv=1e-4, bias_int=0.3, nu_depthinv=3.5, nu_int=6.0)
    o["A"], o["b"] = A, b
    o["sigma_nu"] = np.array(O.sigma_nu_student(d["err"], 0.0, 0.0025, 5.0, O.STUDENT), np.float64)
    o["sigma_pdf_huber"] = np.array(O.sigma_pdf(d["err"], 0.0, 0.0025, O.HUBER), np.float64)
    o["chi"] = np.array(O.chi_square(d["err"] * 1000, d["err"], 5.0, 0.0025, O.STUDENT), np.float64)
    return o


def trajectory():
    from rgbid import synth
    Ks = (131.25, 131.25, 79.5, 59.5)
    seq = synth.make_sequence(6, K=Ks, rows=120, cols=160, trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8))
    d = seq["depth"].numpy().astype(np.uint16); c = seq["rgb"].numpy()
    trk = O.Tracker(O.default_config(rows=120, cols=160, fx=Ks[0], fy=Ks[1], cx=Ks[2], cy=Ks[3]))
    for k in range(6):
        trk.track(d[k], c[k])
    R, t = trk.poses()
    return dict(traj_depth=d, traj_rgb=c, traj_R=R, traj_t=t, traj_kf_checksum=np.array([np.nansum(trk.kf_depthinv().astype(np.float64))]))


if __name__ == "__main__":
    out = {"out_" + k: v for k, v in outputs(inputs()).items()}
    out.update(trajectory())
    path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
