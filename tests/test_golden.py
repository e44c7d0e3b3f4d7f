"""Golden-vector tests (tests/golden/golden_v1.npz, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces the committed vectors bit-for-bit.  GPU: the HIP kernels and the C++ / batched trackers
reproduce them without running the oracle (tolerances as in tests/test_gpu_kernels.py)."""
import os

import numpy as np
import pytest

from tests import util
from tests.golden import make_golden as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))


def test_oracle_reproduces_golden_vectors():
    o = G.outputs(G.inputs())
    for k, v in o.items():
        g = GOLD["out_" + k]
        assert np.array_equal(np.asarray(v), g, equal_nan=True), k


def test_oracle_reproduces_golden_trajectory():
    from oracle import oracle as O
    Ks = (131.25, 131.25, 79.5, 59.5)
    trk = O.Tracker(O.default_config(rows=120, cols=160, fx=Ks[0], fy=Ks[1], cx=Ks[2], cy=Ks[3]))
    for k in range(6):
        trk.track(GOLD["traj_depth"][k], GOLD["traj_rgb"][k])
    R, t = trk.poses()
    assert np.allclose(R, GOLD["traj_R"], atol=1e-12) and np.allclose(t, GOLD["traj_t"], atol=1e-12)


@pytest.mark.gpu
def test_hip_kernels_reproduce_golden_vectors(ctx):
    import torch
    d = G.inputs()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    new = lambda r=G.ROWS, c=G.COLS: torch.full((r, c), float("nan"), device="cuda")
    g = lambda k: GOLD["out_" + k]
    out = new(); ctx.convertDepth2InvDepth(dev(d["depth_u16"].view(np.int16)), out, 1.0); util.assert_bits(out.cpu().numpy(), g("invdepth"), 0, "invdepth")
    out = new(); ctx.computeIntensity(dev(d["rgb"]), out); util.assert_bits(out.cpu().numpy(), g("intensity"), 0, "intensity")
    gx, gy = new(), new(); ctx.computeGradient(dev(d["W0"]), gx, gy)
    util.assert_bits(gx.cpu().numpy(), g("gx"), 0, "gx"); util.assert_bits(gy.cpu().numpy(), g("gy"), 0, "gy")
    p = new(G.ROWS // 2, G.COLS // 2); ctx.pyrDown(dev(d["W0"]), p); util.assert_bits(p.cpu().numpy(), g("pyr"), 4, "pyr")
    b = new(); ctx.bilateralFilter(dev(d["W0"]), b, 0.005); util.assert_bits(b.cpu().numpy(), g("bilat"), 8, "bilat")
    W1 = new(); ctx.warpInvDepthWithTrafo3D(dev(d["Wc"]), W1, dev(d["W0"]), d["Rp"], d["tp"]); util.assert_bits(W1.cpu().numpy(), g("W1"), 0, "W1")
    I1 = new(); ctx.warpIntensityWithTrafo3DInvDepth(dev(d["Ic"]), I1, W1, d["Rp"], d["tp"]); util.assert_bits(I1.cpu().numpy(), g("I1_tex8"), 0, "I1")
    Ww, Wt = new(), torch.zeros((G.ROWS, G.COLS), device="cuda")
    ctx.warpInvDepthWithTrafo3DWeighted(dev(d["Wc"]), Ww, dev(d["W0"]), Wt, d["Rp"], d["tp"])
    util.assert_bits(Ww.cpu().numpy(), g("Ww"), 0, "Ww"); util.assert_bits(Wt.cpu().numpy(), g("Wwt"), 0, "Wwt")
    kf, kw = dev(d["W0"].copy()), torch.ones((G.ROWS, G.COLS), device="cuda")
    ctx.integrateWarpedFrame(Ww, Wt, kf, kw); util.assert_bits(kf.cpu().numpy(), g("fused"), 0, "fused"); util.assert_bits(kw.cpu().numpy(), g("fused_w"), 0, "fw")
    m = torch.zeros((G.ROWS, G.COLS), dtype=torch.uint8, device="cuda")
    ratio = ctx.getVisibilityRatioWithOverlapMask(dev(d["Wc"]), W1, d["Rp"], d["tp"], overlap_mask=m)
    assert ratio == np.float32(g("vis")[0]) and np.array_equal(m.cpu().numpy(), g("mask"))
    gix, giy = new(), new(); ctx.computeGradient(dev(d["I0"]), gix, giy)
    A, bb = ctx.buildSystemStudentNuGridStride(dev(d["W0"]), dev(d["I0"]), gx, gy, gix, giy, W1, I1, 3, 0, 0.003, 6.0, 1e-4, 0.3, 3.5, 6.0, G.K)
    dg = np.sqrt(np.diag(g("A")))
    assert (np.abs(A - g("A")) <= 2e-5 * np.outer(dg, dg)).all()
    bs, ss, nn = ctx.computeSigmaAndNuStudent(dev(d["err"]), d["err"].size, 0.0, 0.0025, 5.0, 3)
    assert nn == g("sigma_nu")[2] and abs(ss - g("sigma_nu")[1]) <= 2e-5 * g("sigma_nu")[1]


@pytest.mark.gpu
def test_trackers_reproduce_golden_trajectory(ctx):
    import torch
    from rgbid import engine as E, host
    Ks = (131.25, 131.25, 79.5, 59.5)
    trk = host.Tracker(host.default_config(rows=120, cols=160, fx=Ks[0], fy=Ks[1], cx=Ks[2], cy=Ks[3]))
    eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=1, K=Ks, record_capacity=6))
    for k in range(6):
        trk.track(GOLD["traj_depth"][k], GOLD["traj_rgb"][k])
        eng.step(torch.from_numpy(GOLD["traj_depth"][k:k + 1].view(np.int16)).cuda(), torch.from_numpy(GOLD["traj_rgb"][k:k + 1]).cuda())
    R, t = trk.poses()
    rec = eng.records()
    for k in range(1, 6):
        for Rk, tk in ((R[k], t[k]), (rec[k, 0]["R"], rec[k, 0]["t"])):
            ang = np.arccos(np.clip((np.trace(Rk.T @ GOLD["traj_R"][k]) - 1) / 2, -1, 1))
            assert ang < 1e-4 and np.linalg.norm(tk - GOLD["traj_t"][k]) < 1e-4
    trk.close(); eng.close()


# ---- golden_v2: the "next" rows of SURVEY 8 (f-1 KeyframeAlign, f-4 dataset I/O, f-5 custom calibration) -----------------------
from tests.golden import make_golden_v2 as G2  # noqa: E402

GOLD2 = np.load(os.path.join(ROOT, "tests", "golden", "golden_v2.npz"))
TUM_MINI = os.path.join(ROOT, "tests", "golden", "tum_mini")


def test_oracle_reproduces_golden_v2():
    for k, v in G2.calib_outputs().items():
        assert np.array_equal(np.asarray(v), GOLD2["cal_" + k], equal_nan=True), k
    ka = G2.kfalign_outputs()
    for k in ("ka_iD0", "ka_iD1", "ka_grey0", "ka_grey1"):
        assert np.array_equal(ka[k], GOLD2[k], equal_nan=True), k
    assert np.allclose(ka["ka_R"], GOLD2["ka_R"], atol=1e-12) and np.allclose(ka["ka_t"], GOLD2["ka_t"], atol=1e-12)
    assert np.allclose(ka["ka_cov"], GOLD2["ka_cov"], rtol=1e-9)


def test_tum_mini_fixture_decodes_to_golden():
    """the committed PNG / association files (written by an independent encoder) through the product's dataset reader and
    trajectory formatter (host code only: no GPU needed)"""
    from rgbid import tum
    ds = tum.Dataset(TUM_MINI)
    n = len(GOLD2["tum_stamps"])
    assert len(ds) == n
    for k in range(n):
        assert ds.stamp(k) == GOLD2["tum_stamps"][k]
        d, c = ds.grab(k, 24, 32)
        assert np.array_equal(d, GOLD2["tum_depth_mm"][k]) and np.array_equal(c, GOLD2["tum_rgb"][k])
        assert tum.format_pose_line(ds.stamp(k), GOLD2["tum_R"][k], GOLD2["tum_t"][k]) == str(GOLD2["tum_lines"][k])


@pytest.mark.gpu
def test_hip_calibration_kernels_reproduce_golden_v2(ctx):
    import torch
    from rgbid import device
    d = G1_inputs = G.inputs()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    new = lambda r=G.ROWS, c=G.COLS: torch.full((r, c), float("nan"), device="cuda")
    g = lambda k: GOLD2["cal_" + k]
    out = new(); ctx.undistortIntensity(dev(d["I0"]), out, G2.KRGB); util.assert_bits(out.cpu().numpy(), g("und_I"), 0, "undistort I")
    corr, und = new(), new(); ctx.undistortDepthInv(dev(d["W0"]), corr, und, G2.KDEPTH, device.depth_dist(**G2.DIST))
    util.assert_bits(corr.cpu().numpy(), g("corr_W"), 0, "depth correction"); util.assert_bits(und.cpu().numpy(), g("und_W"), 0, "undistort iD")
    inter = new(3 * G.ROWS, 3 * G.COLS); inter_i = torch.zeros((3 * G.ROWS, 3 * G.COLS), dtype=torch.int32, device="cuda"); reg = new()
    ctx.registerDepthinv(dev(d["W0"]), inter, inter_i, reg, g("dRc_proj").reshape(9), g("t_proj"), g("cRd_proj").reshape(9))
    util.assert_bits(inter.cpu().numpy(), g("reg_inter"), 0, "registration canvas"); util.assert_bits(reg.cpu().numpy(), g("reg_W"), 0, "registered iD")


@pytest.mark.gpu
def test_keyframe_align_reproduces_golden_v2():
    from rgbid import host
    Ks = (131.25, 131.25, 79.5, 59.5)
    R, t, cov = host.keyframe_align(GOLD2["ka_iD0"], GOLD2["ka_grey0"], GOLD2["ka_iD1"], GOLD2["ka_grey1"], Ks)
    ang = float(np.arccos(np.clip((np.trace(R.T @ GOLD2["ka_R"]) - 1) / 2, -1, 1)))
    assert ang < 1e-4 and np.linalg.norm(t - GOLD2["ka_t"]) < 1e-4
    sc = np.sqrt(np.outer(np.diag(GOLD2["ka_cov"]), np.diag(GOLD2["ka_cov"])))
    assert (np.abs(cov - GOLD2["ka_cov"]) / sc).max() < 1e-2
