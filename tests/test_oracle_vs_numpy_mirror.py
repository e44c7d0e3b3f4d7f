"""Cross-check of the C oracle (oracle/) against tests/np_mirror.py, a second restatement of the same reference kernels written in
vectorised numpy from the reference sources.  The two share no code and evaluate in different precision (fp32 in reference order vs
float64), so they agree to rounding -- and, where a kernel selects pixels by floor / round, on all but the handful of samples that sit
within rounding of a selection boundary.  A transcription error in either shows up as a gross mismatch."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import np_mirror as M
from tests import util

K = (131.25, 131.25, 79.5, 59.5)
ROWS, COLS = 120, 160


def maps(seed, nan_frac=0.05):
    r = util.rng(seed)
    w = util.rand_invdepth(r, ROWS, COLS, nan_frac=nan_frac)
    i = (r.random((ROWS, COLS)) * 255).astype(np.float32)
    return r, w, i


def close(a, b, rtol, atol=0.0, max_bad=0):
    """same NaN pattern and values within tolerance, except at most max_bad entries"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    bad = np.isnan(a) != np.isnan(b)
    both = ~np.isnan(a) & ~np.isnan(b)
    bad |= both & (np.abs(np.where(both, a - b, 0.0)) > atol + rtol * np.abs(np.where(both, b, 0.0)))
    assert bad.sum() <= max_bad, (int(bad.sum()), max_bad)


def test_conversions():
    r = util.rng(1)
    d = r.integers(0, 12000, (ROWS, COLS)).astype(np.uint16); d[r.random((ROWS, COLS)) < 0.1] = 0
    close(O.depth2invdepth(d), M.depth_to_invdepth(d), 1e-6)
    close(O.depth2invdepth(d, 5.0), M.depth_to_invdepth(d, 5.0), 1e-6)
    rgb = r.integers(0, 256, (ROWS, COLS, 3)).astype(np.uint8)
    close(O.intensity(rgb), M.intensity(rgb), 1e-6, 1e-4)


@pytest.mark.parametrize("nan_frac", [0.0, 0.05, 0.5])
def test_pyrdown_sobel_bilateral(nan_frac):
    r, w, i = maps(2, nan_frac)
    close(O.pyr_down(w), M.pyr_down(w), 2e-6)
    close(O.pyr_down(i), M.pyr_down(i), 2e-6, 1e-4)
    gx, gy = O.gradient(w); mx, my = M.sobel(w)
    close(gx, mx, 1e-5, 1e-7); close(gy, my, 1e-5, 1e-7)
    close(O.bilateral(w, 2 * 0.0025), M.bilateral(w, 2 * 0.0025), 2e-5)
    close(O.bilateral(i, 3.0), M.bilateral(i, 3.0), 2e-5, 1e-3)


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_warps_visibility_fusion(seed):
    r, w0, i0 = maps(seed)
    _, wc, _ = maps(seed + 100)
    ic = util.rand_intensity(r, ROWS, COLS)      # smooth texture: fp32 coordinates are good to ~1e-5 px, i.e. ~1e-3 grey levels here
    R, t = util.small_motion(r, K, 0.03, 1.5)
    Rp, tp = util.project(K, R, t)
    # point-sampled inverse-depth warp: a sample within rounding of a pixel boundary may pick the neighbour
    ow = O.warp_invdepth(wc, w0, Rp, tp); mw = M.warp_invdepth(wc, w0, Rp, tp)
    close(ow, mw, 2e-5, max_bad=8)
    for mode, tex8 in ((O.INTERP_EXACT, False), (O.INTERP_TEX8, True)):
        oi = O.warp_intensity(ic, ow, Rp, tp, mode); mi = M.warp_intensity(ic, ow, Rp, tp, tex8)
        close(oi, mi, 1e-5, 5e-3 if not tex8 else 1.0, max_bad=8)        # a 1.8 fixed-point weight flips by 1/256 at its rounding boundary
        assert np.nanmedian(np.abs(oi - mi)) < 1e-3
    ratio, nvis, nval, mask = O.visibility_ratio(w0, wc, Rp, tp, with_mask=True)
    mratio, vis, valid = M.visibility_ratio(w0, wc, Rp, tp)
    assert abs(ratio - mratio) < 5e-4 and nval == valid.sum() and abs(nvis - vis.sum()) <= 4
    assert np.count_nonzero(mask[valid].astype(bool) != vis[valid]) <= 4
    # fusion update
    warped = ow; ww = (r.random((ROWS, COLS)) + 0.5).astype(np.float32)
    kf = util.rand_invdepth(r, ROWS, COLS, nan_frac=0.2); kfw = (r.random((ROWS, COLS)) * 3 + 1).astype(np.float32)
    ok_, okw = O.integrate_warped(warped, ww, kf, kfw)
    mk, mkw = M.integrate(warped, ww, kf, kfw)
    close(ok_, mk, 1e-6, max_bad=2); close(okw, mkw, 1e-6, max_bad=2)


@pytest.mark.parametrize("student_nu,mest,weighting", [(True, 3, 0), (True, 3, 1), (False, 0, 0), (False, 1, 2), (False, 2, 3), (False, 3, 0)])
def test_normal_equations(student_nu, mest, weighting):
    """the 27-term system of estimate_VO.cu against the numpy restatement, for the Student-nu path and every fixed-nu M-estimator / weighting"""
    r, W0, _ = maps(11)
    I0 = util.rand_intensity(r, ROWS, COLS); I0[r.random((ROWS, COLS)) < 0.02] = np.nan
    gwx, gwy = O.gradient(W0); gix, giy = O.gradient(I0)
    W1 = (W0 * (1 + 0.01 * r.standard_normal((ROWS, COLS)))).astype(np.float32); W1[r.random((ROWS, COLS)) < 0.05] = np.nan
    I1 = (I0 + 4 * r.standard_normal((ROWS, COLS))).astype(np.float32)
    kw = dict(sigma_depthinv=0.004, sigma_int=6.0, bias_depthinv=0.0003, bias_int=-0.4, nu_depthinv=3.5, nu_int=7.0)
    A, b = O.build_system(W0, I0, gwx, gwy, gix, giy, W1, I1, K, student_nu=student_nu, mestimator=mest, weighting=weighting, **kw)
    Am, bm = M.build_system(W0, I0, gwx, gwy, gix, giy, W1, I1, K, kw["sigma_depthinv"], kw["sigma_int"], kw["bias_depthinv"], kw["bias_int"],
                            kw["nu_depthinv"], kw["nu_int"], student_nu=student_nu, mestimator=mest, weighting=weighting)
    sc = np.sqrt(np.outer(np.diag(Am), np.diag(Am)))
    assert np.abs(A - Am).max() / sc.max() < 1e-5 and (np.abs(A - Am) / sc).max() < 1e-4, (np.abs(A - Am) / sc).max()
    assert np.abs(b - bm).max() < 2e-5 * np.abs(bm).max(), np.abs(b - bm).max() / np.abs(bm).max()
    assert np.allclose(A, A.T) and np.linalg.eigvalsh(Am).min() > 0


@pytest.mark.parametrize("dof,scale,bias", [(3.0, 0.004, 0.0005), (5.0, 4.0, -0.7), (8.0, 0.01, 0.0), (30.0, 2.0, 0.3), (2.2, 1.0, 0.0)])
def test_sigma_nu_estimator(dof, scale, bias):
    """IRLS scale / bias and the Student-t degrees of freedom by bisection (sigmaFuncs.cu:858-1066) against the numpy restatement, on
    Student-t distributed residuals contaminated with NaN and infinities"""
    r = util.rng(int(dof * 10))
    e = (bias + scale * r.standard_t(dof, 19200)).astype(np.float32)
    e[r.random(e.size) < 0.03] = np.nan
    e[:5] = np.inf; e[5:9] = -np.inf
    start_sigma = 0.0025 if scale < 0.1 else 5.0
    ob, os_, onu = O.sigma_nu_student(e, 0.0, start_sigma)
    mb, ms, mnu = M.sigma_nu_student(e, 0.0, start_sigma)
    assert abs(os_ - ms) < 2e-4 * ms and abs(ob - mb) < 2e-4 * ms, (ob, mb, os_, ms)
    assert onu == mnu, (onu, mnu)
    assert abs(ms - scale * (np.sqrt(dof / (dof - 2)) if dof > 2.5 else 1.0)) / ms < 1.0      # sanity: the right order of magnitude


def test_maps_and_weighted_warp():
    r, w0, _ = maps(21)
    _, wc, _ = maps(22)
    ov = O.vmap(w0, K).reshape(3, ROWS, COLS); mv = M.vmap(w0, K)
    valid = ~np.isnan(w0)
    close(ov[0], mv[0], 2e-6, 1e-6)
    close(ov[1][valid], mv[1][valid], 2e-6, 1e-6); close(ov[2][valid], mv[2][valid], 2e-6)      # planes 1, 2 are untouched where invalid
    gx, gy = O.gradient(w0)
    on = O.nmap_gradients(w0, gx, gy, K).reshape(3, ROWS, COLS); mn = M.nmap_gradients(w0, gx, gy, K)
    keep = ~np.isnan(mn[0])
    assert np.count_nonzero(np.isnan(on[0]) != np.isnan(mn[0])) <= 2                             # the 0.1 cosine gate
    both = keep & ~np.isnan(on[0])
    assert np.abs(on[:, both] - mn[:, both]).max() < 5e-6
    R, t = util.small_motion(r, K, 0.03, 1.5)
    Rp, tp = util.project(K, R, t)
    ow, owt = O.warp_invdepth_weighted(wc, w0, Rp, tp, weight_init=np.full((ROWS, COLS), np.nan, np.float32))
    mw, mwt = M.warp_invdepth_weighted(wc, w0, Rp, tp)
    close(ow, mw, 2e-5, max_bad=8); close(owt, mwt, 1e-4, max_bad=8)


@pytest.mark.parametrize("warp_first", [False, True])
def test_gauss_newton_alignment_of_a_frame_pair(warp_first):
    """the whole coarse-to-fine estimateVisualOdometry (visodo.cpp:1041-1263) rebuilt from the mirror's kernels against the oracle's: the two
    recover the same relative pose (to the float32-vs-float64 difference of two 18-iteration Gauss-Newton runs), and it is the synthetic
    camera's true motion to sensor-noise accuracy"""
    from rgbid import synth
    seq = synth.make_sequence(2, K=K, rows=ROWS, cols=COLS, trans_step=(0.01, 0.015), rot_step_deg=(0.5, 0.8))
    d = seq["depth"].numpy().astype(np.uint16); c = seq["rgb"].numpy()
    cfg = O.default_config(rows=ROWS, cols=COLS, fx=K[0], fy=K[1], cx=K[2], cy=K[3], motion_model=O.NO_MM,
                           warping=O.WARP_FIRST if warp_first else O.PYR_FIRST)
    ok, Ro, to, _ = O.align_pair(cfg, d[0], c[0], d[1], c[1])
    assert ok
    Rm, tm = M.align_pair(d[0], c[0], d[1], c[1], K, warp_first=warp_first)
    ang = lambda A, B: float(np.arccos(np.clip((np.trace(A.T @ B) - 1) / 2, -1, 1)))
    assert ang(Ro, Rm) < 5e-6 and np.linalg.norm(to - tm) < 5e-6, (ang(Ro, Rm), np.linalg.norm(to - tm))     # measured 1e-7 rad / 3e-7 m
    # ground truth: pose of camera 1 in camera 0
    R_wc, t_wc = seq["R_wc"].numpy(), seq["t_wc"].numpy()
    Rg = R_wc[0].T @ R_wc[1]; tg = R_wc[0].T @ (t_wc[1] - t_wc[0])
    assert ang(Rm, Rg) < 3e-3 and np.linalg.norm(tm - tg) < 5e-3


def test_tracker_frame_loop():
    """trackNewFrame over a short sequence that switches both keyframes: the mirror's tracker (constant-velocity prediction, Gauss-Newton,
    covariance pass, covisibility, keyframe switching, keyframe fusion -- all from np_mirror) against the oracle tracker: same keyframe
    decisions, same covisibility ratios, same poses and fused keyframe map"""
    from rgbid import synth
    n = 9
    seq = synth.make_sequence(n, K=K, rows=ROWS, cols=COLS, trans_step=(0.005, 0.012), rot_step_deg=(0.2, 0.6))
    d = seq["depth"].numpy().astype(np.uint16); c = seq["rgb"].numpy()
    th = dict(visratio_odo=0.95, visratio_integr=0.915)          # odometry keyframe switches on 4 frames, integration keyframe on 2, fusion on 6
    orc = O.Tracker(O.default_config(rows=ROWS, cols=COLS, fx=K[0], fy=K[1], cx=K[2], cy=K[3], **th))
    mir = M.Tracker(K, ROWS, COLS, **th)
    ang = lambda A, B: float(np.arccos(np.clip((np.trace(A.T @ B) - 1) / 2, -1, 1)))
    switches = [0, 0]
    for k in range(n):
        orc.track(d[k], c[k]); mir.track(d[k], c[k])
        if k == 0:
            continue
        oi, mi = orc.last_info(), mir.info[-1]
        assert abs(oi.visratio_odo - mi["vis_odo"]) < 5e-4 and abs(oi.visratio_integr - mi["vis_int"]) < 5e-4, k
        assert bool(oi.odo_kf_switched) == mi["sw_odo"] and bool(oi.integr_kf_switched) == mi["sw_int"], k
        switches[0] += mi["sw_odo"]; switches[1] += mi["sw_int"]
        oc = np.array(oi.delta_cov).reshape(6, 6)                      # covariance of the keyframe-relative pose estimateVisualOdometry returned
        sc = np.sqrt(np.outer(np.diag(mi["cov"]), np.diag(mi["cov"])))
        assert (np.abs(oc - mi["cov"]) / sc).max() < 1e-4, k
    assert 1 <= switches[0] < n - 1 and 1 <= switches[1] < n - 1                      # every branch (switch / keep / fuse) exercised
    Ro, to = orc.poses()
    for k in range(n):
        assert ang(Ro[k], mir.poses[k][0]) < 1e-5 and np.linalg.norm(to[k] - mir.poses[k][1]) < 1e-5, k      # measured 1.4e-6 rad / 2.7e-6 m
    kd, kw = orc.kf_depthinv(), orc.kf_weight()
    close(kd, mir.int_w, 1e-4, max_bad=max(16, int(5e-3 * kd.size)))
    close(kw, mir.int_weight, 1e-3, max_bad=max(16, int(1e-2 * kd.size)))


def test_custom_calibration_front_end():
    """undistortIntensity / undistortDepthInv (src/cuda/undistortion.cu) with the lens and depth-distortion parameters of the golden fixture"""
    from tests.golden import make_golden_v2 as G2
    r, w, _ = maps(31)
    i = util.rand_intensity(r, ROWS, COLS)
    krgb = (K[0], K[1], K[2], K[3]) + G2.KD
    for tex8, mode in ((False, O.INTERP_EXACT), (True, O.INTERP_TEX8)):
        close(O.undistort_intensity(i, krgb, mode), M.undistort_intensity(i, krgb, tex8), 1e-5, 5e-3 if not tex8 else 1.0, max_bad=8)
    kd = (K[0] * 1.1, K[1] * 1.1, K[2] - 0.6, K[3] + 0.4) + tuple(-v * 0.5 for v in G2.KD)
    D = G2.DIST
    dd = O.depth_dist(c1=D["c1"], c0=D["c0"], q0=D["q0"], q1=D["q1"])
    ocorr, oout = O.undistort_depthinv(w, kd, dd)
    mcorr, mout = M.undistort_depthinv(w, kd, D["c1"], D["c0"], D["q0"], D["q1"], 4, 4)
    close(ocorr, mcorr, 2e-6, 1e-7); close(oout, mout, 2e-6, 1e-7, max_bad=8)


def test_keyframe_align():
    """KeyframeAlign::alignKeyframes (src/keyframe_align.cpp:115-357) rebuilt from the mirror's kernels against the oracle's restatement:
    same pose and covariance; both find the true relative pose of two frames three steps apart"""
    from rgbid import synth
    rows, cols = 240, 320                      # four pyramid levels down to 30 x 40; the finest lattice is 160 x 120 = 19 200 samples
    Kq = (262.5, 262.5, 159.5, 119.5)
    seq = synth.make_sequence(4, K=Kq, rows=rows, cols=cols, trans_step=(0.008, 0.015), rot_step_deg=(0.3, 0.8))
    d = seq["depth"].numpy().astype(np.uint16); c = seq["rgb"].numpy()
    iD = [O.depth2invdepth(d[k]) for k in (0, 3)]
    grey = [np.clip(np.rint(O.intensity(c[k])), 0, 255).astype(np.uint8) for k in (0, 3)]
    Ro, to, covo = O.keyframe_align(iD[0], grey[0], iD[1], grey[1], Kq)
    Rm, tm, covm = M.keyframe_align(iD[0], grey[0], iD[1], grey[1], Kq)
    ang = lambda A, B: float(np.arccos(np.clip((np.trace(A.T @ B) - 1) / 2, -1, 1)))
    assert ang(Ro, Rm) < 1e-5 and np.linalg.norm(to - tm) < 1e-5, (ang(Ro, Rm), np.linalg.norm(to - tm))
    sc = np.sqrt(np.outer(np.diag(covm), np.diag(covm)))
    assert (np.abs(covo - covm) / sc).max() < 1e-3
    R_wc, t_wc = seq["R_wc"].numpy(), seq["t_wc"].numpy()
    Rg = R_wc[0].T @ R_wc[3]; tg = R_wc[0].T @ (t_wc[3] - t_wc[0])
    assert ang(Rm, Rg) < 5e-3 and np.linalg.norm(tm - tg) < 1.5e-2


def test_depth_to_colour_registration():
    """registerDepthinv (z-buffer splat with dilation + rotation homography) on the golden fixture's stereo calibration"""
    from tests.golden import make_golden_v2 as G2
    d, dRc_proj, t_proj, cRd_proj = G2.calib_case()
    w = d["W0"][:60, :80].copy()                               # the splat is a Python loop here: a crop keeps it to a second
    ointer, oout = O.register_depthinv(w, dRc_proj, t_proj, cRd_proj)
    minter, mout = M.register_depthinv(w, dRc_proj, t_proj, cRd_proj)
    close(ointer, minter, 1e-6, max_bad=4)
    close(oout, mout, 2e-6, max_bad=4)
    assert np.isfinite(oout).mean() > 0.5


@pytest.mark.parametrize("mest", [0, 1, 2, 3])
def test_sigma_pdf_and_chi_square(mest):
    """the fixed-M-estimator scale estimate (computeSigmaPdf) and the chi-square statistic (computeChiSquare) against the numpy restatement"""
    r = util.rng(40 + mest)
    ei = (0.5 + 6.0 * r.standard_t(4, 19200)).astype(np.float32); ed = (0.003 * r.standard_t(4, 19200)).astype(np.float32)
    ei[r.random(ei.size) < 0.02] = np.nan; ed[:3] = np.inf
    ob, os_ = O.sigma_pdf(ei, 0.0, 5.0, mest)[:2]
    mb, ms = M.sigma_pdf(ei, 0.0, 5.0, mest)
    assert abs(os_ - ms) < 2e-4 * ms and abs(ob - mb) < 2e-4 * ms, (ob, mb, os_, ms)
    ochi, otest, on = O.chi_square(ei, ed, 5.0, 0.0025, mest)
    mchi, mtest, mn = M.chi_square(ei, ed, 5.0, 0.0025, mest)
    assert on == mn and abs(ochi - mchi) < 1e-4 * mchi and abs(otest - mtest) < 1e-6


def test_preview_shading():
    """generateImageRGB on the vertex / normal maps of a random surface: identical bytes except where a product lands within rounding of x.5"""
    r, w, _ = maps(51)
    rgb = r.integers(0, 256, (ROWS, COLS, 3)).astype(np.uint8)
    vm = O.vmap(w, K); gx, gy = O.gradient(w); nm = O.nmap_gradients(w, gx, gy, K)
    light = np.array([0.3, -0.2, -0.5], np.float32)
    a = O.generate_image_rgb(vm, nm, rgb, light).astype(int); b = M.generate_image_rgb(vm, nm, rgb, light).astype(int)
    assert np.abs(a - b).max() <= 1 and np.count_nonzero(a != b) <= 1e-3 * a.size
    assert (a.sum(-1) > 0).mean() > 0.5
