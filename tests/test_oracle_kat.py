"""Known-answer tests that pin the CPU oracle (the reference has no tests or golden vectors of its own, SURVEY.md
section 4): closed-form cases, scipy cross-checks for the third-party arithmetic the reference delegates
(boost digamma), hand-computed single-pixel systems, and an end-to-end pose recovery on noise-free synthetic data."""
import numpy as np
import pytest
import scipy.special

from oracle import oracle as O
from tests import util

K = (525.0, 525.0, 319.5, 239.5)


def test_digamma_vs_scipy():
    xs = np.linspace(1.0, 5.5, 20).tolist() + [0.5, 0.75, 10.0, 25.0]
    for x in xs:
        assert abs(O.digamma(x) - scipy.special.digamma(x)) <= 2e-7 * max(1.0, abs(scipy.special.digamma(x)))


def test_depth2invdepth_kat():
    d = np.array([[0, 1, 1000, 2000, 10000, 10001, 65535]], np.uint16)
    w = O.depth2invdepth(d, 1.0)[0]
    assert np.isnan(w[0]) and w[1] == 1000.0 and w[2] == 1.0 and w[3] == 0.5
    assert w[4] == np.float32(0.1) and w[5] == w[4] and w[6] == w[4]            # clamp at 10 m (misc.cu:120)
    assert O.depth2invdepth(d, 0.96)[0][2] == np.float32(np.float32(1.0) / np.float32(0.96)) * np.float32(1000) / np.float32(1000)


def test_intensity_kat():
    rgb = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.uint8)
    i = O.intensity(rgb)[0]
    assert i[0] <= 255.0 and abs(i[0] - 255.0) < 1e-3 and i[1] == 0
    assert np.allclose(i[2:], [0.2126 * 255, 0.7152 * 255, 0.0722 * 255], rtol=1e-6)


def test_sobel_ramp_and_nan():
    v, u = np.mgrid[0:20, 0:30].astype(np.float32)
    gx, gy = O.gradient(3 * u + 2 * v)
    assert np.all(gx[:, 1:-1] == 3.0) and np.all(gy[1:-1, :] == 2.0)
    assert np.all(gx[:, 0] == 1.5) and np.all(gy[0, :] == 1.0)                  # replicate border halves the derivative
    img = (3 * u + 2 * v).copy(); img[10, 10] = np.nan
    gx, gy = O.gradient(img)
    assert np.isnan(gx[9:12, 9:12]).all() and np.isnan(gy[9:12, 9:12]).all()    # incl. the zero-weight centre tap
    assert np.count_nonzero(np.isnan(gx)) == 9


def test_pyrdown_validity_and_constant():
    src = np.full((48, 64), 7.0, np.float32)
    d = O.pyr_down(src)
    assert d.shape == (24, 32)
    nan = np.isnan(d)
    assert nan[0, 0] and nan[0, -1] and nan[-1, 0] and not nan[-1, -1] and nan.sum() == 3    # 9 / 12 / 12 / 16 taps (App. A.3)
    assert np.allclose(d[~nan], 7.0, rtol=1e-6)
    src[10:16, 20:26] = np.nan                     # a 6x6 hole: outputs whose 5x5 window keeps <= 12 taps become NaN
    d2 = O.pyr_down(src)
    for y in range(24):
        for x in range(32):
            ys = slice(max(0, 2 * y - 2), min(2 * y + 3, 48)); xs = slice(max(0, 2 * x - 2), min(2 * x + 3, 64))
            cnt = np.count_nonzero(~np.isnan(src[ys, xs]))
            assert np.isnan(d2[y, x]) == (cnt <= 12)
    assert O.pyr_down(np.zeros((61, 83), np.float32)).shape == (30, 41)


def test_bilateral_constant_and_nan_centre():
    src = np.full((20, 20), 0.5, np.float32); src[5, 5] = np.nan
    d = O.bilateral(src, 0.005)
    assert np.isnan(d[5, 5]) and np.count_nonzero(np.isnan(d)) == 1 and np.allclose(d[~np.isnan(d)], 0.5, rtol=1e-6)


def test_lattice_always_160x120():
    for rows, cols in [(480, 640), (240, 320), (120, 160), (960, 1280)]:
        e, (lr, lc, st) = O.error_lattice(np.zeros((rows, cols), np.float32), np.zeros((rows, cols), np.float32), 10000)
        assert (lr, lc) == (120, 160) and e.size == 19200 and st * 160 == cols
    a = np.arange(48 * 64, dtype=np.float32).reshape(48, 64)
    e, (lr, lc, st) = O.error_lattice(a, np.zeros_like(a), 700)
    assert (lr, lc, st) == (24, 32, 2) and np.array_equal(e.reshape(24, 32), a[::2, ::2])


def test_identity_warp_and_plane_translation():
    rows, cols = 48, 64
    Ks = tuple(v * cols / 640 for v in K)
    r = util.rng(1)
    w0 = util.rand_invdepth(r, rows, cols, 0.1); i0 = util.rand_intensity(r, rows, cols)
    Rp, tp = util.project(Ks, np.eye(3), np.zeros(3))
    w1 = O.warp_invdepth(w0, w0, Rp, tp)
    assert np.array_equal(np.isnan(w1), np.isnan(w0)) and np.allclose(w1[~np.isnan(w0)], w0[~np.isnan(w0)], rtol=2e-6)
    i1 = O.warp_intensity(i0, w0, Rp, tp, O.INTERP_EXACT)
    assert np.allclose(i1[~np.isnan(w0)], i0[~np.isnan(w0)], atol=2e-3)
    # fronto-parallel plane z = 2 m seen by a camera that moved 0.1 m forward: X_cur = X_kf - t => z_cur = 1.9
    # the warp maps the CURRENT map (iD = 1/1.9) into the keyframe: every valid output must be exactly 1/2
    wk = np.full((rows, cols), 0.5, np.float32); wc = np.full((rows, cols), 1 / 1.9, np.float32)
    Rp, tp = util.project(Ks, *util.inv_pose(np.eye(3), np.array([0, 0, 0.1])))
    out = O.warp_invdepth(wc, wk, Rp, tp)
    assert np.allclose(out[~np.isnan(out)], 0.5, rtol=1e-5) and np.count_nonzero(~np.isnan(out)) > 0.8 * out.size


def test_point_sample_rounds_down_at_half():
    """xs = x' + 0.5 and the point sample takes floor(xs): a projection landing exactly on x.5 selects pixel x+1."""
    rows, cols = 8, 16
    src = np.arange(rows * cols, dtype=np.float32).reshape(rows, cols) / 100 + 1
    grid = np.ones((rows, cols), np.float32)
    R = np.eye(3, dtype=np.float32).reshape(9); t = np.array([0.5, 0, 0], np.float32)   # x' = x + 0.5 for iD = 1
    out = O.warp_invdepth(src, grid, R, t)
    assert np.allclose(out[2, 3], src[2, 4]) and np.isnan(out[2, cols - 1])


def test_fusion_and_visibility_gates():
    kf = np.array([[1.0, 1.0, 1.0, np.nan]], np.float32); kw = np.ones((1, 4), np.float32)
    warped = np.array([[1.02, 1.03, np.nan, 0.7]], np.float32); ww = np.full((1, 4), 3.0, np.float32)
    k2, w2 = O.integrate_warped(warped, ww, kf, kw)
    assert np.isclose(k2[0, 0], (1.0 + 3 * 1.02) / 4) and w2[0, 0] == 4.0      # |d| = 0.02 < 0.0225: fused
    assert k2[0, 1] == 1.0 and w2[0, 1] == 1.0                                # |d| = 0.03: occlusion gate, untouched
    assert k2[0, 2] == 1.0 and k2[0, 3] == np.float32(0.7) and w2[0, 3] == 3.0  # NaN warped ignored; NaN KF adopts the sample
    src = np.full((8, 8), 0.5, np.float32)
    dst = src.copy(); dst[4, 4] = 0.5 + 0.019; dst[4, 5] = 0.5 + 0.021
    ratio, nvis, nval, mask = O.visibility_ratio(src, dst, np.eye(3).reshape(9), np.zeros(3), with_mask=True)
    assert nval == 64 and mask[4, 4] == 1 and mask[4, 5] == 0 and mask[0, 0] == 0 and mask[1, 1] == 1   # border excluded: 0 < x' < cols-1
    assert nvis == 36 - 1 and abs(ratio - 35 / 64) < 1e-7


def test_single_pixel_system_packing():
    """One valid pixel: A must equal w_i J_i J_i^T + n w_d J_d J_d^T with the rows of SURVEY 8 a1, in row-major packing."""
    rows, cols = 6, 8
    nanmap = lambda: np.full((rows, cols), np.nan, np.float32)
    W0, I0, gWx, gWy, gIx, gIy, W1, I1 = [nanmap() for _ in range(8)]
    y, x = 2, 5
    vals = dict(w0=0.5, i0=100.0, gwx=0.01, gwy=-0.02, gix=3.0, giy=-1.5, w1=0.52, i1=104.0)
    for m, k in zip((W0, I0, gWx, gWy, gIx, gIy, W1, I1), ("w0", "i0", "gwx", "gwy", "gix", "giy", "w1", "i1")):
        m[y, x] = vals[k]
    k = (50.0, 60.0, 3.5, 2.5)
    sd, si, nud, nui = 0.0025, 5.0, 4.0, 6.0
    A, b = O.build_system(W0, I0, gWx, gWy, gIx, gIy, W1, I1, k, sigma_depthinv=sd, sigma_int=si, nu_depthinv=nud, nu_int=nui)
    p = np.array([(x - k[2]) / k[0], (y - k[3]) / k[1], 1.0])
    def rows_for(gx, gy, w0, extra):
        g = np.array([gx * k[0], gy * k[1], 0.0]); g[2] = -(g[0] * p[0] + g[1] * p[1])
        jt = g * w0; jt[2] += w0 * extra
        g2 = g.copy(); g2[2] += extra
        return np.concatenate([jt, -np.cross(g2, p)]), g
    Jd, g = rows_for(vals["gwx"], vals["gwy"], vals["w0"], vals["w1"]); Jd /= sd
    Ji, _ = rows_for(vals["gix"], vals["giy"], vals["w0"], 0.0); Ji /= si
    ed = -(vals["w1"] - vals["w0"]) / sd; ei = -(vals["i1"] - vals["i0"]) / si
    wd = (nud + 1) / (nud + ed ** 2); wi = (nui + 1) / (nui + ei ** 2)
    n = g / vals["w0"] + np.array([0, 0, 1.0]); nf = abs(n @ p) / np.linalg.norm(n) / np.linalg.norm(p)
    A_ref = wi * np.outer(Ji, Ji) + nf * wd * np.outer(Jd, Jd)
    b_ref = wi * Ji * ei + nf * wd * Jd * ed
    assert np.allclose(A, A_ref, rtol=2e-5) and np.allclose(b, b_ref, rtol=2e-5) and np.array_equal(A, A.T)


def test_student_t_scale_recovery():
    r = util.rng(7)
    for nu, sigma in [(3.0, 0.004), (5.0, 7.0), (8.0, 2.0)]:
        e = (sigma * r.standard_t(nu, 19200)).astype(np.float32)
        b, s, v = O.sigma_nu_student(e, 0.0, sigma * 1.6, 5.0, O.STUDENT)
        assert abs(v - nu) <= 2.0 and 0.6 * sigma < s < 1.7 * sigma and abs(b) < 0.1 * sigma
        assert v in [2.0 + 0.25 * k for k in range(33)]


def test_gauss_newton_step_contracts_noise_free():
    """Jacobian sign/scale KAT: on a noise-free synthetic pair ONE level-0 iteration started 0.15 deg / 3 mm off the true
    pose (inside the linear basin of the finest level) must cut the pose error several-fold, and the full {3,5,10}
    schedule started from identity must recover the ~2.5 cm / 1.3 deg motion to a few 1e-4 (rad / m)."""
    import torch
    from rgbid import synth
    seq = synth.make_sequence(2, noise=False, dropout=0.0, trans_step=(0.02, 0.03), rot_step_deg=(1.0, 1.5))
    d = seq["depth"].numpy().astype(np.uint16); c = seq["rgb"].numpy()
    Rg, tg = [a.numpy() for a in synth.relative_pose(seq["R_wc"][0], seq["t_wc"][0], seq["R_wc"][1], seq["t_wc"][1])]
    dR = O.expmap_rot(np.deg2rad(0.15) * np.array([0.6, -0.64, 0.48])); dt = np.array([0.002, -0.002, 0.001])
    R0, t0 = dR @ Rg, tg + dt
    err0 = np.linalg.norm(t0 - tg) + np.linalg.norm(O.logmap(R0.T @ Rg, np.zeros(3))[3:])
    ok, R1, t1, _ = O.align_pair(O.default_config(iters=[1, 0, 0]), d[0], c[0], d[1], c[1], R0, t0)
    err1 = np.linalg.norm(t1 - tg) + np.linalg.norm(O.logmap(R1.T @ Rg, np.zeros(3))[3:])
    assert ok and err1 < 0.25 * err0, (err0, err1)
    ok, R, t, cov = O.align_pair(O.default_config(), d[0], c[0], d[1], c[1])
    # mm depth quantisation + 8-bit colour bound the achievable accuracy vs ground truth (this is not the parity tolerance)
    assert ok and np.linalg.norm(O.logmap(R.T @ Rg, np.zeros(3))[3:]) < 3e-4 and np.linalg.norm(t - tg) < 5e-4
    assert np.all(np.linalg.eigvalsh(cov) > 0)


def test_calibration_front_end_kats():
    """SURVEY 8 f-5 oracle pins: zero distortion is the identity map; the default depth model (c1=1, c0=0, q=0) only shifts by
    (xshift, yshift); identity extrinsics with equal intrinsics register a fronto-parallel plane onto itself; a pure +z offset of
    the depth camera rescales inverse depth by 1/(1 - w tz)."""
    r = util.rng(41)
    rows, cols = 60, 80
    I = util.rand_intensity(r, rows, cols)
    k0 = (105.0, 105.0, 39.5, 29.5, 0, 0, 0, 0, 0)
    out = O.undistort_intensity(I, k0)
    ok = np.isfinite(out)
    assert ok[1:-1, 1:-1].all()
    np.testing.assert_allclose(out[ok], I[ok], atol=2e-3)
    w = util.rand_invdepth(r, rows, cols, nan_frac=0.0)
    corr, und = O.undistort_depthinv(w, k0, O.depth_dist(q1=(0,) * 9))
    np.testing.assert_array_equal(corr[5:, 5:], w[1:-4, 1:-4])       # res(y, x) = src(y - 4, x - 4), strict > 0
    assert np.isnan(corr[:5]).all() and np.isnan(corr[:, :5]).all()
    plane = np.full((rows, cols), 0.5, np.float32)
    eye = np.eye(3, dtype=np.float32)
    inter, reg = O.register_depthinv(plane, eye, np.zeros(3, np.float32), eye)
    np.testing.assert_array_equal(reg, plane)
    assert np.isnan(inter[:rows - 1]).all() and np.isfinite(inter[rows:2 * rows, cols:2 * cols]).all()   # centred in the 3x canvas
    K = np.array([[105.0, 0, 39.5], [0, 105.0, 29.5], [0, 0, 1]], np.float32)
    tz = 0.2
    _, reg = O.register_depthinv(plane, eye, K @ np.array([0, 0, tz], np.float32), eye)
    ok = np.isfinite(reg)
    assert ok.mean() > 0.5
    np.testing.assert_allclose(reg[ok], 0.5 / (1 - 0.5 * tz), rtol=1e-6)


def test_jacobian_rows_vs_finite_differences():
    """Jacobian KAT: with least-squares weights the right-hand side b of the normal equations is minus the gradient of
    the cost  C(x) = 1/2 sum nfac (W0 - W1)^2 / s_d^2 + 1/2 sum (I0 - I1)^2 / s_i^2  with respect to the update x = (trans, rot)
    applied the way the tracker applies it (visodo.cpp:1252-1263).  Checked per channel by central differences of the warps
    themselves; Sobel gradients + bilinear resampling limit the agreement to a few percent."""
    from rgbid import synth
    K = (131.25, 131.25, 79.875, 59.875)
    rows, cols = 120, 160
    seq = synth.make_sequence(2, K=K, rows=rows, cols=cols, noise=False, dropout=0.0, trans_step=(0.004, 0.006), rot_step_deg=(0.2, 0.3))
    d = seq["depth"].numpy().astype(np.uint16); c = seq["rgb"].numpy()
    W0, Wc = O.depth2invdepth(d[0]), O.depth2invdepth(d[1])
    I0, Ic = O.intensity(c[0]), O.intensity(c[1])
    gwx, gwy = O.gradient(W0); gix, giy = O.gradient(I0)
    sd, si = 0.0025, 5.0
    v, u = np.mgrid[0:rows, 0:cols].astype(np.float64)
    px_, py_ = (u - K[2]) / K[0], (v - K[3]) / K[1]
    gx, gy = gwx * K[0], gwy * K[1]
    gz = -(gx * px_ + gy * py_)
    n = np.stack([gx / W0, gy / W0, gz / W0 + 1.0]); p = np.stack([px_, py_, np.ones_like(px_)])
    nfac = np.abs((n * p).sum(0)) / np.sqrt((n * n).sum(0) * (p * p).sum(0))          # estimate_VO.cu:225-231

    def warps(R, t):
        Ri = R.T; ti = -Ri @ t
        Rp, tp = util.project(K, Ri, ti)
        W1 = O.warp_invdepth(Wc, W0, Rp, tp)
        return W1, O.warp_intensity(Ic, W1, Rp, tp, O.INTERP_EXACT)

    def update(R, t, x):                                                               # visodo.cpp:1252-1263
        Rinc = np.linalg.inv(O.expmap_rot(x[3:]))
        return Rinc @ R, Rinc @ t - Rinc @ x[:3]

    R0, t0 = np.eye(3), np.zeros(3)
    W1, I1 = warps(R0, t0)
    inner = np.zeros((rows, cols), bool); inner[6:-6, 6:-6] = True                     # keep away from the frame border
    # the iD warp is POINT sampled (piecewise constant in the pose): its differences must span more than a pixel (f t / z = 1.3 px)
    for weighting, eps_t, eps_r, cos_min in ((O.PHOT_ONLY, 2e-3, 1.5e-3, 0.995), (O.GEOM_ONLY, 2e-2, 1e-2, 0.98)):
        m = inner & np.isfinite(W1) & np.isfinite(W0) & np.isfinite(gwx) & np.isfinite(gwy) & np.isfinite(I1)
        # restrict both the system and the cost to the same pixel set: everything outside becomes invalid (NaN keyframe iD)
        W0m = np.where(m, W0, np.nan).astype(np.float32)
        A, b = O.build_system(W0m, I0, gwx, gwy, gix, giy, W1, I1, K, student_nu=False, mestimator=O.LSQ, weighting=weighting,
                              sigma_depthinv=sd, sigma_int=si)

        def cost(R, t):
            w1, i1 = warps(R, t)
            ok = m & np.isfinite(w1) & np.isfinite(i1)
            assert ok.sum() > 0.9 * m.sum()
            if weighting == O.PHOT_ONLY:
                return 0.5 * np.sum(((I0[ok].astype(np.float64) - i1[ok]) / si) ** 2) * m.sum() / ok.sum()
            return 0.5 * np.sum(nfac[ok] * ((W0[ok].astype(np.float64) - w1[ok]) / sd) ** 2) * m.sum() / ok.sum()

        g = np.zeros(6)
        for i in range(6):
            e = np.zeros(6); e[i] = eps_t if i < 3 else eps_r
            g[i] = (cost(*update(R0, t0, e)) - cost(*update(R0, t0, -e))) / (2 * e[i])
        cosang = float(np.dot(-g, b) / (np.linalg.norm(g) * np.linalg.norm(b)))
        assert cosang > cos_min, (weighting, cosang, -g, b)
        assert 0.85 < np.linalg.norm(g) / np.linalg.norm(b) < 1.15, (weighting, np.linalg.norm(g) / np.linalg.norm(b))
        assert np.all(np.sign(-g[np.abs(b) > 0.05 * np.abs(b).max()]) == np.sign(b[np.abs(b) > 0.05 * np.abs(b).max()]))


def test_backend_streams_are_self_consistent():
    """f-3 (visodo.cpp:1610-1611, 2128-2129, 2161-2162): known-answer properties of the Pose / PoseConstraint / keyframe streams.
    Chaining the SEQ_ODO constraints from the identity reproduces the pushed poses; every SEQ_KF constraint equals the chain of the
    SEQ_ODO constraints between its two ids; the exported keyframe's relative pose is that constraint, its global pose the pose pushed
    at its id; the keyframe ids chain (end of one = id of the next)."""
    from rgbid import synth
    K = (131.25, 131.25, 79.5, 59.5)
    rows, cols, n = 120, 160, 7
    seq = synth.make_sequence(n, K=K, rows=rows, cols=cols, trans_step=(0.01, 0.02), rot_step_deg=(0.5, 1.0))
    d = seq["depth"].numpy().astype(np.uint16); c = seq["rgb"].numpy()
    trk = O.Tracker(O.default_config(rows=rows, cols=cols, fx=K[0], fy=K[1], cx=K[2], cy=K[3], visratio_odo=0.985, visratio_integr=0.97))
    for k in range(n):
        trk.track(d[k], c[k])
    ids, Rs, ts = trk.sink_poses()
    assert list(ids) == list(range(n))
    cons = trk.constraints()
    odo = {q["end"]: q for q in cons if q["type"] == O.SEQ_ODO}
    assert sorted(odo) == list(range(1, n)) and all(q["ini"] == q["end"] - 1 for q in odo.values())
    R, t = np.eye(3), np.zeros(3)
    for k in range(1, n):
        t = t + R @ odo[k]["t"]; R = R @ odo[k]["R"]
        assert np.allclose(Rs[k], R, atol=1e-12) and np.allclose(ts[k], t, atol=1e-12)
    Rg, tg = trk.poses()                                    # ... and they agree with the tracker's own global trajectory
    assert np.allclose(Rs, Rg, atol=1e-9) and np.allclose(ts, tg, atol=1e-9)
    kfc = [q for q in cons if q["type"] == O.SEQ_KF]
    assert len(kfc) == trk.num_keyframes() >= 2
    prev_end = 0
    for i, q in enumerate(kfc):
        assert q["ini"] == prev_end
        prev_end = q["end"]
        R, t = np.eye(3), np.zeros(3)
        for k in range(q["ini"] + 1, q["end"] + 1):
            t = t + R @ odo[k]["t"]; R = R @ odo[k]["R"]
        assert np.allclose(q["R"], R, atol=1e-9) and np.allclose(q["t"], t, atol=1e-9)
        kf = trk.keyframe(i)
        assert kf["id"] == q["ini"] and np.array_equal(kf["R_rel"], q["R"]) and np.array_equal(kf["t_rel"], q["t"])
        assert np.allclose(kf["R"], Rs[kf["id"]], atol=1e-9) and np.allclose(kf["t"], ts[kf["id"]], atol=1e-9)
        assert np.array_equal(kf["colors"], c[kf["id"]])                   # the keyframe's colours are its frame's
        ev = np.linalg.eigvalsh(0.5 * (q["cov"] + q["cov"].T))
        assert ev.min() > 0                                                # a proper covariance
        assert np.isfinite(kf["depthinv"]).mean() > 0.5 and (i == 0 or kf["overlap_mask"].any())
    trk.close()
