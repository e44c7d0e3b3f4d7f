"""CPU check of the guard band that makes the FAST numerics class selection-exact (csrc/guard_band.h).

The bound |xs_fast - xs_oracle| <= delta = |wc| d1 + d2 is PROVEN in the header; this test holds the proof's constants to a numerical experiment:
both coordinate evaluations are emulated operation by operation in numpy -- the oracle's register_pixel in round-to-nearest fp32, the FAST
projection with its FMAs (an FMA = the exact fp64 product-sum rounded once to fp32: the 24 + 24-bit product and the sum fit fp64 for every
magnitude met here) and a reciprocal perturbed by a random +-1 ulp (v_rcp_f32's documented accuracy) -- on millions of random pixels, camera motions
and image sizes, and compared with the constants the product computes (rgbid_fast_guard, the same make_guard the kernels run).  It also checks
the emulation itself against the C oracle (the emulated oracle path must reproduce orc_warp_invdepth's pixel selection bit for bit), so that
the experiment measures the real thing."""
import numpy as np
import pytest

from oracle import oracle as O
from rgbid import device
from tests import util

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def oracle_xs(x, y, w, R, t):
    """register_pixel (oracle/rgbid_oracle.c, warping_registration.cu:129-146) + 0.5, every operation rounded to fp32"""
    zd = f32(1) / w
    X, Y = x * zd, y * zd
    row = lambda r: ((R[3 * r] * X + R[3 * r + 1] * Y) + R[3 * r + 2] * zd) + t[r]
    X0, X1, X2 = row(0), row(1), row(2)
    wc = f32(1) / X2
    return X0 * wc + f32(0.5), X1 * wc + f32(0.5), wc


def fast_xs(x, y, w, R, t, rng, bias=0.5):
    """fastnum::ray + scaled_point + id_project (csrc/warp_device.h): two FMAs per ray component, one per scaled component, v_rcp_f32, one per coordinate
    (`bias`: the addend of the last FMA -- 0.5 for the bound (3) of guard_band.h, the lane's bL for the form (3') the kernels run)"""
    q = [fma(np.full_like(x, R[3 * r]), x, fma(np.full_like(y, R[3 * r + 1]), y, np.full_like(y, R[3 * r + 2]))) for r in range(3)]
    Y = [fma(np.full_like(w, t[r]), w, q[r]) for r in range(3)]
    wc = (f32(1) / Y[2])
    ulp = np.spacing(np.abs(wc)).astype(f32)
    wc = (wc + rng.integers(-1, 2, wc.shape).astype(f32) * ulp).astype(f32)       # 1 ulp reciprocal
    return fma(Y[0], wc, np.full_like(wc, bias)), fma(Y[1], wc, np.full_like(wc, bias)), wc


def fract(x):
    """v_fract_f32: min(fl(x - floor x), 1 - 2^-24) (verified on the device for all 2^32 inputs by rgbid_selftest_fast_primitives)"""
    return np.minimum((x - np.floor(x)).astype(f32), f32(1) - f32(2.0 ** -24))


@pytest.mark.parametrize("rows,cols", [(480, 640), (960, 1280), (120, 160), (61, 83)])
def test_guard_band_covers_the_measured_distance(rows, cols):
    rng = util.rng(77)
    s = cols / 640.0
    K = (525.0 * s, 525.0 * s, 319.5 * s, 239.5 * s)
    worst, flagged, total, flagged_lane, total_lane = 0.0, 0, 0, 0, 0
    for trial in range(24):
        trans, rot = [(0.03, 1.5), (0.3, 10.0), (0.005, 0.2), (1.0, 25.0)][trial % 4]
        Rm, tv = util.small_motion(rng, K, trans, rot)
        Rp, tp = util.project(K, *util.inv_pose(Rm, tv))
        g = device.fast_guard(Rp, tp, cols, rows)
        if not g["zsafe"]:
            continue                                               # such a lane runs every pixel through the exact path: nothing to bound
        n = 200_000
        x = rng.integers(0, cols, n).astype(f32); y = rng.integers(0, rows, n).astype(f32)
        w = np.exp(rng.uniform(np.log(2.0 ** -6), np.log(2.0 ** 6), n)).astype(f32) if trial % 3 == 0 else rng.uniform(0.1, 4.0, n).astype(f32)
        R = np.asarray(Rp, f32).reshape(-1); t = np.asarray(tp, f32)
        with np.errstate(all="ignore"):
            ox, oy, _ = oracle_xs(x, y, w, R, t)
            fx, fy, wc = fast_xs(x, y, w, R, t, rng)
            delta = np.abs(wc) * f32(g["d1"]) + f32(g["d2"])
            open_ = (np.abs(fx) <= 1.01 * max(rows, cols) + 2) & (np.abs(fy) <= 1.01 * max(rows, cols) + 2) & np.isfinite(delta) & (delta < 0.5)
            ratio = np.maximum(np.abs(fx - ox), np.abs(fy - oy))[open_] / delta[open_]
        worst = max(worst, float(ratio.max()))
        # the pixels the kernels would recompute: a coordinate within delta of an integer
        amb = np.maximum(np.abs(fx - np.floor(fx) - 0.5), np.abs(fy - np.floor(fy) - 0.5))[open_] > 0.5 - delta[open_]
        flagged += int(amb.sum()); total += int(open_.sum())
        # and the decision itself: wherever the guard does not fire, floor() of the two evaluations agrees
        same = (np.floor(fx) == np.floor(ox)) & (np.floor(fy) == np.floor(oy))
        assert same[open_][~amb].all()
        # (3'), the form the kernels run: the coordinate biased down by the lane's band; wherever max3(fract, fract, |wc| kL) < cL the pixel floor()
        # selects is the oracle's (and |wc| is inside the range the lane constant is priced for)
        with np.errstate(all="ignore"):
            bx, by, wcb = fast_xs(x, y, w, R, t, rng, bias=f32(g["bL"]))
            m = np.maximum(np.maximum(fract(bx), fract(by)), (np.abs(wcb) * f32(g["kL"])).astype(f32))
            safe = m < f32(g["cL"])
        assert (np.abs(wcb[safe]) <= 2.0).all()
        assert ((np.floor(bx) == np.floor(ox)) & (np.floor(by) == np.floor(oy)))[safe].all()
        in_range = np.abs(wcb) <= 2.0
        flagged_lane += int((~safe & in_range & open_).sum()); total_lane += int((in_range & open_).sum())
    assert total > 1_000_000
    assert 0 < flagged_lane < 0.02 * total_lane                     # the lane-constant band costs a few pixels per thousand, not per cent
    assert worst < 1.0, worst                                       # the proven bound holds ...
    assert worst > 0.02                                             # ... and is not absurdly loose (typical: 0.1 - 0.3 of the bound)
    print(f"{cols}x{rows}: worst measured |xs_fast - xs_oracle| / delta = {worst:.3f}; pixels inside the guard band: {flagged / total:.2e} "
          f"(per-pixel band), {flagged_lane / max(total_lane, 1):.2e} (lane-constant band, |wc| <= 2)")


def test_emulated_oracle_path_is_the_c_oracle():
    """the numpy restatement of register_pixel above selects the same source pixel as orc_warp_invdepth at every pixel of a 120x160 case"""
    rows, cols = 120, 160
    r = util.rng(5)
    K = (131.25, 131.25, 79.875, 59.875)
    grid = util.rand_invdepth(r, rows, cols, nan_frac=0.0)
    src = (np.arange(rows * cols, dtype=np.float32).reshape(rows, cols) + 1.0) * f32(1e-3)       # the value names its pixel
    Rm, tv = util.small_motion(r, K, 0.03, 1.5)
    Rp, tp = util.project(K, *util.inv_pose(Rm, tv))
    R = np.asarray(Rp, f32).reshape(-1); t = np.asarray(tp, f32)
    yy, xx = np.mgrid[0:rows, 0:cols]
    ox, oy, wc = oracle_xs(xx.astype(f32).ravel(), yy.astype(f32).ravel(), grid.ravel(), R, t)
    ix, iy = np.floor(ox).astype(int), np.floor(oy).astype(int)
    inb = (ix >= 0) & (iy >= 0) & (ix < cols) & (iy < rows)
    w2 = np.where(inb, src[np.clip(iy, 0, rows - 1), np.clip(ix, 0, cols - 1)], np.nan).reshape(rows, cols)
    oW1 = O.warp_invdepth(src, grid, Rp, tp)
    # res = v / (1 - w2 tz) * w2 is monotone in w2 here; compare the SELECTED source value through the oracle's own formula
    tz = t[2]
    v = ((f32(1) / wc.reshape(rows, cols)) - tz) * grid
    res = (v / (f32(1) - w2.astype(f32) * tz)) * w2.astype(f32)
    res = np.where(res > 0, res, np.nan).astype(f32)
    assert np.array_equal(np.isnan(res), np.isnan(oW1))
    assert np.array_equal(res[~np.isnan(res)], oW1[~np.isnan(oW1)])


def test_guard_refuses_motions_outside_the_sign_analysis():
    K = (525.0, 525.0, 319.5, 239.5)
    r = util.rng(9)
    ok = device.fast_guard(*util.project(K, *util.inv_pose(*util.small_motion(r, K, 0.03, 1.5))), 640, 480)
    assert ok["zsafe"] == 1 and 1e-4 < ok["d1"] + ok["d2"] < 3e-3 and np.isfinite(ok["e0"])
    wild = device.fast_guard(*util.project(K, *util.inv_pose(*util.small_motion(r, K, 0.1, 120.0))), 640, 480)
    assert wild["zsafe"] == 0 and np.isinf(wild["d1"]) and np.isinf(wild["db"])
    assert wild["cL"] == -1.0 and wild["wcore"] == 0.0               # the lane-constant forms: never safe, never core
    assert 0.49 < ok["bL"] < 0.5 and 0.99 < ok["cL"] < 1.0 and 100.0 < ok["wcore"] <= 512.0
    nan = device.fast_guard([float("nan")] * 9, [0, 0, 0], 640, 480)
    assert nan["zsafe"] == 0 and np.isinf(nan["d1"]) and nan["cL"] == -1.0 and nan["wcore"] == 0.0


@pytest.mark.parametrize("rows,cols", [(480, 640), (120, 160)])
def test_value_bands_of_the_gates_cover_the_measured_distance(rows, cols):
    """guard_band.h (5): the relative distance between the oracle's and the FAST class's (a) inverse depth of the point in the other frame (covisibility gate
    |w' - D| < 0.020) and (b) warped inverse depth (fusion gate), against the bands eps_w = |rcp(Y_2)| g1 + g0 and eps_res = e0 + e1 |rcp(1 - w2 t_z)| the
    kernels use to decide whether a gate is open.  Same emulation as above."""
    rng = util.rng(78)
    s = cols / 640.0
    K = (525.0 * s, 525.0 * s, 319.5 * s, 239.5 * s)
    worst_w, worst_r, total = 0.0, 0.0, 0
    for trial in range(16):
        trans, rot = [(0.03, 1.5), (0.3, 10.0), (0.005, 0.2), (0.1, 4.0)][trial % 4]
        Rm, tv = util.small_motion(rng, K, trans, rot)
        Rp, tp = util.project(K, *util.inv_pose(Rm, tv))
        g = device.fast_guard(Rp, tp, cols, rows)
        if not g["zsafe"]:
            continue
        n = 200_000
        x = rng.integers(0, cols, n).astype(f32); y = rng.integers(0, rows, n).astype(f32)
        w = rng.uniform(0.1, 4.0, n).astype(f32)
        w2 = rng.uniform(0.1, 4.0, n).astype(f32)                   # the sampled inverse depth of the other frame
        R = np.asarray(Rp, f32).reshape(-1); t = np.asarray(tp, f32)
        with np.errstate(all="ignore"):
            # oracle: register_pixel returns wc = fl(1 / X_2); v = fl(fl(1 / wc) - t_z) * w; res = fl(fl(v / fl(1 - fl(w2 t_z))) * w2)
            _, _, wc_o = oracle_xs(x, y, w, R, t)
            v = ((f32(1) / wc_o) - t[2]) * w
            res_o = (v / (f32(1) - w2 * t[2])) * w2
            # FAST: w' = ws * rcp(Y_2); res = (q_2 * rcp(1 - w2 t_z)) * w2 with the FMA inside the reciprocal's argument
            q2 = fma(np.full_like(x, R[6]), x, fma(np.full_like(y, R[7]), y, np.full_like(y, R[8])))
            Y2 = fma(np.full_like(w, t[2]), w, q2)
            ry = f32(1) / Y2
            ry = (ry + rng.integers(-1, 2, n).astype(f32) * np.spacing(np.abs(ry)).astype(f32)).astype(f32)
            wc_f = w * ry
            wf = fma(-w2, np.full_like(w2, t[2]), np.full_like(w2, 1.0))
            rwf = f32(1) / wf
            rwf = (rwf + rng.integers(-1, 2, n).astype(f32) * np.spacing(np.abs(rwf)).astype(f32)).astype(f32)
            res_f = (q2 * rwf) * w2
            ok = np.isfinite(res_o) & np.isfinite(res_f) & (np.abs(rwf) <= 2.0 ** 19) & (np.abs(ry) <= 2.0 ** 9) & (res_o > 0)
            eps_w = np.abs(ry) * f32(g["g1"]) + f32(g["g0"])
            eps_r = f32(g["e0"]) + f32(g["e1"]) * np.abs(rwf)
            rw = (np.abs(wc_f - wc_o) / np.abs(wc_o))[ok] / eps_w[ok]
            rr = (np.abs(res_f - res_o) / np.abs(res_o))[ok] / eps_r[ok]
        worst_w = max(worst_w, float(rw.max())); worst_r = max(worst_r, float(rr.max())); total += int(ok.sum())
    assert total > 1_000_000
    assert worst_w < 1.0 and worst_r < 1.0, (worst_w, worst_r)
    print(f"{cols}x{rows}: worst measured / bound: inverse depth in the other frame {worst_w:.3f}, warped inverse depth {worst_r:.3f}")
