"""Kernel-level parity of the ENGINE's hot-path kernels, called one at a time through the batched C-ABI (include/rgbid_batched.h ->
csrc/c_api_batched.hip -> the launchers engine.hip calls) and held to the CPU oracle on the same seeded inputs.

These are the kernels bench.py times: the fused Gauss-Newton evaluation k_build_system<ByLane, true, L, 2, 1> / <..., 2, 2> / <..., 2, 0> and its
exact-numerics sibling <..., 1, 0>, the lattice pre-pass + sigma / nu pair, the one-pass keyframe fusion, vertex + normal maps, the
two-direction covisibility and the one-pass frame preparation.  Tolerances:
  * EXACT numerics: integer / index / validity work bit-exact; element-wise fp32 maps 0 ULP; normal equations |dA_ij| <= 2e-5 sqrt(A_ii A_jj)
    against orc_build_system(orc_warp_invdepth, orc_warp_intensity) (estimate_VO.cu:354-439,505-645 on warping_registration.cu:465-546);
  * FAST numerics (the reference build's class of arithmetic): the same bound PLUS the summed contribution of the pixels whose warp selected
    another source pixel (a coordinate within an ulp of a pixel boundary; counted and bounded per test), and -- independently -- the plain
    2e-5 bound against the oracle's normal equations evaluated on the device's own FAST warp pair;
  * sigma: 2e-5 relative, nu: the bisection grid value exactly.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from rgbid import batched as BT
from rgbid._lib import RgbidError
from tests import util
from tests.util import assert_bits

pytestmark = pytest.mark.gpu

GEOMS = [(48, 64, 3), (61, 83, 2), (120, 160, 2), (240, 320, 2), (480, 640, 2), (960, 1280, 1)]   # rows, cols, lanes (120 / 240: pyramid levels 2 / 1 of the shipped configuration)
VEC_GEOMS = [g for g in GEOMS if g[1] % 4 == 0]


@pytest.fixture(scope="module")
def bt(ctx):
    return BT.Batched(ctx)


def K_for(rows, cols):
    s = cols / 640.0
    return (525.0 * s, 525.0 * s, 319.5 * s, 239.5 * s)


def stack(maps, pad=0):
    """[lanes, rows, cols] CUDA tensor (optionally with padded rows: step != cols * 4, still 16-byte aligned when pad % 4 == 0)"""
    a = np.stack([np.ascontiguousarray(m) for m in maps])
    t = torch.from_numpy(a).cuda()
    if pad:
        big = torch.zeros(a.shape[:2] + (a.shape[2] + pad,) + a.shape[3:], dtype=t.dtype, device="cuda")
        v = big[:, :, :a.shape[2]]
        v.copy_(t)
        return v
    return t


def gn_case(rows, cols, lanes, seed, filtered_grads=False):
    """per lane: keyframe maps + gradients, a current frame that overlaps it, the projected inverse pose; plus the oracle's W1 / I1"""
    K = K_for(rows, cols)
    lanes_data = []
    for l in range(lanes):
        r = util.rng(1000 * seed + l)
        W0 = util.rand_invdepth(r, rows, cols, nan_frac=0.05)
        I0 = util.rand_intensity(r, rows, cols)
        if l == 1:
            I0[0, 0] = np.nan; I0[0, -1] = np.nan; I0[-1, 0] = np.nan        # pyramid levels >= 1 carry NaN corners
        src_g_W, src_g_I = (O.bilateral(W0, 2 * 0.0025), O.bilateral(I0, 3.0)) if filtered_grads else (W0, I0)
        gWx, gWy = O.gradient(src_g_W)
        gIx, gIy = O.gradient(src_g_I)
        R, t = util.small_motion(r, K, 0.005 + 0.01 * l, 0.3 + 0.3 * l)
        Rp, tp = util.project(K, *util.inv_pose(R, t))
        Wc = (W0 * np.float32(1.001) + (0.002 * r.standard_normal((rows, cols))).astype(np.float32)).astype(np.float32)
        Wc[r.random((rows, cols)) < 0.03] = np.nan
        Ic = np.clip(I0 + (3 * r.standard_normal((rows, cols))).astype(np.float32), 0, 255).astype(np.float32)
        Ic[np.isnan(Ic)] = 100.0
        W1 = O.warp_invdepth(Wc, W0, Rp, tp)
        I1 = O.warp_intensity(Ic, W1, Rp, tp, O.INTERP_TEX8)
        lanes_data.append(dict(W0=W0, I0=I0, gWx=gWx, gWy=gWy, gIx=gIx, gIy=gIy, Wc=Wc, Ic=Ic, Rp=Rp, tp=tp, W1=W1, I1=I1))
    return K, lanes_data


def dev_maps(L, names, pad=0):
    return [stack([d[n] for d in L], pad) for n in names]


KF_NAMES = ("W0", "I0", "gWx", "gWy", "gIx", "gIy")


def check_system(A, b, oA, ob, extraA=None, extrab=None, rtol_b=2e-5):
    """|dA_ij| <= 2e-5 sqrt(A_ii A_jj) (+ extraA); |db_i| <= rtol_b * sqrt(A_ii) * sum_j sqrt(A_jj) |x_j| (+ extrab), x the Gauss-Newton step.
    Returns the two worst ratios (before the extras) for the log."""
    assert np.array_equal(A, A.T)
    d = np.sqrt(np.diag(oA))
    tolA = 2e-5 * np.outer(d, d) + (0 if extraA is None else extraA)
    ra = (np.abs(A - oA) / np.outer(d, d)).max()
    assert (np.abs(A - oA) <= tolA).all(), ra
    x = np.linalg.solve(oA, ob)
    scale = d * (d @ np.abs(x)) + 1e-30
    tolb = rtol_b * scale + rtol_b * np.abs(ob) + (0 if extrab is None else extrab)
    rb = (np.abs(b - ob) / scale).max()
    assert (np.abs(b - ob) <= tolb).all(), rb
    return ra, rb


FAST_RTOL_B = 2e-5
ARGS = dict(sigma_depthinv=0.003, sigma_int=6.0, bias_depthinv=0.0002, bias_int=-0.5, nu_depthinv=3.5, nu_int=6.25)
COV_ARGS = dict(sigma_depthinv=0.0025, sigma_int=5.0, bias_depthinv=0.0, bias_int=0.0, nu_depthinv=5.0, nu_int=5.0)

# (name, oracle kwargs, rgbid_sys_params kwargs, expected kernel variant under RGBID_WM_AUTO)
CONFIGS = [
    ("gn_student_nu", dict(student_nu=True, mestimator=O.STUDENT, weighting=O.INDEPENDENT, **ARGS), BT.WM_STUDENT_NU),
    ("covariance_pass", dict(student_nu=False, mestimator=O.STUDENT, weighting=O.INDEPENDENT, **COV_ARGS), BT.WM_STUDENT_FIXED),
    ("min_weight", dict(student_nu=True, mestimator=O.STUDENT, weighting=O.MIN_WEIGHT, **ARGS), BT.WM_GENERIC),
    ("huber_fixed", dict(student_nu=False, mestimator=O.HUBER, weighting=O.INDEPENDENT, **COV_ARGS), BT.WM_GENERIC),
    ("geom_only", dict(student_nu=True, mestimator=O.STUDENT, weighting=O.GEOM_ONLY, **ARGS), BT.WM_STUDENT_NU),
]


def sp_from(lanes, kw):
    k = dict(kw)
    k["student_nu"] = int(k["student_nu"])
    return BT.sys_params(lanes, **k)


@pytest.mark.parametrize("rows,cols,lanes", GEOMS)
@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_gn_fused_exact_vs_oracle(bt, rows, cols, lanes, cfg):
    """the exact-numerics fused kernel (<..., 1, 0>; scalar path at 83 columns): A, b of every lane against the oracle's warp -> warp -> system chain"""
    name, okw, _ = cfg
    if rows >= 480 and name not in ("gn_student_nu", "covariance_pass"):
        pytest.skip("full sizes run the two configurations the engine times")
    K, L = gn_case(rows, cols, lanes, 31, filtered_grads=(name == "covariance_pass" and rows < 480))
    dm = dev_maps(L, KF_NAMES + ("Wc", "Ic"), pad=4)
    A, b = bt.gn_fused(*dm, [d["Rp"] for d in L], [d["tp"] for d in L], K, sp_from(lanes, okw), fast=False)
    for l, d in enumerate(L):
        oA, ob = O.build_system(d["W0"], d["I0"], d["gWx"], d["gWy"], d["gIx"], d["gIy"], d["W1"], d["I1"], K, **okw)
        check_system(A[l], b[l], oA, ob)
    A2, b2 = bt.gn_fused(*dm, [d["Rp"] for d in L], [d["tp"] for d in L], K, sp_from(lanes, okw), fast=False)
    assert np.array_equal(A, A2) and np.array_equal(b, b2)        # fixed-order partial sums: deterministic


def _selection_mismatches(d, W1f, I1f):
    """what tests/test_gpu_kernels.py::test_warp_pair_fast_selects_the_oracles_pixels counts: FAST point-sampled another source pixel (warped iD off by
    more than 1e-5 relative) or carries another validity than the oracle.  Zero since round 4 (csrc/guard_band.h)."""
    W1o, I1o = d["W1"], d["I1"]
    nanW = np.isnan(W1o) != np.isnan(W1f)
    nanI = np.isnan(I1o) != np.isnan(I1f)
    with np.errstate(invalid="ignore"):
        dW = np.abs(W1f - W1o) > 1e-5 * np.abs(W1o)
    return nanW | nanI | (dW & ~np.isnan(W1o) & ~np.isnan(W1f))


@pytest.mark.parametrize("rows,cols,lanes", VEC_GEOMS)
@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_gn_fused_fast_vs_oracle(bt, rows, cols, lanes, cfg):
    """THE benchmarked kernel: k_build_system<ByLane, true, 0, 2, WM> (WM = 1: Gauss-Newton iterations, 2: covariance pass, 0: generic).
    (a) against the oracle's normal equations on the device's own FAST warp pair: plain 2e-5 (the row algebra / reduction / weights of the
        fused kernel are those of estimate_VO.cu whatever the warp numerics);
    (b) against the pure oracle chain: the same plain 2e-5, no allowance (round 4: every pixel selects the oracle's source pixel and validity);
    (c) the specialised variant equals the generic variant bit for bit (same arithmetic, fewer branches)."""
    name, okw, wm_expected = cfg
    if rows >= 480 and name not in ("gn_student_nu", "covariance_pass"):
        pytest.skip("full sizes run the two configurations the engine times")
    K, L = gn_case(rows, cols, lanes, 32)
    dm = dev_maps(L, KF_NAMES + ("Wc", "Ic"))
    Rs, ts = [d["Rp"] for d in L], [d["tp"] for d in L]
    sp = sp_from(lanes, okw)
    A, b = bt.gn_fused(*dm, Rs, ts, K, sp, fast=True, weight_mode=BT.WM_AUTO)
    Ag, bg = bt.gn_fused(*dm, Rs, ts, K, sp, fast=True, weight_mode=BT.WM_GENERIC)
    assert np.array_equal(A, Ag) and np.array_equal(b, bg), (wm_expected, np.abs(A - Ag).max())
    if wm_expected != BT.WM_GENERIC:
        Av, bv = bt.gn_fused(*dm, Rs, ts, K, sp, fast=True, weight_mode=wm_expected)   # the variant AUTO must have picked: accepted explicitly
        assert np.array_equal(A, Av) and np.array_equal(b, bv)
    W1f, I1f = torch.empty_like(dm[0]), torch.empty_like(dm[0])
    bt.warp_pair(dm[6], dm[7], dm[0], W1f, I1f, Rs, ts, fast=True)
    W1f, I1f = W1f.cpu().numpy(), I1f.cpu().numpy()
    for l, d in enumerate(L):
        kf = [d[n] for n in KF_NAMES]
        oAf, obf = O.build_system(*kf, W1f[l], I1f[l], K, **okw)
        check_system(A[l], b[l], oAf, obf)                                           # (a)
        nb = int(_selection_mismatches(d, W1f[l], I1f[l]).sum())
        assert nb == 0, (nb, rows * cols)
        oA, ob = O.build_system(*kf, d["W1"], d["I1"], K, **okw)
        # (the 1.8 fixed-point bilinear weights flip a 1/256 step at ~1 % of the pixels -- |dI1| <= local contrast / 256, random sign -- which reaches b,
        # linear in the residual; measured on the MI355X: db <= 9e-6 of the scale at every size, inside the plain 2e-5)
        ra, rb = check_system(A[l], b[l], oA, ob, rtol_b=FAST_RTOL_B)   # (b)
        print(f"{name} {cols}x{rows} lane {l}: {nb} selection mismatches of {rows * cols}; vs pure oracle chain: dA {ra:.2e}, db {rb:.2e}")


@pytest.mark.parametrize("rows,cols,lanes", [VEC_GEOMS[0], VEC_GEOMS[3]])
def test_gn_fused_fast_equals_unfused_fast(bt, rows, cols, lanes):
    """regression (not parity evidence): the fused kernel == normal equations on the stored FAST warp pair, bit for bit (same launch plan)"""
    K, L = gn_case(rows, cols, lanes, 33)
    dm = dev_maps(L, KF_NAMES + ("Wc", "Ic"))
    Rs, ts = [d["Rp"] for d in L], [d["tp"] for d in L]
    okw = CONFIGS[0][1]
    A, b = bt.gn_fused(*dm, Rs, ts, K, sp_from(lanes, okw), fast=True)
    W1f, I1f = torch.empty_like(dm[0]), torch.empty_like(dm[0])
    bt.warp_pair(dm[6], dm[7], dm[0], W1f, I1f, Rs, ts, fast=True)
    A2, b2 = bt.build_system(*dm[:6], W1f, I1f, K, sp_from(lanes, okw))
    assert np.array_equal(A, A2) and np.array_equal(b, b2)


@pytest.mark.parametrize("many", [256, 1024])
def test_gn_fused_long_partial_sums(bt, many):
    """the launch plan of the benchmarked batch sizes: with 256 / 1 024 lanes at 640x480 a workgroup walks 15 / 60 tiles (a thread's fp32 partial
    sums run over 60 / 240 pixels before the workgroup reduction) -- 2 distinct lanes dealt onto `many`, every copy bit-identical, A and b of the
    fused FAST kernel and of the normal equations on stored maps still within the plain 2e-5 of the oracle on the device's warp pair, and the two
    kernels still bit-identical to each other"""
    rows, cols = 480, 640
    K, L = gn_case(rows, cols, 2, 34)
    idx = torch.arange(many, device="cuda") % 2
    dm = [t[idx].contiguous() for t in dev_maps(L, KF_NAMES + ("Wc", "Ic"))]
    Rs, ts = [L[l % 2]["Rp"] for l in range(many)], [L[l % 2]["tp"] for l in range(many)]
    okw = CONFIGS[0][1]
    A, b = bt.gn_fused(*dm, Rs, ts, K, sp_from(many, okw), fast=True)
    for l in range(2, many):
        assert np.array_equal(A[l], A[l % 2]) and np.array_equal(b[l], b[l % 2]), l
    W1f, I1f = torch.empty_like(dm[0]), torch.empty_like(dm[0])
    bt.warp_pair(dm[6], dm[7], dm[0], W1f, I1f, Rs, ts, fast=True)
    A2, b2 = bt.build_system(*dm[:6], W1f, I1f, K, sp_from(many, okw))
    assert np.array_equal(A, A2) and np.array_equal(b, b2)
    W1h, I1h = W1f[:2].cpu().numpy(), I1f[:2].cpu().numpy()
    for l, d in enumerate(L):
        oA, ob = O.build_system(*[d[n] for n in KF_NAMES], W1h[l], I1h[l], K, **okw)
        ra, rb = check_system(A[l], b[l], oA, ob)
        print(f"{many} lanes, lane {l}: dA {ra:.2e}, db {rb:.2e}")


def test_gn_fused_nu_int_from_max(bt):
    """visodo.cpp:1186: nu_int = max(nu_int, nu_depthinv) applied inside the kernel when asked for"""
    rows, cols, lanes = 48, 64, 2
    K, L = gn_case(rows, cols, lanes, 34)
    dm = dev_maps(L, KF_NAMES + ("Wc", "Ic"))
    Rs, ts = [d["Rp"] for d in L], [d["tp"] for d in L]
    kw = dict(CONFIGS[0][1]); kw.update(nu_depthinv=7.5, nu_int=3.0)
    for fast in (False, True):
        A, b = bt.gn_fused(*dm, Rs, ts, K, sp_from(lanes, dict(kw, nu_int_from_max=1)), fast=fast)
        A2, b2 = bt.gn_fused(*dm, Rs, ts, K, sp_from(lanes, dict(kw, nu_int=7.5)), fast=fast)
        assert np.array_equal(A, A2) and np.array_equal(b, b2)


def test_gn_fused_identity_kat(bt):
    """analytic: current frame == keyframe and identity transform => b == 0 exactly, A symmetric PSD -- in both numerics classes"""
    rows, cols, lanes = 48, 64, 2
    K, L = gn_case(rows, cols, lanes, 35)
    Rp, tp = util.project(K, np.eye(3), np.zeros(3))
    for d in L:
        d["I0"] = np.nan_to_num(d["I0"], nan=90.0)
    dm = dev_maps(L, KF_NAMES + ("W0", "I0"))
    for fast in (False, True):
        A, b = bt.gn_fused(*dm, [Rp] * lanes, [tp] * lanes, K, sp_from(lanes, dict(CONFIGS[0][1], bias_depthinv=0.0, bias_int=0.0)), fast=fast)
        for l in range(lanes):
            # TEX8 weights at an integer coordinate are exactly 0 / 1 and the point sample is the pixel itself: the residuals are rounding
            # (W1 = w0 to an ulp), the Gauss-Newton step vanishes
            x = np.linalg.solve(A[l], b[l])
            assert np.abs(x).max() < 1e-6, (fast, x)
            assert np.linalg.eigvalsh(A[l]).min() > -1e-6 * np.abs(A[l]).max()


def test_gn_fused_argument_errors(bt):
    rows, cols, lanes = 61, 83, 2
    K, L = gn_case(rows, cols, lanes, 36)
    dm = dev_maps(L, KF_NAMES + ("Wc", "Ic"))
    Rs, ts = [d["Rp"] for d in L], [d["tp"] for d in L]
    with pytest.raises(RgbidError):      # FAST needs rows of whole 4-pixel groups: never silently the other class
        bt.gn_fused(*dm, Rs, ts, K, sp_from(lanes, CONFIGS[0][1]), fast=True)
    with pytest.raises(RgbidError):      # the caller's guarantee does not hold for these parameters
        bt.gn_fused(*dm, Rs, ts, K, sp_from(lanes, CONFIGS[1][1]), fast=False, weight_mode=BT.WM_STUDENT_NU)
    with pytest.raises(RgbidError):
        bt.gn_fused(*dm, Rs, ts, K, sp_from(lanes, CONFIGS[2][1]), fast=False, weight_mode=BT.WM_STUDENT_FIXED)
    res = torch.zeros((lanes, 2 * rows * cols), device="cuda")
    with pytest.raises(RgbidError):
        bt.lattice_residuals(dm[6], dm[0], dm[7], dm[1], Rs, ts, 1000, res, fast=True)
    w = torch.zeros_like(dm[0])
    with pytest.raises(RgbidError):      # one-pass fusion needs the 16-byte geometry
        bt.fuse_frame(dm[6], dm[0].clone(), w, w.clone(), Rs, ts)
    with pytest.raises(RgbidError):
        bt.kf_maps(K, dm[0], torch.zeros((lanes, 3 * rows, cols), device="cuda"), torch.zeros((lanes, 3 * rows, cols), device="cuda"))


def test_batched_calls_refuse_images_that_do_not_hold_their_rows(bt):
    """a caller-described image whose step is smaller than a row (or not a multiple of the element size), whose lanes overlap, or a lane count beyond
    the kernels' grid dimension, is refused before anything is launched (the kernels would write past rows and lanes)"""
    import ctypes as C
    src = torch.zeros((2, 16, 32), device="cuda"); dst = torch.zeros((2, 8, 16), device="cuda")
    ms = C.c_float()
    good_s, good_d = BT.imgb(src), BT.imgb(dst)
    assert bt.L.rgbid_pyr_down_batched(bt._h, 2, C.byref(good_s), C.byref(good_d), C.byref(ms)) == 0
    for field, value in (("step", 4 * 16 - 4), ("step", 4 * 16 + 2), ("lane_stride", 4 * 16 * 4), ("lane_stride", 4 * 16 * 8 + 2)):
        bad = BT.imgb(dst); setattr(bad, field, value)
        assert bt.L.rgbid_pyr_down_batched(bt._h, 2, C.byref(good_s), C.byref(bad), C.byref(ms)) == -1, (field, value)
    assert bt.L.rgbid_pyr_down_batched(bt._h, 70000, C.byref(good_s), C.byref(good_d), C.byref(ms)) == -1


@pytest.mark.parametrize("rows,cols,lanes,ns", [(48, 64, 3, 500), (61, 83, 2, 1000), (480, 640, 2, 10000), (240, 320, 2, 10000), (960, 1280, 1, 10000)])
@pytest.mark.parametrize("packed", [False, True])
def test_lattice_residuals_and_sigma_pair(bt, rows, cols, lanes, ns, packed):
    """k_lattice_pack + k_lattice_residuals_fused (EXACT): both channels' lattice residuals bit-identical to computeErrorGridStride on the oracle's
    warped maps (sigmaFuncs.cu:701-765); k_sigma_pair_arrays on them: nu exactly, bias / sigma to 2e-5 (sigmaFuncs.cu:858-1066)"""
    K, L = gn_case(rows, cols, lanes, 41)
    dm = dev_maps(L, ("Wc", "W0", "Ic", "I0"), pad=4)
    Rs, ts = [d["Rp"] for d in L], [d["tp"] for d in L]
    from rgbid import device
    n, lr, lc, st = device.error_lattice_size(rows, cols, ns)
    res = torch.full((lanes, 2 * n + 8), 7.0, device="cuda")
    kf_lat = None
    if packed:
        kf_lat = torch.zeros((lanes, 2 * n), device="cuda")
        bt.lattice_pack(dm[1], dm[3], ns, kf_lat)
        for l, d in enumerate(L):
            assert_bits(kf_lat[l, :n].cpu().numpy(), d["W0"][::st, ::st][:lr, :lc].reshape(-1), 0, "packed W0")
            assert_bits(kf_lat[l, n:].cpu().numpy(), d["I0"][::st, ::st][:lr, :lc].reshape(-1), 0, "packed I0")
    bt.lattice_residuals(*dm, Rs, ts, ns, res, fast=False, kf_lat=kf_lat)
    got = res.cpu().numpy()
    assert (got[:, 2 * n:] == 7.0).all()
    errs = []
    for l, d in enumerate(L):
        ed, geo = O.error_lattice(d["W1"], d["W0"], ns)
        ei, _ = O.error_lattice(d["I1"], d["I0"], ns)
        assert ed.size == n and geo == (lr, lc, st)
        assert_bits(got[l, :n], ed, 0, "lattice residual iD")
        assert_bits(got[l, n:2 * n], ei, 0, "lattice residual intensity")
        errs.append((ed, ei))
    for mest in (O.STUDENT, O.HUBER):
        out = bt.sigma_pair(res, n, mest)
        for l, (ed, ei) in enumerate(errs):
            ob, os_, ov = O.sigma_nu_student(ed, 0.0, 0.0025, 5.0, mest)
            o = out[l]
            assert o["nu_depthinv"] == ov, (o, ov)
            assert abs(o["sigma_depthinv"] - os_) <= 2e-5 * os_ and abs(o["bias_depthinv"] - ob) <= 2e-5 * os_, (o, ob, os_)
            ob, os_, ov = O.sigma_nu_student(ei, 0.0, 5.0, 5.0, mest)
            assert o["nu_int"] == ov, (o, ov)
            assert abs(o["sigma_int"] - os_) <= 2e-5 * os_ and abs(o["bias_int"] - ob) <= 2e-5 * os_, (o, ob, os_)


@pytest.mark.parametrize("rows,cols,lanes,ns", [(48, 64, 3, 500), (480, 640, 2, 10000), (960, 1280, 1, 10000)])
def test_lattice_residuals_fast(bt, rows, cols, lanes, ns):
    """FAST lattice residuals: the oracle's SELECTION at every sample (same validity, same point-sampled source pixel: round 4, csrc/guard_band.h), values
    to rounding (intensity: 1/256 weight steps), and bit-identical to the residuals of the device's FAST warp pair -- the lattice and the normal
    equations that follow see the same W1 / I1"""
    K, L = gn_case(rows, cols, lanes, 42)
    dm = dev_maps(L, ("Wc", "W0", "Ic", "I0"))
    Rs, ts = [d["Rp"] for d in L], [d["tp"] for d in L]
    from rgbid import device
    n, lr, lc, st = device.error_lattice_size(rows, cols, ns)
    res = torch.zeros((lanes, 2 * n), device="cuda")
    bt.lattice_residuals(*dm, Rs, ts, ns, res, fast=True)
    got = res.cpu().numpy()
    W1f, I1f = torch.empty_like(dm[0]), torch.empty_like(dm[0])
    bt.warp_pair(dm[0], dm[2], dm[1], W1f, I1f, Rs, ts, fast=True)
    W1f, I1f = W1f.cpu().numpy(), I1f.cpu().numpy()
    for l, d in enumerate(L):
        ed, _ = O.error_lattice(d["W1"], d["W0"], ns)
        ei, _ = O.error_lattice(d["I1"], d["I0"], ns)
        assert_bits(got[l, :n], O.error_lattice(W1f[l], d["W0"], ns)[0], 0, "fast lattice == fast warp pair (iD)")
        assert_bits(got[l, n:], O.error_lattice(I1f[l], d["I0"], ns)[0], 0, "fast lattice == fast warp pair (intensity)")
        nan_mis = int(np.count_nonzero(np.isnan(got[l, :n]) != np.isnan(ed)) + np.count_nonzero(np.isnan(got[l, n:]) != np.isnan(ei)))
        both = ~np.isnan(got[l, :n]) & ~np.isnan(ed)
        moved = int(np.count_nonzero(np.abs(got[l, :n][both] - ed[both]) > 1e-5 * np.abs(d["W1"][::st, ::st][:lr, :lc].reshape(-1)[both])))
        bi = ~np.isnan(got[l, n:]) & ~np.isnan(ei)
        moved_i = int(np.count_nonzero(np.abs(got[l, n:][bi] - ei[bi]) > 0.5))
        assert nan_mis == 0 and moved == 0 and moved_i <= max(4, 2e-3 * n), (nan_mis, moved, moved_i, n)


def _fuse_case(rows, cols, lanes, seed):
    K = K_for(rows, cols)
    L = []
    for l in range(lanes):
        r = util.rng(1000 * seed + l)
        kf = util.rand_invdepth(r, rows, cols, nan_frac=0.08)
        cur = (kf + r.normal(0, 0.004, kf.shape)).astype(np.float32)
        cur[r.random(kf.shape) < 0.05] = np.nan
        kfw = r.uniform(0.5, 4.0, kf.shape).astype(np.float32)
        ww = r.uniform(0.5, 2.0, kf.shape).astype(np.float32)
        R, t = util.small_motion(r, K, 0.004 + 0.004 * l, 0.2 + 0.2 * l)
        Rp, tp = util.project(K, *util.inv_pose(R, t))
        od, ow = O.warp_invdepth_weighted(cur, kf, Rp, tp, weight_init=ww)
        okf, okfw = O.integrate_warped(od, ow, kf, kfw)
        L.append(dict(kf=kf, cur=cur, kfw=kfw, ww=ww, Rp=Rp, tp=tp, okf=okf, okfw=okfw, ow=ow))
    return K, L


@pytest.mark.parametrize("rows,cols,lanes", VEC_GEOMS)
def test_fuse_frame(bt, rows, cols, lanes):
    """k_fuse_frame4: warpInvDepthWithTrafo3DWeighted + integrateWarpedFrame (warping_registration.cu:549-669) in one pass.  EXACT: fused map,
    weight and the warped-weight buffer bit-identical to the oracle's two steps; FAST: the oracle's selection at every pixel (validity, point-sampled
    source pixel, fusion gate: round 4, csrc/guard_band.h), values to rounding"""
    K, L = _fuse_case(rows, cols, lanes, 51)
    Rs, ts = [d["Rp"] for d in L], [d["tp"] for d in L]
    cur, kf, kfw, ww = dev_maps(L, ("cur", "kf", "kfw", "ww"), pad=4)
    bt.fuse_frame(cur, kf, kfw, ww, Rs, ts, fast=False)
    for l, d in enumerate(L):
        assert_bits(kf[l].cpu().numpy(), d["okf"], 0, "fused iD")
        assert_bits(kfw[l].cpu().numpy(), d["okfw"], 0, "fused weight")
        assert_bits(ww[l].cpu().numpy(), d["ow"], 0, "warped weight buffer")
    cur, kf, kfw, ww = dev_maps(L, ("cur", "kf", "kfw", "ww"))
    bt.fuse_frame(cur, kf, kfw, ww, Rs, ts, fast=True)
    n = rows * cols
    for l, d in enumerate(L):
        g, gw = kf[l].cpu().numpy(), kfw[l].cpu().numpy()
        assert int(np.count_nonzero(np.isnan(g) != np.isnan(d["okf"]))) == 0
        both = ~np.isnan(g) & ~np.isnan(d["okf"])
        assert int(np.count_nonzero(np.abs(g[both] - d["okf"][both]) > 1e-5 * np.abs(d["okf"][both]))) == 0        # other source pixel / other gate decision
        assert int(np.count_nonzero(np.abs(gw[both] - d["okfw"][both]) > 1e-4 * np.abs(d["okfw"][both]))) == 0     # fused or not: the weight jumps
        assert np.median(np.abs(g[both] - d["okf"][both]) / np.abs(d["okf"][both])) < 2e-7


@pytest.mark.parametrize("rows,cols,lanes", VEC_GEOMS)
def test_kf_maps(bt, rows, cols, lanes):
    """k_kf_maps4: createVMap + computeGradientDepth + createNMapGradients (maps.cu:63-179, misc.cu:176-220) in one pass: every value the reference
    defines bit-identical (normals 2 ULP: rsqrt chain), planes 1 / 2 of invalid pixels NaN"""
    K = K_for(rows, cols)
    ws = [util.rand_invdepth(util.rng(6000 + l), rows, cols, nan_frac=0.05) for l in range(lanes)]
    w = stack(ws, pad=8)
    vm = torch.zeros((lanes, 3 * rows, cols), device="cuda")
    nm = torch.zeros((lanes, 3 * rows, cols), device="cuda")
    bt.kf_maps(K, w, vm, nm)
    for l in range(lanes):
        ov = O.vmap(ws[l], K)
        gx, gy = O.gradient(ws[l])
        on = O.nmap_gradients(ws[l], gx, gy, K)
        gv, gn = vm[l].cpu().numpy(), nm[l].cpu().numpy()
        valid, nvalid = ~np.isnan(ov[:rows]), ~np.isnan(on[:rows])
        assert_bits(gv[:rows], ov[:rows], 0, "vmap x")
        assert_bits(gn[:rows], on[:rows], 2, "nmap x")
        for p in (1, 2):
            assert_bits(gv[p * rows:(p + 1) * rows][valid], ov[p * rows:(p + 1) * rows][valid], 0, f"vmap plane {p}")
            assert np.isnan(gv[p * rows:(p + 1) * rows][~valid]).all()
            assert_bits(gn[p * rows:(p + 1) * rows][nvalid], on[p * rows:(p + 1) * rows][nvalid], 2, f"nmap plane {p}")
            assert np.isnan(gn[p * rows:(p + 1) * rows][~nvalid]).all()


@pytest.mark.parametrize("rows,cols,lanes", GEOMS)
def test_visibility_pair(bt, rows, cols, lanes):
    """k_visibility_pair: both directions of computeCovisibility (partialVisibilityKernel, warping_registration.cu:297-360) in one kernel.
    EXACT and FAST: the four integer counts equal the oracle's (FAST since round 4: csrc/guard_band.h)"""
    K = K_for(rows, cols)
    a, b, Rab, tab, Rba, tba, ref = [], [], [], [], [], [], []
    for l in range(lanes):
        r = util.rng(7000 + l)
        A_ = util.rand_invdepth(r, rows, cols, nan_frac=0.05)
        R, t = util.small_motion(r, K, 0.01 + 0.01 * l, 0.5 + 0.3 * l)
        Rp, tp = util.project(K, *util.inv_pose(R, t))
        Rq, tq = util.project(K, R, t)
        B_ = O.warp_invdepth(A_, A_, Rq, tq)        # the same surface seen from the other camera: a large share is covisible
        B_[np.isnan(B_) & (r.random(B_.shape) < 0.5)] = 0.6
        _, v1, n1, _ = O.visibility_ratio(A_, B_, Rp, tp)
        _, v2, n2, _ = O.visibility_ratio(B_, A_, Rq, tq)
        a.append(A_); b.append(B_); Rab.append(Rp); tab.append(tp); Rba.append(Rq); tba.append(tq)
        ref.append((int(v1), int(n1), int(v2), int(n2)))
    da, db = stack(a, pad=3), stack(b)
    counts = bt.visibility_pair(da, db, Rab, tab, Rba, tba, fast=False)
    assert [tuple(int(v) for v in c) for c in counts] == ref, (counts, ref)
    assert all(r_[0] > 0.3 * r_[1] for r_ in ref)                      # the case is not degenerate
    cf = bt.visibility_pair(da, db, Rab, tab, Rba, tba, fast=True)
    assert [tuple(int(v) for v in c) for c in cf] == ref, (cf, ref)


@pytest.mark.parametrize("rows,cols,lanes", GEOMS)
@pytest.mark.parametrize("factor", [1.0, 0.96])
def test_prep_frame(bt, rows, cols, lanes, factor):
    """k_prep_frame4 (and the three-kernel fallback at 83 columns): convertDepth2InvDepth + computeIntensity + decomposeRGBInChannels
    (misc.cu:105-172) in one pass, bit-exact"""
    ds, cs = [], []
    for l in range(lanes):
        r = util.rng(8000 + l)
        d = r.integers(0, 12000, (rows, cols)).astype(np.uint16)
        d[r.random((rows, cols)) < 0.1] = 0
        d[0, 0] = 65535
        ds.append(d); cs.append(r.integers(0, 256, (rows, cols, 3)).astype(np.uint8))
    dd = stack([d.view(np.int16) for d in ds])
    dc = stack(cs)
    outs = [torch.zeros((lanes, rows, cols), device="cuda") for _ in range(5)]
    bt.prep_frame(dd, dc, *outs, factor)
    for l in range(lanes):
        assert_bits(outs[0][l].cpu().numpy(), O.depth2invdepth(ds[l], factor), 0, "iD")
        assert_bits(outs[1][l].cpu().numpy(), O.intensity(cs[l]), 0, "luma")
        for got, ref in zip(outs[2:], O.decompose_rgb(cs[l])):
            assert_bits(got[l].cpu().numpy(), ref, 0, "channel")


@pytest.mark.parametrize("rows,cols,lanes", [GEOMS[0], GEOMS[2], GEOMS[4]])
def test_gradient_keep(bt, rows, cols, lanes):
    """k_gradient4<true>, the engine's keyframe-switch pass: Sobel pair of a map + a copy of the map in one launch -- gradients bit-exact against the
    oracle (NaN pattern included), the copy bit-identical to the source (NaN payloads included), padded rows left alone; a geometry off the 16-byte
    path is refused"""
    srcs = [util.rand_invdepth(util.rng(9300 + l), rows, cols, nan_frac=0.1) for l in range(lanes)]
    s = stack(srcs, pad=4)
    full = [torch.full((lanes, rows, cols + 4), 7.0, device="cuda") for _ in range(3)]
    gx, gy, keep = [t[:, :, :cols] for t in full]
    bt.gradient_keep(s, gx, gy, keep)
    for l in range(lanes):
        ogx, ogy = O.gradient(srcs[l])
        assert_bits(gx[l].cpu().numpy(), ogx, 0, "gx"); assert_bits(gy[l].cpu().numpy(), ogy, 0, "gy")
        assert np.array_equal(keep[l].cpu().numpy().view(np.uint32), srcs[l].view(np.uint32))
    assert all(bool((t[:, :, cols:] == 7.0).all()) for t in full)
    odd = torch.zeros((1, 9, 83), device="cuda")
    with pytest.raises(Exception):
        bt.gradient_keep(odd, torch.zeros_like(odd), torch.zeros_like(odd), torch.zeros_like(odd))


@pytest.mark.parametrize("rows,cols,lanes", [GEOMS[0], GEOMS[1], GEOMS[4]])
def test_batched_stencils(bt, rows, cols, lanes):
    """natively batched pyrDown / Sobel / bilateral (EXACT and the engine's FAST class) per lane against the oracle"""
    srcs = [util.rand_invdepth(util.rng(9000 + l), rows, cols, nan_frac=0.1) for l in range(lanes)]
    ints = [util.rand_intensity(util.rng(9100 + l), rows, cols) for l in range(lanes)]
    s, si = stack(srcs, pad=4), stack(ints)
    dst = torch.zeros((lanes, rows // 2, cols // 2), device="cuda")
    bt.pyr_down(s, dst)
    gx, gy = torch.zeros_like(si), torch.zeros_like(si)
    bt.gradient(si, gx, gy)
    be, bf = torch.zeros_like(si), torch.zeros_like(si)
    bt.bilateral(si, be, 3.0, fast=False)
    bt.bilateral(si, bf, 3.0, fast=True)
    de, df = torch.zeros((lanes, rows, cols), device="cuda"), torch.zeros((lanes, rows, cols), device="cuda")
    bt.bilateral(s, de, 2 * 0.0025, fast=False)
    bt.bilateral(s, df, 2 * 0.0025, fast=True)
    for l in range(lanes):
        assert_bits(dst[l].cpu().numpy(), O.pyr_down(srcs[l]), 4, "pyrDown")
        ogx, ogy = O.gradient(ints[l])
        assert_bits(gx[l].cpu().numpy(), ogx, 0, "gx"); assert_bits(gy[l].cpu().numpy(), ogy, 0, "gy")
        for exact, fastm, src, sg in ((be, bf, ints[l], 3.0), (de, df, srcs[l], 2 * 0.0025)):
            ref = O.bilateral(src, sg)
            assert_bits(exact[l].cpu().numpy(), ref, 8, "bilateral")
            g = fastm[l].cpu().numpy()
            assert np.array_equal(np.isnan(g), np.isnan(ref))
            m = ~np.isnan(ref)
            assert (np.abs(g[m] - ref[m]) <= 2e-6 * np.abs(ref[m]) + 1e-30).all()


def test_plain_c_consumer(tmp_path):
    """tests/cpp/test_batched_c.c: rgbid_batched.h, rgbid_engine.h and rgbid_dist.h from plain C (gcc -std=c99): frame preparation -> Sobel -> fused
    Gauss-Newton evaluation with the identity known answer in both numerics classes, the FAST refusal on odd geometry, two engine steps, the
    partition helpers"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "rgbid-slam_amd", "lib")
    exe = str(tmp_path / "test_batched_c")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "test_batched_c.c"),
                           "-L" + lib, "-lrgbid_dist", "-lrgbid_hip", "-lm", "-Wl,-rpath," + lib, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all ok" in r.stdout and "FAILED" not in r.stdout, r.stdout[-2000:] + r.stderr[-500:]
