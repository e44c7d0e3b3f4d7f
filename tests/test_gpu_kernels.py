"""GPU parity tests: every C-ABI kernel wrapper vs the CPU oracle on the same seeded inputs.

Tolerances (stated per test):
  * integer / index / validity work (NaN patterns, u8 masks, counts, lattice geometry): bit-exact;
  * element-wise fp32 kernels whose arithmetic is evaluated operation-by-operation on both sides
    (converters, Sobel, warps, fusion, vmap): bit-exact (0 ULP);
  * kernels through expf / logf (pyrDown, bilateral, sigma/nu): 4 ULP resp. rel 2e-5;
  * the normal equations (fp32 per-thread partial sums, fast reciprocals): |dA_ij| <= 2e-5 sqrt(A_ii A_jj).
All sizes run through the C-ABI (rgbid.device.Context -> librgbid_hip.so); nothing here falls back to the CPU.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import util
from tests.util import assert_bits, assert_rel

pytestmark = pytest.mark.gpu

SIZES = [(48, 64), (61, 83), (480, 640)]
SMALL = [(48, 64), (61, 83)]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def K_for(rows, cols):
    s = cols / 640.0
    return (525.0 * s, 525.0 * s, 319.5 * s, 239.5 * s)


def new(rows, cols, dtype=torch.float32):
    return torch.full((rows, cols), float("nan") if dtype == torch.float32 else 0, dtype=dtype, device="cuda")


@pytest.mark.parametrize("rows,cols", SIZES)
@pytest.mark.parametrize("factor", [1.0, 0.96])
def test_depth_to_invdepth(ctx, rows, cols, factor):
    r = util.rng(1)
    d = r.integers(0, 12000, (rows, cols)).astype(np.uint16)
    d[r.random((rows, cols)) < 0.1] = 0
    d[0, 0] = 65535
    out = new(rows, cols)
    ctx.convertDepth2InvDepth(dev(d.view(np.int16)), out, factor)
    assert_bits(out.cpu().numpy(), O.depth2invdepth(d, factor), 0, "depth2invdepth")


@pytest.mark.parametrize("rows,cols", SIZES)
def test_intensity_and_decompose(ctx, rows, cols):
    r = util.rng(2)
    rgb = r.integers(0, 256, (rows, cols, 3)).astype(np.uint8)
    out = new(rows, cols)
    ctx.computeIntensity(dev(rgb), out)
    assert_bits(out.cpu().numpy(), O.intensity(rgb), 0, "intensity")
    ch = [new(rows, cols) for _ in range(3)]
    ctx.decomposeRGBInChannels(dev(rgb), *ch)
    for a, b in zip(ch, O.decompose_rgb(rgb)):
        assert_bits(a.cpu().numpy(), b, 0, "decompose")


@pytest.mark.parametrize("rows,cols", SIZES)
def test_gradient(ctx, rows, cols):
    r = util.rng(3)
    src = util.rand_invdepth(r, rows, cols, nan_frac=0.03)
    gx, gy = new(rows, cols), new(rows, cols)
    ctx.computeGradient(util.padded(dev(src)), gx, gy)
    ogx, ogy = O.gradient(src)
    assert_bits(gx.cpu().numpy(), ogx, 0, "gradient x")
    assert_bits(gy.cpu().numpy(), ogy, 0, "gradient y")


@pytest.mark.parametrize("rows,cols", SIZES)
def test_pyr_down(ctx, rows, cols):
    r = util.rng(4)
    for src in (util.rand_invdepth(r, rows, cols, nan_frac=0.3), util.rand_intensity(r, rows, cols)):
        dst = new(rows // 2, cols // 2)
        ctx.pyrDown(dev(src), dst)
        ref = O.pyr_down(src)
        got = dst.cpu().numpy()
        # validity (count > 12) is integer work: the NaN pattern must be identical; values go through expf
        assert_bits(got, ref, 4, "pyrDown")
    if rows % 2 or cols % 2:
        return
    # NaN-free input, even size: three corners are invalid at every level
    src = util.rand_intensity(r, rows, cols)
    dst = new(rows // 2, cols // 2)
    ctx.pyrDown(dev(src), dst)
    g = dst.cpu().numpy()
    assert np.isnan(g[0, 0]) and np.isnan(g[0, -1]) and np.isnan(g[-1, 0]) and not np.isnan(g[-1, -1])
    assert np.count_nonzero(np.isnan(g)) == 3


@pytest.mark.parametrize("rows,cols", SIZES)
@pytest.mark.parametrize("sigma", [0.005, 3.0])
def test_bilateral(ctx, rows, cols, sigma):
    r = util.rng(5)
    src = util.rand_invdepth(r, rows, cols, nan_frac=0.05) if sigma < 1 else util.rand_intensity(r, rows, cols)
    dst = new(rows, cols)
    ctx.bilateralFilter(dev(src), dst, sigma)
    assert_bits(dst.cpu().numpy(), O.bilateral(src, sigma), 8, "bilateral")


@pytest.mark.parametrize("rows,cols", SIZES)
@pytest.mark.parametrize("sigma", [2.0 * 0.0025, 3.0])
def test_bilateral_fast_numerics(ctx, rows, cols, sigma):
    """rgbid_ctx_set_numerics(FAST): the bilateral filter in the reference build's class of arithmetic (v_exp_f32 on a folded exponent, what
    __expf is on the reference's GPU) agrees with the IEEE oracle to 2e-6 relative, with the same NaN pattern"""
    r = util.rng(5)
    src = util.rand_invdepth(r, rows, cols, nan_frac=0.05) if sigma < 1 else util.rand_intensity(r, rows, cols)
    dst = new(rows, cols)
    ctx.set_numerics(True)
    try:
        ctx.bilateralFilter(dev(src), dst, sigma)
    finally:
        ctx.set_numerics(False)
    got, ref = dst.cpu().numpy(), O.bilateral(src, sigma)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    m = ~np.isnan(ref)
    assert (np.abs(got[m] - ref[m]) <= 2e-6 * np.abs(ref[m]) + 1e-30).all(), float((np.abs(got[m] - ref[m]) / np.abs(ref[m])).max())


_BIL_WORKER = """
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/rgbid-slam_amd")
from rgbid import device, batched
d = np.load(sys.argv[2])
ctx = device.Context(0); bt = batched.Batched(ctx)
out = {}
for k in d.files:
    a = torch.from_numpy(d[k]).cuda()
    o = torch.full_like(a, 7.0)
    bt.bilateral(a, o, 3.0 if k.startswith("I") else 2 * 0.0025, fast=True)
    out[k] = o.cpu().numpy()
    if a.shape[1] >= 8 and a.shape[2] >= 8:
        h = torch.full((a.shape[0], a.shape[1] // 2, a.shape[2] // 2), 7.0, device="cuda")
        bt.pyr_down(a, h)
        out["pyr_" + k] = h.cpu().numpy()
np.savez(sys.argv[3], **out)
"""


def test_bilateral_shared_weights_equals_the_two_sided_kernel(ctx, tmp_path):
    """round 6: the engine's FAST bilateral filter evaluates every pair weight ONCE (k_bilateral_shared: forward taps, the backward ones from the neighbouring
    lanes / earlier steps) -- the pair weight is symmetric bit for bit and every output adds its taps in the same order, so the result must be the two-sided
    kernel's (k_bilateral<2>, kept behind RGBID_BILATERAL_TWO_SIDED for this test and for A/B timing) BYTE for byte: sizes that are / are not multiples of the
    60-column strips and 30 / 60-row blocks, several lanes, NaN, the sentinel's special values.  (Both are held to the oracle by test_bilateral_fast_numerics,
    tests/test_gpu_batched.py and the fuzz suite.)"""
    import os
    import subprocess
    import sys
    from rgbid import batched
    r = util.rng(606)
    specials = np.array([np.inf, -np.inf, 3e38, 1e19, -1e19, 1e10, 1e9, 9.9e8, -9.9e8, 0.0, -0.0, 1e-45, 1e-39], np.float32)
    maps = {}
    for i, (B, rows, cols) in enumerate([(1, 480, 640), (3, 120, 160), (2, 61, 83), (40, 65, 127), (1, 5, 5), (2, 31, 61), (1, 92, 181)]):
        for kind in ("W", "I"):
            a = np.stack([util.rand_invdepth(r, rows, cols, nan_frac=0.1) if kind == "W" else util.rand_intensity(r, rows, cols, nan_frac=0.02) for _ in range(B)])
            idx = r.integers(0, a.size, size=max(1, a.size // 40))
            a.reshape(-1)[idx] = specials[r.integers(0, specials.size, size=idx.size)]
            maps[f"{kind}{i}"] = a.astype(np.float32)
    maps["I_clean"] = np.stack([util.rand_intensity(r, 480, 640) for _ in range(2)])      # no invalid pixel: the fast paths run everywhere but at the border
    maps["W_clean"] = np.stack([util.rand_invdepth(r, 240, 320, nan_frac=0.0) for _ in range(2)])
    np.savez(tmp_path / "in.npz", **maps)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RGBID_BILATERAL_TWO_SIDED="1", RGBID_PYRDOWN_NO_FASTPATH="1")
    subprocess.run([sys.executable, "-c", _BIL_WORKER, root, str(tmp_path / "in.npz"), str(tmp_path / "two_sided.npz")], check=True, env=env, timeout=600)
    ref = np.load(tmp_path / "two_sided.npz")
    bt = batched.Batched(ctx)
    for k, a in maps.items():
        src = dev(a); o = torch.full_like(src, 7.0)
        bt.bilateral(src, o, 3.0 if k.startswith("I") else 2 * 0.0025, fast=True)
        got = o.cpu().numpy()
        assert got.tobytes() == ref[k].tobytes(), (k, a.shape, int(np.count_nonzero(got.view(np.uint32) != ref[k].view(np.uint32))))
        # ... and the pyramid reduction's all-valid fast path (round 6: tap count and mask-weighted sum skipped where a wave's five window rows hold no
        # invalid tap) against the kernel without it
        if "pyr_" + k in ref.files:
            h = torch.full((a.shape[0], a.shape[1] // 2, a.shape[2] // 2), 7.0, device="cuda")
            bt.pyr_down(src, h)
            assert h.cpu().numpy().tobytes() == ref["pyr_" + k].tobytes(), ("pyrDown", k, a.shape)


def _warp_case(rows, cols, seed, trans=0.03, rot=1.5):
    r = util.rng(seed)
    K = K_for(rows, cols)
    grid = util.rand_invdepth(r, rows, cols, nan_frac=0.05)
    src = util.rand_invdepth(r, rows, cols, nan_frac=0.05)
    inten = util.rand_intensity(r, rows, cols)
    R, t = util.small_motion(r, K, trans, rot)
    Rp, tp = util.project(K, *util.inv_pose(R, t))
    return K, grid, src, inten, Rp, tp


@pytest.mark.parametrize("rows,cols", SIZES)
def test_warp_invdepth(ctx, rows, cols):
    K, grid, src, inten, Rp, tp = _warp_case(rows, cols, 6)
    dst = new(rows, cols)
    ctx.warpInvDepthWithTrafo3D(dev(src), dst, dev(grid), Rp, tp)
    assert_bits(dst.cpu().numpy(), O.warp_invdepth(src, grid, Rp, tp), 0, "warp iD")


@pytest.mark.parametrize("rows,cols", SIZES)
@pytest.mark.parametrize("mode", [O.INTERP_EXACT, O.INTERP_TEX8])
def test_warp_intensity(ctx, rows, cols, mode):
    K, grid, src, inten, Rp, tp = _warp_case(rows, cols, 7)
    # pyramid levels >= 1 carry NaN corners in the intensity map: exercise the NaN -> 255 clamp
    inten[0, 0] = np.nan
    dst = new(rows, cols)
    ctx.set_interp_mode(mode)
    try:
        ctx.warpIntensityWithTrafo3DInvDepth(dev(inten), dst, dev(grid), Rp, tp)
    finally:
        ctx.set_interp_mode(O.INTERP_TEX8)
    assert_bits(dst.cpu().numpy(), O.warp_intensity(inten, grid, Rp, tp, mode), 0, "warp intensity")


@pytest.mark.parametrize("rows,cols", SIZES)
def test_warp_pair_exact_is_the_two_warps(ctx, rows, cols):
    """rgbid_warp_pair with EXACT numerics: both maps bit-identical to the oracle's two warps (the intensity warp sampled on the WARPED iD)"""
    K, grid, src, inten, Rp, tp = _warp_case(rows, cols, 6)
    d1, d2 = new(rows, cols), new(rows, cols)
    ctx.warpPair(dev(src), dev(inten), dev(grid), d1, d2, Rp, tp, fast=False)
    w1 = O.warp_invdepth(src, grid, Rp, tp)
    assert_bits(d1.cpu().numpy(), w1, 0, "pair iD")
    assert_bits(d2.cpu().numpy(), O.warp_intensity(inten, w1, Rp, tp, O.INTERP_TEX8), 0, "pair intensity")


def test_selftest_cvt_flr_exhaustive(ctx):
    """v_cvt_flr_i32_f32 == v_floor_f32 + v_cvt_i32_f32 (saturating) for every non-NaN one of the 2^32 float bit patterns"""
    assert ctx.selftest_cvt_flr(1) == 0


def test_selftest_fast_primitives_exhaustive(ctx):
    """the hardware facts under the guard band (csrc/guard_band.h) over all 2^32 floats: v_rcp_f32 within 1 ulp, v_med3_f32 as the domain clamp (NaN ->
    the lower bound), v_fract_f32(x) == x - floor(x)"""
    assert ctx.selftest_fast_primitives() == 0


@pytest.mark.parametrize("rows,cols", SIZES)
@pytest.mark.parametrize("seed", [6, 21])
def test_warp_pair_fast_selects_the_oracles_pixels(ctx, rows, cols, seed):
    """FAST numerics (csrc/warp_device.h fastnum: v_rcp_f32, FMAs, the scaled point) against the IEEE oracle: the pixel-selection parity statement of
    DESIGN.md section 4.1.  Since round 4 EVERY pixel point-samples the oracle's source pixel and carries the oracle's validity (guard band + exact
    recomputation of the pixels inside it, csrc/guard_band.h): NaN patterns identical, no warped inverse depth off by more than rounding.  What remains
    are float values in their last bits and the 1.8 fixed-point bilinear weight one 1/256 step off where the coordinate sits within the error bound of a
    step (a value change of at most 2/256 of the local contrast; no selection)."""
    K, grid, src, inten, Rp, tp = _warp_case(rows, cols, seed)
    d1, d2 = new(rows, cols), new(rows, cols)
    ctx.warpPair(dev(src), dev(inten), dev(grid), d1, d2, Rp, tp, fast=True)
    g1, g2 = d1.cpu().numpy(), d2.cpu().numpy()
    w1 = O.warp_invdepth(src, grid, Rp, tp)
    i1 = O.warp_intensity(inten, w1, Rp, tp, O.INTERP_TEX8)
    n = rows * cols
    nan_mis = int(np.count_nonzero(np.isnan(g1) != np.isnan(w1)))
    both = ~np.isnan(g1) & ~np.isnan(w1)
    rel = np.abs(g1[both] - w1[both]) / np.abs(w1[both])
    other_px = int(np.count_nonzero(rel > 1e-5))                     # a different source pixel was point-sampled
    assert nan_mis == 0 and other_px == 0, (nan_mis, other_px, n)
    assert np.median(rel) < 2e-7
    # the intensity warp samples at the warped iD: its validity is the oracle's too (the device's W1 has the oracle's NaN pattern and its values to rounding)
    nan_mis_i = int(np.count_nonzero(np.isnan(g2) != np.isnan(i1)))
    assert nan_mis_i == 0, (nan_mis_i, n)
    bi = ~np.isnan(g2) & ~np.isnan(i1)
    di = np.abs(g2[bi] - i1[bi])
    assert np.count_nonzero(di > 0.5) <= max(8, 2e-3 * n), (np.count_nonzero(di > 0.5), n)      # 1/256 weight steps x contrast
    assert di.max() <= 2.0 * 255.0 / 256.0 + 1e-3 and np.median(di) < 1e-3
    # how often the 1.8 fixed-point weight lands on the neighbouring 1/256 step (a VALUE difference, bounded above; DESIGN.md section 4.1 says why it is not guarded)
    step_frac = float(np.count_nonzero(di > 2e-3)) / max(1, di.size)
    assert step_frac < 0.08, step_frac
    print(f"fast vs oracle {cols}x{rows}: NaN-pattern {nan_mis} / {nan_mis_i}, other source pixel {other_px} of {int(both.sum())}; intensity > 0.5 grey levels: {int(np.count_nonzero(di > 0.5))}, max {di.max():.3f}; "
          f"samples whose bilinear weight sits one 1/256 step off: {100 * step_frac:.2f} %")


@pytest.mark.parametrize("rows,cols", SMALL)
def test_warp_identity_is_exact(ctx, rows, cols):
    """Analytic KAT: identity transform => W1 == W0 wherever W0 is valid, I1 == I0 (exact bilinear at integer coords)."""
    r = util.rng(8)
    K = K_for(rows, cols)
    w0 = util.rand_invdepth(r, rows, cols, nan_frac=0.1)
    i0 = util.rand_intensity(r, rows, cols)
    Rp, tp = util.project(K, np.eye(3), np.zeros(3))
    d1, d2 = new(rows, cols), new(rows, cols)
    ctx.warpInvDepthWithTrafo3D(dev(w0), d1, dev(w0), Rp, tp)
    ctx.warpIntensityWithTrafo3DInvDepth(dev(i0), d2, dev(w0), Rp, tp)
    g1, g2 = d1.cpu().numpy(), d2.cpu().numpy()
    valid = ~np.isnan(w0)
    # projection in fp32 may land a hair off the pixel centre; the point sample is still the same pixel
    assert np.array_equal(np.isnan(g1), ~valid)
    np.testing.assert_allclose(g1[valid], w0[valid], rtol=2e-6)
    np.testing.assert_allclose(g2[valid], i0[valid], atol=0.51)  # TEX8 weights: <= 1/512 of a neighbour difference


@pytest.mark.parametrize("rows,cols", SIZES)
def test_warp_weighted_and_integrate(ctx, rows, cols):
    K, grid, src, inten, Rp, tp = _warp_case(rows, cols, 9, trans=0.01, rot=0.5)
    r = util.rng(10)
    winit = r.uniform(0.5, 2.0, (rows, cols)).astype(np.float32)
    dst, w = new(rows, cols), dev(winit.copy())
    ctx.warpInvDepthWithTrafo3DWeighted(dev(src), dst, dev(grid), w, Rp, tp)
    od, ow = O.warp_invdepth_weighted(src, grid, Rp, tp, weight_init=winit)
    assert_bits(dst.cpu().numpy(), od, 0, "weighted warp iD")
    assert_bits(w.cpu().numpy(), ow, 0, "weighted warp weight")
    kf = grid.copy(); kf[r.random((rows, cols)) < 0.05] = np.nan
    kfw = np.ones((rows, cols), np.float32)
    dkf, dkfw = dev(kf.copy()), dev(kfw.copy())
    ctx.integrateWarpedFrame(dst, w, dkf, dkfw)
    okf, okfw = O.integrate_warped(od, ow, kf, kfw)
    assert_bits(dkf.cpu().numpy(), okf, 0, "fused iD")
    assert_bits(dkfw.cpu().numpy(), okfw, 0, "fused weight")


@pytest.mark.parametrize("rows,cols", SIZES)
def test_visibility(ctx, rows, cols):
    K, grid, src, inten, Rp, tp = _warp_case(rows, cols, 11)
    # dst = the surface seen from the other camera: use the warped map so a large share is "visible"
    dstmap = O.warp_invdepth(src, grid, Rp, tp)
    ratio = ctx.getVisibilityRatio(dev(src), dev(dstmap), Rp, tp)
    oratio, nvis, nval, _ = O.visibility_ratio(src, dstmap, Rp, tp)
    assert ratio == oratio, (ratio, oratio, nvis, nval)
    mask0 = util.rng(12).integers(0, 2, (rows, cols)).astype(np.uint8)
    m = dev(mask0.copy())
    ratio2 = ctx.getVisibilityRatioWithOverlapMask(dev(src), dev(dstmap), Rp, tp, overlap_mask=m)
    _, _, _, omask = O.visibility_ratio(src, dstmap, Rp, tp, with_mask=True, mask_init=mask0)
    assert ratio2 == oratio
    assert np.array_equal(m.cpu().numpy(), omask)


@pytest.mark.parametrize("rows,cols", SIZES)
def test_vmap_nmap_image(ctx, rows, cols):
    r = util.rng(13)
    K = K_for(rows, cols)
    w = util.rand_invdepth(r, rows, cols, nan_frac=0.05)
    vm = torch.zeros((3 * rows, cols), device="cuda")
    ctx.createVMap(K, dev(w), vm)
    ov = O.vmap(w, K)
    got = vm.cpu().numpy()
    valid = ~np.isnan(ov[:rows])
    assert_bits(got[:rows], ov[:rows], 0, "vmap x")
    for p in (1, 2):  # planes 1,2 are only written where valid (maps.cu:78-85)
        assert_bits(got[p * rows:(p + 1) * rows][valid], ov[p * rows:(p + 1) * rows][valid], 0, f"vmap plane {p}")
    gx, gy = O.gradient(w)
    nm = torch.zeros((3 * rows, cols), device="cuda")
    ctx.createNMapGradients(K, dev(w), dev(gx), dev(gy), nm)
    on = O.nmap_gradients(w, gx, gy, K)
    gn = nm.cpu().numpy()
    nvalid = ~np.isnan(on[:rows])
    assert_bits(gn[:rows], on[:rows], 2, "nmap x")
    for p in (1, 2):
        assert_bits(gn[p * rows:(p + 1) * rows][nvalid], on[p * rows:(p + 1) * rows][nvalid], 2, f"nmap plane {p}")
    rgb = r.integers(0, 256, (rows, cols, 3)).astype(np.uint8)
    out = torch.zeros((rows, cols, 3), dtype=torch.uint8, device="cuda")
    light = (0.1, -0.05, 0.02)
    ctx.generateImageRGB(dev(ov), dev(on), dev(rgb), light, out)
    oi = O.generate_image_rgb(ov, on, rgb, light)
    diff = np.abs(out.cpu().numpy().astype(int) - oi.astype(int))
    assert diff.max() <= 1 and np.count_nonzero(diff) <= 1e-3 * diff.size, (diff.max(), np.count_nonzero(diff))


@pytest.mark.parametrize("rows,cols,ns", [(480, 640, 10000), (240, 320, 10000), (120, 160, 10000), (480, 640, 9999999), (61, 83, 1000), (960, 1280, 10000)])
def test_error_lattice(ctx, rows, cols, ns):
    r = util.rng(14)
    a, b = util.rand_intensity(r, rows, cols, nan_frac=0.02), util.rand_intensity(r, rows, cols)
    from rgbid import device
    oerr, geo = O.error_lattice(a, b, ns)
    n, lr, lc, st = device.error_lattice_size(rows, cols, ns)
    assert (n, lr, lc, st) == (oerr.size, geo[0], geo[1], geo[2])  # integer geometry: exact
    if (rows, cols, ns) in ((480, 640, 10000), (240, 320, 10000), (120, 160, 10000), (960, 1280, 10000)):
        assert n == 19200  # always the 160x120 lattice
    err = torch.zeros(rows * cols, device="cuda")
    n2 = ctx.computeErrorGridStride(dev(a), dev(b), err, ns)
    assert n2 == n
    assert_bits(err[:n].cpu().numpy(), oerr, 0, "error lattice")


def _student_samples(r, n, nu, sigma, bias, outliers=0.02):
    e = (bias + sigma * r.standard_t(nu, n)).astype(np.float32)
    k = int(outliers * n)
    e[r.integers(0, n, k)] = np.nan
    return e


@pytest.mark.parametrize("n", [19200, 4800, 307200])
@pytest.mark.parametrize("nu,sigma,bias,s0", [(3.0, 0.004, 0.0005, 0.0025), (5.0, 7.0, -1.0, 5.0), (8.0, 2.0, 0.0, 5.0)])
def test_sigma_nu_student(ctx, n, nu, sigma, bias, s0):
    r = util.rng(15)
    e = _student_samples(r, n, nu, sigma, bias)
    b, s, v = ctx.computeSigmaAndNuStudent(dev(e), n, 0.0, s0, 5.0, O.STUDENT)
    ob, os_, ov = O.sigma_nu_student(e, 0.0, s0, 5.0, O.STUDENT)
    assert v == ov, (v, ov)                      # bisection grid value: exact
    assert abs(s - os_) <= 2e-5 * os_ and abs(b - ob) <= 2e-5 * os_, (b, ob, s, os_)
    assert abs(v - nu) <= 2.0                    # recovers the generating nu within the bisection resolution
    v2 = ctx.computeNuStudent(dev(e), n, ob, os_)
    assert v2 == O.nu_student(e, ob, os_)
    for mest in (O.LSQ, O.HUBER, O.TUKEY, O.STUDENT):
        b3, s3 = ctx.computeSigmaPdf(dev(e), n, 0.0, s0, mest)
        ob3, os3 = O.sigma_pdf(e, 0.0, s0, mest)
        assert abs(s3 - os3) <= 2e-5 * os3 and abs(b3 - ob3) <= 2e-5 * os3, (mest, b3, ob3, s3, os3)


@pytest.mark.parametrize("mest", [O.LSQ, O.HUBER, O.TUKEY, O.STUDENT])
def test_chi_square(ctx, mest):
    r = util.rng(16)
    n = 19200
    ei = _student_samples(r, n, 5.0, 5.0, 0.0)
    ed = _student_samples(r, n, 5.0, 0.0025, 0.0)
    x, t, d = ctx.computeChiSquare(dev(ei), dev(ed), n, 5.0, 0.0025, mest)
    ox, ot, od = O.chi_square(ei, ed, 5.0, 0.0025, mest)
    assert d == od
    assert abs(x - ox) <= 2e-5 * abs(ox) and abs(t - ot) <= 1e-5


def _system_case(rows, cols, seed):
    r = util.rng(seed)
    K = K_for(rows, cols)
    W0 = util.rand_invdepth(r, rows, cols, nan_frac=0.05)
    I0 = util.rand_intensity(r, rows, cols)
    gWx, gWy = O.gradient(W0)
    gIx, gIy = O.gradient(I0)
    R, t = util.small_motion(r, K, 0.01, 0.4)
    Rp, tp = util.project(K, *util.inv_pose(R, t))
    Wc = util.rand_invdepth(r, rows, cols, nan_frac=0.05)
    W1 = O.warp_invdepth(W0 * np.float32(1.001), W0, Rp, tp) + (0.002 * r.standard_normal((rows, cols))).astype(np.float32)
    I1 = O.warp_intensity(I0, W1, Rp, tp) + (3 * r.standard_normal((rows, cols))).astype(np.float32)
    return K, (W0, I0, gWx, gWy, gIx, gIy, W1.astype(np.float32), I1.astype(np.float32))


def _check_system(A, b, oA, ob):
    assert np.array_equal(A, A.T)
    d = np.sqrt(np.diag(oA))
    assert (np.abs(A - oA) <= 2e-5 * np.outer(d, d)).all(), np.abs(A - oA) / np.outer(d, d)
    # b_i = sum J_i w e: bound by sqrt(A_ii * sum w e^2) ~ sqrt(A_ii * N); compare on the scale of A_ii * |x|
    x = np.linalg.solve(oA, ob)
    scale = d * (d @ np.abs(x)) + 1e-30
    assert (np.abs(b - ob) <= 2e-5 * scale + 2e-5 * np.abs(ob)).all(), (np.abs(b - ob) / scale)


@pytest.mark.parametrize("rows,cols", SIZES + [(240, 320), (120, 160)])
@pytest.mark.parametrize("weighting", [O.INDEPENDENT, O.MIN_WEIGHT, O.GEOM_ONLY, O.PHOT_ONLY])
def test_build_system_student_nu(ctx, rows, cols, weighting):
    K, maps = _system_case(rows, cols, 17)
    if (rows, cols) == (480, 640) and weighting != O.INDEPENDENT:
        pytest.skip("full size covered by INDEPENDENT")
    args = dict(sigma_depthinv=0.003, sigma_int=6.0, bias_depthinv=0.0002, bias_int=-0.5, nu_depthinv=3.5, nu_int=6.25)
    dm = [util.padded(dev(m), 4) for m in maps]
    A, b = ctx.buildSystemStudentNuGridStride(*dm, O.STUDENT, weighting, args["sigma_depthinv"], args["sigma_int"], args["bias_depthinv"],
                                              args["bias_int"], args["nu_depthinv"], args["nu_int"], K)
    oA, ob = O.build_system(*maps, K, student_nu=True, mestimator=O.STUDENT, weighting=weighting, **args)
    _check_system(A, b, oA, ob)
    # determinism: fixed-order partial sums, no fp atomics
    A2, b2 = ctx.buildSystemStudentNuGridStride(*dm, O.STUDENT, weighting, args["sigma_depthinv"], args["sigma_int"], args["bias_depthinv"],
                                                args["bias_int"], args["nu_depthinv"], args["nu_int"], K)
    assert np.array_equal(A, A2) and np.array_equal(b, b2)


@pytest.mark.parametrize("rows,cols", [(37, 100), (50, 72), (45, 136), (9, 260), (130, 36)])
def test_build_system_ragged_tiles(ctx, rows, cols):
    """the 16-byte path's tile decomposition (kernels_system.hip SysTiles) on widths whose unit count no tile width divides (25, 18, 34, 65, 9 units
    per row: a ragged last strip) and heights that are no multiple of the tile height (ragged bottom tiles)"""
    K, maps = _system_case(rows, cols, 23)
    args = dict(sigma_depthinv=0.003, sigma_int=6.0, bias_depthinv=0.0002, bias_int=-0.5, nu_depthinv=3.5, nu_int=6.25)
    dm = [util.padded(dev(m), 4) for m in maps]
    assert all(d.stride(0) % 4 == 0 and d.data_ptr() % 16 == 0 for d in dm)     # the 16-byte path is the one taken
    A, b = ctx.buildSystemStudentNuGridStride(*dm, O.STUDENT, O.INDEPENDENT, args["sigma_depthinv"], args["sigma_int"], args["bias_depthinv"],
                                              args["bias_int"], args["nu_depthinv"], args["nu_int"], K)
    oA, ob = O.build_system(*maps, K, student_nu=True, mestimator=O.STUDENT, weighting=O.INDEPENDENT, **args)
    _check_system(A, b, oA, ob)


@pytest.mark.parametrize("rows,cols", SMALL + [(480, 640)])
@pytest.mark.parametrize("mest", [O.LSQ, O.HUBER, O.TUKEY, O.STUDENT])
def test_build_system_fixed_nu(ctx, rows, cols, mest):
    K, maps = _system_case(rows, cols, 18)
    dm = [dev(m) for m in maps]   # dense tensors: pitch == cols*4 (83 cols -> scalar path, 64/640 -> float4 path)
    A, b = ctx.buildSystemGridStride(*dm, mest, O.INDEPENDENT, 0.0025, 5.0, 0.0, 0.0, K)
    oA, ob = O.build_system(*maps, K, student_nu=False, mestimator=mest, weighting=O.INDEPENDENT,
                            sigma_depthinv=0.0025, sigma_int=5.0, bias_depthinv=0.0, bias_int=0.0)
    _check_system(A, b, oA, ob)


def test_identity_system_kat(ctx):
    """Analytic KAT: W1 == W0 and I1 == I0 => b == 0 exactly and A is symmetric PSD."""
    rows, cols = 48, 64
    K, maps = _system_case(rows, cols, 19)
    W0, I0, gWx, gWy, gIx, gIy, _, _ = maps
    dm = [dev(m) for m in (W0, I0, gWx, gWy, gIx, gIy, W0, I0)]
    A, b = ctx.buildSystemStudentNuGridStride(*dm, O.STUDENT, O.INDEPENDENT, 0.0025, 5.0, 0.0, 0.0, 5.0, 5.0, K)
    assert np.all(b == 0.0)
    assert np.linalg.eigvalsh(A).min() > -1e-6 * np.abs(A).max()


def test_invalid_arguments(ctx):
    """Error behaviour of the C-ABI: bad sizes return RGBID_E_INVALID (-1), never crash."""
    from rgbid._lib import RgbidError
    a, b = new(48, 64), new(20, 30)
    with pytest.raises(RgbidError):
        ctx.pyrDown(a, b)
    with pytest.raises(RgbidError):
        ctx.copyImage(a, b)
    with pytest.raises(RgbidError):
        ctx.bilateralFilter(a, a, 1.0)  # in-place is not allowed


# ---- bridge functions the reference defines but its tracker no longer calls (kept for a complete C++ surface) -----------------
@pytest.mark.parametrize("rows,cols", SIZES)
def test_unused_bridge_functions(ctx, rows, cols):
    r = util.rng(51)
    K = K_for(rows, cols)
    # convertDepth2Float / convertFloat2RGB
    d = r.integers(0, 12000, (rows, cols)).astype(np.uint16); d[r.random((rows, cols)) < 0.1] = 0
    out = new(rows, cols); ctx.convertDepth2Float(dev(d.view(np.int16)), out)
    assert_bits(out.cpu().numpy(), O.depth2float(d), 0, "depth2float")
    f = util.rand_intensity(r, rows, cols, nan_frac=0.05); f[1, 1] = np.inf; f[2, 2] = -7.3; f[3, 3] = 300.2; f[4, 4] = 127.5
    rgb = torch.zeros((rows, cols, 3), dtype=torch.uint8, device="cuda"); ctx.convertFloat2RGB(dev(f), rgb)
    assert np.array_equal(rgb.cpu().numpy(), O.float2rgb(f))
    # createNMap on a vertex map with holes
    w = util.rand_invdepth(r, rows, cols)
    vm = O.vmap(w, K)
    nm = new(3 * rows, cols); nm[:] = 0
    ctx.createNMap(dev(vm), nm)
    ref = O.nmap_cross(vm)
    got = nm.cpu().numpy()
    assert_bits(got[:rows], ref[:rows], 0, "nmap plane 0")
    ok = np.isfinite(ref[:rows])
    for p in (1, 2):
        assert np.array_equal(got[p * rows:(p + 1) * rows][ok], ref[p * rows:(p + 1) * rows][ok])
    n2 = got[:rows][ok] ** 2 + got[rows:2 * rows][ok] ** 2 + got[2 * rows:][ok] ** 2
    assert np.abs(n2 - 1).max() < 1e-5
    # integrateWarpedRGB
    kf = util.rand_invdepth(r, rows, cols, 0.1); ws = (kf + r.normal(0, 0.004, kf.shape)).astype(np.float32); ws[r.random(kf.shape) < 0.1] = np.nan
    ws[np.isnan(kf) & (r.random(kf.shape) < 0.5)] = 0.7
    cr, cg, cb = [util.rand_intensity(r, rows, cols, nan_frac=0.02) for _ in range(3)]
    qs = r.uniform(0.5, 2.0, kf.shape).astype(np.float32); q = r.uniform(0.5, 4.0, kf.shape).astype(np.float32)
    col = r.integers(0, 256, (rows, cols, 3)).astype(np.uint8)
    kfd, qd, cd = dev(kf.copy()), dev(q.copy()), dev(col.copy())
    ctx.integrateWarpedRGB(dev(ws), dev(cr), dev(cg), dev(cb), dev(qs), kfd, cd, qd)
    rk, rc, rq = O.integrate_warped_rgb(ws, cr, cg, cb, qs, kf, col, q)
    assert_bits(kfd.cpu().numpy(), rk, 0, "fused iD"); assert_bits(qd.cpu().numpy(), rq, 0, "fused weight")
    assert np.array_equal(cd.cpu().numpy(), rc)


def test_exact_reciprocal_is_ieee_for_every_float(ctx):
    """csrc/common.h rcp_exact (v_rcp_f32 + one FMA Newton step, full IEEE sequence outside [2^-126, 2^126)) == 1.0f / x for ALL 2^32
    float bit patterns -- the proof that the shorter instruction sequence cannot change any projected coordinate."""
    assert ctx.selftest_rcp() == 0


def test_bilateral_short_division_is_ieee_for_every_dividend(ctx):
    """csrc/common.h div_const_fast (multiply by the rounded reciprocal + two FMA corrections) == x / sigma for ALL 2^32 dividends x, for the two
    range sigmas the tracker filters with (visodo.cpp:843-844) -- the only divisors the bilateral kernel uses it for; any other divisor takes the
    IEEE division (and the filter is bit-exact against the oracle either way, test_bilateral / the fuzz suite)."""
    for sigma in (np.float32(2.0) * np.float32(0.0025), np.float32(3.0)):
        bad, used = ctx.selftest_div_const(float(sigma))
        assert used and bad == 0, (sigma, bad)
    for other in (0.7, 1e-3, 123.456):
        bad, used = ctx.selftest_div_const(other)
        assert not used


def test_c_abi_rejects_bad_arguments(ctx):
    """error convention of the C-ABI: 0 = ok, negative = RGBID_E_* for bad arguments (never a crash, never a silent no-op)"""
    import ctypes as C
    from rgbid import device, _lib
    L = _lib.lib()
    a = torch.zeros((48, 64), device="cuda"); b = torch.zeros((48, 64), device="cuda"); small = torch.zeros((24, 32), device="cuda")
    ia, ib, ism = device.img(a), device.img(b), device.img(small)
    ms = C.c_float()
    h = ctx._h
    assert L.rgbid_compute_gradient(h, C.byref(ia), C.byref(ib), C.byref(ism), C.byref(ms)) < 0          # size mismatch
    assert L.rgbid_compute_gradient(None, C.byref(ia), C.byref(ib), C.byref(ib), C.byref(ms)) < 0         # no context
    assert L.rgbid_compute_gradient(h, None, C.byref(ib), C.byref(ib), C.byref(ms)) < 0                   # null image
    bad = device.Img(0, 256, 48, 64)
    assert L.rgbid_pyr_down(h, C.byref(bad), C.byref(ism), C.byref(ms)) < 0                               # null data pointer
    assert L.rgbid_pyr_down(h, C.byref(ia), C.byref(ia), C.byref(ms)) < 0                                 # dst must be rows/2 x cols/2
    R = (C.c_float * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1); t = (C.c_float * 3)()
    assert L.rgbid_warp_invdepth(h, C.byref(ia), C.byref(ib), C.byref(ism), R, t, C.byref(ms)) < 0
    assert L.rgbid_warp_invdepth(h, C.byref(ia), C.byref(ib), C.byref(ia), None, t, C.byref(ms)) < 0
    A = (C.c_double * 36)(); bb = (C.c_double * 6)()
    k = device.Intr(50.0, 50.0, 31.5, 23.5)
    args = [C.byref(ia)] * 8
    assert L.rgbid_build_system_student_nu(h, *args, 3, 0, C.c_float(0.0025), C.c_float(5), C.c_float(0), C.c_float(0), C.c_float(5), C.c_float(5), k, None, bb, C.byref(ms)) < 0
    args[7] = C.byref(ism)
    assert L.rgbid_build_system_student_nu(h, *args, 3, 0, C.c_float(0.0025), C.c_float(5), C.c_float(0), C.c_float(0), C.c_float(5), C.c_float(5), k, A, bb, C.byref(ms)) < 0
    assert L.rgbid_error_string(-1) and L.rgbid_error_string(12345)
    e = C.c_void_p()
    cfg = __import__("rgbid.engine", fromlist=["x"]).default_config(rows=2, cols=2, levels=3, lanes=1)
    assert L.rgbid_engine_create(C.byref(e), h, C.byref(cfg)) < 0 and not e.value                         # pyramid too deep for the image
    # and a good call still works afterwards
    assert L.rgbid_compute_gradient(h, C.byref(ia), C.byref(ib), C.byref(ib), C.byref(ms)) == 0


def test_remaining_abi_entry_points(ctx):
    """the small entry points the tracker tests only reach indirectly: copies, fills, keyframe weight, raw memory functions, stream
    adoption, device queries, device-side record ring"""
    import ctypes as C
    from rgbid import device, _lib, engine as E
    L = _lib.lib()
    r = util.rng(61)
    rows, cols = 37, 53
    a = util.rand_invdepth(r, rows, cols); b = util.rand_intensity(r, rows, cols)
    da, db = new(rows, cols), new(rows, cols)
    ctx.copyImages(dev(a), dev(b), da, db)
    assert_bits(da.cpu().numpy(), a, 0, "copyImages depth"); assert_bits(db.cpu().numpy(), b, 0, "copyImages intensity")
    dc = new(rows, cols); ctx.copyImage(dev(a), dc); assert_bits(dc.cpu().numpy(), a, 0, "copyImage")
    rgb = r.integers(0, 256, (rows, cols, 3)).astype(np.uint8)
    drgb = torch.zeros((rows, cols, 3), dtype=torch.uint8, device="cuda"); ctx.copyImageRGB(dev(rgb), drgb)
    assert np.array_equal(drgb.cpu().numpy(), rgb)
    w = new(rows, cols); ctx.initialiseWeightKeyframe(dev(a), w)                      # misc.cu:272-287: weight 1 everywhere
    assert np.array_equal(w.cpu().numpy(), O.init_weight(a)) and (w.cpu().numpy() == 1).all()
    for t, bits, want in ((torch.empty((rows, cols), device="cuda"), 0x40490fdb, np.float32(np.pi)),
                          (torch.empty((rows, cols), dtype=torch.int32, device="cuda"), 0xfffffffe, -2),
                          (torch.empty((rows, cols), dtype=torch.uint8, device="cuda"), 7, 7)):
        ctx.initialiseDeviceMemory2D(t, bits)                                          # initialiseDeviceMemory2D<T>
        assert (t.cpu().numpy() == want).all()
    # raw memory functions (what pcl::gpu::DeviceMemory2D sits on)
    n = C.c_int(); assert L.rgbid_device_count(C.byref(n)) == 0 and n.value >= 1
    assert L.rgbid_set_device(0) == 0 and L.rgbid_set_device(n.value + 7) != 0
    p, step = C.c_void_p(), C.c_size_t()
    assert L.rgbid_malloc_pitch(C.byref(p), C.byref(step), C.c_size_t(cols * 4), C.c_size_t(rows)) == 0 and step.value % 256 == 0 and step.value >= cols * 4
    host_in = np.ascontiguousarray(b); host_out = np.zeros_like(host_in)
    assert L.rgbid_memcpy2d_h2d(ctx._h, p, step, host_in.ctypes.data_as(C.c_void_p), C.c_size_t(cols * 4), C.c_size_t(cols * 4), C.c_size_t(rows)) == 0
    q = C.c_void_p(); assert L.rgbid_malloc(C.byref(q), C.c_size_t(step.value * rows)) == 0
    assert L.rgbid_memcpy2d_d2d(ctx._h, q, step, p, step, C.c_size_t(cols * 4), C.c_size_t(rows)) == 0
    assert L.rgbid_memcpy2d_d2h(ctx._h, host_out.ctypes.data_as(C.c_void_p), C.c_size_t(cols * 4), q, step, C.c_size_t(cols * 4), C.c_size_t(rows)) == 0
    assert np.array_equal(host_out, host_in)
    flat_in = np.arange(100, dtype=np.float32); flat_out = np.zeros_like(flat_in)
    assert L.rgbid_memcpy_h2d(ctx._h, p, flat_in.ctypes.data_as(C.c_void_p), C.c_size_t(400)) == 0
    assert L.rgbid_memcpy_d2d(ctx._h, q, p, C.c_size_t(400)) == 0
    assert L.rgbid_memcpy_d2h(ctx._h, flat_out.ctypes.data_as(C.c_void_p), q, C.c_size_t(400)) == 0 and np.array_equal(flat_in, flat_out)
    assert L.rgbid_free(p) == 0 and L.rgbid_free(q) == 0
    # a context can be moved onto the caller's stream
    s = torch.cuda.Stream()
    c2 = device.Context(0)
    assert L.rgbid_ctx_set_stream(c2._h, C.c_void_p(s.cuda_stream)) == 0 and c2.stream_handle() == s.cuda_stream
    g1, g2 = new(rows, cols), new(rows, cols); c2.computeGradient(dev(a), g1, g2); c2.sync()
    ogx, _ = O.gradient(a); assert_bits(g1.cpu().numpy(), ogx, 0, "gradient on an adopted stream")
    c2.close()
    # device-side record ring of the engine
    K = (131.25, 131.25, 79.5, 59.5)
    eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=2, K=K, use_graph=0, record_capacity=3))
    ptr, cap = C.c_void_p(), C.c_int()
    assert L.rgbid_engine_records_dev(eng._h, C.byref(ptr), C.byref(cap)) == 0 and ptr.value and cap.value == 3 and eng.steps() == 0
    eng.close()
