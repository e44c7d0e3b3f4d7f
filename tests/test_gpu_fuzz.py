"""Randomised edge-case sweep of the pixel-selecting kernels against the oracle (bit-exact): odd sizes, violent motions that put
points behind the camera or far outside the image (negative / infinite / NaN projections, saturating float->int conversions),
extreme inverse depths that leave the fast range of the exact reciprocal (zero, denormal, huge), dense NaN patterns."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import util
from tests.util import assert_bits

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _case(seed):
    r = util.rng(1000 + seed)
    rows, cols = int(r.integers(5, 90)), int(r.integers(5, 130))
    K = (float(r.uniform(20, 200)), float(r.uniform(20, 200)), float(r.uniform(0, cols)), float(r.uniform(0, rows)))
    grid = util.rand_invdepth(r, rows, cols, nan_frac=float(r.uniform(0, 0.5)), smooth=bool(r.integers(0, 2)))
    src = util.rand_invdepth(r, rows, cols, nan_frac=float(r.uniform(0, 0.5)), smooth=bool(r.integers(0, 2)))
    inten = util.rand_intensity(r, rows, cols, nan_frac=float(r.uniform(0, 0.1)))
    # sprinkle values that leave the verified range of the fast reciprocal or make the projection degenerate
    specials = np.array([0.0, -0.0, 1e-45, 1e-39, 1e-38, 3e38, 1e30, -1.0, np.inf, -np.inf, 1e-20, 5e-4], np.float32)
    for m in (grid, src):
        idx = r.integers(0, m.size, size=max(1, m.size // 40))
        m.reshape(-1)[idx] = specials[r.integers(0, specials.size, size=idx.size)]
    mode = seed % 3
    if mode == 0:      # gentle motion
        R, t = util.small_motion(r, K, float(r.uniform(0, 0.05)), float(r.uniform(0, 2)))
    elif mode == 1:    # violent: up to 170 degrees, metres of translation (points behind the camera, huge coordinates)
        R, t = util.small_motion(r, K, float(r.uniform(0, 3)), float(r.uniform(20, 170)))
    else:              # translation that puts X.z near / exactly at zero for part of the image
        R, t = util.small_motion(r, K, 0.0, float(r.uniform(0, 1)))
        t = np.array([0.0, 0.0, -float(1.0 / np.nanmedian(np.abs(grid[np.isfinite(grid)]) + 1e-6))])
    Rp, tp = util.project(K, *util.inv_pose(R, t))
    return rows, cols, grid, src, inten, Rp, tp


@pytest.mark.parametrize("seed", range(36))
def test_fuzz_warps_visibility(ctx, seed):
    rows, cols, grid, src, inten, Rp, tp = _case(seed)
    new = lambda: torch.full((rows, cols), float("nan"), device="cuda")
    W1 = new(); ctx.warpInvDepthWithTrafo3D(dev(src), W1, dev(grid), Rp, tp)
    oW1 = O.warp_invdepth(src, grid, Rp, tp)
    assert_bits(W1.cpu().numpy(), oW1, 0, "warp iD")
    for mode in (O.INTERP_TEX8, O.INTERP_EXACT):
        I1 = new(); ctx.set_interp_mode(mode)
        try:
            ctx.warpIntensityWithTrafo3DInvDepth(dev(inten), I1, dev(grid), Rp, tp)
        finally:
            ctx.set_interp_mode(O.INTERP_TEX8)
        assert_bits(I1.cpu().numpy(), O.warp_intensity(inten, grid, Rp, tp, mode), 0, "warp intensity")
    Ww, Wt = new(), torch.zeros((rows, cols), device="cuda")
    ctx.warpInvDepthWithTrafo3DWeighted(dev(src), Ww, dev(grid), Wt, Rp, tp)
    oWw, oWt = O.warp_invdepth_weighted(src, grid, Rp, tp)
    assert_bits(Ww.cpu().numpy(), oWw, 0, "weighted warp"); assert_bits(Wt.cpu().numpy(), oWt, 0, "warp weight")
    m = torch.zeros((rows, cols), dtype=torch.uint8, device="cuda")
    ratio = ctx.getVisibilityRatioWithOverlapMask(dev(grid), dev(src), Rp, tp, overlap_mask=m)
    oratio, nvis, nval, omask = O.visibility_ratio(grid, src, Rp, tp, with_mask=True)
    assert ratio == np.float32(oratio) and np.array_equal(m.cpu().numpy(), omask)
    vm = torch.zeros((3 * rows, cols), device="cuda"); ctx.createVMap((40.0, 45.0, cols / 2.0, rows / 2.0), dev(grid), vm)
    ov = O.vmap(grid, (40.0, 45.0, cols / 2.0, rows / 2.0))
    assert_bits(vm.cpu().numpy()[:rows], ov[:rows], 0, "vmap plane 0")
    ok = ~np.isnan(ov[:rows])
    assert np.array_equal(vm.cpu().numpy()[rows:2 * rows][ok], ov[rows:2 * rows][ok], equal_nan=True) and np.array_equal(vm.cpu().numpy()[2 * rows:][ok], ov[2 * rows:][ok], equal_nan=True)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_stencils_and_fusion(ctx, seed):
    """Sobel, pyrDown and the fusion update on odd sizes with infinities / zeros / denormals / dense NaN in the maps."""
    r = util.rng(2000 + seed)
    rows, cols = int(r.integers(5, 90)), int(r.integers(5, 130))
    a = util.rand_invdepth(r, rows, cols, nan_frac=float(r.uniform(0, 0.6)), smooth=bool(r.integers(0, 2)))
    specials = np.array([0.0, -0.0, 1e-45, 1e-39, 3e38, -3e38, np.inf, -np.inf, 1.0], np.float32)
    idx = r.integers(0, a.size, size=max(1, a.size // 30))
    a.reshape(-1)[idx] = specials[r.integers(0, specials.size, size=idx.size)]
    gx, gy = torch.empty((rows, cols), device="cuda"), torch.empty((rows, cols), device="cuda")
    ctx.computeGradient(dev(a), gx, gy)
    with np.errstate(all="ignore"):
        ogx, ogy = O.gradient(a)
    assert_bits(gx.cpu().numpy(), ogx, 0, "Sobel x"); assert_bits(gy.cpu().numpy(), ogy, 0, "Sobel y")
    if rows >= 8 and cols >= 8:
        p = torch.empty((rows // 2, cols // 2), device="cuda"); ctx.pyrDown(dev(a), p)
        with np.errstate(all="ignore"):
            op = O.pyr_down(a)
        got = p.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(op)) and np.array_equal(np.isinf(got), np.isinf(op))   # validity rule is integer work
        fin = np.isfinite(op)
        assert_bits(np.where(fin, got, 0), np.where(fin, op, 0), 4, "pyrDown")
    kf = util.rand_invdepth(r, rows, cols, 0.3); ws = (kf + r.normal(0, 0.01, kf.shape)).astype(np.float32)
    ws.reshape(-1)[idx] = specials[r.integers(0, specials.size, size=idx.size)]
    q = r.uniform(0.5, 4, kf.shape).astype(np.float32); qs = r.uniform(0, 2, kf.shape).astype(np.float32); qs.reshape(-1)[idx[: idx.size // 2]] = 0
    kfd, qd = dev(kf.copy()), dev(q.copy())
    ctx.integrateWarpedFrame(dev(ws), dev(qs), kfd, qd)
    with np.errstate(all="ignore"):
        ok_, oq = O.integrate_warped(ws, qs, kf, q)
    assert_bits(kfd.cpu().numpy(), ok_, 0, "fused iD"); assert_bits(qd.cpu().numpy(), oq, 0, "fused weight")
