"""custom-calibration front-end (undistortIntensity / undistortDepthInv / registerDepthinv and
prepareImagesCustomCalibration) -- HIP kernels through the C-ABI against the CPU oracle, bit for bit (the registration splat is
an order-independent atomicMax z-buffer), and the C++ VisodoTracker with a custom-calibration file against the oracle tracker."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from rgbid import device, host, synth
from tests import util
from tests.util import assert_bits

pytestmark = pytest.mark.gpu

SIZES = [(480, 640), (61, 83), (120, 160)]
KD = (0.12, -0.25, 0.0015, -0.0008, 0.09)          # k1 k2 k3(p1) k4(p2) k5


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def intr_k(rows, cols, kd=KD, f=1.0):
    s = cols / 640.0
    return (525.0 * s * f, 527.0 * s * f, 319.5 * s + 1.3, 239.5 * s * rows / (480.0 * s) - 0.7) + tuple(kd)


def dist(xshift=4, yshift=4):
    q0 = (0.002, -0.004, 0.003, -0.001, 0.0007, -0.0005, 0.0011, -0.0009, 0.0004)
    q1 = (0.01, 0.02, -0.015, 0.004, -0.003, 0.002, 0.006, -0.005, 0.001)
    return dict(c1=1.03, c0=-0.004, q0=q0, q1=q1, xshift=xshift, yshift=yshift)


@pytest.mark.parametrize("rows,cols", SIZES)
@pytest.mark.parametrize("mode", [O.INTERP_EXACT, O.INTERP_TEX8])
def test_undistort_intensity(ctx, rows, cols, mode):
    r = util.rng(31)
    src = util.rand_intensity(r, rows, cols)
    k = intr_k(rows, cols)
    dst = torch.empty((rows, cols), device="cuda")
    ctx.set_interp_mode(mode)
    try:
        ctx.undistortIntensity(dev(src), dst, k)
    finally:
        ctx.set_interp_mode(O.INTERP_TEX8)
    ref = O.undistort_intensity(src, k, mode)
    assert_bits(dst.cpu().numpy(), ref, 0, "undistort intensity")
    assert np.isnan(ref).any() and np.isfinite(ref).mean() > 0.8     # barrel distortion pushes some border pixels outside


def test_undistort_zero_distortion_is_identity(ctx):
    """KAT: k = 0 -> the distorted position is the pixel itself (+0.5): bilinear at texel centres returns the texel."""
    r = util.rng(32)
    rows, cols = 96, 128
    src = util.rand_intensity(r, rows, cols)
    k = (525.0 * 0.2, 525.0 * 0.2, 63.5, 47.5, 0, 0, 0, 0, 0)
    dst = torch.empty((rows, cols), device="cuda")
    ctx.undistortIntensity(dev(src), dst, k)
    out = dst.cpu().numpy()
    ok = np.isfinite(out)
    assert ok[1:-1, 1:-1].all()
    np.testing.assert_allclose(out[ok], src[ok], atol=2e-3)          # (x - cx)/fx*fx + cx reproduces x to ~1e-5 px
    np.testing.assert_array_equal(O.undistort_intensity(src, k), out)


@pytest.mark.parametrize("rows,cols", SIZES)
@pytest.mark.parametrize("shift", [(4, 4), (0, 2)])
def test_undistort_depthinv(ctx, rows, cols, shift):
    r = util.rng(33)
    src = util.rand_invdepth(r, rows, cols)
    k = intr_k(rows, cols, f=1.1)
    dd = dist(*shift)
    corr, out = torch.empty((rows, cols), device="cuda"), torch.empty((rows, cols), device="cuda")
    ctx.undistortDepthInv(dev(src), corr, out, k, device.depth_dist(**dd))
    rc, ro = O.undistort_depthinv(src, k, O.depth_dist(**dd))
    assert_bits(corr.cpu().numpy(), rc, 0, "depth-distortion correction")
    assert_bits(out.cpu().numpy(), ro, 0, "undistort iD")
    assert np.isnan(rc[: shift[1] + 1]).all() and np.isnan(rc[:, : shift[0] + 1]).all()   # x - xshift > 0 is strict


@pytest.mark.parametrize("rows,cols", SIZES)
def test_register_depthinv(ctx, rows, cols):
    r = util.rng(34)
    src = util.rand_invdepth(r, rows, cols)
    s = cols / 640.0
    Kc = np.array([[525.0 * s, 0, 319.5 * s], [0, 525.0 * s, 239.5 * s * rows / (480.0 * s)], [0, 0, 1]], np.float32)
    Kd = np.array([[580.0 * s, 0, 315.0 * s], [0, 582.0 * s, 236.0 * s * rows / (480.0 * s)], [0, 0, 1]], np.float32)
    R, _ = util.small_motion(r, util.TUM_K, 0.0, 1.2)
    t_dc = np.array([0.025, -0.003, 0.004], np.float32)
    dRc_proj = ((Kd @ R.astype(np.float32)) @ np.linalg.inv(Kc).astype(np.float32)).astype(np.float32)
    cRd_proj = np.linalg.inv(dRc_proj.astype(np.float64)).astype(np.float32)
    t_proj = (Kd @ t_dc).astype(np.float32)
    inter = torch.empty((3 * rows, 3 * cols), device="cuda"); inter_i = torch.empty((3 * rows, 3 * cols), dtype=torch.int32, device="cuda")
    dst = torch.empty((rows, cols), device="cuda")
    ctx.registerDepthinv(dev(src), inter, inter_i, dst, dRc_proj.reshape(9), t_proj, cRd_proj.reshape(9))
    ri, ro = O.register_depthinv(src, dRc_proj, t_proj, cRd_proj)
    assert_bits(inter.cpu().numpy(), ri, 0, "translation splat (z-buffer)")
    assert_bits(dst.cpu().numpy(), ro, 0, "registered iD")
    assert np.isfinite(ro).mean() > 0.5


def test_register_identity_keeps_the_map(ctx):
    """KAT: identity extrinsics and equal intrinsics -> every valid pixel lands on itself with dilation 1 (3x3 footprint of
    neighbours can only raise a pixel to a nearer surface), so NaN-free smooth input is reproduced up to the z-buffer max."""
    rows, cols = 60, 80
    w = np.full((rows, cols), 0.5, np.float32)
    eye = np.eye(3, dtype=np.float32)
    inter = torch.empty((3 * rows, 3 * cols), device="cuda"); inter_i = torch.empty((3 * rows, 3 * cols), dtype=torch.int32, device="cuda")
    dst = torch.empty((rows, cols), device="cuda")
    ctx.registerDepthinv(dev(w), inter, inter_i, dst, eye.reshape(9), np.zeros(3, np.float32), eye.reshape(9))
    np.testing.assert_array_equal(dst.cpu().numpy(), w)


CALIB_INI = """[RGB_CALIBRATION]
fx = 525.0
fy = 525.0
cx = 319.5
cy = 239.5
kd = 0.02 -0.04 0.0005 -0.0004 0.01
[DEPTH_CALIBRATION]
custom_registration = 1
fx = 571.0
fy = 572.5
cx = 316.0
cy = 241.5
kd = -0.015 0.03 0.0003 0.0002 -0.008
c0 = -0.002
c1 = 1.01
q0 = 0.001 -0.002 0.001 0.0 0.0005 -0.0004 0.0 0.0 0.0
q1 = 0.005 0.01 0.0 0.0 -0.002 0.001 0.0 0.0 0.0
[STEREO_DEPTH2RGB]
dRc = 0.99995 -0.008 0.006 0.00803 0.99995 -0.005 -0.00596 0.00505 0.99997
t_dc = 0.0251 -0.0012 0.0031
"""


def test_cpp_tracker_custom_calibration(tmp_path):
    """prepareImagesCustomCalibration end to end: the calibration file switches the C++ tracker to the undistort + register
    front-end; maps are bit-identical to the oracle's, poses within the north-star tolerance."""
    n = 4
    seq = synth.make_sequence(n, device="cuda")
    d = seq["depth"].cpu().numpy().astype(np.uint16); c = seq["rgb"].cpu().numpy()
    (tmp_path / "calib.ini").write_text(CALIB_INI)
    trk = host.Tracker(host.default_config())
    trk.load_calibration(str(tmp_path / "calib.ini"))
    orc = O.Tracker(O.default_config())
    dRc = [0.99995, -0.008, 0.006, 0.00803, 0.99995, -0.005, -0.00596, 0.00505, 0.99997]
    orc.set_custom_calibration((525.0, 525.0, 319.5, 239.5, 0.02, -0.04, 0.0005, -0.0004, 0.01),
                               (571.0, 572.5, 316.0, 241.5, -0.015, 0.03, 0.0003, 0.0002, -0.008),
                               O.depth_dist(c1=1.01, c0=-0.002, q0=(0.001, -0.002, 0.001, 0.0, 0.0005, -0.0004, 0.0, 0.0, 0.0),
                                            q1=(0.005, 0.01, 0.0, 0.0, -0.002, 0.001, 0.0, 0.0, 0.0)), dRc, (0.0251, -0.0012, 0.0031))
    for k in range(n):
        assert trk.track(d[k], c[k]) == orc.track(d[k], c[k])
        iD, I = trk.current_maps()
        assert_bits(iD, orc.cur_depthinv(), 0, f"registered iD, frame {k}")
        assert_bits(I, orc.cur_intensity(), 0, f"undistorted intensity, frame {k}")
    Ra, ta = trk.poses(); Rb, tb = orc.poses()
    for k in range(n):
        ang = float(np.arccos(np.clip((np.trace(Ra[k].T @ Rb[k]) - 1) / 2, -1, 1)))
        assert ang < 1e-4 and np.linalg.norm(ta[k] - tb[k]) < 1e-4, (k, ang, np.linalg.norm(ta[k] - tb[k]))
    assert np.isfinite(orc.cur_depthinv()).mean() > 0.6
