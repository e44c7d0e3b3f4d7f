"""Shared helpers for the parity tests: seeded inputs with NaN holes, bit-level comparison."""
import numpy as np

TUM_K = (525.0, 525.0, 319.5, 239.5)


def rng(seed):
    return np.random.default_rng(20260928 + seed)


def rand_invdepth(r, rows, cols, nan_frac=0.05, smooth=True):
    """iD ~ U(0.25, 1.25) m^-1 built from a smooth surface + noise, with NaN holes."""
    if smooth:
        v, u = np.mgrid[0:rows, 0:cols].astype(np.float64)
        z = 2.0 + 0.4 * np.sin(u / cols * 5.0) * np.cos(v / rows * 4.0) + 0.3 * (u / cols) + 0.01 * r.standard_normal((rows, cols))
        w = (1.0 / z).astype(np.float32)
    else:
        w = r.uniform(0.25, 1.25, (rows, cols)).astype(np.float32)
    holes = r.random((rows, cols)) < nan_frac
    w[holes] = np.nan
    return w


def rand_intensity(r, rows, cols, nan_frac=0.0):
    v, u = np.mgrid[0:rows, 0:cols].astype(np.float64)
    i = 127 + 60 * np.sin(u / 7.0) * np.cos(v / 5.0) + 40 * np.sin((u + 2 * v) / 13.0) + 8 * r.standard_normal((rows, cols))
    i = np.clip(i, 0, 255).astype(np.float32)
    if nan_frac > 0:
        i[r.random((rows, cols)) < nan_frac] = np.nan
    return i


def small_motion(r, K, trans=0.02, rot_deg=1.0):
    """KF-relative pose (R, t) of the current camera and the projected inverse (K R^-1 K^-1, K t^-1) in float32."""
    w = r.standard_normal(3); w = w / np.linalg.norm(w) * np.deg2rad(rot_deg)
    th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx
    t = r.standard_normal(3); t = t / np.linalg.norm(t) * trans
    return R, t


def project(K, R, t):
    """float32 K R K^-1 (row-major 9) and K t, computed like the host does (float arithmetic)."""
    fx, fy, cx, cy = [np.float32(v) for v in K]
    Km = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)
    Ki = np.array([[1 / fx, 0, -cx / fx], [0, 1 / fy, -cy / fy], [0, 0, 1]], np.float32)
    Rp = (Km @ R.astype(np.float32)) @ Ki
    tp = Km @ t.astype(np.float32)
    return Rp.astype(np.float32).reshape(9), tp.astype(np.float32)


def inv_pose(R, t):
    Ri = R.T
    return Ri, -Ri @ t


def ulp_diff(a, b):
    """max ULP distance over finite entries + count of NaN-pattern mismatches (float32)."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    nan_mismatch = int(np.count_nonzero(na != nb))
    m = ~(na | nb)
    if not m.any():
        return 0, nan_mismatch
    ia = a[m].view(np.int32).astype(np.int64); ib = b[m].view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia); ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return int(np.abs(ia - ib).max()), nan_mismatch


def assert_bits(a, b, max_ulp=0, what=""):
    u, n = ulp_diff(a, b)
    assert n == 0, f"{what}: NaN pattern differs at {n} pixels"
    assert u <= max_ulp, f"{what}: max ULP distance {u} > {max_ulp}"


def assert_rel(a, b, rtol, atol=0.0, what=""):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), f"{what}: NaN pattern differs at {np.count_nonzero(na != nb)} entries"
    m = ~na
    err = np.abs(a[m] - b[m]) - (atol + rtol * np.abs(b[m]))
    assert (err <= 0).all(), f"{what}: max excess error {err.max():.3e} (rtol={rtol}, atol={atol})"


def padded(t, pad_cols=5):
    """A [rows, cols] view into a wider allocation: exercises step != cols*elem_size."""
    import torch
    rows, cols = t.shape[:2]
    big = torch.zeros((rows, cols + pad_cols) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device)
    v = big[:, :cols]
    v.copy_(t)
    return v
