"""A second, independent restatement of a handful of the reference's kernels in vectorised numpy (float64 arithmetic on float32 data),
written from the reference sources -- not from oracle/ -- so that tests/test_oracle_vs_numpy_mirror.py can cross-check the C oracle
against it (to rounding, not bit for bit: the two evaluate in different precision and order).  TEST INFRASTRUCTURE ONLY.

Each function cites the reference lines it follows."""
import numpy as np


def _shift(a, dy, dx, fill=np.nan):
    """a[y+dy, x+dx] with `fill` outside the image"""
    out = np.full(a.shape, fill, a.dtype)
    h, w = a.shape
    ys, yd = (slice(dy, h), slice(0, h - dy)) if dy >= 0 else (slice(0, h + dy), slice(-dy, h))
    xs, xd = (slice(dx, w), slice(0, w - dx)) if dx >= 0 else (slice(0, w + dx), slice(-dx, w))
    out[yd, xd] = a[ys, xs]
    return out


def pyr_down(src):
    """pyrDownKernelGridStridef, src/cuda/pyrdown.cu:84-132 (radius 2, sigma 1 :76-80): Gaussian-weighted mean of the valid samples of the
    5x5 window around (2x, 2y) clipped to the image; NaN unless more than 12 samples are valid."""
    src = src.astype(np.float64)
    rows, cols = src.shape[0] // 2, src.shape[1] // 2
    s1 = np.zeros((rows, cols)); s2 = np.zeros((rows, cols)); cnt = np.zeros((rows, cols), int)
    big = np.pad(src, 2, constant_values=np.nan)                       # out-of-image taps: never valid
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            tap = big[2 + dy:2 + dy + 2 * rows:2, 2 + dx:2 + dx + 2 * cols:2]
            ok = ~np.isnan(tap)
            w = np.exp(-(dx * dx + dy * dy) * 0.5)
            s1 += np.where(ok, tap, 0.0) * w; s2 += ok * w; cnt += ok
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(cnt > 12, s1 / s2, np.nan)


def sobel(src):
    """gradientKernel, src/cuda/misc.cu:176-220: 3x3 Sobel with replicated borders, divided by 8; NaN taps propagate."""
    p = np.pad(src.astype(np.float64), 1, mode="edge")
    h, w = src.shape
    gx = np.zeros((h, w)); gy = np.zeros((h, w))
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            t = p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
            gx = gx + t * (dx * (2 - dy * dy)); gy = gy + t * (dy * (2 - dx * dx))
    return gx / 8.0, gy / 8.0


def bilateral(src, sigma_value, sigma_space=5.0):
    """bilateralKernel, src/cuda/filters.cu:86-135 (RADIUS 2, sigma_space 5 :60-70) with the window clipped to the image (the reference's
    unsigned index arithmetic is documented as a deviation in DESIGN.md): NaN centre -> NaN, NaN taps skipped."""
    a = src.astype(np.float64)
    s1 = np.zeros(a.shape); s2 = np.zeros(a.shape)
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            tap = _shift(a, dy, dx)
            ok = ~np.isnan(tap)
            fn = (a - tap) / sigma_value
            w = np.exp(-((dx * dx + dy * dy) * (0.5 / (sigma_space * sigma_space)) + 0.5 * fn * fn))
            s1 += np.where(ok, tap * w, 0.0); s2 += np.where(ok, w, 0.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(np.isnan(a), np.nan, s1 / s2)


def depth_to_invdepth(depth_u16, factor_depth=1.0):
    """depth2invDepthKernel, src/cuda/misc.cu:104-124: (1 / factor_depth) * 1000 / min(value, 10000) for value > 0 (millimetres), else NaN."""
    v = np.minimum(depth_u16.astype(np.float64), 10000.0)
    with np.errstate(divide="ignore"):
        return np.where(depth_u16 > 0, (1.0 / factor_depth) * 1000.0 / v, np.nan)


def intensity(rgb_u8):
    """intensityKernel, src/cuda/misc.cu:128-148: Rec. 709 luma 0.2126 R + 0.7152 G + 0.0722 B clamped to [0, 255]."""
    r, g, b = [rgb_u8[..., k].astype(np.float64) for k in range(3)]
    return np.clip(0.2126 * r + 0.7152 * g + 0.0722 * b, 0.0, 255.0)


def register_pixel(x, y, w, Rp, tp):
    """registerPixel, src/cuda/warping_registration.cu:129-146: X = R_proj (x, y, 1) / w + t_proj; returns (X.x/X.z, X.y/X.z, 1/X.z)."""
    z = 1.0 / w
    X = Rp[:, 0, None, None] * (x * z) + Rp[:, 1, None, None] * (y * z) + Rp[:, 2, None, None] * z + tp[:, None, None]
    with np.errstate(invalid="ignore", divide="ignore"):
        return X[0] / X[2], X[1] / X[2], 1.0 / X[2]


def warp_invdepth(src, grid, Rp, tp):
    """trafo3DKernelInvDepthGridStride, src/cuda/warping_registration.cu:505-546: project every valid keyframe pixel (inverse depth `grid`)
    into `src`, point-sample at floor(coordinate + 0.5), rescale the sampled inverse depth into the keyframe's frame; NaN unless > 0."""
    h, w_ = grid.shape
    y, x = np.mgrid[0:h, 0:w_].astype(np.float64)
    g = grid.astype(np.float64)
    Rp = np.asarray(Rp, np.float64).reshape(3, 3); tp = np.asarray(tp, np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        xs, ys, w3 = register_pixel(x, y, g, Rp, tp)
        ix = np.floor(xs + 0.5); iy = np.floor(ys + 0.5)
        inb = (ix >= 0) & (iy >= 0) & (ix < w_) & (iy < h) & ~np.isnan(g)
        w2 = src.astype(np.float64)[np.clip(iy, 0, h - 1).astype(int) * inb, np.clip(ix, 0, w_ - 1).astype(int) * inb]
        v = (1.0 / w3 - tp[2]) * g
        res = v / (1.0 - w2 * tp[2]) * w2
        return np.where(inb & (res > 0), res, np.nan)


def bilinear(src, xs, ys, tex8):
    """tex2D<float> with cudaFilterModeLinear / clamp addressing at unnormalised coordinates (CUDA programming guide, texture fetching):
    xB = x - 0.5, weights from frac(xB) -- stored in 1.8 fixed point by the hardware when tex8 -- taps clamped to the image."""
    h, w_ = src.shape
    xb, yb = xs - 0.5, ys - 0.5
    x0, y0 = np.floor(xb), np.floor(yb)
    a, b = xb - x0, yb - y0
    if tex8:
        a, b = np.rint(a * 256.0) / 256.0, np.rint(b * 256.0) / 256.0
    cl = lambda v, n: np.clip(np.nan_to_num(v, nan=0.0, posinf=1e9, neginf=-1e9), 0, n - 1).astype(int)
    s = src.astype(np.float64)
    t00, t10 = s[cl(y0, h), cl(x0, w_)], s[cl(y0, h), cl(x0 + 1, w_)]
    t01, t11 = s[cl(y0 + 1, h), cl(x0, w_)], s[cl(y0 + 1, h), cl(x0 + 1, w_)]
    return (1 - a) * (1 - b) * t00 + a * (1 - b) * t10 + (1 - a) * b * t01 + a * b * t11


def warp_intensity(src, grid, Rp, tp, tex8):
    """trafo3DKernelIntensityWithInvDepthGridStride, src/cuda/warping_registration.cu:465-501: same projection, bilinear texture fetch at
    coordinate + 0.5, clamped to [0, 255]; NaN where the grid is invalid or the point-sample position leaves the image."""
    h, w_ = grid.shape
    y, x = np.mgrid[0:h, 0:w_].astype(np.float64)
    g = grid.astype(np.float64)
    Rp = np.asarray(Rp, np.float64).reshape(3, 3); tp = np.asarray(tp, np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        xs, ys, _ = register_pixel(x, y, g, Rp, tp)
        xs, ys = xs + 0.5, ys + 0.5
        ix, iy = np.floor(xs), np.floor(ys)
        inb = (ix >= 0) & (iy >= 0) & (ix < w_) & (iy < h) & ~np.isnan(g)
        val = np.clip(bilinear(src, np.where(inb, xs, 0.5), np.where(inb, ys, 0.5), tex8), 0.0, 255.0)
        return np.where(inb, val, np.nan)


def visibility_ratio(src, dst, Rp, tp):
    """partialVisibilityKernel + getVisibilityRatio, src/cuda/warping_registration.cu:297-461, 825-869: of the valid pixels of `src`, the
    fraction that projects strictly inside `dst` and agrees with the inverse depth found there (round-to-nearest pixel) to 0.020."""
    h, w_ = src.shape
    y, x = np.mgrid[0:h, 0:w_].astype(np.float64)
    s = src.astype(np.float64)
    Rp = np.asarray(Rp, np.float64).reshape(3, 3); tp = np.asarray(tp, np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        xd, yd, wd = register_pixel(x, y, s, Rp, tp)
        valid = ~np.isnan(s)
        inside = valid & (xd > 0) & (xd < w_ - 1) & (yd > 0) & (yd < h - 1)
        xi = np.where(inside, np.rint(xd), 0).astype(int); yi = np.where(inside, np.rint(yd), 0).astype(int)
        vis = inside & (np.abs(wd - dst.astype(np.float64)[yi, xi]) < 0.020)
    nval = int(valid.sum())
    return (float(vis.sum()) / nval if nval >= 1 else 0.0), vis, valid


def integrate(warped, wweight, kf, kfw):
    """integrateWarpedFrameKernel, src/cuda/warping_registration.cu:637-669: weighted running mean of the keyframe inverse depth, gated at
    3 * 0.0075; an empty keyframe pixel adopts the warped sample."""
    kf = kf.astype(np.float64).copy(); kfw = kfw.astype(np.float64).copy()
    ws, qs = warped.astype(np.float64), wweight.astype(np.float64)
    have = ~np.isnan(ws)
    adopt = have & np.isnan(kf)
    with np.errstate(invalid="ignore"):
        fuse = have & ~np.isnan(kf) & (np.abs(ws - kf) < 3 * 0.0075)
    nw = kfw + qs
    fused = (kf * kfw + ws * qs) / np.where(fuse, nw, 1.0)
    out_kf = np.where(adopt, ws, np.where(fuse, fused, kf))
    out_w = np.where(adopt, qs, np.where(fuse, nw, kfw))
    return out_kf, out_w


def build_system(W0, I0, gWx, gWy, gIx, gIy, W1, I1, K, sigma_d, sigma_i, bias_d=0.0, bias_i=0.0, nu_d=5.0, nu_i=5.0, student_nu=True,
                 mestimator=3, weighting=0):
    """constraintsHandler::{invDepthConstraint, intensityConstraint, computeWeight, computeWeightStudent, computeStudentNuSystemGridStride},
    src/cuda/estimate_VO.cu:141-262, 354-418 and the host-side unpacking :774-786: the 6x6 normal equations A x = b of one Gauss-Newton
    iteration over all pixels.  Rows are [translation | rotation]; weighting: 0 independent, 1 min-weight, 2 geometric only, 3 photometric only;
    mestimator (fixed-nu path): 0 least squares, 1 Huber, 2 Tukey, 3 Student-t with 5 dof."""
    fx, fy, cx, cy = [float(v) for v in K]
    f = lambda a: np.asarray(a, np.float64)
    W0, I0, gWx, gWy, gIx, gIy, W1, I1 = map(f, (W0, I0, gWx, gWy, gIx, gIy, W1, I1))
    h, w_ = W0.shape
    v, u = np.mgrid[0:h, 0:w_].astype(np.float64)
    p = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)])

    def weight(e, nu):
        if student_nu:
            return (nu + 1.0) / (nu + e * e)
        if mestimator == 1:
            return np.where(np.abs(e) > 1.345, 1.345 / np.abs(e), 1.0)
        if mestimator == 2:
            return np.where(np.abs(e) < 4.685, (1.0 - (e / 4.685) ** 2) ** 2, 0.0)
        if mestimator == 3:
            return 6.0 / (5.0 + e * e)
        return np.ones_like(e)

    with np.errstate(invalid="ignore", divide="ignore"):
        # inverse-depth constraint
        g = np.stack([gWx * fx, gWy * fy, np.zeros_like(gWx)]); g[2] = -(g[0] * p[0] + g[1] * p[1])
        n = g / W0; n[2] = n[2] + 1.0
        n = n / np.sqrt((n * n).sum(0)); pu = p / np.sqrt((p * p).sum(0))
        nfac = np.abs((n * pu).sum(0))
        row_t = g * W0; row_t[2] = row_t[2] + W0 * W1
        g2 = g.copy(); g2[2] = g2[2] + W1
        row_r = -np.cross(g2, p, axis=0)
        Jd = np.concatenate([row_t, row_r]) / sigma_d
        ed = -(W1 - W0) / sigma_d
        vd = ~(np.isnan(W0) | np.isnan(W1) | np.isnan(gWx) | np.isnan(gWy))
        wd = np.where(vd, weight(ed - bias_d / sigma_d, nu_d), 0.0) * (0.0 if weighting == 3 else 1.0)
        # intensity constraint
        hI = np.stack([gIx * fx, gIy * fy, np.zeros_like(gIx)]); hI[2] = -(hI[0] * p[0] + hI[1] * p[1])
        Ji = np.concatenate([hI * W0, -np.cross(hI, p, axis=0)]) / sigma_i
        ei = -(I1 - I0) / sigma_i
        vi = ~(np.isnan(W0) | np.isnan(I0) | np.isnan(I1) | np.isnan(gIx) | np.isnan(gIy))
        wi = np.where(vi, weight(ei - bias_i / sigma_i, nu_i), 0.0) * (0.0 if weighting == 2 else 1.0)
        if weighting == 1:
            wi = np.minimum(wd, wi)
    z = lambda a, ok: np.where(ok, a, 0.0)
    Jd, ed, nfac = z(Jd, vd), z(ed, vd), z(nfac, vd)
    Ji, ei = z(Ji, vi), z(ei, vi)
    A = np.einsum("ihw,jhw,hw->ij", Ji, Ji, wi) + np.einsum("ihw,jhw,hw->ij", Jd, Jd, nfac * wd)
    b = np.einsum("ihw,hw,hw->i", Ji, ei, wi) + np.einsum("ihw,hw,hw->i", Jd, ed, nfac * wd)
    return A, b


def sigma_nu_student(err, bias, sigma, mestimator=3):
    """computeSigmaAndNuStudent, src/cuda/sigmaFuncs.cu:858-1066 with partialBiasAndSigmaStudent :281-332, finalReductionBiasAndSigma :361-407,
    partialFuncWeightsNu :410-468 and its final reduction :471-512.  Returns (bias, sigma, nu).
    IRLS: first pass least squares, then Student weights with nu = 5, at most 10 passes, stop when sigma changes by < 10 %;
    nu: bisection of C(nu) = -psi(nu/2) + ln(nu/2) + mean(ln w - w) + 1 + psi((nu+1)/2) - ln((nu+1)/2) on [2, 10], at most 5 steps,
    stopping when the bracket is below 1; returns the last midpoint."""
    from scipy.special import digamma
    e = np.asarray(err, np.float64)
    e = e[np.isfinite(e)]
    sh_bias, sh_sigma, nu_irls, mest = float(bias), float(sigma), 5.0, 0
    for i in range(10):
        if mest == 0:
            w = np.ones_like(e)
        else:
            en = (e - sh_bias) / sh_sigma
            w = (nu_irls + 1.0) / (nu_irls + en * en)
        swr, swsr, sw, n = (e * w).sum(), (e * e * w).sum(), w.sum(), e.size
        b = swr / sw
        s = np.sqrt((swsr - 2.0 * b * swr + b * b * sw) / n)
        prev = sh_sigma
        sh_bias, sh_sigma, mest = b, s, mestimator
        if i > 0 and abs(s - prev) / prev < 0.1:
            break

    return sh_bias, sh_sigma, _nu_bisection(e, sh_bias, sh_sigma)


def _nu_bisection(e, bias, sigma):
    """the second half of computeSigmaAndNuStudent (:934-1039) == computeNuStudent (:1100-1205): e holds the finite residuals"""
    from scipy.special import digamma
    sh_bias, sh_sigma = bias, sigma

    def C(nu):
        en = (e - sh_bias) / sh_sigma
        w = (nu + 1.0) / (nu + en * en)
        fw = (np.log(w).sum() - w.sum()) / e.size
        return -digamma(nu / 2) + np.log(nu / 2) + fw + 1.0 + digamma((nu + 1) / 2) - np.log((nu + 1) / 2)

    up, down = 10.0, 2.0
    c_down, c_up = C(down), C(up)
    if c_up * c_down > 0:
        nu = down if c_down <= 0 else up
    else:
        new = None
        for _ in range(5):
            new = (up + down) / 2
            if up - down < 1.0:
                break
            c_new = C(new)
            if c_new * c_up > 0:
                c_up, up = c_new, new
            else:
                c_down, down = c_new, new
        nu = new
    return nu


def nu_student(err, bias, sigma):
    """computeNuStudent, src/cuda/sigmaFuncs.cu:1068-1222: degrees of freedom only, for a given bias and scale"""
    e = np.asarray(err, np.float64)
    return _nu_bisection(e[np.isfinite(e)], float(bias), float(sigma))


def keyframe_align(iD_ini, grey_ini, iD_end, grey_end, K, R0=None, t0=None):
    """KeyframeAlign::alignKeyframes, src/keyframe_align.cpp:115-357: dense alignment of two keyframes for the loop closer.  Four pyramid
    levels with {5, 5, 3, 0} iterations (:43), fixed scales (0.0025, 5) with nu from computeNuStudent on a 19 200-sample lattice, the
    intensity warp sampled with the KEYFRAME's inverse depth (:233-236), and nu_depthinv passed for both channels (:268).  Returns (R, t, cov)."""
    iters = (5, 5, 3, 0)
    a_w, a_i = [np.asarray(iD_ini, np.float64)], [np.asarray(grey_ini, np.float64)]
    b_w, b_i = [np.asarray(iD_end, np.float64)], [np.asarray(grey_end, np.float64)]
    for _ in range(1, 4):
        a_w.append(pyr_down(a_w[-1])); b_w.append(pyr_down(b_w[-1])); a_i.append(pyr_down(a_i[-1])); b_i.append(pyr_down(b_i[-1]))
    R = np.eye(3) if R0 is None else np.asarray(R0, np.float64); t = np.zeros(3) if t0 is None else np.asarray(t0, np.float64)
    A = np.eye(6)
    for l in range(3, -1, -1):
        Kl = tuple(float(v) / (1 << l) for v in K)
        Km = np.array([[Kl[0], 0, Kl[2]], [0, Kl[1], Kl[3]], [0, 0, 1.0]]); Ki = np.linalg.inv(Km)
        gwx, gwy = sobel(a_w[l]); gix, giy = sobel(a_i[l])
        for _ in range(iters[l]):
            Ri = np.linalg.inv(R)
            Rp, tp = Km @ Ri @ Ki, Km @ (-Ri @ t)
            W1 = warp_invdepth(b_w[l], a_w[l], Rp, tp)
            I1 = warp_intensity(b_i[l], a_w[l], Rp, tp, tex8=True)
            nu_d = nu_student(lattice(W1, a_w[l], 19200), 0.0, 0.0025)
            A, b = build_system(a_w[l], a_i[l], gwx, gwy, gix, giy, W1, I1, Kl, 0.0025, 5.0, 0.0, 0.0, nu_d, nu_d)
            x = np.linalg.solve(A, b)
            Rinc = np.linalg.inv(exp_map_rot(x[3:]))
            t = Rinc @ t - Rinc @ x[:3]; R = Rinc @ R
    return R, t, np.linalg.inv(A)


def vmap(depthinv, K):
    """computeVmapKernel, src/cuda/maps.cu:63-90: back-projection; planes (x, y, z) stacked vertically; only plane 0 is set to NaN when invalid."""
    fx, fy, cx, cy = [float(v) for v in K]
    w = depthinv.astype(np.float64)
    h, w_ = w.shape
    v, u = np.mgrid[0:h, 0:w_].astype(np.float64)
    with np.errstate(divide="ignore"):
        z = 1.0 / w
    return np.stack([z * (u - cx) / fx, z * (v - cy) / fy, z])


def nmap_gradients(depthinv, gx, gy, K):
    """computeNmapGradientsKernel, src/cuda/maps.cu:134-179: normal from the inverse-depth gradients, kept when it faces the viewing ray
    (cosine > 0.1)."""
    fx, fy, cx, cy = [float(v) for v in K]
    w, gx, gy = [a.astype(np.float64) for a in (depthinv, gx, gy)]
    h, w_ = w.shape
    v, u = np.mgrid[0:h, 0:w_].astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        n = np.stack([gx * fx, gy * fy, gx * (cx - u) + gy * (cy - v) + w])
        n = n / np.sqrt((n * n).sum(0))
        z = 1.0 / w
        vt = np.stack([z * (u - cx) / fx, z * (v - cy) / fy, z])
        vt = vt / np.sqrt((vt * vt).sum(0))
        keep = (vt * n).sum(0) > 0.1
    return np.where(keep, n, np.nan)


def warp_invdepth_weighted(src, grid, Rp, tp):
    """trafo3DKernelInvDepthWeightedGridStride, src/cuda/warping_registration.cu:549-594: the inverse-depth warp plus the propagated weight
    (1 - w2 tz)^4 / v^2; returns (warped, weight) with NaN where the kernel writes nothing."""
    h, w_ = grid.shape
    y, x = np.mgrid[0:h, 0:w_].astype(np.float64)
    g = grid.astype(np.float64)
    Rp = np.asarray(Rp, np.float64).reshape(3, 3); tp = np.asarray(tp, np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        xs, ys, w3 = register_pixel(x, y, g, Rp, tp)
        ix = np.floor(xs + 0.5); iy = np.floor(ys + 0.5)
        inb = (ix >= 0) & (iy >= 0) & (ix < w_) & (iy < h) & ~np.isnan(g)
        w2 = src.astype(np.float64)[np.clip(iy, 0, h - 1).astype(int) * inb, np.clip(ix, 0, w_ - 1).astype(int) * inb]
        v = (1.0 / w3 - tp[2]) * g
        wf = 1.0 - w2 * tp[2]
        wt = wf ** 4 / (v * v)
        res = v / wf * w2
        return np.where(inb & (res > 0), res, np.nan), np.where(inb & (wt > 0), wt, np.nan)


def lattice(im1, im0, min_nsamples=10000):
    """computeErrorGridStride, src/cuda/sigmaFuncs.cu:109-131, 701-765: halve the sampling lattice while both sizes stay even and it keeps at
    least min_nsamples points; residual im1 - im0 at (stride y, stride x)."""
    rows, cols = im0.shape
    r, c = rows, cols
    if min_nsamples < rows * cols:
        while True:
            c2, r2 = c // 2, r // 2
            if 2 * c2 != c or 2 * r2 != r or min_nsamples > c2 * r2:
                break
            c, r = c2, r2
    stride = int(round(np.sqrt(rows * cols / (r * c))))
    return (im1.astype(np.float64)[::stride, ::stride][:r, :c] - im0.astype(np.float64)[::stride, ::stride][:r, :c]).reshape(-1)


def exp_map_rot(w):
    """expMapRot, src/util_funcs.cpp:125-147 (Rodrigues; the forced re-orthogonalisation is a no-op to rounding)"""
    w = np.asarray(w, np.float64)
    th = np.linalg.norm(w)
    Om = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-5:
        return np.eye(3) + Om + 0.5 * Om @ Om
    return np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th ** 2 * Om @ Om


def align_pair(depth0, rgb0, depth1, rgb1, K, iters=(10, 5, 3), min_nsamples=10000, warp_first=False):
    """VisodoTracker::estimateVisualOdometry, src/visodo.cpp:1041-1263 for the shipped configuration (PYR_FIRST, Student-t, sigma from the
    residual pdf, independent weights, no filtering, start at the identity): coarse-to-fine Gauss-Newton of the pose of the current frame
    (1) relative to the keyframe (0).  Built only from the mirror's own kernels.  Returns (R, t)."""
    levels = len(iters)
    kf_w, kf_i = [depth_to_invdepth(depth0)], [intensity(rgb0)]
    cu_w, cu_i = [depth_to_invdepth(depth1)], [intensity(rgb1)]
    for l in range(1, levels):
        kf_w.append(pyr_down(kf_w[-1])); kf_i.append(pyr_down(kf_i[-1]))
        cu_w.append(pyr_down(cu_w[-1])); cu_i.append(pyr_down(cu_i[-1]))
    R, t = np.eye(3), np.zeros(3)
    for l in range(levels - 1, -1, -1):
        Kl = tuple(v / (1 << l) for v in K)                                            # getCalibMatrix(level), :1886-1900
        Km = np.array([[Kl[0], 0, Kl[2]], [0, Kl[1], Kl[3]], [0, 0, 1.0]])
        gwx, gwy = sobel(kf_w[l]); gix, giy = sobel(kf_i[l])
        for _ in range(iters[l]):
            Ri = np.linalg.inv(R); ti = -Ri @ t
            if warp_first:                                                             # WARP_FIRST :1078-1105: warp at level 0, pyrDown the warped maps
                K0 = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1.0]])
                Rp, tp = K0 @ Ri @ np.linalg.inv(K0), K0 @ ti
                W1 = warp_invdepth(cu_w[0], kf_w[0], Rp, tp)
                I1 = warp_intensity(cu_i[0], W1, Rp, tp, tex8=True)
                for _ in range(l):
                    I1, W1 = pyr_down(I1), pyr_down(W1)
            else:
                Rp, tp = Km @ Ri @ np.linalg.inv(Km), Km @ ti                          # :1108-1114
                W1 = warp_invdepth(cu_w[l], kf_w[l], Rp, tp)
                I1 = warp_intensity(cu_i[l], W1, Rp, tp, tex8=True)
            bi, si, nui = sigma_nu_student(lattice(I1, kf_i[l], min_nsamples), 0.0, 5.0)        # :1168-1187
            bd, sd, nud = sigma_nu_student(lattice(W1, kf_w[l], min_nsamples), 0.0, 0.0025)
            nui = max(nui, nud)
            A, b = build_system(kf_w[l], kf_i[l], gwx, gwy, gix, giy, W1, I1, Kl, sd, si, bd, bi, nud, nui)
            x = np.linalg.solve(A, b)                                                   # :1249
            Rinc = np.linalg.inv(exp_map_rot(x[3:]))                                    # :1252-1263
            t = Rinc @ t - Rinc @ x[:3]
            R = Rinc @ R
    return R, t


# ---- SE(3) helpers (src/util_funcs.cpp:31-123, include/util_funcs.h:50-58) and the tracker's per-frame logic -------------------------------
def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], np.float64)


def _Q(w):
    th = np.linalg.norm(w); Om = _skew(w)
    if th < 1e-5:
        return np.eye(3) + 0.5 * Om + Om @ Om / 6.0
    return np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (1 - np.sin(th) / th) / th ** 2 * Om @ Om


def exp_map(w, v):
    """expMap, src/util_funcs.cpp:85-123: (R, t) = (exp([w]x), Q(w) v)"""
    return exp_map_rot(w), _Q(np.asarray(w, np.float64)) @ np.asarray(v, np.float64)


def log_map(R, t):
    """logMap, src/util_funcs.cpp:31-83: twist (v, w) of (R, t)"""
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt((r * r).sum() * 0.25)
    th = np.arccos(np.clip((np.trace(R) - 1) * 0.5, -1, 1))
    k = (1.0 + th * th / 6.0 + 7.0 / 360.0 * th ** 4) if s < 1e-5 else th / s
    w = r * (k / 2.0)
    return np.linalg.solve(_Q(w), np.asarray(t, np.float64)), w


class Tracker:
    """VisodoTracker::trackNewFrame, src/visodo.cpp:1967-2247, for the shipped configuration (constant-velocity prediction, PYR_FIRST, Student-t,
    sigma from the residual pdf, independent weights, no gradient filtering), built only from this module's kernels: keyframe bookkeeping
    (:826-893, 1541-1575), the Gauss-Newton loop and its covariance pass (:944-1479), covisibility (:1481-1514), keyframe switching and fusion
    (:2172-2215, 1674-1764).  The tracking-lost branch is not mirrored (the sequences used here never lose track)."""

    def __init__(self, K, rows, cols, iters=(10, 5, 3), visratio_odo=0.9, visratio_integr=0.7, delta_t=0.03333, min_nsamples=10000):
        self.K, self.rows, self.cols, self.iters = tuple(float(v) for v in K), rows, cols, tuple(iters)
        self.th_odo, self.th_int, self.dt, self.nsamp = visratio_odo, visratio_integr, np.float32(delta_t), min_nsamples
        self.time = 0
        self.poses = []
        self.info = []

    # -- images
    def _prepare(self, depth, rgb):
        w, i = [depth_to_invdepth(depth)], [intensity(rgb)]
        for _ in range(1, len(self.iters)):
            w.append(pyr_down(w[-1])); i.append(pyr_down(i[-1]))
        return w, i

    def _save_odo_kf(self):                                            # saveCurrentImagesAsOdoKeyframes :826-878
        self.kf_w, self.kf_i = [a.copy() for a in self.cur_w], [a.copy() for a in self.cur_i]
        fw, fi = bilateral(self.kf_w[0], 2 * 0.0025), bilateral(self.kf_i[0], 3.0)
        self.cov_grads = sobel(fw) + sobel(fi)                          # gWx, gWy, gIx, gIy of the filtered level-0 maps
        self.grads = [sobel(w) + sobel(i) for w, i in zip(self.kf_w, self.kf_i)]

    def _save_integr_kf(self):                                         # saveCurrentImagesAsIntegrationKeyframes :880-893
        self.int_w, self.int_raw = self.cur_w[0].copy(), self.cur_w[0].copy()
        self.int_weight = np.ones_like(self.cur_w[0])

    def _Km(self, level=0):
        fx, fy, cx, cy = [v / (1 << level) for v in self.K]
        return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])

    def _covis(self, R_ab, t_ab, A, B):                                # computeCovisibility :1481-1514
        Km = self._Km(); Ki = np.linalg.inv(Km)
        r_ba, _, _ = visibility_ratio(B, A, Km @ R_ab @ Ki, Km @ t_ab)
        Ri = np.linalg.inv(R_ab)
        r_ab, _, _ = visibility_ratio(A, B, Km @ Ri @ Ki, -Km @ Ri @ t_ab)
        return min(r_ab, r_ba)

    # -- one frame
    def track(self, depth, rgb):
        self.cur_w, self.cur_i = self._prepare(depth, rgb)
        if self.time == 0:                                             # :1994-2045
            self.time = 1
            self.odo_R, self.odo_t = np.eye(3), np.zeros(3); self.int_R, self.int_t = np.eye(3), np.zeros(3)
            self.est_R, self.est_t = np.eye(3), np.zeros(3)
            self.dR, self.dt_, self.dcov = np.eye(3), np.zeros(3), np.zeros((6, 6))
            self.vel, self.omega = np.zeros(3), np.zeros(3)
            self.odo_count = self.int_count = 0
            self._save_odo_kf(); self._save_integr_kf()
            self.poses.append((np.eye(3), np.zeros(3)))
            return
        # ---- estimateVisualOdometry :944-1479
        prev_R, prev_t = self.dR, self.dt_
        if self.time > 1:                                              # constant-velocity prediction :1012-1025
            dRp, dtp = exp_map(self.omega * self.dt, self.vel * self.dt)
            R, t = prev_R @ dRp, prev_R @ dtp + prev_t
        else:
            R, t = prev_R, prev_t
        for l in range(len(self.iters) - 1, -1, -1):
            Kl = tuple(v / (1 << l) for v in self.K); Km = self._Km(l); Ki = np.linalg.inv(Km)
            gwx, gwy, gix, giy = self.grads[l]
            for _ in range(self.iters[l]):
                Ri = np.linalg.inv(R)
                Rp, tp = Km @ Ri @ Ki, Km @ (-Ri @ t)
                W1 = warp_invdepth(self.cur_w[l], self.kf_w[l], Rp, tp)
                I1 = warp_intensity(self.cur_i[l], W1, Rp, tp, tex8=True)
                bi, si, nui = sigma_nu_student(lattice(I1, self.kf_i[l], self.nsamp), 0.0, 5.0)
                bd, sd, nud = sigma_nu_student(lattice(W1, self.kf_w[l], self.nsamp), 0.0, 0.0025)
                nui = max(nui, nud)
                A, b = build_system(self.kf_w[l], self.kf_i[l], gwx, gwy, gix, giy, W1, I1, Kl, sd, si, bd, bi, nud, nui)
                x = np.linalg.solve(A, b)
                Rinc = np.linalg.inv(exp_map_rot(x[3:]))
                t = Rinc @ t - Rinc @ x[:3]; R = Rinc @ R
        # covariance pass :1283-1330 (filtered-map gradients, Student-t with 5 dof, reference sigmas)
        Km = self._Km(0); Ki = np.linalg.inv(Km); Ri = np.linalg.inv(R)
        Rp, tp = Km @ Ri @ Ki, Km @ (-Ri @ t)
        W1 = warp_invdepth(self.cur_w[0], self.kf_w[0], Rp, tp)
        I1 = warp_intensity(self.cur_i[0], W1, Rp, tp, tex8=True)
        cg = self.cov_grads
        A, _ = build_system(self.kf_w[0], self.kf_i[0], cg[0], cg[1], cg[2], cg[3], W1, I1, self.K, 0.0025, 5.0, student_nu=False, mestimator=3)
        self.dR, self.dt_, self.dcov = R, t, np.linalg.inv(A)
        est_cov = self.dcov.copy()
        v, w = log_map(prev_R.T @ R, prev_R.T @ (t - prev_t))          # :1459-1468
        self.vel, self.omega = v / self.dt, w / self.dt
        # ---- trackNewFrame :2059-2215
        self.est_t = self.odo_t + self.odo_R @ self.dt_; self.est_R = self.odo_R @ self.dR
        self.poses.append((self.est_R.copy(), self.est_t.copy()))
        self.odo_count += 1; self.int_count += 1
        vis_odo = self._covis(self.dR, self.dt_, self.kf_w[0], self.cur_w[0])
        sw_odo = vis_odo < self.th_odo
        if sw_odo:                                                     # resetOdometryKeyframe :1541-1575 (pose part)
            self.odo_count = 0
            self.odo_R, self.odo_t = self.est_R.copy(), self.est_t.copy()
            self.dR, self.dt_, self.dcov = np.eye(3), np.zeros(3), np.zeros((6, 6))
            self._save_odo_kf()
        iR = self.int_R.T @ self.est_R; it = self.int_R.T @ (self.est_t - self.int_t)
        vis_int = self._covis(iR, it, self.int_raw, self.cur_w[0])
        sw_int = vis_int < self.th_int
        if sw_int:
            self.int_count = 0
            self.int_R, self.int_t = self.est_R.copy(), self.est_t.copy()
            self._save_integr_kf()
        else:                                                          # integrateImagesIntoKeyframes :1674-1764
            Kd = self._Km(0)
            Rp = Kd @ iR @ np.linalg.inv(Kd); tp = Kd @ it
            Rpi = np.linalg.inv(Rp)
            ws, wt = warp_invdepth_weighted(self.cur_w[0], self.int_w, Rpi, -Rpi @ tp)
            self.int_w, self.int_weight = integrate(ws, np.where(np.isnan(wt), 0.0, wt), self.int_w, self.int_weight)
        self.info.append(dict(vis_odo=vis_odo, vis_int=vis_int, sw_odo=bool(sw_odo), sw_int=bool(sw_int), cov=est_cov))
        self.time += 1


# ---- custom-calibration front-end (src/cuda/undistortion.cu) ----------------------------------------------------------------------------
def _distorted_coords(rows, cols, k):
    """undistortKernel, src/cuda/undistortion.cu:131-160 with distortPixel :96-108: for every undistorted pixel the (x + 0.5, y + 0.5)
    position in the distorted image (radial k1, k2, k5; tangential k3, k4)"""
    fx, fy, cx, cy, k1, k2, k3, k4, k5 = [float(v) for v in k]
    yu, xu = np.mgrid[0:rows, 0:cols].astype(np.float64)
    u, v = (xu - cx) / fx, (yu - cy) / fy
    r2 = u * u + v * v
    fr = 1.0 + k1 * r2 + k2 * r2 * r2 + k5 * r2 ** 3
    ud = fr * u + 2.0 * k3 * u * v + k4 * (r2 + 2.0 * u * u)
    vd = fr * v + 2.0 * k4 * u * v + k3 * (r2 + 2.0 * v * v)
    return fx * ud + cx + 0.5, fy * vd + cy + 0.5


def undistort_intensity(src, k, tex8=True):
    """undistortIntensity, src/cuda/undistortion.cu:208-245: bilinear texture fetch at the distorted position; NaN outside (0, cols) x (0, rows)"""
    rows, cols = src.shape
    xd, yd = _distorted_coords(rows, cols, k)
    ok = ~((xd <= 0) | (yd <= 0) | (xd >= cols) | (yd >= rows))
    return np.where(ok, bilinear(src, np.where(ok, xd, 0.5), np.where(ok, yd, 0.5), tex8), np.nan)


def undistort_depthinv(src, k, c1, c0, q0, q1, xshift, yshift):
    """undistortDepthInv, src/cuda/undistortion.cu:247-312: per-pixel correction (1 + D1(u, v)) (c1 w + c0) + D0(u, v) of the SHIFTED raw inverse
    depth (depthinvCorrectionKernel :162-206, polynomials :110-123), then a point-sampled fetch of the corrected map at the distorted position.
    Returns (corrected, undistorted)."""
    fx, fy, cx, cy = [float(v) for v in k[:4]]
    rows, cols = src.shape
    y, x = np.mgrid[0:rows, 0:cols]
    u, v = (x - cx) / fx, (y - cy) / fy
    r2 = u * u + v * v
    basis = [np.ones_like(u), r2, r2 * r2, r2 ** 3, u, v, u * v, u * u * v, u * v * v]
    D0 = sum(float(q) * b for q, b in zip(q0, basis)); D1 = sum(float(q) * b for q, b in zip(q1, basis))
    xs, ys = x - xshift, y - yshift
    ok = (xs > 0) & (ys > 0)
    val = src.astype(np.float64)[np.where(ok, ys, 0), np.where(ok, xs, 0)]
    corr = np.where(ok, (1.0 + D1) * (c1 * val + c0) + D0, np.nan)
    xd, yd = _distorted_coords(rows, cols, k)
    inside = ~((xd <= 0) | (yd <= 0) | (xd >= cols) | (yd >= rows))
    ix = np.clip(np.floor(np.where(inside, xd, 0)), 0, cols - 1).astype(int); iy = np.clip(np.floor(np.where(inside, yd, 0)), 0, rows - 1).astype(int)
    return corr, np.where(inside, corr[iy, ix], np.nan)


def register_depthinv(src, dRc_proj, t_dc_proj, cRd_proj, scale=3):
    """registerDepthinv, src/cuda/warping_registration.cu:720-822: the depth camera's inverse depth re-rendered in the colour camera.
    (1) translation only (registerPixelTranslationOnly :148-162) splatted with dilation into an enlarged canvas, the nearest surface winning
    (atomicMax on the bits of the positive inverse depth, :236-288; canvas initialised / converted by :164-203); (2) the remaining rotation
    as a homography with a point-sampled fetch from the canvas (:597-635).  Returns (canvas, registered)."""
    s = np.asarray(src, np.float32)
    rows, cols = s.shape
    R, C_ = scale * rows, scale * cols
    ox, oy = (C_ - cols) // 2, (R - rows) // 2
    t = np.asarray(t_dc_proj, np.float32)
    canvas = np.zeros((R, C_), np.float32)                     # 0 = empty (the int view of +0.0f)
    f = np.float32
    for yd in range(rows):
        for xd in range(cols):
            wd = s[yd, xd]
            if np.isnan(wd):
                continue
            zd = f(1) / wd
            X = np.array([f(xd) * zd, f(yd) * zd, zd], np.float32) - t
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                wc = f(1) / X[2]
                xc, yc = X[0] * wc, X[1] * wc
                if not wc > f(0.01):
                    continue
                dil = wc / wd
                lim = lambda v: int(np.clip(np.rint(v), -2 ** 31, 2 ** 31 - 1))
                x0, x1 = lim(xc - f(0.5) * dil) + ox, lim(xc + f(0.5) * dil) + ox
                y0, y1 = lim(yc - f(0.5) * dil) + oy, lim(yc + f(0.5) * dil) + oy
            for x in range(max(0, x0), min(x1 + 1, C_)):
                for y in range(max(0, y0), min(y1 + 1, R)):
                    canvas[y, x] = max(canvas[y, x], wc)
    inter = np.where(canvas != 0, canvas, np.nan).astype(np.float32)
    H = np.asarray(dRc_proj, np.float64).reshape(3, 3); Hi = np.asarray(cRd_proj, np.float64).reshape(3, 3)
    y, x = np.mgrid[0:rows, 0:cols].astype(np.float64)
    p = np.einsum("ij,jhw->ihw", H, np.stack([x, y, np.ones_like(x)]))
    p = p / p[2]
    xs, ys = p[0] + 0.5 + ox, p[1] + 0.5 + oy
    ix, iy = np.floor(xs), np.floor(ys)
    inb = (ix >= 0) & (iy >= 0) & (ix < C_) & (iy < R)
    w = inter.astype(np.float64)[np.where(inb, iy, 0).astype(int), np.where(inb, ix, 0).astype(int)]
    with np.errstate(invalid="ignore", divide="ignore"):
        res = w / np.einsum("j,jhw->hw", Hi[2], p)
        return inter, np.where(inb & (res > 0), res, np.nan)


def _m_weight(en, mest):
    """computeWeight-style M-estimator weights of sigmaFuncs.cu:207-233: returns (weight, is_valid)"""
    ok = np.ones_like(en)
    if mest == 1:
        w = np.where(np.abs(en) > 1.345, 1.345 / np.maximum(np.abs(en), 1e-300), 1.0)
    elif mest == 2:
        inside = np.abs(en) < 4.685
        w = np.where(inside, (1.0 - (en / 4.685) ** 2) ** 2, 0.0); ok = inside.astype(np.float64)
    elif mest == 3:
        w = 6.0 / (5.0 + en * en)
    else:
        w = np.ones_like(en)
    return w, ok


def sigma_pdf(err, bias, sigma, mestimator=3):
    """computeSigmaPdf, src/cuda/sigmaFuncs.cu:773-854 with partialBiasAndSigma :179-255: IRLS bias / scale with a fixed M-estimator (first pass
    least squares), at most 10 passes, stop when sigma changes by < 10 % -- here the comparison is against the PREVIOUS pass's sigma and the
    returned values are those of the last pass."""
    e = np.asarray(err, np.float64); e = e[np.isfinite(e)]
    sh_b, sh_s, mest = float(bias), float(sigma), 0
    b = s = None
    for i in range(10):
        w, ok = _m_weight((e - sh_b) / sh_s, mest)
        swr, swsr, sw, n = (e * w).sum(), (e * e * w).sum(), w.sum(), ok.sum()
        b = swr / sw
        s = np.sqrt((swsr - 2.0 * b * swr + b * b * sw) / n)
        if i > 0 and abs(s - sh_s) / sh_s < 0.1:
            break
        sh_b, sh_s, mest = b, s, mestimator
    return b, s


def chi_square(err_int, err_depth, sigma_int, sigma_depth, mestimator=3):
    """computeChiSquare, src/cuda/sigmaFuncs.cu:1225-1297 with normalizeAndAppendErrorsKernel :137-150 and partialChiSquared :541-600: mean
    robust cost rho of the normalised residuals of both channels, the number of finite residuals, and the Gaussian tail test built from them.
    Returns (chi_squared, chi_test, Ndof)."""
    from scipy.special import erf
    en = np.concatenate([np.asarray(err_int, np.float64) / sigma_int, np.asarray(err_depth, np.float64) / sigma_depth])
    en = en[np.isfinite(en)]
    rho = en * en / 2.0
    if mestimator == 1:
        rho = np.where(np.abs(en) > 1.345, 1.345 * (np.abs(en) - 1.345 / 2.0), rho)
    elif mestimator == 2:
        c = 4.685 * 4.685 / 6.0
        rho = np.where(np.abs(en) < 4.685, c * (1.0 - (1.0 - (en / 4.685) ** 2) ** 3), c)
    elif mestimator == 3:
        rho = 3.0 * np.log(1.0 + en * en / 5.0)
    n = float(en.size)
    chi = rho.sum() / n
    z = (chi - n) / np.sqrt(2.0 * n)
    return chi, 0.5 * (1.0 + erf(z / np.sqrt(2.0))), n


def generate_image_rgb(vmap_, nmap_, rgb, light):
    """ImageGeneratorRGB, src/cuda/image_generator.cu:122-180 with one light (getImage, visodo.cpp:559-580): the keyframe colours shaded by
    |cos| between the normal and the direction to the light, brightness (int)(205 w) + 50 clamped to [0, 255]; black where vertex or normal
    is invalid."""
    rows = rgb.shape[0]
    v = np.asarray(vmap_, np.float64).reshape(3, rows, -1); n = np.asarray(nmap_, np.float64).reshape(3, rows, -1)
    ok = ~np.isnan(v[0]) & ~np.isnan(n[0])
    with np.errstate(invalid="ignore"):
        vec = np.asarray(light, np.float64)[:, None, None] - v
        vec = vec / np.sqrt((vec * vec).sum(0))
        w = np.abs((vec * n).sum(0))
        br = np.clip(np.floor(np.where(ok, 205.0 * w, 0.0)) + 50, 0, 255) / 255.0
    out = np.rint(rgb.astype(np.float64) * br[..., None])
    return np.where(ok[..., None], out, 0).astype(np.uint8)
