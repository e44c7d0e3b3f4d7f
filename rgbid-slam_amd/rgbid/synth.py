"""Synthetic RGB-D sequences with analytic ground truth.

No TUM / ICL-NUIM data exists in this image, so every benchmark and parity input is rendered
from an analytic scene: a smooth height field z = f(x, y) in the world frame (= first camera
frame, +z forward) textured with band-limited value noise + two sinusoid gratings, viewed from a
smooth SE(3) camera path.  Frames are rendered by exact ray / surface intersection (fixed-point
iteration), depth is quantised to u16 millimetres (the tracker's input unit after the TUM x0.2
convention, tools/evaluation.cpp:285), with optional dropout and sensor noise.

Written with torch so the same code renders on the GPU (bench) and on the CPU (tests).
"""
import math

import torch

SEED = 20260928
TUM_K = (525.0, 525.0, 319.5, 239.5)  # config_data/calibration_factory.ini


def _lattice(seed, n=64, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    return torch.rand((n, n), generator=g, dtype=torch.float64).to(device)


def _value_noise(x, y, table):
    n = table.shape[0]
    xf, yf = torch.floor(x), torch.floor(y)
    fx, fy = x - xf, y - yf
    sx, sy = fx * fx * (3 - 2 * fx), fy * fy * (3 - 2 * fy)
    x0 = torch.remainder(xf.long(), n); y0 = torch.remainder(yf.long(), n)
    x1 = torch.remainder(x0 + 1, n); y1 = torch.remainder(y0 + 1, n)
    v00, v10, v01, v11 = table[y0, x0], table[y0, x1], table[y1, x0], table[y1, x1]
    return (v00 * (1 - sx) + v10 * sx) * (1 - sy) + (v01 * (1 - sx) + v11 * sx) * sy


class Scene:
    """Height field + albedo in world coordinates (metres)."""

    def __init__(self, seed=SEED, device="cpu", z0=2.0, amp=0.25, Lx=1.7, Ly=1.3, tilt=(0.08, -0.05)):
        self.device = device
        self.z0, self.amp, self.Lx, self.Ly, self.tilt = z0, amp, Lx, Ly, tilt
        self.tables = [_lattice(seed + 7 * c + o, device=device) for c in range(3) for o in range(3)]
        self.phase = [0.0, 1.1, 2.3]

    def depth(self, x, y):
        return (self.z0 + self.amp * torch.sin(2 * math.pi * x / self.Lx) * torch.cos(2 * math.pi * y / self.Ly)
                + self.tilt[0] * x + self.tilt[1] * y)

    def albedo(self, x, y):
        out = []
        for c in range(3):
            v = 0.0
            for o in range(3):
                f = 6.0 * (2 ** o)
                v = v + _value_noise(x * f + 17.3 * c, y * f + 5.1 * o, self.tables[3 * c + o]) / (2 ** o)
            v = v / 1.75
            g = 0.5 + 0.25 * torch.sin(2 * math.pi * (x * 3.1 + y * 1.7) + self.phase[c]) \
                + 0.25 * torch.sin(2 * math.pi * (-x * 1.3 + y * 4.3) + 2 * self.phase[c])
            out.append(255.0 * torch.clamp(0.65 * v + 0.35 * g, 0.0, 1.0))
        return torch.stack(out, dim=-1)


def rodrigues(w):
    w = torch.as_tensor(w, dtype=torch.float64)
    th = torch.linalg.norm(w)
    K = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=torch.float64)
    if th < 1e-12:
        return torch.eye(3, dtype=torch.float64) + K
    return torch.eye(3, dtype=torch.float64) + torch.sin(th) / th * K + (1 - torch.cos(th)) / th ** 2 * (K @ K)


def camera_path(n_frames, seed=SEED, trans_step=(0.005, 0.03), rot_step_deg=(0.1, 1.5), smooth=0.85):
    """Camera-to-world poses (R_wc, t_wc), frame 0 = identity; per-frame translation step ~U(0.5,3) cm and
    rotation step ~U(0.1,1.5) deg along slowly varying directions."""
    g = torch.Generator().manual_seed(seed + 1000)
    Rs, ts = [torch.eye(3, dtype=torch.float64)], [torch.zeros(3, dtype=torch.float64)]
    dv = torch.randn(3, generator=g, dtype=torch.float64)
    dw = torch.randn(3, generator=g, dtype=torch.float64)
    for _ in range(1, n_frames):
        dv = smooth * dv + (1 - smooth) * torch.randn(3, generator=g, dtype=torch.float64)
        dw = smooth * dw + (1 - smooth) * torch.randn(3, generator=g, dtype=torch.float64)
        # keep the camera looking at the textured surface: damp z translation and roll
        dirv = dv * torch.tensor([1.0, 1.0, 0.5], dtype=torch.float64)
        dirw = dw * torch.tensor([1.0, 1.0, 0.5], dtype=torch.float64)
        step = trans_step[0] + (trans_step[1] - trans_step[0]) * torch.rand((), generator=g, dtype=torch.float64)
        ang = math.radians(1.0) * (rot_step_deg[0] + (rot_step_deg[1] - rot_step_deg[0]) * torch.rand((), generator=g, dtype=torch.float64))
        v = dirv / torch.linalg.norm(dirv) * step
        w = dirw / torch.linalg.norm(dirw) * ang
        # relative motion expressed in the current camera frame: T_w,k+1 = T_w,k * dT
        Rs.append(Rs[-1] @ rodrigues(w))
        ts.append(ts[-1] + Rs[-2] @ v)
    return torch.stack(Rs), torch.stack(ts)


def render(scene, R_wc, t_wc, K=TUM_K, rows=480, cols=640, iters=40):
    """Exact ray/height-field intersection. Returns (depth [m] float64 rows x cols, rgb float64 rows x cols x 3)."""
    dev = scene.device
    fx, fy, cx, cy = K
    v, u = torch.meshgrid(torch.arange(rows, dtype=torch.float64, device=dev),
                          torch.arange(cols, dtype=torch.float64, device=dev), indexing="ij")
    dc = torch.stack([(u - cx) / fx, (v - cy) / fy, torch.ones_like(u)], dim=-1)  # camera ray, z = 1
    R = torch.as_tensor(R_wc, dtype=torch.float64, device=dev)
    t = torch.as_tensor(t_wc, dtype=torch.float64, device=dev)
    dw = dc @ R.T
    lam = torch.full_like(u, scene.z0)
    for _ in range(iters):
        P = t + lam[..., None] * dw
        lam = (scene.depth(P[..., 0], P[..., 1]) - t[2]) / dw[..., 2]
    P = t + lam[..., None] * dw
    return lam, scene.albedo(P[..., 0], P[..., 1])


def make_frame(scene, R_wc, t_wc, K=TUM_K, rows=480, cols=640, seed=0, noise=True, dropout=0.03):
    """Sensor model: depth -> u16 mm with sigma_z = 1.4e-3 z^2 noise, 3% dropout, intensity noise sigma = 2."""
    depth, rgb = render(scene, R_wc, t_wc, K, rows, cols)
    g = torch.Generator(device="cpu").manual_seed(SEED + 31 * seed)
    if noise:
        depth = depth + (1.4e-3 * depth ** 2) * torch.randn(depth.shape, generator=g, dtype=torch.float64).to(depth.device)
        rgb = rgb + 2.0 * torch.randn(rgb.shape, generator=g, dtype=torch.float64).to(rgb.device)
    d_mm = torch.clamp(torch.round(depth * 1000.0), 0, 65535)
    if dropout > 0:
        drop = torch.rand(depth.shape, generator=g, dtype=torch.float64).to(depth.device) < dropout
        d_mm = torch.where(drop, torch.zeros_like(d_mm), d_mm)
    rgb8 = torch.clamp(torch.round(rgb), 0, 255).to(torch.uint8)
    return d_mm.to(torch.int32), rgb8


def make_sequence(n_frames, seed=SEED, K=TUM_K, rows=480, cols=640, device="cpu", noise=True, dropout=0.03,
                  trans_step=(0.005, 0.03), rot_step_deg=(0.1, 1.5)):
    """Returns dict(depth u16 [n,rows,cols] (as int32 tensor), rgb u8 [n,rows,cols,3], R_wc [n,3,3], t_wc [n,3])."""
    scene = Scene(seed=seed, device=device)
    Rs, ts = camera_path(n_frames, seed=seed, trans_step=trans_step, rot_step_deg=rot_step_deg)
    depths, rgbs = [], []
    for k in range(n_frames):
        d, c = make_frame(scene, Rs[k], ts[k], K, rows, cols, seed=seed * 1000 + k, noise=noise, dropout=dropout)
        depths.append(d)
        rgbs.append(c)
    return {"depth": torch.stack(depths), "rgb": torch.stack(rgbs), "R_wc": Rs, "t_wc": ts, "K": K}


def camera_path_bounded(n_frames, seed=SEED):
    """A long hand-held-scan path that STAYS in front of the textured surface (the random walk of camera_path drifts away over thousands of
    frames): position and orientation are sums of two incommensurate sinusoids per axis -- the camera sweeps about +-0.8 m sideways, +-0.5 m
    up / down, +-0.25 m in depth and pans / tilts by about +-17 / +-10 degrees -- with per-frame steps of 0.5-3 cm and 0.1-1 degree, the range
    SURVEY 8d asks for.  Frame 0 = identity (the path is expressed relative to its first pose)."""
    g = torch.Generator().manual_seed(seed + 2000)
    ph = 2 * math.pi * torch.rand(12, generator=g, dtype=torch.float64)
    k = torch.arange(n_frames, dtype=torch.float64)
    amp_t = [(0.60, 397.0, 0.20, 89.0), (0.35, 311.0, 0.15, 71.0), (0.18, 523.0, 0.07, 113.0)]
    amp_r = [(0.17, 283.0, 0.04, 61.0), (0.25, 353.0, 0.05, 53.0), (0.06, 431.0, 0.02, 97.0)]       # tilt (x), pan (y), roll (z) in radians
    tw = torch.stack([a * torch.sin(2 * math.pi * k / p_ + ph[2 * i]) + b * torch.sin(2 * math.pi * k / q + ph[2 * i + 1]) for i, (a, p_, b, q) in enumerate(amp_t)], 1)
    rw = torch.stack([a * torch.sin(2 * math.pi * k / p_ + ph[6 + 2 * i]) + b * torch.sin(2 * math.pi * k / q + ph[7 + 2 * i]) for i, (a, p_, b, q) in enumerate(amp_r)], 1)
    Rs = torch.stack([rodrigues(rw[i]) for i in range(n_frames)])
    R0i = Rs[0].T
    return torch.stack([R0i @ Rs[i] for i in range(n_frames)]), torch.stack([R0i @ (tw[i] - tw[0]) for i in range(n_frames)])


def make_long_sequence(n_frames, seed=SEED, K=TUM_K, rows=480, cols=640, device="cuda", batch=25, dropout=0.03, path="bounded",
                       trans_step=(0.005, 0.03), rot_step_deg=(0.1, 1.5)):
    """A long sequence (thousands of frames: BASELINE config 4) of the same scene / camera-path / sensor model as make_sequence, rendered `batch`
    frames at a time with the noise drawn on the device -- a few milliseconds per frame instead of ~25.  Same dict as make_sequence; the noise
    realisation differs from make_sequence's (CPU generator), so the two are different sequences of the same distribution."""
    scene = Scene(seed=seed, device=device)
    Rs, ts = camera_path_bounded(n_frames, seed=seed) if path == "bounded" else camera_path(n_frames, seed=seed, trans_step=trans_step, rot_step_deg=rot_step_deg)
    fx, fy, cx, cy = K
    v, u = torch.meshgrid(torch.arange(rows, dtype=torch.float64, device=device), torch.arange(cols, dtype=torch.float64, device=device), indexing="ij")
    dc = torch.stack([(u - cx) / fx, (v - cy) / fy, torch.ones_like(u)], dim=-1)          # [rows, cols, 3]
    g = torch.Generator(device=device).manual_seed(SEED + 977 * seed)
    depth_out = torch.empty((n_frames, rows, cols), dtype=torch.int16, device=device)
    rgb_out = torch.empty((n_frames, rows, cols, 3), dtype=torch.uint8, device=device)
    for k0 in range(0, n_frames, batch):
        k1 = min(n_frames, k0 + batch)
        R = Rs[k0:k1].to(device); t = ts[k0:k1].to(device)
        dw = torch.einsum("hwc,fdc->fhwd", dc, R)                                           # ray directions in the world frame
        t_ = t[:, None, None, :]
        lam = torch.full((k1 - k0, rows, cols), scene.z0, dtype=torch.float64, device=device)
        for _ in range(40):
            P = t_ + lam[..., None] * dw
            lam = (scene.depth(P[..., 0], P[..., 1]) - t_[..., 2]) / dw[..., 2]
        P = t_ + lam[..., None] * dw
        rgb = scene.albedo(P[..., 0], P[..., 1])
        depth = lam + (1.4e-3 * lam ** 2) * torch.randn(lam.shape, generator=g, dtype=torch.float64, device=device)
        rgb = rgb + 2.0 * torch.randn(rgb.shape, generator=g, dtype=torch.float64, device=device)
        d_mm = torch.clamp(torch.round(depth * 1000.0), 0, 65535)
        if dropout > 0:
            d_mm = torch.where(torch.rand(lam.shape, generator=g, dtype=torch.float64, device=device) < dropout, torch.zeros_like(d_mm), d_mm)
        depth_out[k0:k1] = d_mm.to(torch.int32).to(torch.int16)      # u16 bit pattern
        rgb_out[k0:k1] = torch.clamp(torch.round(rgb), 0, 255).to(torch.uint8)
    return {"depth": depth_out, "rgb": rgb_out, "R_wc": Rs, "t_wc": ts, "K": K}


def relative_pose(R_wa, t_wa, R_wb, t_wb):
    """Pose of camera b in the frame of camera a (X_a = R X_b + t) -- the tracker's KF-relative convention."""
    R = R_wa.T @ R_wb
    t = R_wa.T @ (t_wb - t_wa)
    return R, t
