"""ctypes binding of include/rgbid_batched.h: the engine's hot-path kernels as single calls over `lanes` images.

Images are torch CUDA tensors [lanes, rows, cols] (float32 / int16 depth) or [lanes, rows, cols, 3] (uint8 rgb) whose last dimension is
dense; the row and lane strides may be padded (rgbid_imgb carries both).  Per-lane transforms are numpy arrays [lanes, 9] / [lanes, 3].
torch is only the device-memory allocator here: every call below runs the hand-written HIP kernels of csrc/ (no CPU path).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Intr, check

NUMERICS_EXACT, NUMERICS_FAST = 0, 1
WM_AUTO, WM_GENERIC, WM_STUDENT_NU, WM_STUDENT_FIXED = -1, 0, 1, 2

BATCHED_EXPORTS = [
    "rgbid_gn_fused_batched", "rgbid_build_system_batched", "rgbid_warp_pair_batched", "rgbid_lattice_pack_batched",
    "rgbid_lattice_residuals_batched", "rgbid_sigma_pair_batched", "rgbid_fuse_frame_batched", "rgbid_kf_maps_batched",
    "rgbid_visibility_pair_batched", "rgbid_prep_frame_batched", "rgbid_pyr_down_batched", "rgbid_compute_gradient_batched",
    "rgbid_bilateral_filter_batched", "rgbid_gradient_keep_batched",
]


class ImgB(C.Structure):
    """rgbid_imgb: lanes images of one geometry in one allocation"""
    _fields_ = [("data", C.c_void_p), ("step", C.c_size_t), ("lane_stride", C.c_size_t), ("rows", C.c_int), ("cols", C.c_int)]


class SysParams(C.Structure):
    _fields_ = [("sigma_depthinv", C.c_float), ("sigma_int", C.c_float), ("bias_depthinv", C.c_float), ("bias_int", C.c_float),
                ("nu_depthinv", C.c_float), ("nu_int", C.c_float), ("mestimator", C.c_int), ("weighting", C.c_int),
                ("student_nu", C.c_int), ("nu_int_from_max", C.c_int)]


class ScalePair(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("bias_depthinv", "sigma_depthinv", "nu_depthinv", "bias_int", "sigma_int", "nu_int")]


def imgb(t):
    """rgbid_imgb view of a CUDA tensor [lanes, rows, cols] or [lanes, rows, cols, 3] (uint8)"""
    if not t.is_cuda:
        raise _lib.RgbidError("rgbid images must live in device memory (no CPU path)")
    es = t.element_size()
    if t.dim() == 4:
        assert t.shape[3] == 3 and t.dtype == torch.uint8 and t.stride(3) == 1 and t.stride(2) == 3
    else:
        assert t.dim() == 3 and t.stride(2) == 1
    return ImgB(t.data_ptr(), t.stride(1) * es, t.stride(0) * es, t.shape[1], t.shape[2])


def _f32(a, lanes, n):
    a = np.ascontiguousarray(np.asarray(a, np.float32).reshape(lanes, n))
    return a, a.ctypes.data_as(C.c_void_p)


def sys_params(lanes, **kw):
    """[lanes] rgbid_sys_params; every keyword is a scalar (all lanes) or a per-lane sequence"""
    arr = (SysParams * lanes)()
    defaults = dict(sigma_depthinv=0.0025, sigma_int=5.0, bias_depthinv=0.0, bias_int=0.0, nu_depthinv=5.0, nu_int=5.0,
                    mestimator=3, weighting=0, student_nu=1, nu_int_from_max=0)
    defaults.update(kw)
    for k, v in defaults.items():
        for l in range(lanes):
            setattr(arr[l], k, v[l] if isinstance(v, (list, tuple, np.ndarray)) else v)
    return arr


class Batched:
    """the batched calls on one rgbid context (rgbid.device.Context)"""

    def __init__(self, ctx):
        self.ctx = ctx
        self.L = ctx.L
        self._h = ctx._h

    def gn_fused(self, W0, I0, gWx, gWy, gIx, gIy, Wcur, Icur, R_proj, t_proj, intr, params, fast=False, weight_mode=WM_AUTO, return_ms=False):
        lanes = W0.shape[0]
        Rk, Rp = _f32(R_proj, lanes, 9); tk, tp = _f32(t_proj, lanes, 3)
        A = np.zeros((lanes, 36)); b = np.zeros((lanes, 6)); ms = C.c_float()
        maps = [C.byref(imgb(t)) for t in (W0, I0, gWx, gWy, gIx, gIy, Wcur, Icur)]
        check(self.L.rgbid_gn_fused_batched(self._h, lanes, *maps, Rp, tp, Intr(*intr), params, int(bool(fast)), int(weight_mode),
                                            A.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.byref(ms)))
        A = A.reshape(lanes, 6, 6)
        return (A, b, ms.value) if return_ms else (A, b)

    def build_system(self, W0, I0, gWx, gWy, gIx, gIy, W1, I1, intr, params, return_ms=False):
        lanes = W0.shape[0]
        A = np.zeros((lanes, 36)); b = np.zeros((lanes, 6)); ms = C.c_float()
        maps = [C.byref(imgb(t)) for t in (W0, I0, gWx, gWy, gIx, gIy, W1, I1)]
        check(self.L.rgbid_build_system_batched(self._h, lanes, *maps, Intr(*intr), params, A.ctypes.data_as(C.c_void_p),
                                                b.ctypes.data_as(C.c_void_p), C.byref(ms)))
        A = A.reshape(lanes, 6, 6)
        return (A, b, ms.value) if return_ms else (A, b)

    def warp_pair(self, src_iD, src_I, grid, dst_iD, dst_I, R_proj, t_proj, fast=False):
        lanes = grid.shape[0]
        Rk, Rp = _f32(R_proj, lanes, 9); tk, tp = _f32(t_proj, lanes, 3)
        ms = C.c_float()
        check(self.L.rgbid_warp_pair_batched(self._h, lanes, *[C.byref(imgb(t)) for t in (src_iD, src_I, grid, dst_iD, dst_I)], Rp, tp,
                                             int(bool(fast)), C.byref(ms)))
        return ms.value

    def lattice_pack(self, W0, I0, min_nsamples, out):
        """out: float32 CUDA tensor [lanes, >= 2 n]"""
        lanes = W0.shape[0]
        ms = C.c_float()
        check(self.L.rgbid_lattice_pack_batched(self._h, lanes, C.byref(imgb(W0)), C.byref(imgb(I0)), int(min_nsamples),
                                                C.c_void_p(out.data_ptr()), C.c_size_t(out.stride(0)), C.byref(ms)))
        return ms.value

    def lattice_residuals(self, Wcur, W0, Icur, I0, R_proj, t_proj, min_nsamples, res, fast=False, kf_lat=None):
        """res: float32 CUDA tensor [lanes, >= 2 n] <- (W1 - W0) | (I1 - I0) at the lattice points"""
        lanes = W0.shape[0]
        Rk, Rp = _f32(R_proj, lanes, 9); tk, tp = _f32(t_proj, lanes, 3)
        ms = C.c_float()
        check(self.L.rgbid_lattice_residuals_batched(self._h, lanes, *[C.byref(imgb(t)) for t in (Wcur, W0, Icur, I0)], Rp, tp, int(min_nsamples),
                                                     int(bool(fast)), C.c_void_p(kf_lat.data_ptr()) if kf_lat is not None else None,
                                                     C.c_size_t(kf_lat.stride(0) if kf_lat is not None else 0), C.c_void_p(res.data_ptr()),
                                                     C.c_size_t(res.stride(0)), C.byref(ms)))
        return ms.value

    def sigma_pair(self, res, n, mestimator=3):
        lanes = res.shape[0]
        out = (ScalePair * lanes)()
        ms = C.c_float()
        check(self.L.rgbid_sigma_pair_batched(self._h, lanes, C.c_void_p(res.data_ptr()), C.c_size_t(res.stride(0)), int(n), int(mestimator), out,
                                              C.byref(ms)))
        return [dict(bias_depthinv=o.bias_depthinv, sigma_depthinv=o.sigma_depthinv, nu_depthinv=o.nu_depthinv, bias_int=o.bias_int,
                     sigma_int=o.sigma_int, nu_int=o.nu_int) for o in out]

    def fuse_frame(self, cur_iD, kf_iD, kf_weight, warped_weight, R_proj, t_proj, fast=False):
        lanes = kf_iD.shape[0]
        Rk, Rp = _f32(R_proj, lanes, 9); tk, tp = _f32(t_proj, lanes, 3)
        ms = C.c_float()
        check(self.L.rgbid_fuse_frame_batched(self._h, lanes, *[C.byref(imgb(t)) for t in (cur_iD, kf_iD, kf_weight, warped_weight)], Rp, tp,
                                              int(bool(fast)), C.byref(ms)))
        return ms.value

    def kf_maps(self, intr, depthinv, vmap, nmap):
        lanes = depthinv.shape[0]
        ms = C.c_float()
        check(self.L.rgbid_kf_maps_batched(self._h, lanes, Intr(*intr), C.byref(imgb(depthinv)), C.byref(imgb(vmap)), C.byref(imgb(nmap)), C.byref(ms)))
        return ms.value

    def visibility_pair(self, a, b, R_ab, t_ab, R_ba, t_ba, fast=False):
        """[lanes, 4] = visible a->b, valid a, visible b->a, valid b"""
        lanes = a.shape[0]
        k1, p1 = _f32(R_ab, lanes, 9); k2, p2 = _f32(t_ab, lanes, 3); k3, p3 = _f32(R_ba, lanes, 9); k4, p4 = _f32(t_ba, lanes, 3)
        counts = np.zeros((lanes, 4), np.uint32); ms = C.c_float()
        check(self.L.rgbid_visibility_pair_batched(self._h, lanes, C.byref(imgb(a)), C.byref(imgb(b)), p1, p2, p3, p4, int(bool(fast)),
                                                   counts.ctypes.data_as(C.c_void_p), C.byref(ms)))
        return counts

    def prep_frame(self, depth_u16, rgb, iD, I, r, g, b, factor_depth=1.0):
        lanes = iD.shape[0]
        ms = C.c_float()
        check(self.L.rgbid_prep_frame_batched(self._h, lanes, *[C.byref(imgb(t)) for t in (depth_u16, rgb, iD, I, r, g, b)], C.c_float(factor_depth),
                                              C.byref(ms)))
        return ms.value

    def pyr_down(self, src, dst):
        ms = C.c_float()
        check(self.L.rgbid_pyr_down_batched(self._h, src.shape[0], C.byref(imgb(src)), C.byref(imgb(dst)), C.byref(ms)))
        return ms.value

    def gradient(self, src, gx, gy):
        ms = C.c_float()
        check(self.L.rgbid_compute_gradient_batched(self._h, src.shape[0], C.byref(imgb(src)), C.byref(imgb(gx)), C.byref(imgb(gy)), C.byref(ms)))
        return ms.value

    def gradient_keep(self, src, gx, gy, keep):
        ms = C.c_float()
        check(self.L.rgbid_gradient_keep_batched(self._h, src.shape[0], C.byref(imgb(src)), C.byref(imgb(gx)), C.byref(imgb(gy)), C.byref(imgb(keep)), C.byref(ms)))
        return ms.value

    def bilateral(self, src, dst, sigma_floatmap, fast=False):
        ms = C.c_float()
        check(self.L.rgbid_bilateral_filter_batched(self._h, src.shape[0], C.byref(imgb(src)), C.byref(imgb(dst)), C.c_float(sigma_floatmap),
                                                    int(bool(fast)), C.byref(ms)))
        return ms.value
