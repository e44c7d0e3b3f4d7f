"""Python mirror of the reference bridge API `RGBID_SLAM::device::*` (src/internal.h:187-453) over the
C-ABI.  Function names and argument meaning follow the reference so the parity tests read like calls
into the original library; images are torch CUDA tensors (torch is only the device-memory allocator and
stream provider -- every operation below runs in the hand-written HIP kernels of csrc/).

A 2-D float32 tensor [rows, cols] is a DeviceArray2D<float>; its row stride is the pitch (tests also use
padded views to exercise step != cols*4).  u16 depth is an int16/uint16 tensor, RGB a uint8 [rows, cols, 3].
"""
import ctypes as C
import weakref

import torch

from . import _lib
from ._lib import Img, Intr, check

LSQ, HUBER, TUKEY, STUDENT = range(4)
NO_MM, CONSTANT_VELOCITY = range(2)
SIGMA_MAD, SIGMA_PDF, SIGMA_CONS = range(3)
INDEPENDENT, MIN_WEIGHT, GEOM_ONLY, PHOT_ONLY = range(4)
WARP_FIRST, PYR_FIRST = range(2)
CHI_SQUARED, ALL_ITERS = range(2)
NO_FILTERS, FILTER_GRADS = range(2)
INTERP_EXACT, INTERP_TEX8 = range(2)


class IntrK(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "k3", "k4", "k5")]


class DepthDist(C.Structure):
    _fields_ = [("c1", C.c_float), ("c0", C.c_float), ("q0", C.c_float * 9), ("q1", C.c_float * 9), ("xshift", C.c_int), ("yshift", C.c_int)]


def depth_dist(c1=1.0, c0=0.0, q0=(0,) * 9, q1=(1,) + (0,) * 8, xshift=4, yshift=4):
    """DepthDist with the constructor defaults of src/internal.h:150-159"""
    return DepthDist(c1, c0, (C.c_float * 9)(*q0), (C.c_float * 9)(*q1), xshift, yshift)


def img(t):
    """rgbid_img view of a CUDA tensor: [rows, cols] (any 1/2/4-byte dtype) or [rows, cols, 3] uint8."""
    if not t.is_cuda:
        raise _lib.RgbidError("rgbid images must live in device memory (no CPU path)")
    if t.dim() == 3:
        assert t.shape[2] == 3 and t.dtype == torch.uint8 and t.stride(2) == 1 and t.stride(1) == 3
    else:
        assert t.dim() == 2 and t.stride(1) == 1
    return Img(t.data_ptr(), t.stride(0) * t.element_size(), t.shape[0], t.shape[1])


def _fa(vals, n):
    arr = (C.c_float * n)(*[float(v) for v in vals])
    return arr


class Context:
    """rgbid_ctx bound to a device and (by default) torch's current stream on that device."""

    def __init__(self, device=0, use_torch_stream=True):
        # use_torch_stream: adopt torch's CURRENT stream when it is a real stream object; torch's default stream is the null
        # stream (handle 0), in which case the context creates its own non-blocking stream -- order against it with
        # Context.sync() / Engine.records(), or create the Context inside `with torch.cuda.stream(s):` to share stream s.
        L = _lib.lib()
        self._h = C.c_void_p()
        stream = None
        if use_torch_stream:
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        check(L.rgbid_ctx_create(C.byref(self._h), int(device), stream))
        self.device = device
        self.L = L
        self._dependents = weakref.WeakSet()   # engines created on this context: they must go before the stream does

    def close(self):
        if self._h:
            for d in list(self._dependents):
                d.close()
            self.L.rgbid_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_async(self, on):
        check(self.L.rgbid_ctx_set_async(self._h, int(on)))

    def set_interp_mode(self, mode):
        check(self.L.rgbid_ctx_set_interp_mode(self._h, int(mode)))

    def set_numerics(self, fast):
        check(self.L.rgbid_ctx_set_numerics(self._h, 1 if fast else 0))

    def sync(self):
        check(self.L.rgbid_ctx_sync(self._h))

    def selftest_rcp(self):
        """mismatches of the kernels' exact reciprocal vs IEEE 1.0f/x over all 2^32 inputs (must be 0)"""
        n = C.c_ulonglong(1)
        check(self.L.rgbid_selftest_rcp(self._h, C.byref(n)))
        return n.value

    def selftest_cvt_flr(self, stride=1):
        n = C.c_ulonglong()
        check(self.L.rgbid_selftest_cvt_flr(self._h, C.c_uint(int(stride)), C.byref(n)))
        return n.value

    def selftest_fast_primitives(self):
        """v_rcp_f32 within 1 ulp, v_med3_f32 as the domain clamp, v_fract_f32 == x - floor(x): the hardware facts under csrc/guard_band.h (must be 0)"""
        n = C.c_ulonglong(1)
        check(self.L.rgbid_selftest_fast_primitives(self._h, C.byref(n)))
        return n.value

    def selftest_div_const(self, divisor):
        """(mismatches of the bilateral filter's short division by `divisor` vs IEEE over all 2^32 dividends, whether the filter uses it)"""
        n, used = C.c_ulonglong(1), C.c_int(0)
        check(self.L.rgbid_selftest_div_const(self._h, C.c_float(divisor), C.byref(n), C.byref(used)))
        return n.value, bool(used.value)

    def stream_handle(self):
        s = C.c_void_p()
        check(self.L.rgbid_ctx_get_stream(self._h, C.byref(s)))
        return s.value or 0

    def wait_torch_stream(self, stream=None):
        """order this context's stream after everything enqueued so far on a torch stream (default: torch's current stream)"""
        stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        if (stream.cuda_stream or 0) == self.stream_handle():
            return
        ev = torch.cuda.Event()
        ev.record(stream)
        check(self.L.rgbid_ctx_wait_event(self._h, C.c_void_p(ev.cuda_event)))

    # ---- frame preparation (src/cuda/misc.cu) ----
    def convertDepth2InvDepth(self, src, dst, factor_depth=1.0):
        check(self.L.rgbid_depth_to_invdepth(self._h, C.byref(img(src)), C.byref(img(dst)), C.c_float(factor_depth)))

    def computeIntensity(self, rgb, dst):
        check(self.L.rgbid_compute_intensity(self._h, C.byref(img(rgb)), C.byref(img(dst))))

    def decomposeRGBInChannels(self, rgb, r, g, b):
        check(self.L.rgbid_decompose_rgb(self._h, C.byref(img(rgb)), C.byref(img(r)), C.byref(img(g)), C.byref(img(b))))

    def computeGradient(self, src, dst_hor, dst_vert):
        ms = C.c_float()
        check(self.L.rgbid_compute_gradient(self._h, C.byref(img(src)), C.byref(img(dst_hor)), C.byref(img(dst_vert)), C.byref(ms)))
        return ms.value

    computeGradientIntensity = computeGradient
    computeGradientDepth = computeGradient

    def copyImages(self, src_depth, src_int, dst_depth, dst_int):
        check(self.L.rgbid_copy_images(self._h, C.byref(img(src_depth)), C.byref(img(src_int)), C.byref(img(dst_depth)), C.byref(img(dst_int))))

    def copyImage(self, src, dst):
        check(self.L.rgbid_copy_image(self._h, C.byref(img(src)), C.byref(img(dst))))

    def copyImageRGB(self, src, dst):
        check(self.L.rgbid_copy_image_rgb(self._h, C.byref(img(src)), C.byref(img(dst))))

    def initialiseWeightKeyframe(self, src_depth, dst_weight):
        check(self.L.rgbid_init_weight_keyframe(self._h, C.byref(img(src_depth)), C.byref(img(dst_weight))))

    def initialiseDeviceMemory2D(self, t, bits):
        check(self.L.rgbid_fill_2d(self._h, C.byref(img(t)), t.element_size(), C.c_uint32(bits)))

    # ---- pyramid / filter ----
    def pyrDown(self, src, dst):
        ms = C.c_float()
        check(self.L.rgbid_pyr_down(self._h, C.byref(img(src)), C.byref(img(dst)), C.byref(ms)))
        return ms.value

    pyrDownIntensity = pyrDown
    pyrDownDepth = pyrDown

    def bilateralFilter(self, src, dst, sigma_floatmap):
        ms = C.c_float()
        check(self.L.rgbid_bilateral_filter(self._h, C.byref(img(src)), C.byref(img(dst)), C.c_float(sigma_floatmap), C.byref(ms)))
        return ms.value

    # ---- bridge functions the reference defines but no longer calls ----
    def convertDepth2Float(self, src, dst):
        check(self.L.rgbid_depth_to_float(self._h, C.byref(img(src)), C.byref(img(dst))))

    def convertFloat2RGB(self, src, dst):
        check(self.L.rgbid_float_to_rgb(self._h, C.byref(img(src)), C.byref(img(dst))))

    def createNMap(self, vmap, nmap):
        check(self.L.rgbid_create_nmap(self._h, C.byref(img(vmap)), C.byref(img(nmap))))

    def integrateWarpedRGB(self, warped, r, g, b, warped_weight, depth_dst, colors_dst, weight_dst):
        ms = C.c_float()
        check(self.L.rgbid_integrate_warped_rgb(self._h, C.byref(img(warped)), C.byref(img(r)), C.byref(img(g)), C.byref(img(b)),
                                                C.byref(img(warped_weight)), C.byref(img(depth_dst)), C.byref(img(colors_dst)),
                                                C.byref(img(weight_dst)), C.byref(ms)))
        return ms.value

    # ---- custom-calibration front-end (undistortion.cu, warping_registration.cu:720-822) ----
    def undistortIntensity(self, src, dst, intr_k):
        ms = C.c_float()
        k = IntrK(*[float(v) for v in intr_k])
        check(self.L.rgbid_undistort_intensity(self._h, C.byref(img(src)), C.byref(img(dst)), C.byref(k), C.byref(ms)))
        return ms.value

    def undistortDepthInv(self, src, src_corr, dst, intr_k, depth_dist):
        ms = C.c_float()
        k = IntrK(*[float(v) for v in intr_k])
        check(self.L.rgbid_undistort_depthinv(self._h, C.byref(img(src)), C.byref(img(src_corr)), C.byref(img(dst)), C.byref(k),
                                              C.byref(depth_dist), C.byref(ms)))
        return ms.value

    def registerDepthinv(self, src, intermediate, intermediate_as_int, dst, dRc_proj, t_dc_proj, cRd_proj):
        ms = C.c_float()
        check(self.L.rgbid_register_depthinv(self._h, C.byref(img(src)), C.byref(img(intermediate)), C.byref(img(intermediate_as_int)),
                                             C.byref(img(dst)), _fa(dRc_proj, 9), _fa(t_dc_proj, 3), _fa(cRd_proj, 9), C.byref(ms)))
        return ms.value

    # ---- warps / fusion / visibility ----
    def warpInvDepthWithTrafo3D(self, src, dst, depthinv_prev, R_proj, t_proj):
        ms = C.c_float()
        check(self.L.rgbid_warp_invdepth(self._h, C.byref(img(src)), C.byref(img(dst)), C.byref(img(depthinv_prev)),
                                         _fa(R_proj, 9), _fa(t_proj, 3), C.byref(ms)))
        return ms.value

    def warpIntensityWithTrafo3DInvDepth(self, src, dst, depthinv_prev, R_proj, t_proj):
        ms = C.c_float()
        check(self.L.rgbid_warp_intensity(self._h, C.byref(img(src)), C.byref(img(dst)), C.byref(img(depthinv_prev)),
                                          _fa(R_proj, 9), _fa(t_proj, 3), C.byref(ms)))
        return ms.value

    def warpPair(self, src_iD, src_I, grid_iD, dst_iD, dst_I, R_proj, t_proj, fast=False):
        ms = C.c_float()
        check(self.L.rgbid_warp_pair(self._h, C.byref(img(src_iD)), C.byref(img(src_I)), C.byref(img(grid_iD)), C.byref(img(dst_iD)), C.byref(img(dst_I)),
                                     _fa(R_proj, 9), _fa(t_proj, 3), int(bool(fast)), C.byref(ms)))
        return ms.value

    def warpInvDepthWithTrafo3DWeighted(self, src, dst, depthinv_prev, weight_warped, R_proj, t_proj):
        ms = C.c_float()
        check(self.L.rgbid_warp_invdepth_weighted(self._h, C.byref(img(src)), C.byref(img(dst)), C.byref(img(depthinv_prev)),
                                                  C.byref(img(weight_warped)), _fa(R_proj, 9), _fa(t_proj, 3), C.byref(ms)))
        return ms.value

    def integrateWarpedFrame(self, warped_depth, warped_weight, depth_dst, weight_dst):
        ms = C.c_float()
        check(self.L.rgbid_integrate_warped_frame(self._h, C.byref(img(warped_depth)), C.byref(img(warped_weight)),
                                                  C.byref(img(depth_dst)), C.byref(img(weight_dst)), C.byref(ms)))
        return ms.value

    def getVisibilityRatio(self, depth_src, depth_dst, R_proj, t_proj, overlap_mask=None):
        ratio, ms = C.c_float(), C.c_float()
        m = C.byref(img(overlap_mask)) if overlap_mask is not None else None
        check(self.L.rgbid_visibility_ratio(self._h, C.byref(img(depth_src)), C.byref(img(depth_dst)), _fa(R_proj, 9), _fa(t_proj, 3),
                                            m, C.byref(ratio), C.byref(ms)))
        return ratio.value

    getVisibilityRatioWithOverlapMask = getVisibilityRatio

    # ---- maps ----
    def createVMap(self, intr, depthinv, vmap):
        check(self.L.rgbid_create_vmap(self._h, Intr(*intr), C.byref(img(depthinv)), C.byref(img(vmap))))

    def createNMapGradients(self, intr, depthinv, gx, gy, nmap):
        check(self.L.rgbid_create_nmap_gradients(self._h, Intr(*intr), C.byref(img(depthinv)), C.byref(img(gx)), C.byref(img(gy)), C.byref(img(nmap))))

    def generateImageRGB(self, vmap, nmap, rgb, light, dst):
        r = C.byref(img(rgb)) if rgb is not None else None
        check(self.L.rgbid_generate_image(self._h, C.byref(img(vmap)), C.byref(img(nmap)), r, _fa(light, 3), C.byref(img(dst))))

    # ---- residual lattice + sigma / nu ----
    def computeErrorGridStride(self, im1, im0, error, Nsamples=9999999):
        """error: 1-D float32 CUDA tensor with room for rows*cols samples; returns the number of samples written."""
        n, ms = C.c_int(), C.c_float()
        check(self.L.rgbid_compute_error(self._h, C.byref(img(im1)), C.byref(img(im0)), C.c_void_p(error.data_ptr()), int(Nsamples),
                                         C.byref(n), C.byref(ms)))
        return n.value

    def computeSigmaAndNuStudent(self, error, n, bias, sigma, nu, Mestimator=STUDENT):
        b, s, v, ms = C.c_float(bias), C.c_float(sigma), C.c_float(nu), C.c_float()
        check(self.L.rgbid_sigma_nu_student(self._h, C.c_void_p(error.data_ptr()), int(n), C.byref(b), C.byref(s), C.byref(v), int(Mestimator), C.byref(ms)))
        return b.value, s.value, v.value

    def computeNuStudent(self, error, n, bias, sigma):
        v, ms = C.c_float(0), C.c_float()
        check(self.L.rgbid_nu_student(self._h, C.c_void_p(error.data_ptr()), int(n), C.c_float(bias), C.c_float(sigma), C.byref(v), C.byref(ms)))
        return v.value

    def computeSigmaPdf(self, error, n, bias, sigma, Mestimator=STUDENT):
        b, s, ms = C.c_float(bias), C.c_float(sigma), C.c_float()
        check(self.L.rgbid_sigma_pdf(self._h, C.c_void_p(error.data_ptr()), int(n), C.byref(b), C.byref(s), int(Mestimator), C.byref(ms)))
        return b.value, s.value

    def computeChiSquare(self, error_int, error_depth, n, sigma_int, sigma_depth, Mestimator=STUDENT):
        x, t, d, ms = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        check(self.L.rgbid_chi_square(self._h, C.c_void_p(error_int.data_ptr()), C.c_void_p(error_depth.data_ptr()), int(n),
                                      C.c_float(sigma_int), C.c_float(sigma_depth), int(Mestimator), C.byref(x), C.byref(t), C.byref(d), C.byref(ms)))
        return x.value, t.value, d.value

    # ---- normal equations ----
    def buildSystemGridStride(self, W0, I0, gradW0_x, gradW0_y, gradI0_x, gradI0_y, W1, I1, Mestimator, weighting,
                              sigma_depthinv, sigma_int, bias_depthinv, bias_int, intr, return_ms=False):
        import numpy as np
        A = np.zeros(36); b = np.zeros(6); ms = C.c_float()
        maps = [C.byref(img(t)) for t in (W0, I0, gradW0_x, gradW0_y, gradI0_x, gradI0_y, W1, I1)]
        check(self.L.rgbid_build_system(self._h, *maps, int(Mestimator), int(weighting), C.c_float(sigma_depthinv), C.c_float(sigma_int),
                                        C.c_float(bias_depthinv), C.c_float(bias_int), Intr(*intr),
                                        A.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.byref(ms)))
        return (A.reshape(6, 6), b, ms.value) if return_ms else (A.reshape(6, 6), b)

    def buildSystemStudentNuGridStride(self, W0, I0, gradW0_x, gradW0_y, gradI0_x, gradI0_y, W1, I1, Mestimator, weighting,
                                       sigma_depthinv, sigma_int, bias_depthinv, bias_int, nu_depthinv, nu_int, intr, return_ms=False):
        import numpy as np
        A = np.zeros(36); b = np.zeros(6); ms = C.c_float()
        maps = [C.byref(img(t)) for t in (W0, I0, gradW0_x, gradW0_y, gradI0_x, gradI0_y, W1, I1)]
        check(self.L.rgbid_build_system_student_nu(self._h, *maps, int(Mestimator), int(weighting), C.c_float(sigma_depthinv),
                                                   C.c_float(sigma_int), C.c_float(bias_depthinv), C.c_float(bias_int),
                                                   C.c_float(nu_depthinv), C.c_float(nu_int), Intr(*intr),
                                                   A.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.byref(ms)))
        return (A.reshape(6, 6), b, ms.value) if return_ms else (A.reshape(6, 6), b)


def error_lattice_size(rows, cols, min_nsamples):
    n, lr, lc, st = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    check(_lib.lib().rgbid_error_lattice_size(int(rows), int(cols), int(min_nsamples), C.byref(n), C.byref(lr), C.byref(lc), C.byref(st)))
    return n.value, lr.value, lc.value, st.value


def fast_guard(R_proj, t_proj, cols, rows):
    """guard-band constants of one projection (csrc/guard_band.h, host evaluation; no GPU): dict d1, c2, d2, q0, q1, db, g0, g1, e0, e1, zsafe + the lane-constant forms bL, cL, kL, wcore"""
    from ._lib import lib
    out = (C.c_float * 10)()
    z = C.c_int()
    check(lib().rgbid_fast_guard(_fa(R_proj, 9), _fa(t_proj, 3), int(cols), int(rows), out, C.byref(z)))
    d = dict(zip(("d1", "c2", "d2", "q0", "q1", "db", "g0", "g1", "e0", "e1"), [float(v) for v in out]))
    d["zsafe"] = int(z.value)
    out4 = (C.c_float * 4)()
    check(lib().rgbid_fast_guard_lane(_fa(R_proj, 9), _fa(t_proj, 3), int(cols), int(rows), out4))
    d.update(zip(("bL", "cL", "kL", "wcore"), [float(v) for v in out4]))
    return d
