"""Python binding of the batched device-resident tracker (include/rgbid_engine.h).

`Engine` runs VisodoTracker::trackNewFrame (src/visodo.cpp:1967-2247) for `lanes` independent trackers per step;
inputs are CUDA tensors depth [lanes, rows, cols] (int16/uint16 bits, millimetres) and rgb [lanes, rows, cols, 3]
uint8.  Results come back as pose records (numpy structured array mirroring rgbid_pose_record).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Img, check
from .device import Context, DepthDist, IntrK

ST_TRACKED, ST_LOST, ST_ODO_KF, ST_INTEGR_KF, ST_FIRST, ST_KF_EXPORTED = 1, 2, 4, 8, 16, 32


class EngineConfig(C.Structure):
    _fields_ = [
        ("rows", C.c_int), ("cols", C.c_int), ("levels", C.c_int), ("lanes", C.c_int),
        ("iters", C.c_int * 8),
        ("mestimator", C.c_int), ("motion_model", C.c_int), ("sigma_estimator", C.c_int), ("weighting", C.c_int),
        ("max_odoKF_count", C.c_int), ("finest_level", C.c_int), ("image_filtering", C.c_int),
        ("visratio_odo", C.c_float), ("visratio_integr", C.c_float),
        ("max_integrKF_count", C.c_int), ("nsamples", C.c_int),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("factor_depth", C.c_float),
        ("interp_mode", C.c_int), ("delta_t", C.c_float),
        ("use_graph", C.c_int), ("fused_gn", C.c_int), ("chi_square_stats", C.c_int), ("preview", C.c_int),
        ("record_capacity", C.c_int), ("warping", C.c_int), ("keyframe_capacity", C.c_int), ("fast_numerics", C.c_int),
        ("defer_keyframe_maps", C.c_int),
        ("termination", C.c_int), ("custom_registration", C.c_int), ("rgb_dist", C.c_float * 5),
        ("depth_intr", IntrK), ("depth_dist", DepthDist),
        ("dRc_proj", C.c_float * 9), ("t_dc_proj", C.c_float * 3), ("cRd_proj", C.c_float * 9),
    ]


class KeyframeHeader(C.Structure):
    _fields_ = [("id", C.c_int), ("end_id", C.c_int), ("lane", C.c_int), ("seq", C.c_int), ("R", C.c_double * 9), ("t", C.c_double * 3),
                ("R_rel", C.c_double * 9), ("t_rel", C.c_double * 3), ("cov_rel", C.c_double * 36)]


RECORD_DTYPE = np.dtype([
    ("frame", np.int32), ("status", np.int32), ("vis_odo", np.float32), ("vis_integr", np.float32),
    ("sigma_int", np.float32), ("sigma_depthinv", np.float32), ("nu_int", np.float32), ("nu_depthinv", np.float32),
    ("R", np.float64, (3, 3)), ("t", np.float64, (3,)),
    ("odo_R", np.float64, (3, 3)), ("odo_t", np.float64, (3,)), ("odo_cov", np.float64, (6, 6)),
    ("kf_R", np.float64, (3, 3)), ("kf_t", np.float64, (3,)), ("kf_cov", np.float64, (6, 6)),
], align=True)


def default_config(**kw):
    c = EngineConfig()
    _lib.lib().rgbid_engine_default_config(C.byref(c))
    for k, v in kw.items():
        if k == "iters":
            for i in range(8):
                c.iters[i] = int(v[i]) if i < len(v) else 0
        elif k == "K":
            c.fx, c.fy, c.cx, c.cy = [float(x) for x in v]
        else:
            setattr(c, k, v)
    return c


class Engine:
    def __init__(self, ctx: Context, cfg: EngineConfig = None, **kw):
        self.ctx = ctx
        self.cfg = cfg if cfg is not None else default_config(**kw)
        self.L = _lib.lib()
        self.L.rgbid_engine_config_size.restype = C.c_size_t
        if self.L.rgbid_engine_config_size() != C.sizeof(EngineConfig):
            raise _lib.RgbidError(f"rgbid_engine_config: the library has {self.L.rgbid_engine_config_size()} bytes, this binding {C.sizeof(EngineConfig)} (rebuild / update rgbid/engine.py)")
        self._h = C.c_void_p()
        check(self.L.rgbid_engine_create(C.byref(self._h), ctx._h, C.byref(self.cfg)))
        self._inflight = []
        ctx._dependents.add(self)
        assert RECORD_DTYPE.itemsize == 4 * 8 + 8 * (9 + 3 + 9 + 3 + 36 + 9 + 3 + 36), RECORD_DTYPE.itemsize

    @property
    def lanes(self):
        return self.cfg.lanes

    def close(self):
        if self._h:
            if self.ctx._h:   # a context that is already gone took its stream with it; its close() destroys the engines first
                self.L.rgbid_engine_destroy(self._h)
            self._h = None
            self._inflight = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(self.L.rgbid_engine_reset(self._h))

    def set_active(self, active=None):
        """lanes fed by the following steps (sequence of 0/1 per lane; None = all)"""
        if active is None:
            check(self.L.rgbid_engine_set_active(self._h, None))
        else:
            a = np.ascontiguousarray(active, np.int32)
            assert a.shape == (self.cfg.lanes,)
            check(self.L.rgbid_engine_set_active(self._h, a.ctypes.data_as(C.c_void_p)))

    def reset_lane(self, lane):
        check(self.L.rgbid_engine_reset_lane(self._h, int(lane)))

    def step(self, depth, rgb):
        """depth: CUDA tensor [lanes, rows, cols] of 16-bit ints; rgb: CUDA uint8 [lanes, rows, cols, 3] (contiguous)."""
        c = self.cfg
        assert depth.is_cuda and rgb.is_cuda and depth.is_contiguous() and rgb.is_contiguous()
        assert depth.element_size() == 2 and tuple(depth.shape) == (c.lanes, c.rows, c.cols), depth.shape
        assert rgb.dtype == torch.uint8 and tuple(rgb.shape) == (c.lanes, c.rows, c.cols, 3), rgb.shape
        # The engine consumes its inputs asynchronously on the context's stream: order that stream after the producer of the two
        # tensors (torch's current stream) and keep them alive until the next host synchronisation (records() / Context.sync()),
        # otherwise torch may recycle a temporary while the engine's staging copy is still pending.
        self.ctx.wait_torch_stream()
        self._inflight.append((depth, rgb))
        if len(self._inflight) > 64:
            self.ctx.sync()
            self._inflight.clear()
        check(self.L.rgbid_engine_step(self._h, C.c_void_p(depth.data_ptr()), C.c_void_p(rgb.data_ptr())))

    def set_delta_t(self, delta_t):
        """inter-frame time of the constant-velocity model for the following steps (also with use_graph = 1: the kernels read it through a device pointer)"""
        self.L.rgbid_engine_set_delta_t.argtypes = [C.c_void_p, C.c_float]
        check(self.L.rgbid_engine_set_delta_t(self._h, C.c_float(delta_t)))

    def steps(self):
        return self.L.rgbid_engine_steps(self._h)

    def records(self, first_step=0, n_steps=None):
        n = self.steps() - first_step if n_steps is None else n_steps
        out = np.zeros((n, self.cfg.lanes), RECORD_DTYPE)
        check(self.L.rgbid_engine_read_records(self._h, int(first_step), int(n), out.ctypes.data_as(C.c_void_p)))
        self._inflight.clear()   # read_records synchronises the stream
        return out

    def keyframe_counts(self):
        """keyframes each lane has exported for the back-end so far (cfg.keyframe_capacity > 0)"""
        out = np.zeros(self.cfg.lanes, np.int32)
        check(self.L.rgbid_engine_keyframe_counts(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def read_keyframe(self, lane, seq, images=True):
        """export `seq` of `lane`: header fields + (overlap_mask, colors, depthinv, normals) host arrays"""
        rows, cols = self.cfg.rows, self.cfg.cols
        h = KeyframeHeader()
        mask = np.empty((rows, cols), np.uint8); colors = np.empty((rows, cols, 3), np.uint8)
        iD = np.empty((rows, cols), np.float32); nrm = np.empty((3, rows, cols), np.float32)
        ptr = (lambda a: a.ctypes.data_as(C.c_void_p)) if images else (lambda a: None)
        check(self.L.rgbid_engine_read_keyframe(self._h, int(lane), int(seq), C.byref(h), ptr(mask), ptr(colors), ptr(iD), ptr(nrm)))
        out = dict(id=h.id, end_id=h.end_id, lane=h.lane, seq=h.seq, R=np.array(h.R).reshape(3, 3), t=np.array(h.t),
                   R_rel=np.array(h.R_rel).reshape(3, 3), t_rel=np.array(h.t_rel), cov_rel=np.array(h.cov_rel).reshape(6, 6))
        if images:
            out.update(overlap_mask=mask, colors=colors, depthinv=iD, normals=nrm)
        return out

    def profile_begin(self, max_launches):
        check(self.L.rgbid_engine_profile_begin(self._h, int(max_launches)))

    def profile_end(self):
        """-> (total_ms, n_launches, bytes_per_launch) of the level-0 residual + normal-equation kernel."""
        t, n, b = C.c_double(), C.c_int(), C.c_double()
        check(self.L.rgbid_engine_profile_end(self._h, C.byref(t), C.byref(n), C.byref(b)))
        return t.value, n.value, b.value

    def bytes(self):
        b = C.c_size_t()
        check(self.L.rgbid_engine_bytes(self._h, C.byref(b)))
        return b.value

    def launches_per_step(self):
        return self.L.rgbid_engine_launches_per_step(self._h)

    def step_bytes(self):
        """algorithmic HBM bytes per lane of the last step's launch list: [every tracked frame, + per odometry-KF switch, + per integration-KF
        switch, + per fused frame] (rgbid_engine_step_bytes)"""
        out = (C.c_double * 4)()
        check(self.L.rgbid_engine_step_bytes(self._h, out))
        return [float(v) for v in out]

    def preview(self, lane):
        """Host copies of a lane's preview image and keyframe colours (u8 [rows, cols, 3]); needs cfg.preview = 1."""
        imgs = [Img(), Img()]
        check(self.L.rgbid_engine_preview(self._h, int(lane), C.byref(imgs[0]), C.byref(imgs[1])))
        self.ctx.sync()
        outs = []
        for im in imgs:
            host = np.empty((im.rows, im.cols, 3), np.uint8)
            check(self.L.rgbid_memcpy2d_d2h(self.ctx._h, host.ctypes.data_as(C.c_void_p), C.c_size_t(im.cols * 3), C.c_void_p(im.data), C.c_size_t(im.step),
                                            C.c_size_t(im.cols * 3), C.c_size_t(im.rows)))
            outs.append(host)
        return outs

    def current_maps(self, lane):
        """Host copies of a lane's current-frame level-0 maps: inverse depth, intensity (rgbid_engine_current_maps)."""
        imgs = [Img() for _ in range(2)]
        check(self.L.rgbid_engine_current_maps(self._h, int(lane), *[C.byref(i) for i in imgs]))
        self.ctx.sync()
        outs = []
        for im in imgs:
            host = np.empty((im.rows, im.cols), np.float32)
            check(self.L.rgbid_memcpy2d_d2h(self.ctx._h, host.ctypes.data_as(C.c_void_p), C.c_size_t(im.cols * 4), C.c_void_p(im.data), C.c_size_t(im.step),
                                            C.c_size_t(im.cols * 4), C.c_size_t(im.rows)))
            outs.append(host)
        return outs

    def keyframe_maps(self, lane):
        """Host copies of a lane's fused keyframe maps: depthinv, weight, vmap, nmap, overlap mask."""
        imgs = [Img() for _ in range(5)]
        check(self.L.rgbid_engine_keyframe_maps(self._h, int(lane), *[C.byref(i) for i in imgs]))
        self.ctx.sync()
        outs = []
        for im, dt in zip(imgs, (np.float32, np.float32, np.float32, np.float32, np.uint8)):
            host = np.empty((im.rows, im.cols), dt)
            check(self.L.rgbid_memcpy2d_d2h(self.ctx._h, host.ctypes.data_as(C.c_void_p), C.c_size_t(im.cols * host.itemsize),
                                            C.c_void_p(im.data), C.c_size_t(im.step), C.c_size_t(im.cols * host.itemsize), C.c_size_t(im.rows)))
            outs.append(host)
        return outs
