"""Python binding of the batched, device-resident keyframe alignment (include/rgbid_kfalign.h): KeyframeAlign::alignKeyframes
(src/keyframe_align.cpp:115-357) for `pairs` keyframe pairs in lock-step.  Test / bench harness only."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check
from .device import Context


class KfAlign:
    def __init__(self, ctx: Context, rows, cols, max_pairs):
        self.ctx, self.rows, self.cols, self.cap = ctx, int(rows), int(cols), int(max_pairs)
        self.L = _lib.lib()
        self._h = C.c_void_p()
        check(self.L.rgbid_kfalign_create(C.byref(self._h), ctx._h, self.rows, self.cols, self.cap))
        ctx._dependents.add(self)

    def close(self):
        if self._h:
            self.L.rgbid_kfalign_destroy(self._h)
            self._h = None
            self.ctx._dependents.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def align(self, iD_ini, grey_ini, iD_end, grey_end, K, R0=None, t0=None):
        """iD_*: float32 [pairs, rows, cols], grey_*: uint8 [pairs, rows, cols] -- CUDA tensors (device entry point) or numpy arrays (host entry point);
        K: [pairs, 4]; returns (R [pairs, 3, 3], t [pairs, 3], cov [pairs, 6, 6])"""
        n = int(iD_ini.shape[0])
        K = np.ascontiguousarray(np.broadcast_to(np.asarray(K, np.float32), (n, 4)))
        R = np.ascontiguousarray(np.broadcast_to(np.eye(3), (n, 3, 3)) if R0 is None else np.asarray(R0, np.float64).reshape(n, 3, 3)).copy()
        t = np.zeros((n, 3)) if t0 is None else np.ascontiguousarray(np.asarray(t0, np.float64).reshape(n, 3)).copy()
        cov = np.zeros((n, 6, 6))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        if isinstance(iD_ini, torch.Tensor):
            ts = [iD_ini.contiguous(), grey_ini.contiguous(), iD_end.contiguous(), grey_end.contiguous()]
            assert ts[0].dtype == torch.float32 and ts[1].dtype == torch.uint8 and all(x.is_cuda for x in ts)
            check(self.L.rgbid_kfalign_batched(self._h, n, *[C.c_void_p(x.data_ptr()) for x in ts], p(K), p(R), p(t), p(cov)))
        else:
            hs = [np.ascontiguousarray(iD_ini, np.float32), np.ascontiguousarray(grey_ini, np.uint8), np.ascontiguousarray(iD_end, np.float32), np.ascontiguousarray(grey_end, np.uint8)]
            check(self.L.rgbid_kfalign_batched_host(self._h, n, *[p(x) for x in hs], p(K), p(R), p(t), p(cov)))
        return R, t, cov

    def launches(self):
        return int(self.L.rgbid_kfalign_launches(self._h))
