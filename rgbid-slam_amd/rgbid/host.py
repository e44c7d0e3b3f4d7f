"""ctypes binding of librgbid_host.so (include/rgbid_host.h): the C++ VisodoTracker mirror, SE(3) helpers, INI settings."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _lib
from ._lib import check

HOST_LIB = os.path.join(os.path.dirname(_lib.LIB_PATH), "librgbid_host.so")
HOST_SRC = os.path.join(os.path.dirname(os.path.dirname(_lib.LIB_PATH)), "host")


class TrackerConfig(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("levels", C.c_int), ("iters", C.c_int * 8),
                ("mestimator", C.c_int), ("motion_model", C.c_int), ("sigma_estimator", C.c_int), ("weighting", C.c_int), ("warping", C.c_int),
                ("max_odoKF_count", C.c_int), ("finest_level", C.c_int), ("termination", C.c_int), ("visratio_odo", C.c_float),
                ("image_filtering", C.c_int), ("visratio_integr", C.c_float), ("max_integrKF_count", C.c_int), ("nsamples", C.c_int),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("factor_depth", C.c_float),
                ("interp_mode", C.c_int), ("preview", C.c_int)]


class TrackerInfo(C.Structure):
    _fields_ = [("lost", C.c_int), ("odo_kf_switched", C.c_int), ("integr_kf_switched", C.c_int), ("visratio_odo", C.c_float),
                ("visratio_integr", C.c_float), ("sigma_int", C.c_float), ("sigma_depthinv", C.c_float), ("nu_int", C.c_float), ("nu_depthinv", C.c_float)]


class KeyframeInfo(C.Structure):
    _fields_ = [("id", C.c_int), ("rows", C.c_int), ("cols", C.c_int), ("K", C.c_float * 9), ("kd", C.c_float * 5),
                ("R", C.c_double * 9), ("t", C.c_double * 3), ("R_rel", C.c_double * 9), ("t_rel", C.c_double * 3)]


SEQ_ODO, SEQ_KF = 0, 1

_h = None


def build(force=False):
    if force or not os.path.exists(HOST_LIB):
        subprocess.check_call(["make", "-C", HOST_SRC, "-j8"])
    return HOST_LIB


def lib():
    global _h
    if _h is None:
        _lib.lib()  # librgbid_hip.so first (RTLD_GLOBAL not needed: rpath $ORIGIN)
        if not os.path.exists(HOST_LIB):
            raise _lib.RgbidError(f"{HOST_LIB} is missing: run __graft_entry__.build()")
        _h = C.CDLL(HOST_LIB)
    return _h


def _d(a, n):
    return np.ascontiguousarray(a, np.float64).reshape(n)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def expmap_rot(w):
    w = _d(w, 3); R = np.empty(9)
    lib().rgbid_expmap_rot(_p(w), _p(R))
    return R.reshape(3, 3)


def expmap(w, v):
    w, v = _d(w, 3), _d(v, 3); R = np.empty(9); t = np.empty(3)
    lib().rgbid_expmap(_p(w), _p(v), _p(R), _p(t))
    return R.reshape(3, 3), t


def logmap(R, t):
    R, t = _d(R, 9), _d(t, 3); tw = np.empty(6)
    lib().rgbid_logmap(_p(R), _p(t), _p(tw))
    return tw


def force_orthogonal(M):
    M = _d(M, 9); R = np.empty(9)
    lib().rgbid_force_orthogonal(_p(M), _p(R))
    return R.reshape(3, 3)


def llt_solve6(A, b):
    A, b = _d(A, 36), _d(b, 6); x = np.empty(6)
    lib().rgbid_llt_solve6(_p(A), _p(b), _p(x))
    return x


def inverse6(A):
    A = _d(A, 36); out = np.empty(36)
    lib().rgbid_inverse6(_p(A), _p(out))
    return out.reshape(6, 6)


def settings_get(path, section, key):
    buf = C.create_string_buffer(4096)
    n = lib().rgbid_settings_get(path.encode(), section.encode(), key.encode(), buf, 4096)
    return None if n < 0 else buf.value.decode()


def default_config(**kw):
    c = TrackerConfig()
    lib().rgbid_tracker_default_config(C.byref(c))
    for k, v in kw.items():
        if k == "iters":
            for i, it in enumerate(v):
                c.iters[i] = int(it)
        else:
            setattr(c, k, v)
    return c


class Tracker:
    """RGBID_SLAM::VisodoTracker (C++, host-driven) on device `device`; frames are passed as host numpy arrays."""

    def __init__(self, cfg=None, device=0, engine_backed=None, **kw):
        """engine_backed: None = the class default (the device-resident engine, host-driven fallback decided at the first frame), True / False = chosen"""
        self.cfg = cfg if cfg is not None else default_config(**kw)
        self._h = C.c_void_p()
        check(lib().rgbid_tracker_create(C.byref(self._h), C.byref(self.cfg), int(device)))
        if engine_backed is not None:
            self.set_engine_backed(bool(engine_backed))

    def set_engine_backed(self, on):
        """VisodoTracker::setEngineBacked: each trackNewFrame is one step of a one-lane device-resident engine (bit-exact numerics class)."""
        check(lib().rgbid_tracker_set_engine_backed(self._h, int(bool(on))))

    def engine_backed(self):
        on = C.c_int()
        check(lib().rgbid_tracker_get_engine_backed(self._h, C.byref(on)))
        return bool(on.value)

    def close(self):
        if self._h:
            lib().rgbid_tracker_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(lib().rgbid_tracker_reset(self._h))

    def load_settings(self, path):
        check(lib().rgbid_tracker_load_settings(self._h, path.encode()))

    def set_async_bridge(self, on):
        check(lib().rgbid_tracker_set_async_bridge(self._h, int(bool(on))))

    def load_calibration(self, path):
        check(lib().rgbid_tracker_load_calibration(self._h, path.encode()))

    def scene_view(self):
        """(rgb u8 [rows, cols, 3], intensity f32, keyframe inverse depth f32, changed) of the last tracked frame (preview on)"""
        r, c = self.cfg.rows, self.cfg.cols
        rgb = np.empty((r, c, 3), np.uint8); i = np.empty((r, c), np.float32); d = np.empty((r, c), np.float32)
        ch = C.c_int()
        check(lib().rgbid_tracker_scene_view(self._h, rgb.ctypes.data_as(C.c_void_p), _p(i), _p(d), C.byref(ch)))
        return rgb, i, d, bool(ch.value)

    def current_maps(self):
        """level-0 (inverse depth, intensity) of the last prepared frame"""
        d = np.empty((self.cfg.rows, self.cfg.cols), np.float32); i = np.empty_like(d)
        check(lib().rgbid_tracker_current_maps(self._h, _p(d), _p(i)))
        return d, i

    # ---- the streams handed to the back-end (poses_, constraints_, buffer_keyframes_ of the reference's KeyframeManager) ----
    def collect(self, keyframe_capacity=100):
        check(lib().rgbid_tracker_collect(self._h, int(keyframe_capacity)))

    def sink_poses(self):
        n = lib().rgbid_tracker_num_sink_poses(self._h)
        ids = np.empty(n, np.int32); Rs = np.empty((n, 9)); ts = np.empty((n, 3))
        for i in range(n):
            v = C.c_int()
            check(lib().rgbid_tracker_get_sink_pose(self._h, i, C.byref(v), _p(Rs[i]), _p(ts[i])))
            ids[i] = v.value
        return ids, Rs.reshape(n, 3, 3), ts

    def set_sink_pose(self, i, R, t):
        R, t = _d(R, 9), _d(t, 3)
        check(lib().rgbid_tracker_set_sink_pose(self._h, int(i), _p(R), _p(t)))

    def constraints(self):
        """list of dicts {ini, end, type, R, t, cov}"""
        out = []
        for i in range(lib().rgbid_tracker_num_constraints(self._h)):
            a, b, ty = C.c_int(), C.c_int(), C.c_int()
            R = np.empty(9); t = np.empty(3); cov = np.empty(36)
            check(lib().rgbid_tracker_get_constraint(self._h, i, C.byref(a), C.byref(b), C.byref(ty), _p(R), _p(t), _p(cov)))
            out.append(dict(ini=a.value, end=b.value, type=ty.value, R=R.reshape(3, 3), t=t, cov=cov.reshape(6, 6)))
        return out

    def num_keyframes(self):
        return lib().rgbid_tracker_num_keyframes(self._h)

    def peek_keyframe(self, i):
        rows, cols = self.cfg.rows, self.cfg.cols
        info = KeyframeInfo()
        mask = np.empty((rows, cols), np.uint8); colors = np.empty((rows, cols, 3), np.uint8)
        iD = np.empty((rows, cols), np.float32); nrm = np.empty((3, rows, cols), np.float32)
        check(lib().rgbid_tracker_peek_keyframe(self._h, int(i), C.byref(info), _p(mask), _p(colors), _p(iD), _p(nrm)))
        assert (info.rows, info.cols) == (rows, cols)
        return dict(id=info.id, K=np.array(info.K).reshape(3, 3), kd=np.array(info.kd), R=np.array(info.R).reshape(3, 3), t=np.array(info.t),
                    R_rel=np.array(info.R_rel).reshape(3, 3), t_rel=np.array(info.t_rel), overlap_mask=mask, colors=colors, depthinv=iD, normals=nrm)

    def pop_keyframe(self):
        return lib().rgbid_tracker_pop_keyframe(self._h) == 0

    def track(self, depth_u16, rgb_u8):
        d = np.ascontiguousarray(depth_u16, np.uint16); r = np.ascontiguousarray(rgb_u8, np.uint8)
        ok = C.c_int()
        check(lib().rgbid_tracker_track(self._h, _p(d), _p(r), C.byref(ok)))
        return bool(ok.value)

    def poses(self):
        n = lib().rgbid_tracker_num_poses(self._h)
        Rs, ts = np.empty((n, 9)), np.empty((n, 3))
        for i in range(n):
            check(lib().rgbid_tracker_get_pose(self._h, i, _p(Rs[i]), _p(ts[i])))
        return Rs.reshape(n, 3, 3), ts

    def odometry(self):
        n = lib().rgbid_tracker_num_odo(self._h)
        Rs, ts, cs = np.empty((n, 9)), np.empty((n, 3)), np.empty((n, 36))
        for i in range(n):
            check(lib().rgbid_tracker_get_odo(self._h, i, _p(Rs[i]), _p(ts[i]), _p(cs[i])))
        return Rs.reshape(n, 3, 3), ts, cs.reshape(n, 6, 6)

    def last_info(self):
        info = TrackerInfo()
        check(lib().rgbid_tracker_last_info(self._h, C.byref(info)))
        return info

    def keyframe_maps(self):
        d = np.empty((self.cfg.rows, self.cfg.cols), np.float32); w = np.empty_like(d)
        check(lib().rgbid_tracker_keyframe_maps(self._h, _p(d), _p(w)))
        return d, w


def keyframe_align(depthinv_ini, grey_ini, depthinv_end, grey_end, K, R0=None, t0=None, device=0, host_driven=False):
    """KeyframeAlign::alignKeyframes on host arrays; returns (R, t, cov).  host_driven: the reference's call sequence through the bridge instead of the
    class default (the 1-pair case of the device-resident aligner, rgbid_kfalign.h)"""
    a = np.ascontiguousarray(depthinv_ini, np.float32); b = np.ascontiguousarray(depthinv_end, np.float32)
    ga = np.ascontiguousarray(grey_ini, np.uint8); gb = np.ascontiguousarray(grey_end, np.uint8)
    R = np.eye(3).reshape(9).copy() if R0 is None else _d(R0, 9).copy()
    t = np.zeros(3) if t0 is None else _d(t0, 3).copy()
    cov = np.zeros(36)
    check(lib().rgbid_keyframe_align_mode(int(device), a.shape[0], a.shape[1], _p(a), _p(ga), _p(b), _p(gb), C.c_float(K[0]), C.c_float(K[1]),
                                          C.c_float(K[2]), C.c_float(K[3]), _p(R), _p(t), _p(cov), int(bool(host_driven))))
    return R.reshape(3, 3), t, cov.reshape(6, 6)
