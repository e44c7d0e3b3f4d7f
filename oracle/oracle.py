"""ctypes binding of the CPU oracle (oracle/librgbid_oracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package.  See oracle/rgbid_oracle.h for the
file:line citations of every function and for the "parity unpinned" statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librgbid_oracle.so")

LSQ, HUBER, TUKEY, STUDENT = range(4)
NO_MM, CONSTANT_VELOCITY = range(2)
SIGMA_MAD, SIGMA_PDF, SIGMA_CONS = range(3)
INDEPENDENT, MIN_WEIGHT, GEOM_ONLY, PHOT_ONLY = range(4)
WARP_FIRST, PYR_FIRST = range(2)
SEQ_ODO, SEQ_KF = 0, 1   # PoseConstraint types of the streams to the back-end
CHI_SQUARED, ALL_ITERS = range(2)
NO_FILTERS, FILTER_GRADS = range(2)
INTERP_EXACT, INTERP_TEX8 = range(2)


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "librgbid_oracle.so"])
    return _SO


class Intr(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class TrackerConfig(C.Structure):
    _fields_ = [
        ("rows", C.c_int), ("cols", C.c_int), ("levels", C.c_int),
        ("iters", C.c_int * 8),
        ("mestimator", C.c_int), ("motion_model", C.c_int), ("sigma_estimator", C.c_int),
        ("weighting", C.c_int), ("warping", C.c_int),
        ("max_odoKF_count", C.c_int), ("finest_level", C.c_int), ("termination", C.c_int),
        ("visratio_odo", C.c_float), ("image_filtering", C.c_int), ("visratio_integr", C.c_float),
        ("max_integrKF_count", C.c_int), ("nsamples", C.c_int),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("factor_depth", C.c_float), ("interp_mode", C.c_int), ("delta_t", C.c_float),
    ]


class FrameInfo(C.Structure):
    _fields_ = [
        ("lost", C.c_int), ("odo_kf_switched", C.c_int), ("integr_kf_switched", C.c_int),
        ("odometry_success", C.c_int),
        ("visratio_odo", C.c_float), ("visratio_integr", C.c_float),
        ("sigma_int", C.c_float), ("sigma_depthinv", C.c_float), ("nu_int", C.c_float),
        ("nu_depthinv", C.c_float), ("bias_int", C.c_float), ("bias_depthinv", C.c_float),
        ("delta_R", C.c_double * 9), ("delta_t", C.c_double * 3), ("delta_cov", C.c_double * 36),
        ("odo_kf_natural", C.c_int), ("integr_kf_natural", C.c_int),
        ("sigma_stop_margin_int", C.c_float), ("sigma_stop_margin_depthinv", C.c_float), ("sigma_stop_margin_frame", C.c_float),
        ("chi_stop_margin_frame", C.c_float), ("chi_stops_frame", C.c_int),
    ]


_lib = None
_lib_cn = None
_SO_CN = os.path.join(_HERE, "librgbid_oracle_cudanum.so")


def _restypes(L):
    L.orc_tracker_create.restype = C.c_void_p
    for f in ("kf_depthinv", "kf_weight", "kf_normals", "kf_vertices", "kf_overlap_mask", "cur_depthinv", "cur_intensity"):
        getattr(L, "orc_tracker_" + f).restype = C.c_void_p
    return L


def cpu_has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


def lib_cudanum():
    """the oracle built under the model of the reference's nvcc numerics (--ftz --prec-div=false --prec-sqrt=false, fmad, __expf):
    only the Tracker is meant to be driven through it (Tracker(cfg, numerics="cuda")).  Needs a host CPU with FMA."""
    global _lib_cn
    if _lib_cn is None:
        if not cpu_has_fma():
            raise RuntimeError("librgbid_oracle_cudanum.so is compiled with -mfma and this CPU has no FMA")
        if not os.path.exists(_SO_CN):
            subprocess.check_call(["make", "-C", _HERE, "librgbid_oracle_cudanum.so"])
        _lib_cn = _restypes(C.CDLL(_SO_CN))
        assert _lib_cn.orc_cuda_numerics() == 1
    return _lib_cn


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _restypes(_lib)
        _lib.orc_visibility_ratio.restype = C.c_float
        _lib.orc_digamma.restype = C.c_float
        _lib.orc_digamma.argtypes = [C.c_float]
        _lib.orc_tracker_create.restype = C.c_void_p
        _lib.orc_tracker_kf_depthinv.restype = C.c_void_p
        _lib.orc_tracker_kf_weight.restype = C.c_void_p
        _lib.orc_tracker_kf_normals.restype = C.c_void_p
        _lib.orc_tracker_kf_vertices.restype = C.c_void_p
        _lib.orc_tracker_kf_overlap_mask.restype = C.c_void_p
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _intr(k):
    return Intr(*[float(v) for v in k])


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def depth2invdepth(depth_u16, factor_depth=1.0):
    d = np.ascontiguousarray(depth_u16, dtype=np.uint16)
    out = np.empty(d.shape, np.float32)
    lib().orc_depth2invdepth(_p(d), _p(out), d.shape[0], d.shape[1], C.c_float(factor_depth))
    return out


def intensity(rgb_u8):
    r = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
    out = np.empty(r.shape[:2], np.float32)
    lib().orc_intensity(_p(r), _p(out), r.shape[0], r.shape[1])
    return out


def decompose_rgb(rgb_u8):
    r = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
    outs = [np.empty(r.shape[:2], np.float32) for _ in range(3)]
    lib().orc_decompose_rgb(_p(r), _p(outs[0]), _p(outs[1]), _p(outs[2]), r.shape[0], r.shape[1])
    return outs


def gradient(src):
    s = _f(src)
    gx, gy = np.empty_like(s), np.empty_like(s)
    lib().orc_gradient(_p(s), s.shape[0], s.shape[1], _p(gx), _p(gy))
    return gx, gy


def pyr_down(src):
    s = _f(src)
    out = np.empty((s.shape[0] // 2, s.shape[1] // 2), np.float32)
    lib().orc_pyr_down(_p(s), s.shape[0], s.shape[1], _p(out))
    return out


def bilateral(src, sigma_floatmap):
    s = _f(src)
    out = np.empty_like(s)
    lib().orc_bilateral(_p(s), s.shape[0], s.shape[1], C.c_float(sigma_floatmap), _p(out))
    return out


def _rt(R, t):
    return _f(np.asarray(R).reshape(9)), _f(np.asarray(t).reshape(3))


def warp_invdepth(src, grid, R, t):
    s, g = _f(src), _f(grid)
    Rf, tf = _rt(R, t)
    out = np.empty_like(g)
    lib().orc_warp_invdepth(_p(s), _p(g), g.shape[0], g.shape[1], _p(Rf), _p(tf), _p(out))
    return out


def warp_intensity(src, grid, R, t, interp_mode=INTERP_TEX8):
    s, g = _f(src), _f(grid)
    Rf, tf = _rt(R, t)
    out = np.empty_like(g)
    lib().orc_warp_intensity(_p(s), _p(g), g.shape[0], g.shape[1], _p(Rf), _p(tf), int(interp_mode), _p(out))
    return out


def warp_invdepth_weighted(src, grid, R, t, weight_init=None):
    s, g = _f(src), _f(grid)
    Rf, tf = _rt(R, t)
    out = np.empty_like(g)
    w = np.zeros_like(g) if weight_init is None else _f(weight_init).copy()
    lib().orc_warp_invdepth_weighted(_p(s), _p(g), g.shape[0], g.shape[1], _p(Rf), _p(tf), _p(out), _p(w))
    return out, w


def integrate_warped(warped, warped_weight, kf, kf_weight):
    a, b = _f(warped), _f(warped_weight)
    k, kw = _f(kf).copy(), _f(kf_weight).copy()
    lib().orc_integrate_warped(_p(a), _p(b), _p(k), _p(kw), k.shape[0], k.shape[1])
    return k, kw


def visibility_ratio(src, dst, R, t, with_mask=False, mask_init=None):
    s, d = _f(src), _f(dst)
    Rf, tf = _rt(R, t)
    mask = None
    if with_mask:
        mask = np.zeros(s.shape, np.uint8) if mask_init is None else np.ascontiguousarray(mask_init, np.uint8).copy()
    nv, nval = C.c_float(), C.c_float()
    ratio = lib().orc_visibility_ratio(_p(s), _p(d), s.shape[0], s.shape[1], _p(Rf), _p(tf),
                                       _p(mask) if mask is not None else None, C.byref(nv), C.byref(nval))
    return float(ratio), nv.value, nval.value, mask


def vmap(depthinv, k):
    s = _f(depthinv)
    out = np.zeros((3 * s.shape[0], s.shape[1]), np.float32)
    lib().orc_vmap(_p(s), s.shape[0], s.shape[1], _intr(k), _p(out))
    return out


def nmap_gradients(depthinv, gx, gy, k):
    s, a, b = _f(depthinv), _f(gx), _f(gy)
    out = np.zeros((3 * s.shape[0], s.shape[1]), np.float32)
    lib().orc_nmap_gradients(_p(s), _p(a), _p(b), s.shape[0], s.shape[1], _intr(k), _p(out))
    return out


def generate_image_rgb(vmap_, nmap_, rgb, light):
    v, n = _f(vmap_), _f(nmap_)
    r = np.ascontiguousarray(rgb, np.uint8)
    lf = _f(light)
    out = np.empty_like(r)
    lib().orc_generate_image_rgb(_p(v), _p(n), _p(r), _p(lf), r.shape[0], r.shape[1], _p(out))
    return out


def error_lattice(im1, im0, min_nsamples=9999999):
    a, b = _f(im1), _f(im0)
    err = np.empty(a.size, np.float32)
    r, c, s = C.c_int(), C.c_int(), C.c_int()
    n = lib().orc_error_lattice(_p(a), _p(b), a.shape[0], a.shape[1], int(min_nsamples), _p(err),
                                C.byref(r), C.byref(c), C.byref(s))
    return err[:n].copy(), (r.value, c.value, s.value)


def sigma_nu_student(err, bias, sigma, nu=5.0, mestimator=STUDENT):
    e = _f(err)
    b, s, n = C.c_float(bias), C.c_float(sigma), C.c_float(nu)
    lib().orc_sigma_nu_student(_p(e), e.size, C.byref(b), C.byref(s), C.byref(n), int(mestimator))
    return b.value, s.value, n.value


def nu_student(err, bias, sigma):
    e = _f(err)
    n = C.c_float(0)
    lib().orc_nu_student(_p(e), e.size, C.c_float(bias), C.c_float(sigma), C.byref(n))
    return n.value


def sigma_pdf(err, bias, sigma, mestimator=STUDENT):
    e = _f(err)
    b, s = C.c_float(bias), C.c_float(sigma)
    lib().orc_sigma_pdf(_p(e), e.size, C.byref(b), C.byref(s), int(mestimator))
    return b.value, s.value


def chi_square(err_int, err_depth, sigma_int, sigma_depth, mestimator=STUDENT):
    a, b = _f(err_int), _f(err_depth)
    x, t, n = C.c_float(), C.c_float(), C.c_float()
    lib().orc_chi_square(_p(a), _p(b), a.size, C.c_float(sigma_int), C.c_float(sigma_depth), int(mestimator),
                         C.byref(x), C.byref(t), C.byref(n))
    return x.value, t.value, n.value


def digamma(x):
    return float(lib().orc_digamma(C.c_float(x)))


def build_system(W0, I0, gW0x, gW0y, gI0x, gI0y, W1, I1, k, student_nu=True, mestimator=STUDENT,
                 weighting=INDEPENDENT, sigma_depthinv=0.0025, sigma_int=5.0, bias_depthinv=0.0,
                 bias_int=0.0, nu_depthinv=5.0, nu_int=5.0):
    arrs = [_f(a) for a in (W0, I0, gW0x, gW0y, gI0x, gI0y, W1, I1)]
    A = np.zeros(36, np.float64)
    b = np.zeros(6, np.float64)
    rows, cols = arrs[0].shape
    lib().orc_build_system(*[_p(a) for a in arrs], rows, cols, int(bool(student_nu)), int(mestimator),
                           int(weighting), C.c_float(sigma_depthinv), C.c_float(sigma_int),
                           C.c_float(bias_depthinv), C.c_float(bias_int), C.c_float(nu_depthinv),
                           C.c_float(nu_int), _intr(k), _p(A), _p(b))
    return A.reshape(6, 6), b


def _d(a, n):
    a = np.ascontiguousarray(a, np.float64).reshape(n)
    return a


def force_orthogonal(M):
    m = _d(M, 9); r = np.empty(9)
    lib().orc_force_orthogonal(_p(m), _p(r))
    return r.reshape(3, 3)


def expmap_rot(w):
    w = _d(w, 3); r = np.empty(9)
    lib().orc_expmap_rot(_p(w), _p(r))
    return r.reshape(3, 3)


def expmap(w, v):
    w, v = _d(w, 3), _d(v, 3); r = np.empty(9); t = np.empty(3)
    lib().orc_expmap(_p(w), _p(v), _p(r), _p(t))
    return r.reshape(3, 3), t


def logmap(R, t):
    R, t = _d(R, 9), _d(t, 3); tw = np.empty(6)
    lib().orc_logmap(_p(R), _p(t), _p(tw))
    return tw


def llt_solve6(A, b):
    A, b = _d(A, 36), _d(b, 6); x = np.empty(6)
    ok = lib().orc_llt_solve6(_p(A), _p(b), _p(x))
    return x, bool(ok)


def inverse6(A):
    A = _d(A, 36); out = np.empty(36)
    lib().orc_inverse6(_p(A), _p(out))
    return out.reshape(6, 6)


def default_config(**kw):
    c = TrackerConfig()
    lib().orc_tracker_default_config(C.byref(c))
    for k, v in kw.items():
        if k == "iters":
            for i, it in enumerate(v):
                c.iters[i] = int(it)
        else:
            setattr(c, k, v)
    return c


class Tracker:
    """orc_tracker: CPU restatement of VisodoTracker::trackNewFrame (src/visodo.cpp:1967-2247)."""

    def __init__(self, cfg=None, numerics="ieee", **kw):
        self.cfg = cfg if cfg is not None else default_config(**kw)
        self._L = lib() if numerics == "ieee" else lib_cudanum()
        self._h = C.c_void_p(self._L.orc_tracker_create(C.byref(self.cfg)))

    def close(self):
        if self._h and self._L is not None:
            self._L.orc_tracker_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown
            pass

    def track(self, depth_u16, rgb_u8):
        d = np.ascontiguousarray(depth_u16, np.uint16)
        r = np.ascontiguousarray(rgb_u8, np.uint8)
        return bool(self._L.orc_tracker_track(self._h, _p(d), _p(r)))

    def force_kf_decisions(self, odo_switch=-1, integr_switch=-1):
        """test hook: impose the keyframe decisions of the NEXT tracked frame (-1 natural, 0 keep, 1 switch)"""
        self._L.orc_tracker_force_kf_decisions(self._h, int(odo_switch), int(integr_switch))

    def poses(self):
        n = self._L.orc_tracker_num_poses(self._h)
        Rs, ts = np.empty((n, 9)), np.empty((n, 3))
        for i in range(n):
            self._L.orc_tracker_get_pose(self._h, i, _p(Rs[i]), _p(ts[i]))
        return Rs.reshape(n, 3, 3), ts

    def odometry(self):
        n = self._L.orc_tracker_num_odo(self._h)
        Rs, ts, cs = np.empty((n, 9)), np.empty((n, 3)), np.empty((n, 36))
        for i in range(n):
            self._L.orc_tracker_get_odo(self._h, i, _p(Rs[i]), _p(ts[i]), _p(cs[i]))
        return Rs.reshape(n, 3, 3), ts, cs.reshape(n, 6, 6)

    def last_info(self):
        info = FrameInfo()
        self._L.orc_tracker_last_info(self._h, C.byref(info))
        return info

    def _map(self, fn, shape, dtype):
        ptr = fn(self._h)
        n = int(np.prod(shape))
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()

    def set_custom_calibration(self, rgb_k, depth_k, dist, dRc, t_dc):
        """prepareImagesCustomCalibration path (custom_registration=1); dist = oracle.depth_dist(...)"""
        cc = CustomCalib(IntrK(*[float(v) for v in rgb_k]), IntrK(*[float(v) for v in depth_k]), dist,
                         (C.c_float * 9)(*[float(v) for v in np.asarray(dRc).reshape(9)]), (C.c_float * 3)(*[float(v) for v in t_dc]))
        self._L.orc_tracker_set_custom_calibration(self._h, C.byref(cc))

    # ---- streams to the back-end (f-3) ----
    def sink_poses(self):
        n = self._L.orc_tracker_num_sink_poses(self._h)
        ids = np.empty(n, np.int32); Rs = np.empty((n, 9)); ts = np.empty((n, 3))
        for i in range(n):
            v = C.c_int()
            self._L.orc_tracker_get_sink_pose(self._h, i, C.byref(v), _p(Rs[i]), _p(ts[i]))
            ids[i] = v.value
        return ids, Rs.reshape(n, 3, 3), ts

    def constraints(self):
        out = []
        for i in range(self._L.orc_tracker_num_constraints(self._h)):
            a, b, ty = C.c_int(), C.c_int(), C.c_int()
            R = np.empty(9); t = np.empty(3); cov = np.empty(36)
            self._L.orc_tracker_get_constraint(self._h, i, C.byref(a), C.byref(b), C.byref(ty), _p(R), _p(t), _p(cov))
            out.append(dict(ini=a.value, end=b.value, type=ty.value, R=R.reshape(3, 3), t=t, cov=cov.reshape(6, 6)))
        return out

    def num_keyframes(self):
        return self._L.orc_tracker_num_keyframes(self._h)

    def keyframe(self, i):
        rows, cols = self.cfg.rows, self.cfg.cols
        v = C.c_int(); R = np.empty(9); t = np.empty(3); Rr = np.empty(9); tr = np.empty(3)
        pm, pc, pd, pn = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._L.orc_tracker_get_keyframe(self._h, int(i), C.byref(v), _p(R), _p(t), _p(Rr), _p(tr), C.byref(pm), C.byref(pc), C.byref(pd), C.byref(pn))
        n = rows * cols
        grab = lambda ptr, ct, cnt, shape, dt: np.frombuffer((ct * cnt).from_address(ptr.value), dt).reshape(shape).copy()
        return dict(id=v.value, R=R.reshape(3, 3), t=t, R_rel=Rr.reshape(3, 3), t_rel=tr,
                    overlap_mask=grab(pm, C.c_uint8, n, (rows, cols), np.uint8), colors=grab(pc, C.c_uint8, 3 * n, (rows, cols, 3), np.uint8),
                    depthinv=grab(pd, C.c_float, n, (rows, cols), np.float32), normals=grab(pn, C.c_float, 3 * n, (3, rows, cols), np.float32))

    def cur_depthinv(self):
        return self._map(self._L.orc_tracker_cur_depthinv, (self.cfg.rows, self.cfg.cols), np.float32)

    def cur_intensity(self):
        return self._map(self._L.orc_tracker_cur_intensity, (self.cfg.rows, self.cfg.cols), np.float32)

    def kf_depthinv(self):
        return self._map(self._L.orc_tracker_kf_depthinv, (self.cfg.rows, self.cfg.cols), np.float32)

    def kf_weight(self):
        return self._map(self._L.orc_tracker_kf_weight, (self.cfg.rows, self.cfg.cols), np.float32)

    def kf_normals(self):
        return self._map(self._L.orc_tracker_kf_normals, (3 * self.cfg.rows, self.cfg.cols), np.float32)

    def kf_vertices(self):
        return self._map(self._L.orc_tracker_kf_vertices, (3 * self.cfg.rows, self.cfg.cols), np.float32)

    def kf_overlap_mask(self):
        return self._map(self._L.orc_tracker_kf_overlap_mask, (self.cfg.rows, self.cfg.cols), np.uint8)


def align_pair(cfg, depth0, rgb0, depth1, rgb1, R0=None, t0=None):
    R = np.eye(3).reshape(9).copy() if R0 is None else _d(R0, 9).copy()
    t = np.zeros(3) if t0 is None else _d(t0, 3).copy()
    cov = np.zeros(36)
    d0 = np.ascontiguousarray(depth0, np.uint16); r0 = np.ascontiguousarray(rgb0, np.uint8)
    d1 = np.ascontiguousarray(depth1, np.uint16); r1 = np.ascontiguousarray(rgb1, np.uint8)
    ok = lib().orc_align_pair(C.byref(cfg), _p(d0), _p(r0), _p(d1), _p(r1), _p(R), _p(t), _p(cov))
    return bool(ok), R.reshape(3, 3), t, cov.reshape(6, 6)


def keyframe_align(depthinv_ini, grey_ini, depthinv_end, grey_end, k, interp_mode=INTERP_TEX8, R0=None, t0=None):
    a, b = _f(depthinv_ini), _f(depthinv_end)
    ga = np.ascontiguousarray(grey_ini, np.uint8); gb = np.ascontiguousarray(grey_end, np.uint8)
    R = np.eye(3).reshape(9).copy() if R0 is None else _d(R0, 9).copy()
    t = np.zeros(3) if t0 is None else _d(t0, 3).copy()
    cov = np.zeros(36)
    lib().orc_keyframe_align(a.shape[0], a.shape[1], _p(a), _p(ga), _p(b), _p(gb), _intr(k), int(interp_mode), _p(R), _p(t), _p(cov))
    return R.reshape(3, 3), t, cov.reshape(6, 6)


# ---- custom-calibration front-end
class IntrK(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "k3", "k4", "k5")]


class DepthDist(C.Structure):
    _fields_ = [("c1", C.c_float), ("c0", C.c_float), ("q0", C.c_float * 9), ("q1", C.c_float * 9), ("xshift", C.c_int), ("yshift", C.c_int)]


class CustomCalib(C.Structure):
    _fields_ = [("rgb", IntrK), ("depth", IntrK), ("dist", DepthDist), ("dRc", C.c_float * 9), ("t_dc", C.c_float * 3)]


def depth_dist(c1=1.0, c0=0.0, q0=(0,) * 9, q1=(1,) + (0,) * 8, xshift=4, yshift=4):
    return DepthDist(c1, c0, (C.c_float * 9)(*q0), (C.c_float * 9)(*q1), xshift, yshift)


def undistort_intensity(src, k, interp_mode=INTERP_TEX8):
    s = _f(src); out = np.empty_like(s)
    lib().orc_undistort_intensity(_p(s), s.shape[0], s.shape[1], IntrK(*[float(v) for v in k]), int(interp_mode), _p(out))
    return out


def undistort_depthinv(src, k, dd):
    s = _f(src); corr = np.empty_like(s); out = np.empty_like(s)
    lib().orc_undistort_depthinv(_p(s), s.shape[0], s.shape[1], IntrK(*[float(v) for v in k]), dd, _p(corr), _p(out))
    return corr, out


def register_depthinv(src, dRc_proj, t_dc_proj, cRd_proj, scale=3):
    s = _f(src)
    inter = np.empty((scale * s.shape[0], scale * s.shape[1]), np.float32); out = np.empty_like(s)
    a, b, c = _f(np.asarray(dRc_proj).reshape(9)), _f(np.asarray(t_dc_proj).reshape(3)), _f(np.asarray(cRd_proj).reshape(9))
    lib().orc_register_depthinv(_p(s), s.shape[0], s.shape[1], inter.shape[0], inter.shape[1], _p(a), _p(b), _p(c), _p(inter), _p(out))
    return inter, out


# ---- bridge functions the reference defines but does not call any more
def nmap_cross(vmap_):
    v = _f(vmap_); out = np.zeros_like(v)
    lib().orc_nmap_cross(_p(v), v.shape[0] // 3, v.shape[1], _p(out))
    return out


def integrate_warped_rgb(warped, r, g, b, warped_weight, kf, colors, kf_weight):
    a, rr, gg, bb, w = _f(warped), _f(r), _f(g), _f(b), _f(warped_weight)
    k, kw = _f(kf).copy(), _f(kf_weight).copy()
    c = np.ascontiguousarray(colors, np.uint8).copy()
    lib().orc_integrate_warped_rgb(_p(a), _p(rr), _p(gg), _p(bb), _p(w), _p(k), _p(c), _p(kw), k.shape[0], k.shape[1])
    return k, c, kw


def depth2float(depth_u16):
    d = np.ascontiguousarray(depth_u16, np.uint16); out = np.empty(d.shape, np.float32)
    lib().orc_depth2float(_p(d), _p(out), d.shape[0], d.shape[1])
    return out


def float2rgb(src):
    s_ = _f(src); out = np.empty(s_.shape + (3,), np.uint8)
    lib().orc_float2rgb(_p(s_), _p(out), s_.shape[0], s_.shape[1])
    return out


def init_weight(src_depth):
    s = _f(src_depth); out = np.empty_like(s)
    lib().orc_init_weight(_p(s), _p(out), s.shape[0], s.shape[1])
    return out
