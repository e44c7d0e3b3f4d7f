"""TUM RGB-D / ICL-NUIM dataset playback and trajectory files (binding of the I/O part of include/rgbid_host.h;
reference: tools/evaluation.cpp:122-351,380-439).  All decoding / formatting happens in librgbid_host.so."""
import ctypes as C
import os

import numpy as np

from ._lib import check
from .host import lib, _d, _p


def png_info(path):
    r, c, ch, bd = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    check(lib().rgbid_png_info(os.fsencode(path), C.byref(r), C.byref(c), C.byref(ch), C.byref(bd)))
    return r.value, c.value, ch.value, bd.value


def read_png(path):
    rows, cols, ch, bd = png_info(path)
    a = np.empty((rows, cols, ch), np.uint16 if bd == 16 else np.uint8)
    check(lib().rgbid_png_read(os.fsencode(path), _p(a), C.c_size_t(a.nbytes)))
    return a[:, :, 0] if ch == 1 else a


def write_png(path, img):
    a = np.ascontiguousarray(img)
    if a.dtype not in (np.uint8, np.uint16):
        raise TypeError("write_png takes uint8 or uint16 samples")
    ch = 1 if a.ndim == 2 else a.shape[2]
    check(lib().rgbid_png_write(os.fsencode(path), _p(a), a.shape[0], a.shape[1], ch, 8 * a.dtype.itemsize))


def format_pose_line(stamp, R, t):
    buf = C.create_string_buffer(512)
    R = _d(R, 9); t = _d(t, 3)
    n = lib().rgbid_format_pose_line(C.c_double(stamp), _p(R), _p(t), buf, C.c_size_t(512))
    if n < 0:
        check(n)
    return buf.value.decode()


def write_trajectory(path, stamps, Rs, ts):
    """<stamp tx ty tz qx qy qz qw> per line -- the file evaluate_ate.py / evaluate_rpe.py of the TUM benchmark read."""
    with open(path, "w") as f:
        for s, R, t in zip(stamps, Rs, ts):
            f.write(format_pose_line(float(s), R, t) + "\n")


class Dataset:
    """Evaluation (tools/evaluation.h:70): <folder>/depth_associated.txt + rgb_associated.txt, or a 4-column match file."""

    def __init__(self, folder, match_file=""):
        self._h = C.c_void_p()
        check(lib().rgbid_dataset_open(C.byref(self._h), os.fsencode(folder), os.fsencode(match_file or "")))
        lib().rgbid_dataset_stamp.restype = C.c_double

    def close(self):
        if self._h:
            lib().rgbid_dataset_close(self._h)
            self._h = None

    __del__ = close

    def __len__(self):
        return lib().rgbid_dataset_size(self._h)

    def stamp(self, i):
        return lib().rgbid_dataset_stamp(self._h, int(i))

    def grab(self, i, rows=480, cols=640):
        """-> (depth_mm u16 [rows,cols], rgb u8 [rows,cols,3]) or None when the pair cannot be read."""
        d = np.empty((rows, cols), np.uint16); c = np.empty((rows, cols, 3), np.uint8)
        ok = C.c_int()
        check(lib().rgbid_dataset_grab(self._h, int(i), _p(d), _p(c), rows, cols, C.byref(ok)))
        return (d, c) if ok.value else None

    def save_poses(self, tracker, poses_logfile, misc_logfile, frame_number=-1):
        check(lib().rgbid_tracker_save_poses(tracker._h, self._h, int(frame_number), os.fsencode(poses_logfile), os.fsencode(misc_logfile)))

    def save_kf_times(self, tracker, path):
        check(lib().rgbid_tracker_save_kf_times(tracker._h, self._h, os.fsencode(path)))
