"""ctypes loader for the HIP library (rgbid-slam_amd/lib/librgbid_hip.so) -- the C-ABI of include/rgbid.h.

There is no CPU fallback anywhere in this package: if the shared library is missing, or no HIP device
is usable, every entry point raises.  The oracle under /oracle is never imported from here.
"""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(_PKG)
# RGBID_HIP_LIB: an alternative build of the same library (A/B experiments with other -D flags: tools/kernel_bench.py); never a CPU path
LIB_PATH = os.environ.get("RGBID_HIP_LIB") or os.path.join(_PKG, "lib", "librgbid_hip.so")
CSRC = os.path.join(_PKG, "csrc")


class RgbidError(RuntimeError):
    pass


class Img(C.Structure):
    """rgbid_img: device pointer + pitch + size (PtrStepSz of the reference)."""
    _fields_ = [("data", C.c_void_p), ("step", C.c_size_t), ("rows", C.c_int), ("cols", C.c_int)]


class Intr(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


def build(force=False):
    """Compile the gfx950 kernels + C-ABI in-tree (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", CSRC, "-j8"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RgbidError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.rgbid_version.restype = C.c_char_p
        L.rgbid_error_string.restype = C.c_char_p
        L.rgbid_error_string.argtypes = [C.c_int]
        _lib = L
    return _lib


def check(err):
    if err != 0:
        raise RgbidError(f"rgbid error {err}: {lib().rgbid_error_string(err).decode()}")


# every symbol include/rgbid.h declares (checked by the CPU test-suite against the built library)
EXPORTS = [
    "rgbid_version", "rgbid_error_string", "rgbid_device_count", "rgbid_get_device_prop", "rgbid_set_device", "rgbid_ctx_create", "rgbid_ctx_destroy",
    "rgbid_ctx_set_stream", "rgbid_ctx_set_async", "rgbid_ctx_get_async", "rgbid_ctx_set_interp_mode", "rgbid_ctx_get_interp_mode", "rgbid_ctx_set_numerics", "rgbid_ctx_sync", "rgbid_selftest_rcp", "rgbid_selftest_div_const", "rgbid_selftest_cvt_flr", "rgbid_selftest_fast_primitives", "rgbid_fast_guard", "rgbid_fast_guard_lane", "rgbid_ctx_wait_event", "rgbid_ctx_get_stream", "rgbid_mem_info",
    "rgbid_malloc", "rgbid_malloc_pitch", "rgbid_free", "rgbid_malloc_host", "rgbid_free_host", "rgbid_memcpy_h2d", "rgbid_memcpy_d2h", "rgbid_memcpy_d2d",
    "rgbid_memcpy2d_h2d", "rgbid_memcpy2d_d2h", "rgbid_memcpy2d_d2d",
    "rgbid_depth_to_float", "rgbid_float_to_rgb", "rgbid_create_nmap", "rgbid_integrate_warped_rgb",
    "rgbid_undistort_intensity", "rgbid_undistort_depthinv", "rgbid_register_depthinv",
    "rgbid_depth_to_invdepth", "rgbid_compute_intensity", "rgbid_decompose_rgb", "rgbid_compute_gradient",
    "rgbid_copy_images", "rgbid_copy_image", "rgbid_copy_image_rgb", "rgbid_init_weight_keyframe", "rgbid_fill_2d",
    "rgbid_pyr_down", "rgbid_bilateral_filter", "rgbid_warp_invdepth", "rgbid_warp_intensity", "rgbid_warp_pair",
    "rgbid_warp_invdepth_weighted", "rgbid_integrate_warped_frame", "rgbid_visibility_ratio",
    "rgbid_create_vmap", "rgbid_create_nmap_gradients", "rgbid_generate_image",
    "rgbid_error_lattice_size", "rgbid_compute_error", "rgbid_sigma_nu_student", "rgbid_nu_student",
    "rgbid_sigma_pdf", "rgbid_chi_square", "rgbid_build_system", "rgbid_build_system_student_nu",
]
