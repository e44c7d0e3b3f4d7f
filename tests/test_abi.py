"""CPU checks of the drop-in boundary: the in-tree HIP library loads and exports every symbol that
include/rgbid.h declares (no compute call is made: there is no GPU in the authoring container)."""
import ctypes
import os
import re

from rgbid import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rgbid_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared("rgbid.h")
    names += _declared("rgbid_engine.h")
    names += _declared("rgbid_kfalign.h")
    batched = _declared("rgbid_batched.h")
    from rgbid import batched as BT
    assert set(batched) == set(BT.BATCHED_EXPORTS), set(batched) ^ set(BT.BATCHED_EXPORTS)
    names += batched
    assert len(names) >= 90
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # the Python binding's own list must not drift from the header
    assert set(_lib.EXPORTS) <= set(names)


def test_version_and_error_strings():
    L = _lib.lib()
    assert b"gfx950" in L.rgbid_version()
    assert L.rgbid_error_string(0) == b"ok"
    assert b"invalid" in L.rgbid_error_string(-1)


def test_no_device_is_a_loud_error():
    """Without a HIP device the context cannot be created: the product never falls back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        return
    L = _lib.lib()
    h = ctypes.c_void_p()
    assert L.rgbid_ctx_create(ctypes.byref(h), 0, None) != 0
    assert not h.value


def test_lattice_geometry_host_logic():
    """rgbid_error_lattice_size is pure host logic (sigmaFuncs.cu:712-741): check against the oracle on CPU."""
    from rgbid import device
    from oracle import oracle as O
    import numpy as np
    for rows, cols, ns in [(480, 640, 10000), (240, 320, 10000), (120, 160, 10000), (960, 1280, 10000), (480, 640, 9999999),
                           (61, 83, 100), (64, 64, 1000), (480, 640, 1), (30, 40, 300)]:
        a = np.zeros((rows, cols), np.float32)
        e, geo = O.error_lattice(a, a, ns)
        assert device.error_lattice_size(rows, cols, ns) == (e.size, geo[0], geo[1], geo[2])


def test_host_library_exports_every_declared_symbol():
    """librgbid_host.so (C++ VisodoTracker / KeyframeAlign / settings / SE(3)) exports all of include/rgbid_host.h."""
    from rgbid import host
    L = host.lib()
    names = _declared("rgbid_host.h")
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_dist_library_exports_every_declared_symbol():
    """librgbid_dist.so (multi-GPU helpers over librccl, include/rgbid_dist.h) loads without a GPU and exports every declared symbol"""
    import pytest
    from rgbid import dist as D
    if not os.path.exists(D.DIST_LIB_PATH) and not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("librgbid_dist.so is not built on hosts without RCCL (csrc/Makefile says so); every other library is")
    L = D.dlib()
    names = [n for n in _declared("rgbid_dist.h") if n.startswith("rgbid_dist_")]
    assert len(names) == 15 and set(names) == set(D.DIST_EXPORTS), names
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # the 392-byte record of SURVEY 8e, as the Python harness mirrors it
    assert D.GATHER_DTYPE.itemsize == 392 and D.GATHER_DTYPE.fields["R"][1] == 8 and D.GATHER_DTYPE.fields["cov"][1] == 104


def test_headers_are_valid_c(tmp_path):
    """every C-ABI header compiles as C99 on its own (no C++-isms leaked into the boundary)"""
    import subprocess
    for h in ("rgbid.h", "rgbid_batched.h", "rgbid_engine.h", "rgbid_dist.h", "rgbid_host.h"):
        src = tmp_path / (h + ".c")
        src.write_text(f'#include "{h}"\nint main(void) {{ return 0; }}\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)])


def test_python_engine_config_matches_the_library():
    """the ctypes mirror of rgbid_engine_config has the size the library was built with (fields are only appended; a stale mirror or a stale library is an error
    the engine constructors report instead of corrupting memory)"""
    import ctypes as C
    from rgbid import engine as E
    L = _lib.lib()
    L.rgbid_engine_config_size.restype = C.c_size_t
    assert L.rgbid_engine_config_size() == C.sizeof(E.EngineConfig)
