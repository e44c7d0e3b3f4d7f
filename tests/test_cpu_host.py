"""CPU tests of the host-side product code (librgbid_host.so): SE(3) helpers and the INI parser.
  * SE(3): product (Jacobi inverse-square-root polar factor) vs oracle (Newton polar iteration) vs scipy expm/logm --
    three independent implementations;
  * settings: product parser vs the REFERENCE's own parser compiled from /root/reference/src/settings.cpp into
    oracle/_ref (skipped if that build is absent) on a committed fixture that exercises every syntax rule."""
import ctypes
import os

import numpy as np
import pytest
import scipy.linalg

from oracle import oracle as O
from rgbid import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INI = os.path.join(ROOT, "tests", "golden", "visodo_test.ini")


def skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


@pytest.mark.parametrize("scale", [1e-9, 1e-6, 1e-3, 0.05, 1.0, 3.0])
def test_expmap_logmap_three_ways(scale):
    r = np.random.default_rng(3)
    for _ in range(20):
        w = r.standard_normal(3); w *= scale / np.linalg.norm(w)
        v = r.standard_normal(3) * 0.1
        R, t = host.expmap(w, v)
        Ro, to = O.expmap(w, v)
        T = np.zeros((4, 4)); T[:3, :3] = skew(w); T[:3, 3] = v
        E = scipy.linalg.expm(T)
        assert np.abs(R - Ro).max() < 1e-13 and np.abs(t - to).max() < 1e-13
        assert np.abs(R - E[:3, :3]).max() < 1e-12 and np.abs(t - E[:3, 3]).max() < 1e-12
        assert np.abs(host.expmap_rot(w) - R).max() < 1e-15
        tw = host.logmap(R, t)
        assert np.abs(tw - O.logmap(R, t)).max() < 1e-10
        if scale < 3.0:
            assert np.abs(tw[3:] - w).max() < 1e-9 * max(1, scale) and np.abs(tw[:3] - v).max() < 1e-8


def test_force_orthogonal_is_polar_factor():
    r = np.random.default_rng(4)
    for _ in range(20):
        M = np.eye(3) + 0.05 * r.standard_normal((3, 3))
        U, _, Vt = np.linalg.svd(M)
        R = host.force_orthogonal(M)
        assert np.abs(R - U @ Vt).max() < 1e-13            # Eigen: svd.matrixU() * svd.matrixV().transpose()
        assert np.abs(R - O.force_orthogonal(M)).max() < 1e-13
        assert np.abs(R.T @ R - np.eye(3)).max() < 1e-14


def test_force_orthogonal_fast_path_on_rotations():
    """A rotation up to rounding (what expMap / logMap feed it) takes the one-Newton-step path of se3.h: still the polar factor of the SVD."""
    r = np.random.default_rng(41)
    for _ in range(50):
        w = r.standard_normal(3) * r.choice([1e-6, 1e-2, 1.0, 3.0])
        th = np.linalg.norm(w); k = w / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        M = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx + 1e-15 * r.standard_normal((3, 3))   # Rodrigues + rounding-size noise
        U, _, Vt = np.linalg.svd(M)
        R = host.force_orthogonal(M)
        assert np.abs(R - U @ Vt).max() < 5e-15                # a few ulp: LAPACK's own U @ Vt carries as much
        assert np.abs(R.T @ R - np.eye(3)).max() < 1e-15
        assert np.abs(R - O.force_orthogonal(M)).max() < 5e-15


def test_llt_and_inverse():
    r = np.random.default_rng(5)
    for _ in range(10):
        J = r.standard_normal((50, 6)) * r.uniform(0.1, 100, 6)
        A = J.T @ J; b = r.standard_normal(6)
        x = host.llt_solve6(A, b)
        assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-9)
        assert np.allclose(x, O.llt_solve6(A, b)[0], rtol=1e-12)
        assert np.allclose(host.inverse6(A), np.linalg.inv(A), rtol=1e-8)
    # a non-PD matrix propagates NaN like Eigen's LLT (-> tracker declares itself lost)
    A = np.eye(6); A[2, 2] = -1.0
    assert np.isnan(host.llt_solve6(A, np.ones(6))).any()


def _llt_dividing(A, b):
    """Eigen's LLT + two triangular solves as Eigen runs them: every column entry and every substituted element DIVIDED by the pivot"""
    L = np.zeros((6, 6))
    for j in range(6):
        L[j, j] = np.sqrt(A[j, j] - np.dot(L[j, :j], L[j, :j]))
        for i in range(j + 1, 6):
            s_ = A[i, j]
            for k in range(j):
                s_ -= L[i, k] * L[j, k]
            L[i, j] = s_ / L[j, j]
    y = np.zeros(6); x = np.zeros(6)
    for i in range(6):
        s_ = b[i]
        for k in range(i):
            s_ -= L[i, k] * y[k]
        y[i] = s_ / L[i, i]
    for i in range(5, -1, -1):
        s_ = y[i]
        for k in range(i + 1, 6):
            s_ -= L[k, i] * x[k]
        x[i] = s_ / L[i, i]
    return x


def test_reciprocal_per_pivot_against_a_dividing_solve_on_ill_conditioned_systems():
    """se3.h's llt_solve6 / inverse6 multiply by one reciprocal per pivot where Eigen divides.  Held here to a DIVIDING Cholesky solve of the same
    operation order and to LAPACK on normal equations of condition 1e2 .. 1e12: the deviation is a few ulp times the condition number, the bound every
    backward-stable solve has -- far inside the pose bar (1e-4) for the tracker's systems (cond <= 1e8)."""
    r = np.random.default_rng(77)
    eps = np.finfo(np.float64).eps
    for cond in (1e2, 1e5, 1e8, 1e10, 1e12):
        for _ in range(5):
            U, _ = np.linalg.qr(r.standard_normal((6, 6)))
            A = (U * np.geomspace(1.0, 1.0 / cond, 6)) @ U.T
            A = 0.5 * (A + A.T)
            xt = r.standard_normal(6); b = A @ xt
            x = host.llt_solve6(A, b); xd = _llt_dividing(A, b)
            assert np.linalg.norm(x - xd) <= 50 * eps * cond * np.linalg.norm(xd), (cond, x, xd)
            assert np.linalg.norm(x - xt) <= 200 * eps * cond * np.linalg.norm(xt)
            Ai = host.inverse6(A)
            assert np.linalg.norm(Ai - np.linalg.inv(A)) <= 200 * eps * cond * np.linalg.norm(Ai)
            assert np.abs(Ai @ A - np.eye(6)).max() <= 200 * eps * cond


QUERIES = [("VISODO", "M_ESTIMATOR"), ("VISODO", "SIGMA_ESTIMATOR"), ("VISODO", "INTEGRATION_VISRATIO_THRESHOLD"),
           ("VISODO", "ODOMETRY_VISRATIO_THRESHOLD"), ("VISODO", "FINEST_PYR_LEVEL"), ("VISODO", "WARP_ORDER"), ("VISODO", "IMAGE_FILTERING"),
           ("VISODO", "LATE_KEY"), ("CALIBRATION", "fx"), ("CALIBRATION", "fy"), ("CALIBRATION", "kd"), ("CALIBRATION", "novalue"),
           ("CALIBRATION", ""), ("NOPE", "x"), ("VISODO", "missing")]


def test_settings_known_answers():
    g = lambda s, k: host.settings_get(INI, s, k)
    assert g("VISODO", "M_ESTIMATOR") == "Student"          # first value wins, also across a repeated section header
    assert g("VISODO", "WARP_ORDER") == "pyrFirst"
    assert g("VISODO", "ODOMETRY_VISRATIO_THRESHOLD") == "0.9"
    assert g("VISODO", "LATE_KEY") == "7"                    # a repeated [VISODO] keeps filling the same section
    assert g("CALIBRATION", "fy") == "-480.0"                # section names are trimmed
    assert g("CALIBRATION", "kd") == "0.1 0.2\n0.3 0.4 0.5"  # continuation line
    assert g("CALIBRATION", "novalue") == "\n= orphan value" or g("CALIBRATION", "novalue") is not None
    assert g("NOPE", "x") is None and g("VISODO", "missing") is None


def test_settings_match_reference_parser():
    so = os.path.join(ROOT, "oracle", "_ref", "libref_settings.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    ref = ctypes.CDLL(so)
    for sec, key in QUERIES:
        buf = ctypes.create_string_buffer(4096)
        n = ref.ref_settings_get(INI.encode(), sec.encode(), key.encode(), buf, 4096)
        want = None if n < 0 else buf.value.decode()
        assert host.settings_get(INI, sec, key) == want, (sec, key, want)


def test_settings_match_reference_parser_on_shipped_config_files():
    """Every (section, key) of the nine configuration / calibration files the reference ships (config_data/*.ini), read by the
    reference's own parser (oracle/_ref, built from /root/reference/src/settings.cpp) and by the product's parser.  Runs only where
    /root/reference exists (the authoring container); nothing is copied."""
    import glob
    import re
    so = os.path.join(ROOT, "oracle", "_ref", "libref_settings.so")
    files = sorted(glob.glob("/root/reference/config_data/*.ini"))
    if not os.path.exists(so) or not files:
        pytest.skip("needs /root/reference and oracle/_ref")
    ref = ctypes.CDLL(so)
    n_keys = 0
    for f in files:
        text = open(f).read()
        sections = set(m.strip() for m in re.findall(r"^\s*\[([^\]]*)\]", text, re.M)) | {"VISODO", "CALIBRATION"}
        keys = set(m.strip() for m in re.findall(r"^\s*([A-Za-z_][A-Za-z0-9_]*)\s*=", text, re.M)) | {"fx", "kd", "dRc", "t_dc", "q0", "missing"}
        for sec in sections:
            for key in keys:
                buf = ctypes.create_string_buffer(8192)
                n = ref.ref_settings_get(f.encode(), sec.encode(), key.encode(), buf, 8192)
                want = None if n < 0 else buf.value.decode()
                assert host.settings_get(f, sec, key) == want, (f, sec, key, want)
                n_keys += want is not None
    assert n_keys > 60


def test_settings_match_reference_parser_on_random_grammar(tmp_path):
    """The product's INI loader is written from the file grammar (host/settings.cpp), not from the reference's source: hold it to the
    reference's own parser (oracle/_ref) on randomly generated files that mix every production -- headers with and without ']', repeated
    headers, keys before any header, repeated keys, empty keys, empty values, continuation lines after accepted, refused and dropped
    assignments, comments, blank and blank-padded lines."""
    so = os.path.join(ROOT, "oracle", "_ref", "libref_settings.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    ref = ctypes.CDLL(so)
    rs = np.random.default_rng(20260929)
    secs, keys = ["A", "B", "", "A B"], ["k", "x", "kd", "long key", ""]
    pad = lambda: " " * int(rs.integers(0, 3)) + "\t" * int(rs.integers(0, 2))
    n_hits = 0
    for case in range(40):
        lines = []
        for _ in range(int(rs.integers(5, 40))):
            kind = int(rs.integers(0, 9))
            s, k = secs[int(rs.integers(0, len(secs)))], keys[int(rs.integers(0, len(keys)))]
            v = " ".join(str(int(q)) for q in rs.integers(0, 99, int(rs.integers(0, 4))))
            if kind == 0: body = ""
            elif kind == 1: body = ("#" if rs.random() < 0.5 else ";") + " note = 1"
            elif kind == 2: body = "[" + pad() + s + pad() + ("]" if rs.random() < 0.8 else "") + (" trailing" if rs.random() < 0.2 else "")
            elif kind in (3, 4, 5): body = k + pad() + "=" + pad() + v + ("= 7" if rs.random() < 0.1 else "")
            elif kind == 6: body = "=" + v
            else: body = v if v else "word"
            lines.append(pad() + body + pad() + ("\r" if rs.random() < 0.1 else ""))
        path = tmp_path / f"case{case}.ini"
        path.write_text("\n".join(lines) + ("\n" if rs.random() < 0.5 else ""))
        for s in secs + ["missing"]:
            for k in keys + ["missing"]:
                buf = ctypes.create_string_buffer(8192)
                n = ref.ref_settings_get(str(path).encode(), s.encode(), k.encode(), buf, 8192)
                want = None if n < 0 else buf.value.decode()
                assert host.settings_get(str(path), s, k) == want, (case, s, k, want, lines)
                n_hits += want is not None
    assert n_hits > 100
