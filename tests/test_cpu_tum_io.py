"""on-disk formats either side of the tracking path (TUM / ICL-NUIM PNG frames, association files,
trajectory writer).  The PNG decoder is checked against an independent pure-Python encoder using every filter type;
the writer against scipy's quaternion and printf-style formatting.  No GPU needed."""
import os
import struct
import zlib

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from rgbid import tum


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)


def _py_png(a, bit_depth, ctype, filters, palette=None, idat_split=1):
    """independent PNG encoder: a = [rows][cols][ch] samples; per-row filter types cycle through `filters`."""
    rows, cols = a.shape[:2]
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    if bit_depth == 16:
        rowbytes = [a[y].astype(">u2").tobytes() for y in range(rows)]
    elif bit_depth == 8:
        rowbytes = [a[y].astype(np.uint8).tobytes() for y in range(rows)]
    else:  # packed sub-byte samples
        rowbytes = []
        for y in range(rows):
            bits = "".join(format(int(v), f"0{bit_depth}b") for v in a[y].reshape(-1))
            bits += "0" * (-len(bits) % 8)
            rowbytes.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    bpp = max(1, ch * bit_depth // 8)
    raw = b""
    prev = bytes(len(rowbytes[0]))
    for y, cur in enumerate(rowbytes):
        ft = filters[y % len(filters)]
        out = bytearray(len(cur))
        for i in range(len(cur)):
            left = cur[i - bpp] if i >= bpp else 0
            up = prev[i]
            ul = prev[i - bpp] if i >= bpp else 0
            if ft == 0: pred = 0
            elif ft == 1: pred = left
            elif ft == 2: pred = up
            elif ft == 3: pred = (left + up) >> 1
            else:
                p = left + up - ul
                pa, pb, pc = abs(p - left), abs(p - up), abs(p - ul)
                pred = left if (pa <= pb and pa <= pc) else (up if pb <= pc else ul)
            out[i] = (cur[i] - pred) & 255
        raw += bytes([ft]) + bytes(out)
        prev = cur
    z = zlib.compress(raw, 9)
    parts = [z[i * len(z) // idat_split:(i + 1) * len(z) // idat_split] for i in range(idat_split)]
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", cols, rows, bit_depth, ctype, 0, 0, 0))
    png += _chunk(b"tEXt", b"Comment\x00ancillary chunks are skipped")
    if palette is not None:
        png += _chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    for p in parts:
        png += _chunk(b"IDAT", p)
    return png + _chunk(b"IEND", b"")


@pytest.mark.parametrize("bit_depth,ctype", [(16, 0), (8, 0), (8, 2), (16, 2), (8, 6), (8, 4)])
def test_png_decoder_all_filters(tmp_path, bit_depth, ctype):
    rng = np.random.default_rng(bit_depth * 10 + ctype)
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    a = rng.integers(0, 1 << bit_depth, size=(23, 37, ch))
    a[5:9] = a[4:5]                                   # some structure for the predictors
    p = tmp_path / "t.png"
    p.write_bytes(_py_png(a, bit_depth, ctype, filters=[0, 1, 2, 3, 4], idat_split=3))
    got = tum.read_png(str(p))
    assert tum.png_info(str(p)) == (23, 37, ch, bit_depth)
    np.testing.assert_array_equal(got.reshape(23, 37, ch), a)


def test_png_palette_and_subbyte(tmp_path):
    rng = np.random.default_rng(3)
    pal = rng.integers(0, 256, size=(16, 3))
    idx = rng.integers(0, 16, size=(9, 13, 1))
    p = tmp_path / "p.png"
    p.write_bytes(_py_png(idx, 4, 3, filters=[0], palette=pal))
    np.testing.assert_array_equal(tum.read_png(str(p)), pal[idx[:, :, 0]])
    g = rng.integers(0, 4, size=(7, 10, 1))
    p.write_bytes(_py_png(g, 2, 0, filters=[0, 2]))
    np.testing.assert_array_equal(tum.read_png(str(p)), g[:, :, 0] * 85)


def test_png_writer_round_trip_and_errors(tmp_path):
    rng = np.random.default_rng(4)
    d = rng.integers(0, 65536, size=(48, 64)).astype(np.uint16)
    c = rng.integers(0, 256, size=(48, 64, 3)).astype(np.uint8)
    tum.write_png(str(tmp_path / "d.png"), d); tum.write_png(str(tmp_path / "c.png"), c)
    np.testing.assert_array_equal(tum.read_png(str(tmp_path / "d.png")), d)
    np.testing.assert_array_equal(tum.read_png(str(tmp_path / "c.png")), c)
    # the written file is a valid PNG for an independent decoder (filter 0 rows, big-endian samples)
    b = (tmp_path / "d.png").read_bytes()
    assert b[:8] == b"\x89PNG\r\n\x1a\n" and b[12:16] == b"IHDR"
    pos, idat = 8, b""
    while pos < len(b):
        n = struct.unpack(">I", b[pos:pos + 4])[0]; t = b[pos + 4:pos + 8]
        assert zlib.crc32(b[pos + 4:pos + 8 + n]) & 0xffffffff == struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0]
        if t == b"IDAT": idat += b[pos + 8:pos + 8 + n]
        pos += 12 + n
    raw = zlib.decompress(idat)
    rows = [raw[y * (1 + 128):(y + 1) * (1 + 128)] for y in range(48)]
    assert all(r[0] == 0 for r in rows)
    np.testing.assert_array_equal(np.frombuffer(b"".join(r[1:] for r in rows), ">u2").reshape(48, 64), d)
    # corrupt / truncated / missing files are loud errors, not silent garbage
    bad = bytearray(b); bad[60] ^= 0xff
    (tmp_path / "bad.png").write_bytes(bytes(bad))
    for name in ("bad.png", "missing.png"):
        with pytest.raises(Exception):
            tum.read_png(str(tmp_path / name))
    (tmp_path / "trunc.png").write_bytes(b[:len(b) // 2])
    with pytest.raises(Exception):
        tum.read_png(str(tmp_path / "trunc.png"))


def _make_dataset(root, n=5, rows=24, cols=32, seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(root / "depth"); os.makedirs(root / "rgb")
    dl, cl, frames = [], [], []
    for k in range(n):
        d = rng.integers(0, 50001, size=(rows, cols)).astype(np.uint16); d[0, :5] = [0, 2, 3, 7, 65535]
        c = rng.integers(0, 256, size=(rows, cols, 3)).astype(np.uint8)
        td, tc = 1305031102.175304 + k / 30.0, 1305031102.160407 + k / 30.0
        tum.write_png(str(root / "depth" / f"{td:.6f}.png"), d); tum.write_png(str(root / "rgb" / f"{tc:.6f}.png"), c)
        dl.append(f"{td:.6f} depth/{td:.6f}.png"); cl.append(f"{tc:.6f} rgb/{tc:.6f}.png"); frames.append((td, d, c))
    hdr = "# depth maps\n# file: 'rgbd_dataset.bag'\n# timestamp filename\n"
    (root / "depth_associated.txt").write_text(hdr + "\n".join(dl) + "\n")
    (root / "rgb_associated.txt").write_text(hdr.replace("depth maps", "color images") + "\n".join(cl) + "\n")
    (root / "matches.txt").write_text("\n".join(f"{a} {b}" for a, b in zip(dl, cl)) + "\n")
    return frames


@pytest.mark.parametrize("match_file", ["", "matches.txt"])
def test_dataset_playback(tmp_path, match_file):
    frames = _make_dataset(tmp_path)
    ds = tum.Dataset(str(tmp_path), match_file)        # folder without trailing slash: one is appended (evaluation.cpp:126)
    assert len(ds) == len(frames)
    for k, (td, d, c) in enumerate(frames):
        assert ds.stamp(k) == float("%.6f" % td)
        got = ds.grab(k, 24, 32)
        assert got is not None
        # PNG stores depth*5000; x0.2 with round-to-nearest -> millimetres (evaluation.cpp:296); 65535 -> 13107
        np.testing.assert_array_equal(got[0], np.rint(d.astype(np.float64) * 0.2).astype(np.uint16))
        np.testing.assert_array_equal(got[1], c)       # r,g,b byte order (imread BGR -> cvtColor BGR2RGB)
    assert ds.grab(len(frames), 24, 32) is None        # past the end -> grab() false
    os.remove(tmp_path / "depth" / f"{frames[2][0]:.6f}.png")
    assert ds.grab(2, 24, 32) is None                  # unreadable pair -> false (playback counts it as a failure)
    with pytest.raises(Exception):
        tum.Dataset(str(tmp_path / "nowhere"))


def test_depth_must_be_16_bit(tmp_path):
    frames = _make_dataset(tmp_path, n=1)
    tum.write_png(str(tmp_path / "depth" / f"{frames[0][0]:.6f}.png"), np.zeros((24, 32), np.uint8))
    with pytest.raises(Exception):
        tum.Dataset(str(tmp_path)).grab(0, 24, 32)


def test_pose_line_format_and_quaternion():
    rng = np.random.default_rng(7)
    for k in range(200):
        if k < 150:
            rv = rng.normal(size=3); rv *= rng.uniform(0, np.pi) / np.linalg.norm(rv)
        else:                                          # rotations by ~pi about each axis hit the three non-trace branches
            rv = np.eye(3)[k % 3] * (np.pi - 1e-3 * rng.uniform()) + 1e-3 * rng.normal(size=3)
        R = Rotation.from_rotvec(rv).as_matrix()
        t = rng.normal(size=3) * 3
        stamp = 1305031102.175304 + k
        line = tum.format_pose_line(stamp, R, t)
        f = line.split(" ")
        assert len(f) == 8 and all(len(x.split(".")[1]) == 6 for x in f)      # fixed notation, 6 decimals
        assert f[0] == "%.6f" % stamp
        assert [float(x) for x in f[1:4]] == pytest.approx(np.float32(t), abs=1.01e-6)
        q = np.array([float(x) for x in f[4:]])
        qs = Rotation.from_matrix(R).as_quat()
        if np.dot(q, qs) < 0: qs = -qs
        assert np.abs(q - qs).max() < 2e-6
        assert abs(np.linalg.norm(q) - 1) < 3e-6
    assert tum.format_pose_line(0.0, np.eye(3), np.zeros(3)) == "0.000000 0.000000 0.000000 0.000000 0.000000 0.000000 0.000000 1.000000"


def test_trajectory_file_is_tum_readable(tmp_path):
    Rs = [Rotation.from_euler("xyz", [0.1 * k, -0.05 * k, 0.02 * k]).as_matrix() for k in range(6)]
    ts = [np.array([0.01 * k, 0.2, -0.3 * k]) for k in range(6)]
    st = [100.5 + k / 30 for k in range(6)]
    tum.write_trajectory(str(tmp_path / "traj.txt"), st, Rs, ts)
    a = np.loadtxt(tmp_path / "traj.txt")               # what the TUM evaluate_ate.py reader does
    assert a.shape == (6, 8)
    np.testing.assert_allclose(a[:, 0], st, atol=1e-6); np.testing.assert_allclose(a[:, 1:4], ts, atol=1e-6)
    np.testing.assert_allclose(Rotation.from_quat(a[:, 4:]).as_matrix(), Rs, atol=3e-6)
