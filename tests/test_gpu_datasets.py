"""Sequences on disk, tracked by the product and scored with tools/ate.py -- the external pin of the trajectory (the reference holds
no fixtures; ATE / RPE against the published ground truth of TUM fr1/desk, ICL-NUIM lr-kt2 and TUM fr3/long_office is the anchor that does not
depend on the oracle).  The datasets are not in this image: set RGBID_TUM_DIR to a folder holding one or more sequences in the layout the
reference's evaluation mode reads (depth_associated.txt + rgb_associated.txt [+ groundtruth.txt], tools/evaluation.cpp:122-351) and the
gated test below runs; without it that test SKIPS LOUDLY and BASELINE configs 2-4 stay exercised by the synthetic stand-ins only.
The ungated test renders a synthetic sequence INTO that on-disk layout and runs the identical chain, so the chain itself is always tested."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from rgbid import synth, tum

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ate as A  # noqa: E402


def write_tum_layout(root, seq, t0=1305031102.175304, fps=30.0):
    """a synthetic sequence as a TUM-layout folder: 16-bit PNG depth x5000, 8-bit RGB PNG, association files, groundtruth.txt"""
    from scipy.spatial.transform import Rotation
    d = seq["depth"].cpu().numpy().astype(np.uint16); c = seq["rgb"].cpu().numpy()
    os.makedirs(root / "depth"); os.makedirs(root / "rgb")
    hdr = "# line 1\n# line 2\n# timestamp filename\n"
    dl, cl, gl = [], [], ["# ground truth trajectory", "# timestamp tx ty tz qx qy qz qw"]
    Rw, tw = seq["R_wc"].numpy(), seq["t_wc"].numpy()
    for k in range(d.shape[0]):
        st = t0 + k / fps
        tum.write_png(str(root / "depth" / f"{st:.6f}.png"), (d[k].astype(np.uint32) * 5).astype(np.uint16))
        tum.write_png(str(root / "rgb" / f"{st:.6f}.png"), c[k])
        dl.append(f"{st:.6f} depth/{st:.6f}.png"); cl.append(f"{st:.6f} rgb/{st:.6f}.png")
        q = Rotation.from_matrix(Rw[k]).as_quat()
        gl.append(f"{st:.4f} {tw[k][0]:.6f} {tw[k][1]:.6f} {tw[k][2]:.6f} {q[0]:.6f} {q[1]:.6f} {q[2]:.6f} {q[3]:.6f}")
    (root / "depth_associated.txt").write_text(hdr + "\n".join(dl) + "\n")
    (root / "rgb_associated.txt").write_text(hdr + "\n".join(cl) + "\n")
    (root / "groundtruth.txt").write_text("\n".join(gl) + "\n")


def track_and_score(folder, out, chunks, K=None, max_frames=-1):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "track_dataset.py"), str(folder), "--chunks", str(chunks), "--out", str(out), "--max-frames", str(max_frames)]
    if K is not None:
        cmd += ["--K"] + [str(v) for v in K]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=3600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    gt, est = A.read_trajectory(os.path.join(folder, "groundtruth.txt")), A.read_trajectory(out)
    return A.ate(gt, est, 0.02), A.rpe(gt, est, 1, "f", 0.02), len(est[0])


def test_synthetic_sequence_through_the_dataset_chain(tmp_path):
    """TUM-layout folder -> product dataset reader (PNG decode, x0.2 depth convention) -> chunk-sharded batched engine -> record gather ->
    composed trajectory file -> ATE / RPE against the ground-truth file: millimetres on the noise-limited synthetic sequence, and the
    4-chunk run scores like the sequential one"""
    n = 25
    seq = synth.make_sequence(n, seed=synth.SEED + 3, device="cuda")
    root = tmp_path / "synth_office"
    write_tum_layout(root, seq)
    a1, r1, n1 = track_and_score(root, tmp_path / "traj1.txt", 1)
    a4, r4, n4 = track_and_score(root, tmp_path / "traj4.txt", 4)
    print(f"synthetic TUM-layout sequence, {n} frames: ATE rmse {a1['rmse'] * 1e3:.2f} mm sequential / {a4['rmse'] * 1e3:.2f} mm in 4 chunks; "
          f"RPE/frame {r1['trans_rmse'] * 1e3:.2f} mm, {np.degrees(r1['rot_rmse']):.4f} deg")
    assert n1 == n4 == n and a1["pairs"] == n
    assert a1["rmse"] < 3e-3 and a4["rmse"] < 5e-3
    assert r1["trans_rmse"] < 1.5e-3 and r1["rot_rmse"] < 1e-3


def run_cpp_driver(folder, out, chunks, K=None, max_frames=-1, world=1, extra=()):
    """tools/rgbid_track_sequence (C++: dataset reader -> rgbid_dist_track_sequence -> trajectory file) as `world` processes on this box's one GPU
    (world > 1: records exchanged over the library's TCP transport -- RCCL refuses two ranks on one device)"""
    from rgbid import dist as D
    port = D.free_port()
    procs = []
    for r in reversed(range(world)):
        cmd = [D.TRACK_SEQUENCE_BIN, "-eval", str(folder) + "/", "-chunks", str(chunks), "-out", str(out), "-max_frames", str(max_frames), "-gpu", "0",
               "-world", str(world), "-rank", str(r), "-master_addr", "127.0.0.1", "-master_port", str(port)] + (["-exchange", "tcp"] if world > 1 else []) + list(extra)
        if K is not None:
            cmd += ["-K"] + [str(v) for v in K]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = [p.communicate(timeout=3600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [r_[0][-800:] + r_[1][-800:] for r_ in res]
    return json.loads([l for l in res[-1][0].splitlines() if l.startswith("{")][-1])


def test_cpp_sequence_driver_equals_the_python_harness(tmp_path):
    """the C++ host of the sharded path (tools/rgbid_track_sequence.cpp over rgbid_engine.h + rgbid_dist.h) writes, byte for byte, the trajectory
    file tools/track_dataset.py writes for the same TUM-layout folder and chunk count; cut over two processes (a 2-rank run on the one GPU) it
    composes the same poses"""
    n = 25
    seq = synth.make_sequence(n, seed=synth.SEED + 3, device="cuda")
    root = tmp_path / "synth_office"
    write_tum_layout(root, seq)
    a_py, _, n_py = track_and_score(root, tmp_path / "py4.txt", 4)
    rep = run_cpp_driver(root, tmp_path / "cpp4.txt", 4)
    assert rep["frames"] == n and rep["chunks"] == 4 and rep["lanes_per_gpu"] == 4 and rep["world"] == 1
    assert (tmp_path / "cpp4.txt").read_bytes() == (tmp_path / "py4.txt").read_bytes()
    rep1 = run_cpp_driver(root, tmp_path / "cpp1.txt", 1)
    track_and_score(root, tmp_path / "py1.txt", 1)
    assert (tmp_path / "cpp1.txt").read_bytes() == (tmp_path / "py1.txt").read_bytes()
    rep2 = run_cpp_driver(root, tmp_path / "cpp4w2.txt", 4, world=2)
    assert rep2["world"] == 2 and rep2["lanes_per_gpu"] == 2
    e1, e2 = A.read_trajectory(str(tmp_path / "cpp4.txt")), A.read_trajectory(str(tmp_path / "cpp4w2.txt"))
    assert np.abs(np.asarray(e1[1]) - np.asarray(e2[1])).max() < 5e-6          # 6-decimal text; lanes per launch differ (another launch plan of the normal equations: partial-sum grouping)
    gt = A.read_trajectory(str(root / "groundtruth.txt"))
    assert A.ate(gt, e2, 0.02)["rmse"] < 5e-3
    print("C++ driver:", rep, rep2)


def _sequences():
    base = os.environ.get("RGBID_TUM_DIR", "")
    if not base or not os.path.isdir(base):
        return []
    cands = [base] + [os.path.join(base, d) for d in sorted(os.listdir(base))]
    return [c for c in cands if os.path.exists(os.path.join(c, "depth_associated.txt")) and os.path.exists(os.path.join(c, "groundtruth.txt"))]


def test_real_sequences_ate_against_published_ground_truth(tmp_path):
    """BASELINE configs 2-4 on the real data, whenever it is mounted: every sequence under RGBID_TUM_DIR is tracked sequentially and in 8
    chunks and scored against its groundtruth.txt.  Bounds are sanity bounds for frame-to-keyframe dense odometry without loop closure
    (the published RGBiD-SLAM figures include its back-end); the numbers are printed for DESIGN.md."""
    seqs = _sequences()
    if not seqs:
        pytest.skip("LOUD SKIP: RGBID_TUM_DIR is not set (TUM fr1/desk, ICL-NUIM lr-kt2, TUM fr3/long_office are not in this image): "
                    "BASELINE configs 2-4 are exercised on synthetic stand-ins only; mount the sequences (association files + groundtruth.txt) to run this")
    report = {}
    for s in seqs:
        name = os.path.basename(os.path.normpath(s))
        icl = "kt" in name.lower()
        K = (481.2, -480.0, 319.5, 239.5) if icl else None          # config_data/calibration_syntheticHanda.ini
        a1, r1, n1 = track_and_score(s, tmp_path / f"{name}_1.txt", 1, K)
        a8, r8, n8 = track_and_score(s, tmp_path / f"{name}_8.txt", 8, K)
        run_cpp_driver(s, tmp_path / f"{name}_8_cpp.txt", 8, K)                       # the C++ host of the same path: identical file
        assert (tmp_path / f"{name}_8_cpp.txt").read_bytes() == (tmp_path / f"{name}_8.txt").read_bytes()
        report[name] = dict(frames=n1, ate_rmse_sequential=a1["rmse"], ate_rmse_8_chunks=a8["rmse"], rpe_trans_rmse=r1["trans_rmse"], rpe_rot_rmse_deg=float(np.degrees(r1["rot_rmse"])))
        assert a1["rmse"] < 0.25 and r1["trans_rmse"] < 0.03, (name, a1["rmse"], r1["trans_rmse"])
    print(json.dumps(report))
