"""tools/ate.py (own implementation of the TUM benchmark's ATE / RPE) on synthetic trajectories with known answers."""
import os
import sys

import numpy as np
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ate as A  # noqa: E402


def make_traj(n=200, seed=0):
    r = np.random.default_rng(seed)
    R = [np.eye(3)]; t = [np.zeros(3)]
    for _ in range(1, n):
        R.append(R[-1] @ Rotation.from_rotvec(0.02 * r.standard_normal(3)).as_matrix())
        t.append(t[-1] + R[-2] @ (0.02 * r.standard_normal(3) + np.array([0.01, 0, 0])))
    st = 1305031102.0 + np.arange(n) / 30.0
    return st, np.array(t), np.array(R)


def write(path, st, t, R):
    q = Rotation.from_matrix(R).as_quat()   # x y z w
    with open(path, "w") as f:
        f.write("# timestamp tx ty tz qx qy qz qw\n")
        for k in range(len(st)):
            f.write(f"{st[k]:.6f} {t[k][0]:.9f} {t[k][1]:.9f} {t[k][2]:.9f} {q[k][0]:.9f} {q[k][1]:.9f} {q[k][2]:.9f} {q[k][3]:.9f}\n")


def test_ate_is_zero_under_a_rigid_transform_and_measures_added_noise(tmp_path):
    st, t, R = make_traj()
    Rg = Rotation.from_rotvec([0.3, -0.2, 0.5]).as_matrix(); tg = np.array([1.0, -2.0, 0.5])
    # the estimate lives in another world frame: T_est = G^-1 T_gt
    te = (t - tg) @ Rg; Re = np.einsum("ji,njk->nik", Rg, R)
    write(tmp_path / "gt.txt", st, t, R); write(tmp_path / "est.txt", st + 0.004, te, Re)      # 4 ms stamp jitter: still associated
    a = A.ate(A.read_trajectory(tmp_path / "gt.txt"), A.read_trajectory(tmp_path / "est.txt"))
    assert a["pairs"] == len(st) and a["rmse"] < 1e-7
    assert np.abs(a["R"] - Rg).max() < 1e-6 and np.abs(a["t"] - tg).max() < 1e-6
    r = A.rpe(A.read_trajectory(tmp_path / "gt.txt"), A.read_trajectory(tmp_path / "est.txt"), 1, "f")
    assert r["trans_rmse"] < 1e-7 and r["rot_rmse"] < 1e-6                                      # relative motions do not see the world frame
    noise = 0.01 * np.random.default_rng(1).standard_normal(te.shape)
    write(tmp_path / "noisy.txt", st, te + noise, Re)
    a = A.ate(A.read_trajectory(tmp_path / "gt.txt"), A.read_trajectory(tmp_path / "noisy.txt"))
    expect = np.sqrt((noise ** 2).sum(1).mean())
    assert abs(a["rmse"] - expect) < 0.05 * expect                                              # alignment absorbs 6 of 600 degrees of freedom


def test_rpe_of_a_constant_drift(tmp_path):
    """the estimate gains 1 mm per frame along its own x axis and 0.01 deg about z: RPE(1 frame) returns exactly that"""
    st, t, R = make_traj(120, seed=3)
    dR = Rotation.from_euler("z", 0.01, degrees=True).as_matrix(); dt = np.array([0.001, 0, 0])
    Te = [np.eye(4)]
    for k in range(1, len(st)):
        Tg0 = np.eye(4); Tg0[:3, :3] = R[k - 1]; Tg0[:3, 3] = t[k - 1]
        Tg1 = np.eye(4); Tg1[:3, :3] = R[k]; Tg1[:3, 3] = t[k]
        D = np.eye(4); D[:3, :3] = dR; D[:3, 3] = dt
        Te.append(Te[-1] @ (np.linalg.inv(Tg0) @ Tg1) @ D)
    Te = np.array(Te)
    write(tmp_path / "gt.txt", st, t, R); write(tmp_path / "est.txt", st, Te[:, :3, 3], Te[:, :3, :3])
    r = A.rpe(A.read_trajectory(tmp_path / "gt.txt"), A.read_trajectory(tmp_path / "est.txt"), 1, "f")
    assert abs(r["trans_rmse"] - 0.001) < 1e-7 and abs(np.degrees(r["rot_rmse"]) - 0.01) < 1e-5
    rs = A.rpe(A.read_trajectory(tmp_path / "gt.txt"), A.read_trajectory(tmp_path / "est.txt"), 1.0, "s")   # 30 frames
    assert 0.02 < rs["trans_rmse"] < 0.04 and rs["pairs"] > 50


def test_association_and_cli(tmp_path):
    st, t, R = make_traj(50, seed=5)
    write(tmp_path / "gt.txt", st, t, R)
    keep = np.arange(0, 50, 2)                                   # the estimate has every second frame, stamps off by up to 10 ms
    write(tmp_path / "est.txt", st[keep] + 0.01 * np.sin(keep), t[keep], R[keep])
    pairs = A.associate(st, st[keep] + 0.01 * np.sin(keep), 0.02)
    assert [i for i, _ in pairs] == list(keep)
    assert A.associate(st, st[keep] + 100.0, 0.02) == []
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ate.py"), str(tmp_path / "gt.txt"), str(tmp_path / "est.txt"), "--json"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    import json
    d = json.loads(out.stdout)
    assert d["ate"]["pairs"] == 25 and d["ate"]["rmse"] < 1e-6
