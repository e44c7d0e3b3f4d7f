"""Multi-process CPU tests (gloo, world_size 2 and 3) of the N>1 path: chunk sharding, the exchange of the 392-byte per-frame records
(frame id, status, frame-to-frame R | t, 6x6 covariance -- SURVEY.md section 8e) and the trajectory composition, all through the C
functions of librgbid_dist.so, launched with the SAME launcher bench.py --gpus N uses (rgbid.dist.spawn_local ->
python -m torch.distributed.run).  The data path has no other collective, so this is everything ranks exchange.  RCCL itself needs
GPUs: on CPU the records travel through torch.distributed (gloo); the RCCL transport is covered on the GPU box (tests/test_gpu_dist.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from rgbid import dist as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_chunk_ranges_cover_sequence():
    for F, n in [(9, 8), (100, 8), (101, 16), (17, 2), (5, 1)]:
        rg = D.chunk_ranges(F, n)
        assert rg[0][0] == 0 and rg[-1][1] == F - 1 and len(rg) == n
        assert all(rg[i][1] == rg[i + 1][0] for i in range(n - 1))
        lens = [b - a for a, b in rg]
        assert max(lens) - min(lens) <= 1 and min(lens) >= 1
    owned = sum((D.rank_chunks(16, 3, r) for r in range(3)), [])
    assert owned == list(range(16))
    with pytest.raises(Exception):
        D.chunk_ranges(4, 4)            # needs n_frames >= n_chunks + 1


def test_compose_single_rank_matches_reference_composition():
    """world = 1: records laid out chunk-major; the composed poses equal the direct product of the frame-to-frame motions"""
    from scipy.spatial.transform import Rotation
    r = np.random.default_rng(5)
    F, n_chunks = 23, 4
    dR = [np.eye(3)] + [Rotation.from_rotvec(0.03 * r.standard_normal(3)).as_matrix() for _ in range(F - 1)]
    dt = [np.zeros(3)] + [0.02 * r.standard_normal(3) for _ in range(F - 1)]
    Rg, tg = [np.eye(3)], [np.zeros(3)]
    for k in range(1, F):
        tg.append(Rg[-1] @ dt[k] + tg[-1]); Rg.append(Rg[-1] @ dR[k])
    ranges = D.chunk_ranges(F, n_chunks)
    L = max(b - a + 1 for a, b in ranges)
    rec = np.zeros((1, n_chunks, L), D.GATHER_DTYPE)
    for c, (a, b) in enumerate(ranges):
        for j in range(b - a + 1):
            rec[0, c, j]["R"] = dR[a + j] if j else np.eye(3)
            rec[0, c, j]["t"] = dt[a + j] if j else 0
            rec[0, c, j]["status"] = 1 if j else 16
            rec[0, c, j]["frame_id"] = j
    R, t, st, cov = D.compose_trajectory(rec, 1, n_chunks, ranges)
    assert np.abs(R - np.array(Rg)).max() < 1e-14 and np.abs(t - np.array(tg)).max() < 1e-14
    assert st[0] == 16 and (st[1:] == 1).all()
    # a lane that did not track its chunk (zeroed / padded / shifted records) is refused, not composed silently
    bad = rec.copy(); bad[0, 2]["frame_id"] = 0
    with pytest.raises(Exception):
        D.compose_trajectory(bad, 1, n_chunks, ranges)
    lost = rec.copy(); lost[0, 1, 3]["frame_id"] = 2       # a frame that stayed lost repeats its id: accepted
    lost[0, 1, 4:]["frame_id"] -= 1
    D.compose_trajectory(lost, 1, n_chunks, ranges)


@pytest.mark.parametrize("world,chunks", [(2, 8), (2, 7), (3, 7)])
def test_record_gather_and_composition_multi_rank(world, chunks):
    """N ranks through the bench launcher; uneven chunk ownership (7 chunks on 2 or 3 ranks) pads with lanes nobody reads"""
    p = D.spawn_local(world, [os.path.join(ROOT, "tools", "dist_selftest.py"), "--backend", "gloo", "--frames", "41", "--chunks", str(chunks),
                              "--expect-world", str(world)], timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert res["world"] == world and res["compose_err"] < 1e-12 and res["cov_ok"] and res["status_ok"] and res["tcp_rendezvous_ok"], res


def test_bench_gpus_flag_fails_loudly_without_devices():
    """`python bench.py --gpus 2` must not quietly run one rank: with fewer than 2 HIP devices it exits non-zero with a clear message
    (here: no device at all)."""
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0
    assert "--gpus 2" in p.stderr and "device" in p.stderr, p.stderr[-500:]
    assert "{" not in p.stdout          # no result line
