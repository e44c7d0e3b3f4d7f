"""Multi-process CPU test (gloo, world_size 2) of the N>1 path: chunk sharding, the pose all_gather and trajectory
composition.  The data path has no other collective, so this is everything ranks exchange."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rgbid import dist as D


def _rand_chain(n, seed):
    from scipy.spatial.transform import Rotation
    r = np.random.default_rng(seed)
    R = [np.eye(3)]; t = [np.zeros(3)]
    for _ in range(1, n):
        dR = Rotation.from_rotvec(0.02 * r.standard_normal(3)).as_matrix(); dt = 0.02 * r.standard_normal(3)
        t.append(R[-1] @ dt + t[-1]); R.append(R[-1] @ dR)
    return np.array(R), np.array(t)


def test_chunk_ranges_cover_sequence():
    for F, n in [(9, 8), (100, 8), (101, 16), (17, 2), (5, 1)]:
        rg = D.chunk_ranges(F, n)
        assert rg[0][0] == 0 and rg[-1][1] == F - 1 and len(rg) == n
        assert all(rg[i][1] == rg[i + 1][0] for i in range(n - 1))
        lens = [b - a for a, b in rg]
        assert max(lens) - min(lens) <= 1 and min(lens) >= 1
    owned = sum((D.rank_chunks(16, 3, r) for r in range(3)), [])
    assert owned == list(range(16))


def _worker(rank, world, port, F, n_chunks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Rg, tg = _rand_chain(F, 11)
    ranges = D.chunk_ranges(F, n_chunks)
    L = max(b - a + 1 for a, b in ranges)
    mine = D.rank_chunks(n_chunks, world, rank)
    local = np.full((len(mine), L, 12), np.nan)
    for i, c in enumerate(mine):                      # what each rank's engine lanes would produce: chunk-relative poses
        a, b = ranges[c]
        for j in range(b - a + 1):
            Rc = Rg[a].T @ Rg[a + j]; tc = Rg[a].T @ (tg[a + j] - tg[a])
            local[i, j, :9] = Rc.reshape(9); local[i, j, 9:] = tc
    allp = D.gather_pose_records(local)
    R, t = D.compose_trajectory(allp, ranges)
    err = max(np.abs(R - Rg).max(), np.abs(t - tg).max())
    if rank == 0:
        q.put(float(err))
    dist.barrier()
    dist.destroy_process_group()


def test_pose_gather_and_composition_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 41, 8, q)) for r in range(2)]
    for p in procs:
        p.start()
    err = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-12, err
