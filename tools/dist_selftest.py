#!/usr/bin/env python
"""Self-test of the N>1 path, launched as N ranks by rgbid.dist.spawn_local (the launcher bench.py --gpus N uses):
chunk partitioning, the 392-byte record exchange and the trajectory composition of librgbid_dist.so.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/dist_selftest.py [--backend gloo|nccl] [--frames F] [--chunks C]

backend gloo (CPU, used by tests/test_dist_cpu.py): the records travel through torch.distributed; backend nccl (one GPU per rank): through
the C-ABI helper over RCCL (rgbid_dist_gather_records), cross-checked against torch.distributed's all_gather.  Both also run the
library's own TCP rendezvous (rgbid_dist_broadcast_bytes).  Rank 0 prints one JSON line."""
import argparse
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before the HIP runtime comes up: the host driver only supports dmabuf IPC (RCCL across processes)
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rgbid-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist


def rand_chain(n, seed):
    from scipy.spatial.transform import Rotation
    r = np.random.default_rng(seed)
    R = [np.eye(3)]; t = [np.zeros(3)]
    for _ in range(1, n):
        dR = Rotation.from_rotvec(0.02 * r.standard_normal(3)).as_matrix(); dt = 0.02 * r.standard_normal(3)
        t.append(R[-1] @ dt + t[-1]); R.append(R[-1] @ dR)
    return np.array(R), np.array(t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--frames", type=int, default=41)
    ap.add_argument("--chunks", type=int, default=7)
    ap.add_argument("--expect-world", type=int, default=0)
    args = ap.parse_args()
    from rgbid import dist as D
    world = int(os.environ["WORLD_SIZE"]); rank = int(os.environ["RANK"]); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group("gloo")
    assert dist.get_world_size() == world and (args.expect_world in (0, world)), (dist.get_world_size(), world, args.expect_world)
    F, n_chunks = args.frames, args.chunks
    Rg, tg = rand_chain(F, 11)
    ranges = D.chunk_ranges(F, n_chunks)
    L = max(b - a + 1 for a, b in ranges)
    lanes = D.lanes_per_rank(n_chunks, world)
    mine = D.rank_chunks(n_chunks, world, rank)
    # what this rank's engine lanes would produce: frame-to-frame odometry of every frame of its chunks
    local = np.zeros((lanes, L), D.GATHER_DTYPE)
    local["R"] = np.eye(3); local["frame_id"] = -1
    rng = np.random.default_rng(100 + rank)
    for i, c in enumerate(mine):
        a, b = ranges[c]
        for j in range(b - a + 1):
            rec = local[i, j]
            rec["frame_id"] = j
            rec["status"] = 16 if j == 0 else 1
            if j:
                rec["R"] = Rg[a + j - 1].T @ Rg[a + j]; rec["t"] = Rg[a + j - 1].T @ (tg[a + j] - tg[a + j - 1])
                rec["cov"] = np.eye(6) * (a + j)
    for i in range(len(mine), lanes):            # padding lanes carry garbage nobody may read
        local[i]["t"] = rng.standard_normal((L, 3))
    res = {}
    if args.backend == "nccl":
        from rgbid import device
        ctx = device.Context(local_rank)
        comm = D.Comm(ctx, world, rank)          # unique id through torch.distributed's broadcast
        assert comm.world() == world and comm.rank() == rank
        mine_dev = torch.from_numpy(local.view(np.uint8).reshape(-1).copy()).cuda()
        allb = comm.gather(mine_dev, lanes * L)
        comm.barrier()
        allrec = allb.cpu().numpy().view(D.GATHER_DTYPE).reshape(world, lanes, L)
        ref = D.gather_records_torch(local)
        res["rccl_equals_torch_gather"] = bool(allrec.tobytes() == ref.tobytes())
        res["rccl_world"] = comm.world()
        comm.close(); ctx.close()
    else:
        allrec = D.gather_records_torch(local)
    R, t, st, cov = D.compose_trajectory(allrec, world, n_chunks, ranges)
    err = float(max(np.abs(R - Rg).max(), np.abs(t - tg).max()))
    cov_ok = bool(all(np.array_equal(cov[k], np.eye(6) * k) for k in range(1, F)))
    # the library's own rendezvous transport (what a C++ host uses to ship the RCCL id): rank 0's bytes reach everyone
    blob = (C.c_char * 128)()
    if rank == 0:
        C.memmove(blob, bytes(range(128)), 128)
    port = [D.free_port() if rank == 0 else 0]
    dist.broadcast_object_list(port, src=0)
    D.check(D.dlib().rgbid_dist_broadcast_bytes(b"127.0.0.1", int(port[0]), world, rank, blob, C.c_size_t(128)))
    tcp_ok = torch.tensor([int(bytes(blob) == bytes(range(128)))])
    if args.backend == "nccl":
        tcp_ok = tcp_ok.cuda()
    dist.all_reduce(tcp_ok, op=dist.ReduceOp.MIN)
    dist.barrier()
    if rank == 0:
        res.update(world=world, backend=args.backend, frames=F, chunks=n_chunks, lanes_per_rank=lanes, compose_err=err, cov_ok=cov_ok,
                   status_ok=bool(st[0] == 16 and (st[1:] == 1).all()), tcp_rendezvous_ok=bool(int(tcp_ok.item())))
        print(json.dumps(res))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
