#!/usr/bin/env python
"""Chunk-sharded tracking of a TUM / ICL-NUIM layout dataset with the batched engine (BASELINE config 4).

    python tools/track_dataset.py <dataset_folder> [--match-file f] [--chunks 8] [--out traj.txt] [--max-frames N]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/track_dataset.py <folder> --chunks 8N

Frames are read with the product's own dataset reader (librgbid_host.so: association files, 16-bit PNG depth x0.2 -> mm), the
sequence is cut into `chunks` contiguous chunks with one frame of overlap, every chunk is one lane of a rank's engine, ranks exchange
the 392-byte per-frame records only (one all-gather over RCCL through librgbid_dist.so), rank 0 writes the trajectory in the TUM format (`stamp tx ty tz qx qy qz qw`)."""
import argparse
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before the HIP runtime comes up: the host driver only supports dmabuf IPC (RCCL across processes)
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rgbid-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("folder")
    ap.add_argument("--match-file", default="")
    ap.add_argument("--chunks", type=int, default=8)
    ap.add_argument("--out", default="trajectory.txt")
    ap.add_argument("--max-frames", type=int, default=-1)
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with a single rank")
    ap.add_argument("--K", type=float, nargs=4, default=[525.0, 525.0, 319.5, 239.5], help="fx fy cx cy (tools/evaluation.cpp:64-67)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "needs a HIP device (there is no CPU path)"
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from rgbid import device, sequence, tum

    ds = tum.Dataset(args.folder, args.match_file)
    n = len(ds) if args.max_frames < 0 else min(len(ds), args.max_frames)
    frames, stamps = [], []
    for k in range(n):
        g = ds.grab(k, args.rows, args.cols)
        if g is None:                      # unreadable pair: the reference's playback skips it too
            continue
        frames.append(g); stamps.append(ds.stamp(k))
    assert len(frames) >= args.chunks + 1, "need at least chunks + 1 readable frames"
    depth = torch.from_numpy(np.stack([f[0] for f in frames]).view(np.int16)).cuda()
    rgb = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    ctx = device.Context(local_rank)
    ctx.set_async(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    comm = None
    if use_dist:
        from rgbid import dist as D
        try:
            comm = D.Comm(ctx, world, rank)          # the C-ABI RCCL helper (librgbid_dist.so); the id travels through torch's store
        except Exception as e:                       # transport problem: say so, gather through torch.distributed instead
            sys.stderr.write(f"[track_dataset] WARNING: C-ABI RCCL communicator failed ({e}); gathering through torch.distributed\n")
    R, t, ranges = sequence.track_chunked(ctx, depth, rgb, args.chunks, tuple(args.K), comm=comm, use_graph=0)
    el = time.perf_counter() - t0
    if comm is not None:
        comm.close()
    if rank == 0:
        tum.write_trajectory(args.out, stamps, R, t)
        print(f"{len(frames)} frames in {args.chunks} chunks on {world} GPU(s): {el:.3f} s ({len(frames) / el:.1f} frames/s incl. engine set-up) -> {args.out}")
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
