"""Tracking a recorded sequence with the batched engine: every lane of the engine tracks one contiguous chunk
(rgbid.dist.chunk_ranges), all chunks advance in lock-step, and the per-frame records of all chunks are composed into one
trajectory.  With torch.distributed initialised, each rank takes its block of chunks (rgbid.dist.rank_chunks) and the
392-byte records {frame id, status, frame-to-frame R | t, covariance} are all-gathered -- the only collective on the path --
through the C-ABI helper over RCCL (rgbid_dist_gather_records) when `comm` is given, else through torch.distributed."""
import numpy as np
import torch

from . import dist as D
from . import engine as E


def track_chunked(ctx, depth, rgb, n_chunks, K, group=None, comm=None, **cfg_kw):
    """depth [T, rows, cols] 16-bit, rgb [T, rows, cols, 3] uint8 CUDA tensors of ONE sequence.
    Returns (R [T,3,3], t [T,3], ranges); the per-frame status / covariance are in track_chunked.last = (status, cov)."""
    T, rows, cols = depth.shape
    ranges = D.chunk_ranges(T, n_chunks)
    distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
    world = torch.distributed.get_world_size(group) if distributed else 1
    rank = torch.distributed.get_rank(group) if distributed else 0
    mine = D.rank_chunks(n_chunks, world, rank)
    lanes = D.lanes_per_rank(n_chunks, world)          # ranks owning one chunk fewer pad with a lane that re-tracks their last chunk (never read)
    owned = mine + [mine[-1] if mine else 0] * (lanes - len(mine))
    L = max(b - a + 1 for a, b in ranges)
    eng = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=lanes, K=K, record_capacity=L, **cfg_kw))
    # Lane-major staging of the whole run, built once and kept alive until the records are read: the engine consumes its inputs
    # asynchronously on its own HIP stream, so per-step temporaries (torch would recycle them on ITS stream) must not be used.
    idx = torch.tensor([[min(ranges[c][0] + j, ranges[c][1]) for c in owned] for j in range(L)], device=depth.device)  # [L, lanes]
    depth_l = depth[idx.reshape(-1)].reshape(L, lanes, rows, cols).contiguous()   # shorter chunks repeat their last frame (unused)
    rgb_l = rgb[idx.reshape(-1)].reshape(L, lanes, rows, cols, 3).contiguous()
    torch.cuda.synchronize(depth.device)
    for j in range(L):
        eng.step(depth_l[j], rgb_l[j])
    packed = D.pack_engine_records(eng, 0, L)         # device: [lanes][L] records
    if comm is not None:
        allb = comm.gather(packed, lanes * L)
        ctx.sync()
        allrec = allb.cpu().numpy().view(D.GATHER_DTYPE).reshape(world, lanes, L)
    else:
        ctx.sync()
        local = packed.cpu().numpy().view(D.GATHER_DTYPE).reshape(lanes, L)
        allrec = D.gather_records_torch(local, group) if distributed else local[None]
    eng.close()
    R, t, st, cov = D.compose_trajectory(allrec, world, n_chunks, ranges)
    track_chunked.last = (st, cov)
    return R, t, ranges
