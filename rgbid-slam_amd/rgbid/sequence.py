"""Tracking a recorded sequence with the batched engine: every lane of the engine tracks one contiguous chunk
(rgbid.dist.chunk_ranges), all chunks advance in lock-step, and the chunk-relative poses are composed into one
trajectory.  With torch.distributed initialised, each rank takes its block of chunks (rgbid.dist.rank_chunks) and the
pose records are all-gathered (the only collective on the path)."""
import numpy as np
import torch

from . import dist as D
from . import engine as E


def track_chunked(ctx, depth, rgb, n_chunks, K, group=None, **cfg_kw):
    """depth [T, rows, cols] 16-bit, rgb [T, rows, cols, 3] uint8 CUDA tensors of ONE sequence.
    Returns (R [T,3,3], t [T,3], ranges)."""
    T, rows, cols = depth.shape
    ranges = D.chunk_ranges(T, n_chunks)
    distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
    world = torch.distributed.get_world_size(group) if distributed else 1
    rank = torch.distributed.get_rank(group) if distributed else 0
    assert n_chunks % world == 0, "equal chunk count per rank keeps the gathered record tensor rectangular"
    mine = D.rank_chunks(n_chunks, world, rank)
    L = max(b - a + 1 for a, b in ranges)
    eng = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=len(mine), K=K, record_capacity=L, **cfg_kw))
    # Lane-major staging of the whole run, built once and kept alive until the records are read: the engine consumes its inputs
    # asynchronously on its own HIP stream, so per-step temporaries (torch would recycle them on ITS stream) must not be used.
    idx = torch.tensor([[min(ranges[c][0] + j, ranges[c][1]) for c in mine] for j in range(L)], device=depth.device)  # [L, lanes]
    depth_l = depth[idx.reshape(-1)].reshape(L, len(mine), rows, cols).contiguous()   # shorter chunks repeat their last frame (unused)
    rgb_l = rgb[idx.reshape(-1)].reshape(L, len(mine), rows, cols, 3).contiguous()
    torch.cuda.synchronize(depth.device)
    for j in range(L):
        eng.step(depth_l[j], rgb_l[j])
    rec = eng.records()
    eng.close()
    local = np.full((len(mine), L, 12), np.nan)
    for i, c in enumerate(mine):
        n = ranges[c][1] - ranges[c][0] + 1
        local[i, :n, :9] = rec["R"][:n, i].reshape(n, 9)
        local[i, :n, 9:] = rec["t"][:n, i]
    allp = D.gather_pose_records(local, group) if distributed else local
    R, t = D.compose_trajectory(allp, ranges)
    return R, t, ranges
