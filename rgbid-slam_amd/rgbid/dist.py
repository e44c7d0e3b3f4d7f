"""Multi-GPU sharding of a recorded sequence.

The tracker is sequential inside a sequence, so the independent unit is a CHUNK: a contiguous sub-sequence tracked with
the full keyframe / fusion logic starting from identity.  A sequence of F frames is cut into `n_chunks` chunks that
overlap by one frame (chunk c ends on the frame chunk c+1 starts on); every chunk is one lane of one GPU's batched
engine; ranks exchange ONLY pose records (one all_gather over RCCL / xGMI, ~0.9 KB per frame), and every rank (or rank 0)
composes the global trajectory T_w,k = T_w,start(c) * T_chunk(k).  No image data ever crosses GPUs.

One process per GPU: init torch.distributed with backend "nccl" (= RCCL on ROCm); the CPU tests use "gloo".
"""
import numpy as np
import torch
import torch.distributed as dist


def chunk_ranges(n_frames, n_chunks):
    """[(first, last)] inclusive frame ranges, consecutive chunks share one frame; lengths differ by at most 1."""
    assert n_frames >= n_chunks + 1 and n_chunks >= 1
    steps = n_frames - 1                       # frame-to-frame transitions to distribute
    base, extra = divmod(steps, n_chunks)
    out, s = [], 0
    for c in range(n_chunks):
        n = base + (1 if c < extra else 0)
        out.append((s, s + n))
        s += n
    return out


def rank_chunks(n_chunks, world, rank):
    """chunk ids owned by `rank`: contiguous blocks, so neighbouring chunks mostly live on the same GPU."""
    per, extra = divmod(n_chunks, world)
    start = rank * per + min(rank, extra)
    return list(range(start, start + per + (1 if rank < extra else 0)))


def gather_pose_records(local, group=None):
    """all_gather of fixed-size pose records.  local: float64 array [n_local_chunks, chunk_len, 12] (R row-major | t),
    identical shape on every rank (pad shorter chunks with NaN).  Returns [world * n_local_chunks, chunk_len, 12]."""
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.as_tensor(np.ascontiguousarray(local), dtype=torch.float64).to(dev)
    out = torch.empty((world,) + tuple(mine.shape), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out.view(-1), mine.view(-1), group=group)
    return out.reshape((-1,) + tuple(mine.shape[1:])).cpu().numpy()


def compose_trajectory(chunk_poses, ranges):
    """chunk_poses[c][j] = (R, t) pose of the chunk's j-th frame relative to the chunk's first frame (identity at j = 0).
    Returns global (R[F,3,3], t[F,3]) with frame 0 = identity."""
    F = ranges[-1][1] + 1
    R = np.zeros((F, 3, 3)); t = np.zeros((F, 3))
    Rw, tw = np.eye(3), np.zeros(3)
    for c, (a, b) in enumerate(ranges):
        for j in range(b - a + 1):
            Rc = np.asarray(chunk_poses[c][j][:9]).reshape(3, 3); tc = np.asarray(chunk_poses[c][j][9:12])
            R[a + j] = Rw @ Rc
            t[a + j] = Rw @ tc + tw
        Rw, tw = R[b].copy(), t[b].copy()
    return R, t
