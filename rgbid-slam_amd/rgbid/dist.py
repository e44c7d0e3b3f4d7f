"""Multi-GPU sharding of a recorded sequence -- Python harness over the C-ABI of include/rgbid_dist.h.

The tracker is sequential inside a sequence, so the independent unit is a CHUNK: a contiguous sub-sequence tracked with
the full keyframe / fusion logic starting from identity.  A sequence of F frames is cut into `n_chunks` chunks that
overlap by one frame (chunk c ends on the frame chunk c+1 starts on); every chunk is one lane of one GPU's batched
engine; ranks exchange ONLY the 392-byte per-frame records {frame id, status, frame-to-frame R | t, 6x6 covariance} (one
all-gather over RCCL / xGMI), and every rank composes the global trajectory T_w,k = T_w,k-1 * dT_k.  No image data crosses GPUs.

Partitioning, the record layout, the RCCL all-gather (`Comm`) and the composition are the C functions of librgbid_dist.so -- what a
C++ host calls; this module only binds them.  One process per GPU.  Process-group plumbing (launch, barrier, max-over-ranks
timing) is torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests, where the records travel through
torch.distributed instead of RCCL because RCCL needs GPUs).
"""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np

from . import _lib

GATHER_DTYPE = np.dtype([("frame_id", np.int32), ("status", np.int32), ("R", np.float64, (3, 3)), ("t", np.float64, (3,)),
                         ("cov", np.float64, (6, 6))], align=True)
assert GATHER_DTYPE.itemsize == 392

DIST_LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "librgbid_dist.so")
DIST_EXPORTS = ["rgbid_dist_chunk_ranges", "rgbid_dist_rank_chunks", "rgbid_dist_new_id", "rgbid_dist_exchange_id", "rgbid_dist_broadcast_bytes",
                "rgbid_dist_allgather_bytes_tcp", "rgbid_dist_init", "rgbid_dist_destroy", "rgbid_dist_world", "rgbid_dist_rank", "rgbid_dist_gather_records",
                "rgbid_dist_barrier", "rgbid_dist_compose_trajectory", "rgbid_dist_renumber_warmed_chunks", "rgbid_dist_track_sequence"]
TRACK_SEQUENCE_BIN = os.path.join(os.path.dirname(os.path.dirname(_lib.LIB_PATH)), "bin", "rgbid_track_sequence")
EXCHANGE_RCCL, EXCHANGE_TCP = 0, 1
_dl = None


def dlib():
    global _dl
    if _dl is None:
        if not os.path.exists(DIST_LIB_PATH):
            raise _lib.RgbidError(f"{DIST_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib.lib()                       # librgbid_hip.so first (librgbid_dist.so links against it)
        _dl = C.CDLL(DIST_LIB_PATH)
    return _dl


def check(err):
    if err != 0:
        if err <= -100:
            raise _lib.RgbidError(f"RCCL error {-100 - err} (ncclResult_t)")
        if err == -90:
            raise _lib.RgbidError("rendezvous (socket) failure")
        _lib.check(err)


def _ip(a):
    return a.ctypes.data_as(C.c_void_p)


def chunk_ranges(n_frames, n_chunks):
    """[(first, last)] inclusive frame ranges, consecutive chunks share one frame; lengths differ by at most 1."""
    first = np.zeros(n_chunks, np.int32); last = np.zeros(n_chunks, np.int32)
    check(dlib().rgbid_dist_chunk_ranges(int(n_frames), int(n_chunks), _ip(first), _ip(last)))
    return [(int(a), int(b)) for a, b in zip(first, last)]


def rank_chunks(n_chunks, world, rank):
    """chunk ids owned by `rank`: contiguous blocks, so neighbouring chunks mostly live on the same GPU."""
    s, n = C.c_int(), C.c_int()
    check(dlib().rgbid_dist_rank_chunks(int(n_chunks), int(world), int(rank), C.byref(s), C.byref(n)))
    return list(range(s.value, s.value + n.value))


def lanes_per_rank(n_chunks, world):
    return -(-n_chunks // world)


def compose_trajectory(all_records, world, n_chunks, ranges):
    """all_records: GATHER_DTYPE array [world, lanes_per_rank, chunk_len] (the gathered buffer).
    Returns global (R [F,3,3], t [F,3], status [F], cov [F,6,6]); frame 0 = identity."""
    a = np.ascontiguousarray(all_records)
    assert a.dtype == GATHER_DTYPE and a.ndim == 3 and a.shape[0] == world
    F = ranges[-1][1] + 1
    first = np.array([r[0] for r in ranges], np.int32); last = np.array([r[1] for r in ranges], np.int32)
    R = np.zeros((F, 3, 3)); t = np.zeros((F, 3)); st = np.zeros(F, np.int32); cov = np.zeros((F, 6, 6))
    check(dlib().rgbid_dist_compose_trajectory(_ip(a), int(world), int(a.shape[1]), int(n_chunks), int(a.shape[2]), _ip(first), _ip(last),
                                                _ip(R), _ip(t), _ip(st), _ip(cov)))
    return R, t, st, cov


def renumber_warmed_chunks(all_records, world, n_chunks, ranges, warmup_frames):
    """in place: the ids of chunks that ran a warm-up, renumbered from their own head (rgbid_dist_renumber_warmed_chunks)"""
    a = all_records
    assert a.dtype == GATHER_DTYPE and a.ndim == 3 and a.shape[0] == world and a.flags["C_CONTIGUOUS"]
    first = np.array([r[0] for r in ranges], np.int32); last = np.array([r[1] for r in ranges], np.int32)
    check(dlib().rgbid_dist_renumber_warmed_chunks(_ip(a), int(world), int(a.shape[1]), int(n_chunks), int(a.shape[2]), _ip(first), _ip(last), int(warmup_frames)))
    return a


class Comm:
    """RCCL communicator bound to an rgbid context (rgbid_dist_init); the unique id is created on rank 0 and shipped either through
    the library's own TCP rendezvous (`addr`, `port`) or through an initialised torch.distributed group (its store / broadcast)."""

    def __init__(self, ctx, world, rank, addr=None, port=None, group=None):
        import torch
        self.ctx = ctx
        ident = (C.c_char * 128)()
        L = dlib()
        if addr is not None:
            check(L.rgbid_dist_exchange_id(addr.encode(), int(port), int(world), int(rank), ident))
        else:
            import torch.distributed as dist
            if rank == 0:
                check(L.rgbid_dist_new_id(ident))
            if world > 1:
                box = [bytes(ident)]
                dist.broadcast_object_list(box, src=0, group=group)
                C.memmove(ident, box[0], 128)
        self._h = C.c_void_p()
        check(L.rgbid_dist_init(C.byref(self._h), ctx._h, ident, int(world), int(rank)))
        self._torch = torch

    def world(self):
        return dlib().rgbid_dist_world(self._h)

    def rank(self):
        return dlib().rgbid_dist_rank(self._h)

    def gather(self, local_dev, n_local):
        """local_dev: CUDA uint8 tensor holding n_local records; returns a CUDA uint8 tensor with world * n_local records (async on the ctx stream)"""
        out = self._torch.empty(self.world() * n_local * 392, dtype=self._torch.uint8, device=local_dev.device)
        check(dlib().rgbid_dist_gather_records(self._h, C.c_void_p(local_dev.data_ptr()), int(n_local), C.c_void_p(out.data_ptr())))
        return out

    def barrier(self):
        check(dlib().rgbid_dist_barrier(self._h))

    def close(self):
        if self._h:
            dlib().rgbid_dist_destroy(self._h)
            self._h = None


class SeqReport(C.Structure):
    _fields_ = [("lanes", C.c_int), ("chunk_len", C.c_int), ("n_chunks", C.c_int), ("world", C.c_int), ("rccl_ranks", C.c_int),
                ("setup_ms", C.c_double), ("track_ms", C.c_double), ("gather_ms", C.c_double), ("compose_ms", C.c_double), ("total_ms", C.c_double),
                ("staged_bytes", C.c_ulonglong), ("engine_bytes", C.c_ulonglong)]


def track_sequence(ctx, engine_cfg, depth_host, rgb_host, n_chunks, world=1, rank=0, exchange=EXCHANGE_RCCL, master_addr="127.0.0.1", master_port=0,
                   inject=None, n_frames=None, warmup_frames=0):
    """rgbid_dist_track_sequence: the C++ sharded-sequence driver (csrc/dist.cpp) on host frames depth [T, rows, cols] uint16 / rgb [T, rows, cols, 3]
    uint8 (numpy arrays, or CPU torch tensors -- pinned ones upload asynchronously).  inject (GATHER_DTYPE [n_chunks, chunk_len]) + n_frames: compose
    the given per-chunk records instead of running the engine (ctx / frames may be None).  Returns (R [T,3,3], t [T,3], status [T], cov [T,6,6], report)."""
    from .engine import EngineConfig

    class SeqConfig(C.Structure):
        _fields_ = [("engine", EngineConfig), ("n_chunks", C.c_int), ("world", C.c_int), ("rank", C.c_int), ("exchange", C.c_int),
                    ("master_addr", C.c_char_p), ("master_port", C.c_int), ("inject_chunk_len", C.c_int), ("warmup_frames", C.c_int)]

    def ptr(a):
        if a is None:
            return None
        return C.c_void_p(a.data_ptr()) if hasattr(a, "data_ptr") else a.ctypes.data_as(C.c_void_p)

    cfg = SeqConfig()
    C.memmove(C.byref(cfg.engine), C.byref(engine_cfg), C.sizeof(EngineConfig))
    cfg.n_chunks = int(n_chunks); cfg.world = int(world); cfg.rank = int(rank); cfg.exchange = int(exchange)
    cfg.master_addr = master_addr.encode(); cfg.master_port = int(master_port)
    cfg.warmup_frames = int(warmup_frames)
    T = int(n_frames) if n_frames is not None else int(depth_host.shape[0])
    inj = None
    if inject is not None:
        inject = np.ascontiguousarray(inject)
        assert inject.dtype == GATHER_DTYPE and inject.ndim == 2 and inject.shape[0] == n_chunks
        cfg.inject_chunk_len = int(inject.shape[1])
        inj = _ip(inject)
    R = np.zeros((T, 3, 3)); t = np.zeros((T, 3)); st = np.zeros(T, np.int32); cov = np.zeros((T, 6, 6))
    rep = SeqReport()
    check(dlib().rgbid_dist_track_sequence(ctx._h if ctx is not None else None, C.byref(cfg), ptr(depth_host), ptr(rgb_host), T, inj,
                                           _ip(R), _ip(t), _ip(st), _ip(cov), C.byref(rep)))
    return R, t, st, cov, {n: getattr(rep, n) for n, _ in SeqReport._fields_}


def pack_engine_records(eng, first_step, n_steps):
    """device-side pack of an engine's pose-record ring into [lanes][n_steps] gather records -> CUDA uint8 tensor"""
    import torch
    out = torch.empty(eng.cfg.lanes * n_steps * 392, dtype=torch.uint8, device="cuda")
    _lib.check(_lib.lib().rgbid_engine_pack_gather_records(eng._h, int(first_step), int(n_steps), C.c_void_p(out.data_ptr())))
    return out


def gather_records_torch(local, group=None):
    """the same exchange through torch.distributed (CPU tests with gloo, and the cross-check of the RCCL path): local is a GATHER_DTYPE
    array [lanes_per_rank, chunk_len], identical shape on every rank.  Returns [world, lanes_per_rank, chunk_len]."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(np.ascontiguousarray(local).view(np.uint8).reshape(-1).copy()).to(dev)
    out = torch.empty(world * mine.numel(), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu().numpy().view(GATHER_DTYPE).reshape((world,) + tuple(local.shape))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def spawn_local(n, argv, env=None, timeout=None):
    """Launch `argv` (a script + its arguments) as n ranks of ONE node, one process per GPU -- the same command line the driver uses:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port P script args...
    Returns the CompletedProcess (stdout/stderr captured as text)."""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port())] + list(argv)
    return subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
