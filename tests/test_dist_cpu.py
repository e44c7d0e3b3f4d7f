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


@pytest.mark.parametrize("world,chunks,frames", [(2, 8, 41), (2, 7, 41), (3, 7, 41), (8, 248, 2500)])   # the last: the shape bench.py --gpus 8 runs config 4 in
def test_record_gather_and_composition_multi_rank(world, chunks, frames):
    """N ranks through the bench launcher; uneven chunk ownership (7 chunks on 2 or 3 ranks) pads with lanes nobody reads"""
    p = D.spawn_local(world, [os.path.join(ROOT, "tools", "dist_selftest.py"), "--backend", "gloo", "--frames", str(frames), "--chunks", str(chunks),
                              "--expect-world", str(world)], timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert res["world"] == world and res["compose_err"] < 1e-13 * frames and res["cov_ok"] and res["status_ok"] and res["tcp_rendezvous_ok"], res


def _chain_records(F, n_chunks, seed=11):
    """what the engine lanes of a sharded run would produce for a random ground-truth chain: [n_chunks][L] records + the chain"""
    from scipy.spatial.transform import Rotation
    r = np.random.default_rng(seed)
    Rg, tg = [np.eye(3)], [np.zeros(3)]
    for _ in range(1, F):
        dR = Rotation.from_rotvec(0.02 * r.standard_normal(3)).as_matrix(); dt = 0.02 * r.standard_normal(3)
        tg.append(Rg[-1] @ dt + tg[-1]); Rg.append(Rg[-1] @ dR)
    Rg, tg = np.array(Rg), np.array(tg)
    ranges = D.chunk_ranges(F, n_chunks)
    L = max(b - a + 1 for a, b in ranges)
    rec = np.zeros((n_chunks, L), D.GATHER_DTYPE)
    rec["R"] = np.eye(3); rec["frame_id"] = -1
    for c, (a, b) in enumerate(ranges):
        for j in range(b - a + 1):
            rec[c, j]["frame_id"] = j; rec[c, j]["status"] = 16 if j == 0 else 1
            if j:
                rec[c, j]["R"] = Rg[a + j - 1].T @ Rg[a + j]; rec[c, j]["t"] = Rg[a + j - 1].T @ (tg[a + j] - tg[a + j - 1])
                rec[c, j]["cov"] = np.eye(6) * (a + j)
    return rec, Rg, tg, ranges, L


@pytest.mark.parametrize("world,chunks,F", [(1, 5, 41), (2, 8, 41), (2, 7, 41), (3, 7, 41), (8, 248, 2500)])   # the last: the shape of bench.py --gpus 8 (31 chunks of 11 frames per GPU)
def test_cpp_sequence_driver_partition_exchange_compose(world, chunks, F, tmp_path):
    """The C++ sharded-sequence driver (tools/rgbid_track_sequence.cpp -> rgbid_dist_track_sequence) as `world` PROCESSES with the per-chunk records
    injected (no GPU here): its partition (uneven ownership, padded lanes), its exchange (the library's TCP rendezvous: hello / nonce, all-gather)
    and its composition give the trajectory file rgbid/dist.py + rgbid/tum.py give for the same records, byte for byte.  RCCL itself needs one
    GPU per rank and stays unmeasured here (tests/test_gpu_dist.py covers world 1; tools/dist_selftest.py --backend nccl a multi-GPU node)."""
    from rgbid import tum
    rec, Rg, tg, ranges, L = _chain_records(F, chunks)
    inj = tmp_path / "records.bin"
    rec.tofile(str(inj))
    port = D.free_port()
    env = dict(os.environ, RGBID_DIST_NONCE="12345", RGBID_DIST_TIMEOUT_S="60")
    outs = [tmp_path / f"traj_{r}.txt" for r in range(world)]
    procs = [subprocess.Popen([D.TRACK_SEQUENCE_BIN, "-inject", str(inj), "-frames", str(F), "-chunks", str(chunks), "-world", str(world), "-rank", str(r),
                               "-exchange", "tcp", "-master_addr", "localhost", "-master_port", str(port), "-out", str(outs[r])],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in reversed(range(world))]   # rank 0 last: the others retry
    res = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [r[1][-500:] for r in res]
    rep = json.loads([l for l in res[-1][0].splitlines() if l.startswith("{")][-1])
    assert rep["world"] == world and rep["chunks"] == chunks and rep["lanes_per_gpu"] == D.lanes_per_rank(chunks, world) and rep["chunk_len"] == L
    # the Python harness on the same records: the layout a gather would produce, composed and written by the same library functions
    lanes = D.lanes_per_rank(chunks, world)
    allrec = np.zeros((world, lanes, L), D.GATHER_DTYPE)
    for r in range(world):
        for i, c in enumerate(D.rank_chunks(chunks, world, r)):
            allrec[r, i] = rec[c]
    R, t, st, cov = D.compose_trajectory(allrec, world, chunks, ranges)
    assert np.abs(R - Rg).max() < 1e-13 * F and np.abs(t - tg).max() < 1e-13 * F      # the composed chain against the generating chain: rounding of F products
    ref = tmp_path / "ref.txt"
    tum.write_trajectory(str(ref), [k / 30.0 for k in range(F)], R, t)
    assert outs[0].read_bytes() == ref.read_bytes()
    assert not any(o.exists() for o in outs[1:])          # rank 0 writes


def test_sequence_driver_refuses_injected_records_of_another_chunk_length():
    """rgbid_dist_track_sequence reads inject[chunk * chunk_len ...]: a buffer laid out for another chunk length (a records file made for another
    -frames) must be refused, not read out of bounds or mis-strided"""
    from rgbid import engine as E
    from rgbid._lib import RgbidError
    F, chunks = 25, 4
    good, _, _, _, L = _chain_records(F, chunks)
    D.track_sequence(None, E.default_config(rows=48, cols=64, lanes=1), None, None, chunks, inject=good, n_frames=F)
    for bad_len in (L - 1, L + 1, 2 * L):
        bad = np.zeros((chunks, bad_len), D.GATHER_DTYPE)
        with pytest.raises(RgbidError):
            D.track_sequence(None, E.default_config(rows=48, cols=64, lanes=1), None, None, chunks, inject=bad, n_frames=F)


def test_sequence_driver_refuses_a_warm_up_on_injected_records():
    """rgbid_seq_config.warmup_frames (round 5) makes every chunk but the first track frames before its own first one: injected records carry no such
    frames, so the combination is refused (and w = 0 is the driver as it was); absurd values are refused too"""
    from rgbid import engine as E
    from rgbid._lib import RgbidError
    F, chunks = 25, 4
    good, _, _, _, L = _chain_records(F, chunks)
    R0, t0, _, _, _ = D.track_sequence(None, E.default_config(rows=48, cols=64, lanes=1), None, None, chunks, inject=good, n_frames=F, warmup_frames=0)
    for w in (1, 4, -1, 1000):
        with pytest.raises(RgbidError):
            D.track_sequence(None, E.default_config(rows=48, cols=64, lanes=1), None, None, chunks, inject=good, n_frames=F, warmup_frames=w)


def test_warmed_chunks_renumber_and_a_lane_lost_across_the_chunk_boundary():
    """records of a chunk that ran w warm-up frames number their head w: rgbid_dist_renumber_warmed_chunks brings them back to 0 .. n - 1 and the composition
    is what it is without warm-up.  ADVICE r5: a lane that lost tracking during the warm-up and is still lost on the chunk's first transitions repeats the
    head's id -- the sequence must compose (those frames carry LOST), not be refused; a lane that never ran its warm-up (head 0) is refused."""
    from rgbid._lib import RgbidError
    F, chunks, world, w = 41, 5, 2, 3
    rec, Rg, tg, ranges, L = _chain_records(F, chunks)
    lanes = D.lanes_per_rank(chunks, world)

    def gathered(r):
        a = np.zeros((world, lanes, L), D.GATHER_DTYPE)
        for rk in range(world):
            for i, c in enumerate(D.rank_chunks(chunks, world, rk)):
                a[rk, i] = r[c]
        return a
    R0, t0, st0, _ = D.compose_trajectory(gathered(rec), world, chunks, ranges)
    warmed = rec.copy()
    for c, (a, b) in enumerate(ranges):
        if c:
            warmed[c]["frame_id"][: b - a + 1] += min(w, a)
    R1, t1, st1, _ = D.compose_trajectory(D.renumber_warmed_chunks(gathered(warmed), world, chunks, ranges, w), world, chunks, ranges)
    assert np.array_equal(R0, R1) and np.array_equal(t0, t1) and np.array_equal(st0, st1)
    # chunk 2: lost at the second warm-up frame, still lost on its first two transitions (the id does not advance while lost), then tracking again
    LOST = 2
    lost = warmed.copy()
    a2, b2 = ranges[2]
    ids = lost[2]["frame_id"][: b2 - a2 + 1].copy()
    ids[:] = [1, 1, 1] + list(range(2, b2 - a2))
    lost[2]["frame_id"][: b2 - a2 + 1] = ids
    lost[2]["status"][1:3] = LOST
    R2, t2, st2, _ = D.compose_trajectory(D.renumber_warmed_chunks(gathered(lost), world, chunks, ranges, w), world, chunks, ranges)
    assert st2[a2 + 1] == LOST and st2[a2 + 2] == LOST and np.array_equal(R2, R0) and np.array_equal(t2, t0)
    never = warmed.copy()
    never[3]["frame_id"][: ranges[3][1] - ranges[3][0] + 1] = np.arange(ranges[3][1] - ranges[3][0] + 1)
    with pytest.raises(RgbidError):
        D.renumber_warmed_chunks(gathered(never), world, chunks, ranges, w)


def test_tcp_rendezvous_ignores_strangers_and_duplicates():
    """rank 0 keeps accepting when something that is not a rank of this job connects (a port scanner, a stale process with another nonce, a
    rank that was served already); host names resolve (getaddrinfo)"""
    import ctypes as C
    import socket
    import struct
    import threading
    import time
    L = D.dlib()
    port = D.free_port()
    os.environ["RGBID_DIST_NONCE"] = "777"; os.environ["RGBID_DIST_TIMEOUT_S"] = "30"
    try:
        out = {}

        def rank0():
            blob = (C.c_char * 16)(*b"0123456789abcdef")
            out["rc0"] = L.rgbid_dist_broadcast_bytes(b"localhost", port, 2, 0, blob, C.c_size_t(16))
        th = threading.Thread(target=rank0); th.start()
        time.sleep(0.3)
        H = lambda rank, nonce, kind=0, seq=0, n=16: struct.pack("<IiQIIQ", 0x52474245, rank, nonce, kind, seq, n)
        for hello in (b"GET / HTTP/1.0\r\n\r\n", H(1, 999), H(7, 777), H(1, 777, kind=1), H(1, 777, seq=5), H(1, 777, n=17), struct.pack("<IiQ", 0x52474244, 1, 777), b""):
            s = socket.create_connection(("127.0.0.1", port)); s.sendall(hello); time.sleep(0.05); s.close()
        blob = (C.c_char * 16)()
        rc1 = L.rgbid_dist_broadcast_bytes(b"localhost", port, 2, 1, blob, C.c_size_t(16))
        th.join(60)
        assert out.get("rc0") == 0 and rc1 == 0 and bytes(blob) == b"0123456789abcdef"
        # all-gather transport, 3 ranks in threads
        port2 = D.free_port()
        allb = [np.zeros(3 * 8, np.uint8) for _ in range(3)]
        rcs = [None] * 3

        def rk(r):
            mine = np.full(8, r + 1, np.uint8)
            rcs[r] = L.rgbid_dist_allgather_bytes_tcp(b"127.0.0.1", port2, 3, r, mine.ctypes.data_as(C.c_void_p), C.c_size_t(8), allb[r].ctypes.data_as(C.c_void_p))
        ths = [threading.Thread(target=rk, args=(r,)) for r in (2, 1, 0)]
        [t_.start() for t_ in ths]; [t_.join(60) for t_ in ths]
        assert rcs == [0, 0, 0] and all(np.array_equal(a, np.repeat([1, 2, 3], 8)) for a in allb)
        # two exchanges back to back on ONE port, 3 ranks (what rgbid_dist_exchange_id followed by the TCP all-gather does): rank 1 finishes the
        # broadcast at once and reaches rank 0's listener while it still waits for the late rank 2 -- it is told "not yet" and comes back
        port3 = D.free_port()
        got = [None] * 3

        def both(r):
            if r == 2:
                time.sleep(1.0)
            blob = (C.c_char * 8)(*(b"ABCDEFGH" if r == 0 else b"\0" * 8))
            rc_a = L.rgbid_dist_broadcast_bytes(b"127.0.0.1", port3, 3, r, blob, C.c_size_t(8))
            mine = np.full(4, 10 + r, np.uint8); allv = np.zeros(12, np.uint8)
            rc_b = L.rgbid_dist_allgather_bytes_tcp(b"127.0.0.1", port3, 3, r, mine.ctypes.data_as(C.c_void_p), C.c_size_t(4), allv.ctypes.data_as(C.c_void_p))
            got[r] = (rc_a, bytes(blob), rc_b, allv.tolist())
        ths = [threading.Thread(target=both, args=(r,)) for r in (1, 2, 0)]
        [t_.start() for t_ in ths]; [t_.join(90) for t_ in ths]
        assert all(g == (0, b"ABCDEFGH", 0, [10] * 4 + [11] * 4 + [12] * 4) for g in got), got
    finally:
        os.environ.pop("RGBID_DIST_NONCE"); os.environ.pop("RGBID_DIST_TIMEOUT_S")


def test_bench_gpus_flag_fails_loudly_without_devices():
    """`python bench.py --gpus 2` must not quietly run one rank: with fewer than 2 HIP devices it exits non-zero with a clear message
    (here: no device at all)."""
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0
    assert "--gpus 2" in p.stderr and "device" in p.stderr, p.stderr[-500:]
    assert "{" not in p.stdout          # no result line
