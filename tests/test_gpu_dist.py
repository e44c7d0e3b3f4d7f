"""GPU side of the multi-GPU helpers (include/rgbid_dist.h) on the one GPU a test box has: the RCCL communicator of the C-ABI with
world = 1 (ncclCommInitRank / ncclAllGather / all-reduce barrier are real RCCL calls on the engine's stream), the device-side record
pack, and chunk-sharded tracking gathered through it.  world > 1 is covered by the gloo tests (tests/test_dist_cpu.py) and by
tools/dist_selftest.py --backend nccl on a multi-GPU node."""
import numpy as np
import pytest
import torch

from rgbid import dist as D
from rgbid import engine as E
from rgbid import sequence, synth

pytestmark = pytest.mark.gpu


def rot_angle(Ra, Rb):
    return float(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


def test_pack_gather_records_and_rccl_world1(ctx):
    K = (131.25, 131.25, 79.5, 59.5)
    B, T = 3, 6
    seqs = [synth.make_sequence(T, seed=synth.SEED + 17 * l, K=K, rows=120, cols=160, device="cuda", trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8)) for l in range(B)]
    depth = torch.stack([s["depth"] for s in seqs], 1).to(torch.int16).contiguous()
    rgb = torch.stack([s["rgb"] for s in seqs], 1).contiguous()
    eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=B, K=K, use_graph=0, record_capacity=T))
    for k in range(T):
        eng.step(depth[k], rgb[k])
    rec = eng.records()
    packed = D.pack_engine_records(eng, 1, T - 1)
    ctx.sync()
    g = packed.cpu().numpy().view(D.GATHER_DTYPE).reshape(B, T - 1)
    for l in range(B):
        for k in range(T - 1):
            r = rec[1 + k, l]
            assert g[l, k]["frame_id"] == r["frame"] and g[l, k]["status"] == r["status"]
            assert np.array_equal(g[l, k]["R"], r["odo_R"]) and np.array_equal(g[l, k]["t"], r["odo_t"]) and np.array_equal(g[l, k]["cov"], r["odo_cov"])
    comm = D.Comm(ctx, 1, 0)
    assert comm.world() == 1 and comm.rank() == 0
    allb = comm.gather(packed, B * (T - 1))
    comm.barrier()
    assert torch.equal(allb, packed)
    comm.close()
    eng.close()


def test_chunked_tracking_through_the_cabi_gather(ctx):
    """the frame-to-frame records composed by rgbid_dist_compose_trajectory reproduce the engine's own global poses (single chunk), and
    the 4-chunk run through the RCCL gather equals the 4-chunk run without it"""
    K = (131.25, 131.25, 79.5, 59.5)
    T = 13
    seq = synth.make_sequence(T, K=K, rows=120, cols=160, device="cuda", trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8))
    depth = seq["depth"].to(torch.int16).contiguous(); rgb = seq["rgb"].contiguous()
    R1, t1, _ = sequence.track_chunked(ctx, depth, rgb, 1, K)
    st, cov = sequence.track_chunked.last
    eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=1, K=K, record_capacity=T))
    for k in range(T):
        eng.step(depth[k:k + 1], rgb[k:k + 1])
    rec = eng.records()
    eng.close()
    for k in range(1, T):
        assert np.abs(R1[k] - rec[k, 0]["R"]).max() < 1e-12 and np.linalg.norm(t1[k] - rec[k, 0]["t"]) < 1e-12, k   # product of the dT_k = the engine's own pose
        assert st[k] == rec[k, 0]["status"] and np.array_equal(cov[k], rec[k, 0]["odo_cov"])
    comm = D.Comm(ctx, 1, 0)
    Ra, ta, ranges = sequence.track_chunked(ctx, depth, rgb, 4, K, comm=comm)
    Rb, tb, _ = sequence.track_chunked(ctx, depth, rgb, 4, K)
    comm.close()
    assert ranges == [(0, 3), (3, 6), (6, 9), (9, 12)]
    assert np.array_equal(Ra, Rb) and np.array_equal(ta, tb)


def test_sharded_sequence_with_warm_up_frames(ctx):
    """rgbid_seq_config.warmup_frames (round 5): every chunk but the first tracks w frames before its own first frame, so that its first recorded transition
    has a velocity prior and a settled keyframe.  w = 0 is the old behaviour bit for bit; with w > 0 the chunk heads come closer to the unsharded run, no
    frame is lost, the run costs w more lock-step steps."""
    from rgbid import device
    K = (131.25, 131.25, 79.5, 59.5)
    T, chunks = 41, 5
    seq = synth.make_sequence(T, K=K, rows=120, cols=160, device="cuda", trans_step=(0.004, 0.012), rot_step_deg=(0.2, 0.8))
    depth = seq["depth"].to(torch.int16).cpu().contiguous(); rgb = seq["rgb"].cpu().contiguous()
    cfg = E.default_config(rows=120, cols=160, K=K)
    R1, t1, st1, _, _ = D.track_sequence(ctx, cfg, depth, rgb, 1)
    out = {}
    for w in (0, 2, 4):
        R, t, st, cov, rep = D.track_sequence(ctx, cfg, depth, rgb, chunks, warmup_frames=w)
        assert rep["chunk_len"] == 9 and all(int(x) & E.ST_TRACKED for x in st[1:])
        heads = [8 * c + 1 for c in range(1, chunks)]          # the first transition of every chunk but the first
        dr = dt = 0.0
        for k in heads:
            dRa = R[k - 1].T @ R[k]; dRb = R1[k - 1].T @ R1[k]
            dta = R[k - 1].T @ (t[k] - t[k - 1]); dtb = R1[k - 1].T @ (t1[k] - t1[k - 1])
            dr = max(dr, rot_angle(dRa, dRb)); dt = max(dt, float(np.linalg.norm(dta - dtb)))
        out[w] = (dr, dt, R, t)
    Rz, tz, _ = sequence.track_chunked(ctx, seq["depth"].to(torch.int16).contiguous(), seq["rgb"].contiguous(), chunks, K)
    assert np.array_equal(out[0][2], Rz) and np.array_equal(out[0][3], tz)          # w = 0: the driver as it was
    print("chunk-head deviation vs the unsharded run (rad, m):", {w: (round(v[0], 7), round(v[1], 7)) for w, v in out.items()})
    assert out[2][0] <= out[0][0] and out[2][1] <= out[0][1] and out[4][1] <= 1.2 * out[2][1] + 1e-6
    assert out[4][0] < 2e-3 and out[4][1] < 5e-3


def _run_bench(extra_args, env_extra=None, timeout=1500):
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **(env_extra or {}))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra_args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_small_run_reports_every_section():
    """bench.py end to end on a small batch: the result line carries roofline, the engine's own bytes per frame, the oracle check of every
    distinct stream, the EXACT-class variants, configs 1 / 5 and config 4 through the C++ sequence driver"""
    import json
    p = _run_bench(["--lanes", "8", "--streams", "4", "--steps", "3", "--warmup", "1", "--reps", "2", "--seq-frames", "61", "--seq-chunks-per-gpu", "4", "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["roofline"]["frac"] > 0 and r["roofline"]["bound"] == "hbm"
    fl = r["frame_level"]
    assert 1.2e8 < fl["engine_bytes_per_frame"] < 2.4e8 and "u3_equivalent_gbs" in fl and "achieved_gbs" not in fl
    assert r["parity"]["oracle_checked_lanes"] == [0, 1, 2, 3] and r["parity"]["within_1e-4"] and r["parity"]["lanes_bit_identical"]
    ex = {e["config"].split(":")[0]: e for e in r["extra_configs"]}
    assert not any("error" in e for e in r["extra_configs"]), [e for e in r["extra_configs"] if "error" in e]
    assert ex["exact-fused"]["parity"]["within_1e-4"] and ex["exact-unfused"]["parity"]["within_1e-4"]
    assert ex["5"]["parity"]["within_1e-4"]
    c4 = ex["4"]
    assert c4["scaling"] == "strong" and c4["frames"] == 61 and c4["chunks"] == 4 and c4["lanes_per_gpu"] == 4 and c4["frames_lost"] == 0
    assert c4["ate_rmse_m"]["sharded"] < 5e-3 and c4["ate_rmse_m"]["unsharded"] < 5e-3
    assert c4["chunk_head_deviation_vs_unsharded"]["max_rot_rad"] < 5e-3 and c4["chunk_head_deviation_vs_unsharded"]["max_trans_m"] < 1e-2


def test_bench_fails_loudly_when_the_cabi_communicator_fails():
    """the product's record exchange is the C-ABI RCCL helper: if it cannot come up the run ENDS non-zero instead of reporting a number measured
    through torch.distributed; asked for explicitly (--gather torch) that transport is allowed"""
    args = ["--lanes", "4", "--streams", "2", "--steps", "1", "--warmup", "0", "--reps", "1", "--no-extras", "--no-cpu-baseline", "--check-streams", "0"]
    p = _run_bench(args, {"RGBID_FORCE_DIST": "1", "RGBID_BENCH_FORCE_COMM_FAILURE": "1"})
    assert p.returncode == 3 and "refusing to report" in p.stderr and "{" not in p.stdout, (p.returncode, p.stderr[-800:])
    p = _run_bench(args + ["--gather", "torch"], {"RGBID_FORCE_DIST": "1", "RGBID_BENCH_FORCE_COMM_FAILURE": "1"})
    assert p.returncode == 0, p.stderr[-1500:]
    import json
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert "asked for with --gather torch" in r["multi_gpu"]["gather"] and r["multi_gpu"]["gather_equals_torch_all_gather"]
    # and the product path itself on one rank: the C-ABI communicator comes up, gathers inside the timed region, the line says so per rank
    p = _run_bench(args, {"RGBID_FORCE_DIST": "1"})
    assert p.returncode == 0, p.stderr[-1500:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    mg = r["multi_gpu"]
    assert mg["rccl_ranks"] == 1 and mg["rccl_ranks_per_rank"] == [1] and "rgbid_dist_gather_records" in mg["gather"] and mg["gather_equals_torch_all_gather"]
    assert len(mg["gather_us_per_repetition_rank0"]) == 1 and len(mg["per_rank_frames_per_s"]) == 1 and mg["per_rank_frames_per_s"][0] >= r["value"] * 0.999
