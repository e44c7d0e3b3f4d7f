#!/usr/bin/env python
"""Benchmark of the MI355X-native dense RGB-iD alignment front-end (BASELINE.json metric).

One "step" = one VisodoTracker::trackNewFrame for every lane of the batched engine: `lanes` independent
640x480 synthetic RGB-D streams (stand-ins for TUM fr1/desk, which is not in this image) are each advanced by
one frame: frame preparation, 3-level pyramid, {3,5,10} Gauss-Newton iterations with Student-t sigma/nu
estimation, covariance pass, covisibility checks, keyframe inverse-depth fusion.  Inputs are resident in HBM when
the timed region starts.  `value` = aligned frames / s over all GPUs (weak scaling: lanes per GPU is fixed).

    python bench.py --gpus N --steps 8 --warmup 2        (N > 1 without a launcher: re-executes itself as N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Protocol: `--reps` (5) repetitions, each = reset, keyframe frame, W untimed warm-up steps, then EXACTLY K timed steps bracketed by a
barrier + device synchronisation on both sides (max over ranks); `value` is the median repetition, all of them are listed.  Prints ONE
JSON line (rank 0) with `roofline` (level-0 residual + normal-equation kernel, timed with HIP events inside the timed regions),
`parity` (duplicate lanes bit-identical; one lane of EVERY distinct stream against the CPU oracle over all timed steps -- the oracle is the
checker here, after the timed regions), `extra_configs` (the headline workload in the EXACT numerics class, fused and as the reference's kernel
sequence; BASELINE configs 1, 4 and 5) and `cpu_baseline` (the CPU oracle -- a scalar port of the reference algorithm -- on the host cores;
baseline, not target).  Config 4 (ONE long sequence cut into chunks, strong-scaled over the ranks) runs through the C++ driver
rgbid_dist_track_sequence (csrc/dist.cpp; command line: rgbid-slam_amd/bin/rgbid_track_sequence).
"""
import argparse
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before the HIP runtime comes up: the host driver only supports dmabuf IPC (RCCL across processes)
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "rgbid-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured float4-copy ceiling is 6290 GB/s
U3_BYTES_PER_FRAME = 275.6e6   # algorithmic bytes of one aligned 640x480 frame (3 levels, {10,5,3})


def make_inputs(lanes, n_frames, rows, cols, K, device, n_unique=32):
    """[T, B] stacks of `n_unique` distinct synthetic streams (different scene seed, different camera path), dealt round-robin onto the
    lanes: lane l carries stream l % n_unique."""
    from rgbid import synth
    n = min(lanes, n_unique)
    seqs = [synth.make_sequence(n_frames, seed=synth.SEED + 17 * i, K=K, rows=rows, cols=cols, device=device) for i in range(n)]
    depth_u = torch.stack([s["depth"].to(torch.int16) for s in seqs], 1)   # [T, n, rows, cols]
    rgb_u = torch.stack([s["rgb"] for s in seqs], 1)
    idx = torch.arange(lanes, device=depth_u.device) % n
    depth = depth_u[:, idx].contiguous()                                    # [T, B, rows, cols]
    rgb = rgb_u[:, idx].contiguous()                                        # [T, B, rows, cols, 3]
    return seqs, depth, rgb


def auto_lanes(ctx, dev, rows, cols, levels, n_frames, keyframes, world, target=2048):
    """lanes per GPU: `target`, or the largest multiple of 256 whose engine state + resident input frames (n_frames x lanes x 5 B/px) fit 85 % of
    the device's free memory (a long --steps run must not run out of HBM); every rank takes the minimum over ranks"""
    from rgbid import engine as E
    probe = E.Engine(ctx, E.default_config(rows=rows, cols=cols, levels=levels, lanes=1, K=(525.0, 525.0, 319.5, 239.5), record_capacity=n_frames,
                                           keyframe_capacity=keyframes))
    per_lane = probe.bytes() + n_frames * rows * cols * 5 * 1.05      # u16 depth + rgb24 per frame; 5 % for the unique-stream copies
    probe.close()
    free, _total = torch.cuda.mem_get_info(dev)
    lanes = target
    while lanes > 256 and lanes * per_lane > 0.85 * free:
        lanes -= 256
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([lanes], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        lanes = int(t.item())
    return lanes


def usable_cores():
    """host cores this process may actually use: affinity mask, capped by the cgroup CPU quota (cpu.max / cfs_quota_us)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(seq, rows, cols, K, budget_s=12.0):
    """The oracle's VisodoTracker restatement on the host cores, same frames, same config.  Two ways of using the cores are timed
    and the better one is reported: (a) one tracker with OpenMP over image rows (stops scaling at ~16 threads: 480 rows, a fork/join
    per kernel), (b) one single-threaded tracker per core on independent copies of the sequence -- the CPU analogue of the GPU's
    lanes."""
    import subprocess
    import tempfile
    from oracle import oracle as O
    d = seq["depth"].cpu().numpy().astype(np.uint16)
    c = seq["rgb"].cpu().numpy()
    cfg = O.default_config(rows=rows, cols=cols, fx=K[0], fy=K[1], cx=K[2], cy=K[3])
    ncpu = usable_cores()
    # (a) OpenMP over rows
    O.set_num_threads(min(ncpu, 16))
    frames = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s / 2:
        trk = O.Tracker(cfg)
        trk.track(d[0], c[0])  # first frame = keyframe creation (not an aligned frame)
        for k in range(1, d.shape[0]):
            trk.track(d[k], c[k])
            frames += 1
        trk.close()
    el = time.perf_counter() - t0
    omp = {"value": frames / el, "cores": int(O.num_threads()), "frames": frames, "seconds": el}
    # (b) one single-threaded tracker per core
    best = dict(omp, how="one tracker, OpenMP over image rows")
    try:
        with tempfile.TemporaryDirectory() as tmp:
            npz = os.path.join(tmp, "frames.npz")
            np.savez(npz, depth=d, rgb=c, K=np.array(K, np.float64))
            worker = os.path.join(ROOT, "tools", "cpu_worker.py")
            env = dict(os.environ, OMP_NUM_THREADS="1")
            procs = [subprocess.Popen([sys.executable, worker, npz, str(budget_s)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
                     for _ in range(ncpu)]
            outs = [p.communicate(timeout=budget_s * 6 + 120)[0].split() for p in procs]
        fr = sum(int(o[0]) for o in outs if len(o) == 2)
        tm = max(float(o[1]) for o in outs if len(o) == 2)
        multi = {"value": fr / tm, "cores": ncpu, "frames": fr, "seconds": tm}
        if multi["value"] > best["value"]:
            best = dict(multi, how="one single-threaded tracker per host core on independent copies of the sequence")
    except Exception as e:  # the baseline is informative only: never fail the benchmark because of it
        best["note"] = f"per-core instances not run ({type(e).__name__})"
    return {"value": best["value"], "unit": "frames/s", "cores": int(best["cores"]), "kind": "port",
            "sample": f"{best['frames']} aligned {cols}x{rows} frames (same synthetic frames, full per-frame pipeline) in {best['seconds']:.1f} s; "
                      f"oracle C restatement, {best['how']}; OpenMP-over-rows figure: {omp['value']:.1f} frames/s on {omp['cores']} threads"}


def pmc_traffic(lanes, rows, cols, fused):
    """HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE collected in separate
    rocprofv3 --pmc runs by tools/profile_bench.sh, gfx950 corrections applied; committed under profiles/).  Counters
    cannot be read from inside this process, so the figure is only reported when a committed PMC summary matches the
    configuration being run (and says which file it came from); otherwise null."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if (d.get("lanes"), d.get("rows"), d.get("cols"), bool(d.get("fused_gn"))) == (lanes, rows, cols, bool(fused)):
            return d["traffic_bytes_per_launch"], "profiles/" + os.path.basename(f) + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not measured in this run)"
    return None, None


def selection_check(ctx, dev, depth, rgb, rec_step, step, n_lanes, rows, cols, K):
    """Size-independent property of the headline numerics class, on the benchmark's own frames: the FAST gather kernels (cheap float arithmetic) must take
    the same DISCRETE decisions as the EXACT kernels (bit-identical to the oracle, tests/) -- validity of every pixel, the point-sampled source pixel
    (a neighbouring pixel moves the value by far more than rounding), the four covisibility counts, the fusion gate.  Frames step - 1 -> step of the first
    n_lanes lanes, warped with the engine's own frame-to-frame odometry of that step (rec_step: the pose records of frame `step`).  Runs after the timed regions; device kernels only."""
    from rgbid import batched as BT
    bt = BT.Batched(ctx)
    L = n_lanes
    f32 = lambda: torch.empty((L, rows, cols), device=dev)
    W, I = [f32(), f32()], [f32(), f32()]
    ch = [f32() for _ in range(3)]
    for j, k in enumerate((step - 1, step)):
        bt.prep_frame(depth[k][:L].contiguous(), rgb[k][:L].contiguous(), W[j], I[j], *ch, 1.0)
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float64)
    Ki = np.linalg.inv(Km)
    Rab, tab, Rba, tba = [], [], [], []
    for l in range(L):
        R = np.asarray(rec_step[l]["odo_R"], np.float64).reshape(3, 3); t = np.asarray(rec_step[l]["odo_t"], np.float64)
        Ri = R.T; ti = -Ri @ t
        Rab.append((Km @ Ri @ Ki).astype(np.float32).reshape(9)); tab.append((Km @ ti).astype(np.float32))
        Rba.append((Km @ R @ Ki).astype(np.float32).reshape(9)); tba.append((Km @ t).astype(np.float32))
    out = {"lanes": L, "pixels": int(L * rows * cols), "frames": [int(step - 1), int(step)]}
    # warp pair: frame `step` sampled on the grid of frame `step - 1`
    res = {}
    for fast in (False, True):
        d1, d2 = f32(), f32()
        bt.warp_pair(W[1], I[1], W[0], d1, d2, Rab, tab, fast=fast)
        res[fast] = (d1.cpu().numpy(), d2.cpu().numpy())
    (e1, e2), (g1, g2) = res[False], res[True]
    both = ~np.isnan(e1) & ~np.isnan(g1)
    out["warp_validity_mismatches"] = int(np.count_nonzero(np.isnan(e1) != np.isnan(g1)) + np.count_nonzero(np.isnan(e2) != np.isnan(g2)))
    out["warp_other_source_pixel"] = int(np.count_nonzero(np.abs(g1[both] - e1[both]) > 1e-5 * np.abs(e1[both])))
    out["warp_valid_pixels"] = int(both.sum())
    # covisibility counts, both directions
    ce = bt.visibility_pair(W[1], W[0], Rab, tab, Rba, tba, fast=False); cf = bt.visibility_pair(W[1], W[0], Rab, tab, Rba, tba, fast=True)
    out["covisibility_counts_equal"] = bool(np.array_equal(ce, cf))
    out["covisibility_counted"] = int(ce.sum())
    # one-pass fusion of frame `step` into frame `step - 1` as keyframe
    fe = {}
    for fast in (False, True):
        kf, kw, ww = W[0].clone(), torch.ones((L, rows, cols), device=dev), torch.zeros((L, rows, cols), device=dev)
        bt.fuse_frame(W[1], kf, kw, ww, Rab, tab, fast=fast)
        fe[fast] = (kf.cpu().numpy(), kw.cpu().numpy())
    (ke, we), (kg, wg) = fe[False], fe[True]
    both = ~np.isnan(ke) & ~np.isnan(kg)
    out["fusion_validity_mismatches"] = int(np.count_nonzero(np.isnan(ke) != np.isnan(kg)))
    out["fusion_other_decision"] = int(np.count_nonzero(np.abs(kg[both] - ke[both]) > 1e-5 * np.abs(ke[both])) + np.count_nonzero(np.abs(wg[both] - we[both]) > 1e-4 * np.abs(we[both])))
    out["selection_identical"] = bool(out["warp_validity_mismatches"] == 0 and out["warp_other_source_pixel"] == 0 and out["covisibility_counts_equal"]
                                      and out["fusion_validity_mismatches"] == 0 and out["fusion_other_decision"] == 0)
    out["note"] = "FAST vs EXACT device kernels on the benchmark's own frames (the EXACT kernels are bit-identical to the oracle in tests/): discrete decisions only; float values differ by rounding"
    return out


def oracle_check(depth, rgb, lanes_to_check, rows, cols, K, levels, iters, rec, first_step, n_steps):
    # rec: the engine's records of ALL steps [T, B] (status bits of every frame are imposed on the oracle where a ratio sits on its threshold)
    """Parity of the surface that was just benchmarked: for each lane in `lanes_to_check` run the CPU oracle tracker over the SAME frames
    (keyframe frame, warm-up, timed steps) and hold the engine's records of the timed steps to 1e-4 rad / 1e-4 m; keyframe decisions
    must agree unless a covisibility ratio sits on its threshold (the oracle then continues with the engine's decision imposed).
    The oracle is the checker here and runs after every timed region.  One worker process per lane (the host cores run them in parallel)."""
    import subprocess
    import tempfile
    from rgbid import engine as E
    T = first_step + n_steps
    worst_r = worst_t = 0.0
    imposed = 0
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for l in lanes_to_check:
            npz = os.path.join(tmp, f"lane{l}.npz")
            np.savez(npz, depth=depth[:T, l].cpu().numpy().view(np.uint16), rgb=rgb[:T, l].cpu().numpy(), K=np.array(K, np.float64),
                     levels=levels, iters=np.array(iters), status=rec["status"][:, l])
            env = dict(os.environ, OMP_NUM_THREADS=str(max(1, usable_cores() // max(1, len(lanes_to_check)))))
            procs.append((l, subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "oracle_lane_worker.py"), npz], stdout=subprocess.PIPE,
                                              stderr=subprocess.PIPE, text=True, env=env)))
        for l, p in procs:
            out, err = p.communicate(timeout=1200)
            assert p.returncode == 0, f"oracle worker for lane {l} failed: {err[-800:]}"
            o = json.loads(out.strip().splitlines()[-1])
            R = np.array(o["R"]); t = np.array(o["t"])
            imposed += o["imposed"]
            for k in range(n_steps):
                g = rec[first_step + k, l]
                c = (np.trace(R[first_step + k].T @ g["R"]) - 1) / 2
                er = float(np.arccos(np.clip(c, -1, 1))); et = float(np.linalg.norm(t[first_step + k] - g["t"]))
                worst_r, worst_t = max(worst_r, er), max(worst_t, et)
    return worst_r, worst_t, imposed


def run_config(ctx, dev, work_stream, rows, cols, levels, iters, B, Kst, W, reps, n_unique, graph, fused, keyframes, K, dist_env, check_streams=0,
               fast_numerics=1, inputs=None, defer_maps=0):
    """the timed protocol on one engine configuration; returns (result dict, inputs kept for the PCIe leg)"""
    from rgbid import engine as E
    T = 1 + W + Kst
    seqs, depth, rgb = inputs if inputs is not None else make_inputs(B, T, rows, cols, K, dev, n_unique)
    cfg_kw = dict(rows=rows, cols=cols, levels=levels, lanes=B, K=K, iters=iters, use_graph=graph, fused_gn=fused, record_capacity=T,
                  keyframe_capacity=keyframes)
    if hasattr(E.EngineConfig, "fast_numerics"):
        cfg_kw["fast_numerics"] = fast_numerics
    if defer_maps:
        cfg_kw["defer_keyframe_maps"] = 1
    eng = E.Engine(ctx, E.default_config(**cfg_kw))
    gn_l0 = iters[0] + 1                      # level-0 launches of the dominant kernel per step (10 GN + covariance pass)
    profile_in_timed = not graph
    use_dist, comm = dist_env["use_dist"], dist_env.get("comm")
    times, recs, k_ms_tot, k_n_tot, k_bytes = [], [], 0.0, 0, 0.0
    gathered = None
    gather_us, rank_times = [], []
    for rep in range(reps):
        if rep:
            eng.reset()
        eng.step(depth[0], rgb[0])                # frame 0: keyframe creation
        for k in range(1, 1 + W):                 # untimed warm-up steps
            eng.step(depth[k], rgb[k])
        torch.cuda.synchronize(dev)
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        if profile_in_timed:
            eng.profile_begin(gn_l0 * Kst)
        t0 = time.perf_counter()
        for k in range(1 + W, 1 + W + Kst):       # EXACTLY K timed steps
            eng.step(depth[k], rgb[k])
        rec = eng.records(1 + W, Kst)             # synchronises; pose records of the timed steps
        if use_dist:
            # the only collective on the path: all-gather of the 392-byte per-frame records of every rank's lanes (RCCL over xGMI)
            from rgbid import dist as D
            packed = D.pack_engine_records(eng, 1 + W, Kst)
            ctx.sync()
            tg = time.perf_counter()
            if comm is not None:
                gathered = comm.gather(packed, B * Kst)
                ctx.sync()
            else:                                  # --gather torch (asked for explicitly): the same exchange through torch.distributed
                gathered = torch.empty(dist_env["world"] * packed.numel(), dtype=torch.uint8, device=dev)
                dist.all_gather_into_tensor(gathered, packed)
                torch.cuda.synchronize(dev)
            gather_us.append((time.perf_counter() - tg) * 1e6)
        torch.cuda.synchronize(dev)
        el_own = time.perf_counter() - t0          # this rank's own clock, before it waits for the others
        if use_dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
            own = torch.zeros(dist_env["world"], dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(own, torch.tensor([el_own], dtype=torch.float64, device=dev))
            rank_times.append([float(v) for v in own.cpu()])
        if profile_in_timed:
            ms, n, k_bytes = eng.profile_end()
            k_ms_tot += ms; k_n_tot += n
        times.append(el)
        recs.append(rec.copy())
    if not profile_in_timed:
        # graph replay cannot be event-bracketed per kernel: time the same kernel in an extra eager pass over the same frames
        eng.reset()
        eng.step(depth[0], rgb[0])
        eng.profile_begin(gn_l0 * (W + Kst))
        for k in range(1, 1 + W + Kst):
            eng.step(depth[k], rgb[k])
        k_ms_tot, k_n_tot, k_bytes = eng.profile_end()
    world = dist_env["world"]
    order = np.argsort(times)
    med = int(order[len(times) // 2])
    el = times[med]
    rec = recs[med]
    tracked = int(np.count_nonzero(rec["status"] & E.ST_TRACKED))
    avg_s = (k_ms_tot / max(k_n_tot, 1)) * 1e-3
    achieved = k_bytes / avg_s / 1e9 if k_n_tot else 0.0
    n_streams = min(B, n_unique)
    # ---- parity of the benchmarked surface (after the timed regions) ----
    reps_identical = all(recs[i].tobytes() == recs[0].tobytes() for i in range(1, len(recs)))
    ref = rec[:, :n_streams]
    lanes_identical = all(rec[:, l].tobytes() == ref[:, l % n_streams].tobytes() for l in range(B))
    parity = {"lanes_bit_identical": bool(lanes_identical), "duplicate_lanes_per_stream": B // n_streams if n_streams else 0,
              "repetitions_bit_identical": bool(reps_identical)}
    if check_streams > 0:
        chk = list(range(min(check_streams, n_streams)))
        full = eng.records(0, T)   # every step of the last repetition (keyframe frame, warm-up, timed steps): the ring holds T records
        assert full[1 + W:].tobytes() == recs[-1].tobytes()
        wr, wt, imposed = oracle_check(depth, rgb, chk, rows, cols, K, levels, iters, full, 1 + W, Kst)
        parity.update({"oracle_checked_lanes": chk, "oracle_checked_steps": Kst, "worst_rot_rad": wr, "worst_trans_m": wt,
                       "keyframe_decisions_on_threshold_imposed": imposed, "within_1e-4": bool(wr < 1e-4 and wt < 1e-4)})
    # what the engine's own launch list has to move per frame (rgbid_engine_step_bytes) at the switch rates of these records
    sbytes = eng.step_bytes()
    st_all = rec["status"]
    n_tr = max(1, int(np.count_nonzero(st_all & E.ST_TRACKED)))
    p_odo = float(np.count_nonzero(((st_all & E.ST_TRACKED) != 0) & ((st_all & E.ST_ODO_KF) != 0))) / n_tr
    p_int = float(np.count_nonzero(((st_all & E.ST_TRACKED) != 0) & ((st_all & E.ST_INTEGR_KF) != 0))) / n_tr
    engine_bytes_per_frame = sbytes[0] + p_odo * sbytes[1] + p_int * sbytes[2] + (1.0 - p_int) * sbytes[3]
    res = {
        "value": B * Kst * world / el, "ms_per_step": el / Kst * 1e3,
        "engine_bytes_per_frame": engine_bytes_per_frame, "odo_kf_switch_rate": p_odo, "integr_kf_switch_rate": p_int, "step_bytes_model": sbytes,
        "gather_us": gather_us, "rank_seconds": rank_times,
        "repetitions": {"n": len(times), "frames_per_s": [B * Kst * world / t for t in times], "protocol": "each: reset, keyframe frame, warm-up, K timed steps; value = median"},
        "launches_per_step": eng.launches_per_step(), "engine_hbm_bytes": eng.bytes(), "tracked": tracked, "expected": B * Kst,
        "keyframes_exported": int(np.count_nonzero(rec["status"] & E.ST_KF_EXPORTED)), "n_unique_streams": n_streams,
        "u1": {"achieved": achieved, "avg_launch_us": avg_s * 1e6, "launches_timed": k_n_tot, "bytes_per_launch": k_bytes,
               "timed_in": "timed regions" if profile_in_timed else "separate eager pass"},
        "parity": parity,
    }
    keep = (seqs, depth, rgb, eng, rec, gathered)
    return res, keep


def pcie_leg(ctx, dev, work, eng, depth, rgb, W, Kst, ref_rec=None):
    """the timed steps again with the frames streamed from pinned host memory: frame k + 1 is uploaded on a copy stream while step k computes
    (double-buffered, event-ordered against the engine's stream)"""
    T = 1 + W + Kst
    B = depth.shape[1]
    depth_h, rgb_h = depth.cpu().pin_memory(), rgb.cpu().pin_memory()
    bufs = [(torch.empty_like(depth[0]), torch.empty_like(rgb[0])) for _ in range(2)]
    copy_stream = torch.cuda.Stream(dev)
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]

    def upload(k, slot):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(free[slot])
            bufs[slot][0].copy_(depth_h[k], non_blocking=True)
            bufs[slot][1].copy_(rgb_h[k], non_blocking=True)
            ready[slot].record(copy_stream)

    eng.reset()
    for ev in free:
        ev.record(work)
    upload(0, 0)
    t1 = None
    for k in range(T):
        slot = k % 2
        if k + 1 < T:
            upload(k + 1, 1 - slot)
        if k == 1 + W:
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
        work.wait_event(ready[slot])
        eng.step(bufs[slot][0], bufs[slot][1])
        free[slot].record(work)
    rec_h = eng.records(1 + W, Kst)
    torch.cuda.synchronize(dev)
    el_h = time.perf_counter() - t1
    per_step = (depth_h[0].numel() * 2 + rgb_h[0].numel()) / 1e9
    out = {"value": B * Kst / el_h, "unit": "frames/s", "lanes": B, "ms_per_step": el_h / Kst * 1e3, "h2d_gb_per_step": per_step,
           "h2d_gbs_needed": per_step / (el_h / Kst),
           "note": "frames streamed from pinned host memory on a copy stream, double-buffered, overlapped with the previous step"}
    if ref_rec is not None:
        out["records_identical_to_resident_run"] = bool(rec_h.tobytes() == ref_rec.tobytes())
    del depth_h, rgb_h, bufs
    return out


def extra_config1(ctx, dev, K):
    """BASELINE config 1 on the GPU: the residual + 27-term normal equations (unit U1) of ONE 640x480 pair through the single-image C-ABI
    call (cache-resident: 9.8 MB of maps live in L2 / Infinity Cache), device time from the call's own hipEvent pair."""
    from rgbid import device as DV
    rows, cols = 480, 640
    g = torch.Generator(device="cpu").manual_seed(1)
    maps = []
    for _ in range(8):       # smooth positive maps with 5 % invalid pixels: the kernel's time does not depend on the values
        m = 0.25 + torch.rand((rows, cols), generator=g)
        m[torch.rand((rows, cols), generator=g) < 0.05] = float("nan")
        maps.append(m.to(dev))
    ms = []
    for _ in range(60):
        _, _, m = ctx.buildSystemStudentNuGridStride(*maps, DV.STUDENT, DV.INDEPENDENT, 0.0025, 5.0, 0.0, 0.0, 5.0, 5.0, K, return_ms=True)
        ms.append(m)
    us = float(np.median(ms[10:])) * 1e3
    return {"config": "1: single 640x480 pair, residual + 6x6 normal equations (U1), cache-resident", "u1_device_us": us,
            "u1_gbs": 32.0 * rows * cols / (us * 1e-6) / 1e9, "calls_timed": len(ms) - 10,
            "note": "working set 9.8 MB < L2 + Infinity Cache: a latency figure, not an HBM-roofline figure; the batched U1 (all lanes of the headline run) is `roofline`"}


def _ate_rmse(t_est, t_gt):
    """RMSE of the translational residual after the least-squares rigid alignment of the estimate onto the ground truth (Horn; tools/ate.py)"""
    mu_e, mu_g = t_est.mean(0), t_gt.mean(0)
    U, _, Vt = np.linalg.svd((t_est - mu_e).T @ (t_gt - mu_g))
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = Vt.T @ S @ U.T
    res = (t_est @ R.T + (mu_g - R @ mu_e)) - t_gt
    return float(np.sqrt((res ** 2).sum(1).mean()))


def _rot_angle(Ra, Rb):
    return float(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


def extra_config4(ctx, dev, K, world, rank, n_frames, chunks_per_gpu, fused, fast):
    """BASELINE config 4 as a workload: ONE long 640x480 sequence (stand-in for TUM fr3/long_office, ~2 500 frames) cut into chunks_per_gpu x N
    chunks, STRONG-scaled over the N ranks through the C++ driver (rgbid_dist_track_sequence: partition -> uploads from pinned host memory behind
    the previous step -> lock-step engine steps -> record pack -> ONE RCCL all-gather -> composition), timed end to end per call.  Rank 0 also tracks
    the sequence unsharded (one lane, every frame in turn) and reports what sharding costs at the chunk heads (no velocity prior, fresh keyframe)
    and the ATE of both against the synthetic ground truth."""
    from rgbid import dist as D, engine as E, synth
    rows, cols = 480, 640
    t0 = time.perf_counter()
    seq = synth.make_long_sequence(n_frames, seed=synth.SEED + 4, K=K, rows=rows, cols=cols, device=dev)
    depth_h = seq["depth"].cpu().pin_memory(); rgb_h = seq["rgb"].cpu().pin_memory()
    Rg, tg = seq["R_wc"].numpy(), seq["t_wc"].numpy()
    del seq
    torch.cuda.empty_cache()
    render_s = time.perf_counter() - t0
    cfg = E.default_config(rows=rows, cols=cols, K=K, fused_gn=fused, fast_numerics=fast)
    if chunks_per_gpu <= 0:
        # measured on one GPU (tools/shard_sweep.py, profiles/r04_shard_sweep.json): throughput peaks at 128 lanes per GPU, and chunks
        # shorter than ~10 frames cost accuracy (every chunk head starts without a velocity prior on a fresh keyframe: ATE 6.3 mm unsharded, 7.0 at 128
        # chunks of 20 frames, 8.6 at 256 x 10, 9.4 at 512 x 5) and spend a growing share of the steps on chunk heads (1.08 / 1.13 / 1.23 x)
        chunks_per_gpu = max(1, min(128, (n_frames // 10) // world))
    chunks = chunks_per_gpu * world
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1"); port = int(os.environ.get("MASTER_PORT", "29541")) + 2
    runs = []
    for i in range(2):     # the first call also pays the first touch of the freshly allocated engine / staging memory; the second is the steady figure
        R, t, st, cov, rep = D.track_sequence(ctx, cfg, depth_h, rgb_h, chunks, world, rank, D.EXCHANGE_RCCL, addr, port + i)
        runs.append(rep)
    out = None
    if rank == 0:
        Rs, ts, sts, _, rep1 = D.track_sequence(ctx, cfg, depth_h, rgb_h, 1)       # unsharded: one lane, n_frames sequential steps
        ranges = D.chunk_ranges(n_frames, chunks)
        head_r = head_t = 0.0
        for (a, b) in ranges[1:]:          # frame-to-frame motion of the first tracked frame of every chunk, sharded vs unsharded
            k = a + 1
            dRc, dtc = R[k - 1].T @ R[k], R[k - 1].T @ (t[k] - t[k - 1])
            dRs, dts = Rs[k - 1].T @ Rs[k], Rs[k - 1].T @ (ts[k] - ts[k - 1])
            head_r = max(head_r, _rot_angle(dRc, dRs)); head_t = max(head_t, float(np.linalg.norm(dtc - dts)))
        lost = int(np.count_nonzero(st & E.ST_LOST))
        best = runs[-1]
        out = {"config": f"4: ONE {n_frames}-frame synthetic 640x480 sequence (stand-in for TUM fr3/long_office: dataset not in image) cut into {chunks} chunks "
                         f"({chunks_per_gpu} per GPU), strong-scaled over {world} GPU(s); C++ driver rgbid_dist_track_sequence, frames uploaded from pinned host memory",
               "scaling": "strong", "value": 1e3 * n_frames / best["total_ms"], "unit": "frames/s", "frames": n_frames, "chunks": chunks,
               "lanes_per_gpu": best["lanes"], "chunk_len": best["chunk_len"], "world": world, "rccl_ranks": best["rccl_ranks"],
               "timing_ms": {k_: best[k_] for k_ in ("setup_ms", "track_ms", "gather_ms", "compose_ms", "total_ms")},
               "first_call_total_ms": runs[0]["total_ms"], "staged_bytes_per_gpu": best["staged_bytes"], "engine_hbm_bytes": best["engine_bytes"],
               "timed": "total_ms = uploads + chunk_len engine steps + record pack + all-gather + D2H + composition (setup_ms -- communicator, engine, staging allocation -- listed, not included)",
               "frames_lost": lost,
               "unsharded": {"frames_per_s": 1e3 * n_frames / rep1["total_ms"], "total_ms": rep1["total_ms"], "note": "one lane, every frame in turn (latency-bound: ~126 dependent launches per frame)"},
               "chunk_head_deviation_vs_unsharded": {"max_rot_rad": head_r, "max_trans_m": head_t, "note": "frame-to-frame motion of each chunk's first tracked frame: no velocity prior, fresh keyframe"},
               "trajectory_vs_unsharded": {"max_rot_rad": max(_rot_angle(R[k], Rs[k]) for k in range(n_frames)), "max_trans_m": float(np.abs(t - ts).max())},
               "ate_rmse_m": {"sharded": _ate_rmse(t, tg), "unsharded": _ate_rmse(ts, tg), "vs": "synthetic ground truth (exact camera path)"},
               "render_s": render_s}
        if world == 1:
            # what other chunk counts would have given on this GPU (the full sweep: tools/shard_sweep.py -> profiles/r04_shard_sweep.json)
            sweep = []
            for c in (8, 64, 256, 512):
                if c == chunks or n_frames < c + 1:
                    continue
                Rc, tc, stc, _, repc = None, None, None, None, None
                for _ in range(2):
                    Rc, tc, stc, _, repc = D.track_sequence(ctx, cfg, depth_h, rgb_h, c)
                sweep.append({"chunks": c, "frames_per_s": 1e3 * n_frames / repc["total_ms"], "chunk_len": repc["chunk_len"], "ms_per_step": repc["track_ms"] / repc["chunk_len"],
                              "lane_steps_per_transition": c * repc["chunk_len"] / (n_frames - 1), "traj_max_trans_m_vs_unsharded": float(np.abs(tc - ts).max()),
                              "ate_rmse_m": _ate_rmse(tc, tg), "frames_lost": int(np.count_nonzero(stc & E.ST_LOST))})
            out["chunk_count_sweep_1gpu"] = sweep
            # round 5: the chunk OVERLAP (rgbid_seq_config.warmup_frames: every chunk but the first tracks w frames before its own first frame, so its first
            # recorded transition has a velocity prior and a settled keyframe), at the chunk count of the headline run.  Full sweep: tools/shard_sweep.py --warmup
            # -> profiles/r05_shard_warmup.json; DESIGN section 6 states the rule that picks the recommended value
            heads = np.array([a + 1 for (a, b) in ranges[1:]])
            wsweep = []
            for w in (0, 2, 4):
                Rw = tw = stw = repw = None
                for _ in range(2):
                    Rw, tw, stw, _, repw = D.track_sequence(ctx, cfg, depth_h, rgb_h, chunks, warmup_frames=w)
                hr = np.array([_rot_angle(Rw[k - 1].T @ Rw[k], Rs[k - 1].T @ Rs[k]) for k in heads])
                ht = np.array([float(np.linalg.norm(Rw[k - 1].T @ (tw[k] - tw[k - 1]) - Rs[k - 1].T @ (ts[k] - ts[k - 1]))) for k in heads])
                wsweep.append({"warmup_frames": w, "frames_per_s": 1e3 * n_frames / repw["total_ms"], "frames_per_lane_step": (n_frames - 1) / (chunks * (repw["chunk_len"] + w)),
                               "chunk_head_vs_unsharded": {"max_rot_rad": float(hr.max()), "max_trans_m": float(ht.max()), "median_rot_rad": float(np.median(hr)),
                                                           "median_trans_m": float(np.median(ht)), "share_within_1e-4": float(np.mean((hr < 1e-4) & (ht < 1e-4)))},
                               "trajectory_vs_unsharded": {"max_rot_rad": max(_rot_angle(Rw[k], Rs[k]) for k in range(n_frames)), "max_trans_m": float(np.abs(tw - ts).max())},
                               "ate_rmse_m": _ate_rmse(tw, tg), "frames_lost": int(np.count_nonzero(stw & E.ST_LOST))})
            out["chunk_warmup_sweep_1gpu"] = {"points": wsweep, "bar": "north_star 1e-4 rad / 1e-4 m, chunk-head transition vs the unsharded run",
                                              "rule": "the product default stays 0 (throughput: every warm-up frame is one more lock-step step); where chunk-head accuracy matters the "
                                                      "recommended value is the one with the best ATE among those that cost <= 15 % of the frames/s of w = 0 (DESIGN section 6)"}
    del depth_h, rgb_h
    return out


def extra_kfalign(ctx, dev, K):
    """SURVEY 8 f-1 as a workload (round 5): KeyframeAlign::alignKeyframes (src/keyframe_align.cpp:115-357: 4 levels, {5,5,3,0} iterations, nu by bisection, intensity
    sampled on the keyframe inverse depth) for N candidate pairs in lock-step through rgbid_kfalign_batched -- the loop closer's dense verification as a batch.
    Pairs/s with the keyframes resident in HBM at 1 / 64 / 1 024 pairs; two pairs of every run checked against the CPU oracle (orc_keyframe_align)."""
    from oracle import oracle as O
    from rgbid import kfalign, synth
    rows, cols = 480, 640
    n_distinct = 8
    iDa, ga, iDb, gb = [], [], [], []
    for i in range(n_distinct):
        seq = synth.make_sequence(4, seed=synth.SEED + 31 * i, K=K, rows=rows, cols=cols, device=dev, trans_step=(0.008, 0.02), rot_step_deg=(0.3, 1.0))
        d = seq["depth"].cpu().numpy().astype(np.uint16); c = seq["rgb"].cpu().numpy()
        for k, (iD, g) in zip((0, 3), ((iDa, ga), (iDb, gb))):
            iD.append(O.depth2invdepth(d[k])); g.append(np.clip(np.rint(O.intensity(c[k])), 0, 255).astype(np.uint8))
    iDa, ga, iDb, gb = [torch.from_numpy(np.stack(x)).to(dev) for x in (iDa, ga, iDb, gb)]
    points = []
    worst_r = worst_t = 0.0
    for n in (1, 64, 1024):
        idx = torch.arange(n, device=dev) % n_distinct
        ins = [x[idx].contiguous() for x in (iDa, ga, iDb, gb)]
        al = kfalign.KfAlign(ctx, rows, cols, n)
        for _ in range(2):
            al.align(*ins, K)                           # first touch of the aligner's buffers
        times = []
        for _ in range(5 if n >= 64 else 15):
            t0 = time.perf_counter()
            R, t, cov = al.align(*ins, K)               # synchronous on return (poses read back)
            times.append(1e3 * (time.perf_counter() - t0))
        ms = float(np.median(times))
        for i in (0, min(n, n_distinct) - 1):
            Ro, to, _ = O.keyframe_align(iDa[i].cpu().numpy(), ga[i].cpu().numpy(), iDb[i].cpu().numpy(), gb[i].cpu().numpy(), K)
            worst_r = max(worst_r, _rot_angle(R[i], Ro)); worst_t = max(worst_t, float(np.linalg.norm(t[i] - to)))
        nbytes = C_size(al)
        points.append({"pairs": n, "ms_per_call": ms, "ms_per_call_min_max": [min(times), max(times)], "pairs_per_s": 1e3 * n / ms, "launches": al.launches(), "aligner_hbm_bytes": nbytes})
        al.close()
        del ins
        torch.cuda.empty_cache()
    return {"config": "f-1: KeyframeAlign (keyframe-to-keyframe dense alignment of the loop closer) for N pairs of 640x480 keyframes in lock-step, device-resident "
                      "(rgbid_kfalign_batched; RGBID_SLAM::KeyframeAlign rides on the 1-pair case, bit-identical to its host-driven loop)",
            "unit": "pairs/s", "value": points[-1]["pairs_per_s"], "points": points,
            "parity": {"vs": "oracle (orc_keyframe_align), two pairs of every batch size", "max_rot_err_rad": worst_r, "max_trans_err_m": worst_t,
                       "within_1e-4": bool(worst_r < 1e-4 and worst_t < 1e-4)}}


def C_size(al):
    import ctypes
    n = ctypes.c_size_t()
    al.L.rgbid_kfalign_bytes(al._h, ctypes.byref(n))
    return int(n.value)


def dataset_configs(K_default):
    """BASELINE configs 2-4 on the REAL sequences, the moment they are mounted (RGBID_TUM_DIR: folders with depth_associated.txt, rgb_associated.txt,
    groundtruth.txt): tracked by the C++ driver (rgbid-slam_amd/bin/rgbid_track_sequence, 8 chunks and unsharded) and scored with tools/ate.py --
    the external pin of the trajectory.  None when nothing is mounted."""
    base = os.environ.get("RGBID_TUM_DIR", "")
    if not base or not os.path.isdir(base):
        return None
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ate as A
    from rgbid import dist as D
    cands = [base] + [os.path.join(base, d) for d in sorted(os.listdir(base))]
    out = []
    for s_ in [c for c in cands if os.path.exists(os.path.join(c, "depth_associated.txt")) and os.path.exists(os.path.join(c, "groundtruth.txt"))]:
        name = os.path.basename(os.path.normpath(s_))
        Kd = (481.2, -480.0, 319.5, 239.5) if "kt" in name.lower() else K_default      # config_data/calibration_syntheticHanda.ini
        entry = {"sequence": name}
        with tempfile.TemporaryDirectory() as tmp:
            for chunks in (1, 8):
                traj = os.path.join(tmp, f"{name}_{chunks}.txt")
                p = subprocess.run([D.TRACK_SEQUENCE_BIN, "-eval", s_ + "/", "-chunks", str(chunks), "-out", traj, "-K"] + [str(v) for v in Kd],
                                   capture_output=True, text=True, timeout=3600)
                if p.returncode != 0:
                    entry[f"chunks_{chunks}"] = {"error": (p.stdout + p.stderr)[-400:]}
                    continue
                rep = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
                gt, est = A.read_trajectory(os.path.join(s_, "groundtruth.txt")), A.read_trajectory(traj)
                a, r = A.ate(gt, est, 0.02), A.rpe(gt, est, 1, "f", 0.02)
                entry[f"chunks_{chunks}"] = {"frames": rep["frames"], "frames_per_s": rep["frames_per_s"], "ate_rmse_m": a["rmse"],
                                             "rpe_trans_rmse_m": r["trans_rmse"], "rpe_rot_rmse_deg": float(np.degrees(r["rot_rmse"]))}
        out.append(entry)
    return {"config": "2-4 on the mounted sequences (RGBID_TUM_DIR): C++ driver, ATE / RPE against the published ground truth", "sequences": out}


def flush_c_stdio():
    """RCCL prints its version banner to the C stdout when the first communicator of a process comes up (NCCL_DEBUG=VERSION in this
    image); pushing it out right then keeps the result line the LAST line of stdout."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside a launcher: become the launcher.  Fails loudly when the node has fewer devices."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        sys.stderr.write(f"bench.py --gpus {n}: this node exposes {have} HIP device(s); refusing to run fewer ranks than requested\n")
        sys.exit(2)
    from rgbid import dist as D
    p = D.spawn_local(n, [os.path.abspath(__file__)] + argv)
    sys.stderr.write(p.stderr)
    sys.stdout.write(p.stdout)
    sys.exit(p.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5, help="timed repetitions of the K steps (value = median)")
    ap.add_argument("--lanes", type=int, default=0, help="independent RGB-D streams per GPU; 0 = 2 048 (126 GB of engine state at 640x480; per-lane cost 512 -> 2 048: -5 %), "
                    "reduced in steps of 256 if the engine + the resident input frames of W + K + 1 steps would not fit the device's free memory")
    ap.add_argument("--streams", type=int, default=32, help="distinct synthetic input streams dealt onto the lanes")
    ap.add_argument("--check-streams", type=int, default=32, help="lanes (one per distinct stream) held to the CPU oracle over all timed steps, after the timed regions (default: every distinct stream)")
    ap.add_argument("--seq-frames", type=int, default=2500, help="frames of the ONE long sequence of extra config 4 (BASELINE config 4: fr3/long_office has ~2 500)")
    ap.add_argument("--seq-chunks-per-gpu", type=int, default=0, help="chunks (= engine lanes) per GPU of extra config 4; 0 = the rule of DESIGN section 6 / tools/shard_sweep.py: "
                                                                       "chunks of >= 10 frames (ATE within ~35 %% of the unsharded run), at most 128 lanes per GPU")
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--graph", type=int, default=0, help="replay each step as one hipGraph (roofline events then need a 2nd pass)")
    ap.add_argument("--fused", type=int, default=1, help="1: one fused warp + residual + normal-equation kernel per GN iteration (engine default); 0: the reference's kernel sequence")
    ap.add_argument("--fast", type=int, default=1, help="engine numerics of the gather kernels: 1 = reference-build class (FMA contraction, v_rcp), 0 = IEEE-exact")
    ap.add_argument("--keyframes", type=int, default=2, help="per-lane capacity of the keyframe export ring: the outgoing keyframe is handed to the back-end at every switch, as trackNewFrame does (0 = no export)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip extra_configs (BASELINE configs 1 and 5)")
    ap.add_argument("--gather", default="rccl-cabi", choices=["rccl-cabi", "torch"], help="transport of the record all-gather at N > 1: the product's C-ABI helper over RCCL "
                    "(a failure of it ENDS the run with a non-zero exit code), or torch.distributed when asked for explicitly")
    ap.add_argument("--h2d", type=int, default=-1, help="also time the same steps with the frames streamed from pinned host memory (PCIe-inclusive rate at the headline lane count; never `value`); "
                                                          "default: on for the single-GPU run (costs ~13 s: 31 GB of frames are copied to pinned host memory)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py --gpus {args.gpus}: launched with WORLD_SIZE={world}; the two must agree\n")
        sys.exit(2)
    use_dist = world > 1 or bool(os.environ.get("RGBID_FORCE_DIST"))  # the env override exercises the RCCL path on one rank
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        sys.stderr.write(f"bench.py --gpus {args.gpus}: rank {rank} needs HIP device {local_rank}, this node exposes "
                         f"{torch.cuda.device_count() if torch.cuda.is_available() else 0} device(s) (there is no CPU path)\n")
        sys.exit(2)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from rgbid import device, engine as E
    rows, cols, B, Kst, W = args.rows, args.cols, args.lanes, args.steps, args.warmup
    s = cols / 640.0
    K = (525.0 * s, 525.0 * s, (319.5 + 0.5) * s - 0.5, (239.5 + 0.5) * s - 0.5)

    work = torch.cuda.Stream(dev)                 # the engine's HIP stream (a torch stream so torch events can order against it)
    with torch.cuda.stream(work):
        ctx = device.Context(local_rank)
    ctx.set_async(1)
    if B <= 0:
        B = auto_lanes(ctx, dev, rows, cols, args.levels, 1 + W + Kst, args.keyframes, world)
    iters = [10, 5, 3] + [3] * (args.levels - 3) if args.levels >= 3 else [10, 5, 3][:args.levels]
    dist_env = {"use_dist": use_dist, "world": world, "comm": None}
    gather_how = None
    if use_dist:
        import torch.distributed as dist
        from rgbid import dist as D
        gather_how = "torch.distributed all_gather_into_tensor (nccl = RCCL) -- asked for with --gather torch"
        dist.barrier()            # brings torch's communicator up (and RCCL's banner out) before anything is timed or printed
        if args.gather == "rccl-cabi":
            # The product's exchange is the C-ABI helper (librgbid_dist.so over librccl).  If it cannot come up, the scaling figure would be
            # measured through a transport that is not the product's: that is a FAILED run, not a degraded one.
            try:
                if os.environ.get("RGBID_BENCH_FORCE_COMM_FAILURE"):
                    raise RuntimeError("forced by RGBID_BENCH_FORCE_COMM_FAILURE (test hook)")
                dist_env["comm"] = D.Comm(ctx, world, rank)      # the unique id travels through torch's store
                dist_env["comm"].barrier()                       # first collective of the communicator, outside every timed region
                gather_how = "rgbid_dist_gather_records (C-ABI, ncclAllGather on the engine's stream)"
            except Exception as e:
                sys.stderr.write(f"[bench] rank {rank}: the C-ABI RCCL communicator failed ({type(e).__name__}: {e}); refusing to report a multi-GPU number "
                                 f"through another transport (rerun with --gather torch to measure torch.distributed's all-gather instead)\n")
                sys.stderr.flush()
                os._exit(3)
        flush_c_stdio()
    res, keep = run_config(ctx, dev, work, rows, cols, args.levels, iters, B, Kst, W, max(1, args.reps), args.streams, args.graph, args.fused,
                           args.keyframes, K, dist_env, check_streams=args.check_streams if rank == 0 else 0, fast_numerics=args.fast)
    seqs, depth, rgb, eng, rec, gathered = keep
    rccl_ranks = rccl_ranks_all = None
    gather_ok = None
    if use_dist:
        import torch.distributed as dist
        from rgbid import dist as D
        # cross-check of the gathered records (outside the timed regions): every rank's block equals what torch.distributed gathers
        packed = D.pack_engine_records(eng, 1 + W, Kst)
        ctx.sync()
        ref = torch.empty(world * packed.numel(), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(ref, packed)
        gather_ok = bool(torch.equal(ref, gathered))
        rccl_ranks = dist_env["comm"].world() if dist_env["comm"] is not None else dist.get_world_size()
        rr = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(rr, torch.tensor([rccl_ranks], dtype=torch.int64, device=dev))
        rccl_ranks_all = [int(v) for v in rr.cpu()]

    pcie = None
    if args.h2d > 0 or (args.h2d < 0 and world == 1 and not args.no_extras):
        # PCIe-inclusive leg at the headline lane count: the same timed steps with every frame uploaded from pinned host memory
        pcie = pcie_leg(ctx, dev, work, eng, depth, rgb, W, Kst, rec)
    selection = None
    if rank == 0 and world == 1 and args.fast and args.check_streams > 0 and Kst >= 2:
        try:
            selection = selection_check(ctx, dev, depth, rgb, rec[1], 1 + W + 1, min(8, B, res["n_unique_streams"]), rows, cols, K)
        except Exception as e_:
            selection = {"error": f"{type(e_).__name__}: {e_}"}
    eng.close()

    result = None
    if rank == 0:
        traffic, traffic_from = pmc_traffic(B, rows, cols, args.fused)
        u1 = res["u1"]
        headline = rows == 480 and cols == 640 and args.levels == 3
        result = {
            "metric": "aligned RGB-D frames/sec @640x480, 3-level pyr; achieved HBM GB/s vs roofline",
            "value": res["value"], "unit": "frames/s", "n_gpus": world, "steps": Kst, "warmup": W,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic TUM-like {cols}x{rows} RGB-D streams (stand-in for TUM fr1/desk: dataset not in image), "
                                   f"{B} lanes/GPU carrying {res['n_unique_streams']} distinct streams, {args.levels}-level pyramid, GN iterations {iters}, "
                                   f"Student-t + sigmaML, pyrFirst, keyframe iD fusion + keyframe export on, preview off, full trackNewFrame per lane per step",
                       "lanes_per_gpu": B, "graph": bool(args.graph), "fused_gn": bool(args.fused), "fast_numerics": bool(args.fast),
                       "launches_per_step": res["launches_per_step"], "engine_hbm_bytes": res["engine_hbm_bytes"],
                       "tracked_frames_rank0": res["tracked"], "expected_rank0": res["expected"],
                       "keyframe_export_capacity": args.keyframes, "keyframes_exported_in_timed_steps_rank0": res["keyframes_exported"],
                       "n_unique_streams": res["n_unique_streams"]},
            "repetitions": res["repetitions"],
            "frame_level": ({"engine_bytes_per_frame": res["engine_bytes_per_frame"],
                             "engine_gbs": res["value"] / world * res["engine_bytes_per_frame"] / 1e9,
                             "engine_frac_of_hbm_peak": res["value"] / world * res["engine_bytes_per_frame"] / 1e9 / HBM_PEAK_GBS,
                             "engine_bytes_model": {"every_tracked_frame": res["step_bytes_model"][0], "per_odometry_kf_switch": res["step_bytes_model"][1],
                                                    "per_integration_kf_switch": res["step_bytes_model"][2], "per_fused_frame": res["step_bytes_model"][3],
                                                    "odo_kf_switch_rate": res["odo_kf_switch_rate"], "integr_kf_switch_rate": res["integr_kf_switch_rate"],
                                                    "from": "rgbid_engine_step_bytes: algorithmic bytes of the engine's own launch list, per lane"},
                             "u3_bytes_per_frame": U3_BYTES_PER_FRAME,
                             "u3_equivalent_gbs": res["value"] / world * U3_BYTES_PER_FRAME / 1e9,
                             "u3_equivalent_frac_of_hbm_peak": res["value"] / world * U3_BYTES_PER_FRAME / 1e9 / HBM_PEAK_GBS,
                             "note": "engine_*: what the (fused) engine actually has to move per aligned frame x frames/s per GPU -- the fraction of the HBM roofline "
                                     "a step draws; u3_equivalent_*: frames/s priced at SURVEY 8d's UNFUSED budget U3 (275.6 MB/frame), i.e. the rate an unfused "
                                     "implementation would need -- not traffic that happened"} if headline else None),
            "roofline": {"bound": "hbm", "achieved": u1["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": u1["achieved"] / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_from": traffic_from,
                         "kernel": ("rgbid::k_build_system<ByLane<SysParams>, true, 0, 2, 1> (the level-0 Gauss-Newton iterations) and <..., 0, 2, 2> (the covariance pass: same kernel, fixed-nu weights; "
                                    "1 launch in 11): level-0 Gauss-Newton evaluation, FUSED -- warp of the current frame (gathers) + residual rows + "
                                    "27-term normal equations; reads the same 8 fp32 maps as unit U1 (6 keyframe-side streams + the 2 current-frame maps through the gathers), writes nothing"
                                    if args.fused else "rgbid::k_build_system<ByLane<SysParams>, true, 0, 0, 0> (level-0 residual + 27-term normal equations on stored W1 / I1)"),
                         "algorithmic_bytes_per_launch": u1["bytes_per_launch"], "launches_timed": u1["launches_timed"], "avg_launch_us": u1["avg_launch_us"],
                         "timed_in": u1["timed_in"]},
            "parity": res["parity"],
            "lanes_bit_identical": res["parity"]["lanes_bit_identical"],
        }
        if selection is not None:
            result["parity"]["fast_class_selection_vs_exact_kernels"] = selection
        if use_dist:
            med_i = int(np.argsort(res["repetitions"]["frames_per_s"])[::-1][len(res["repetitions"]["frames_per_s"]) // 2])
            result["multi_gpu"] = {"rccl_ranks": rccl_ranks, "rccl_ranks_per_rank": rccl_ranks_all, "gather": gather_how, "record_bytes": 392,
                                   "gathered_bytes_per_rank_per_repetition": B * Kst * 392, "gather_equals_torch_all_gather": gather_ok,
                                   "gather_us_per_repetition_rank0": res["gather_us"],
                                   "per_rank_frames_per_s": [B * Kst / t_ for t_ in res["rank_seconds"][med_i]] if res["rank_seconds"] else None,
                                   "per_rank_note": "each rank's own frames / its own wall time of the median repetition, before the closing barrier; `value` uses the max over ranks"}
        if pcie is not None:
            result["pcie_inclusive"] = pcie
    # ---- extra configurations.  Single-GPU run: the headline workload in the EXACT numerics class (fused, and as the reference's kernel
    # sequence), BASELINE configs 1, 5 and 4, the mounted datasets.  Multi-GPU run: config 4 only (every rank takes part: it is strong-scaled).
    headline_shape = rows == 480 and cols == 640 and args.levels == 3
    extras = []
    if world == 1 and rank == 0 and not args.no_extras and headline_shape:
        inputs = (seqs, depth, rgb)
        for name, fu, fa in (("exact-fused: the headline workload with the gather / filter kernels in the EXACT numerics class (bit-exact pixel selection; the fused kernel k_build_system<..., 1, 0>)", 1, 0),
                             ("exact-unfused: the reference's kernel sequence (warp pair, sigma / nu on stored maps, normal equations k_build_system<..., 0, 0>) in the EXACT class", 0, 0)):
            try:
                rx, keepx = run_config(ctx, dev, work, rows, cols, args.levels, iters, B, Kst, W, 1, args.streams, args.graph, fu, args.keyframes, K,
                                       {"use_dist": False, "world": 1}, check_streams=min(2, args.check_streams), fast_numerics=fa, inputs=inputs)
                keepx[3].close()
                del keepx
                extras.append({"config": name, "value": rx["value"], "unit": "frames/s", "ms_per_step": rx["ms_per_step"], "steps": Kst, "warmup": W, "repetitions": 1,
                               "lanes": B, "fused_gn": bool(fu), "fast_numerics": bool(fa), "tracked": rx["tracked"], "expected": rx["expected"],
                               "u1_achieved_gbs": rx["u1"]["achieved"], "u1_frac_of_hbm_peak": rx["u1"]["achieved"] / HBM_PEAK_GBS, "u1_avg_launch_us": rx["u1"]["avg_launch_us"],
                               "engine_bytes_per_frame": rx["engine_bytes_per_frame"], "parity": rx["parity"]})
            except Exception as e:
                extras.append({"config": name, "error": f"{type(e).__name__}: {e}"})
        # an output-equivalent schedule: vertex / normal maps of the fused keyframe only where they are consumed (rgbid_engine_config.defer_keyframe_maps; the
        # headline keeps the reference's per-frame schedule).  Same records, same exported keyframes (tests/test_gpu_engine.py)
        try:
            rd, keepd = run_config(ctx, dev, work, rows, cols, args.levels, iters, B, Kst, W, 3, args.streams, args.graph, args.fused, args.keyframes, K,
                                   {"use_dist": False, "world": 1}, check_streams=0, fast_numerics=args.fast, inputs=inputs, defer_maps=1)
            same = bool(rd["parity"]["lanes_bit_identical"]) and keepd[4].tobytes() == rec.tobytes()
            keepd[3].close()
            del keepd
            extras.append({"config": "deferred-keyframe-maps: the headline workload with the fused keyframe's vertex / normal maps computed only where consumed "
                                     "(keyframe export, preview, accessor) instead of after every fusion step; opt-in (defer_keyframe_maps), NOT the headline",
                           "value": rd["value"], "unit": "frames/s", "ms_per_step": rd["ms_per_step"], "lanes": B, "steps": Kst, "warmup": W, "repetitions": 3,
                           "records_identical_to_headline_run": same, "engine_bytes_per_frame": rd["engine_bytes_per_frame"]})
        except Exception as e:
            extras.append({"config": "deferred-keyframe-maps", "error": f"{type(e).__name__}: {e}"})
        # what smaller batches deliver (the headline needs 2 048 concurrent streams), and the PCIe-inclusive rate
        for Bs in (1, 8, 64, 512):     # 1 lane: ms_per_step is the latency of one stream's frame (what an 8-GPU strong-scaling run approaches)
            if Bs >= B:
                continue
            try:
                sub = (seqs, depth[:, :Bs], rgb[:, :Bs])
                rs, keeps = run_config(ctx, dev, work, rows, cols, args.levels, iters, Bs, Kst, W, 3, args.streams, args.graph, args.fused, args.keyframes, K,
                                       {"use_dist": False, "world": 1}, check_streams=0, fast_numerics=args.fast, inputs=sub)
                entry = {"config": f"lanes-{Bs}: the headline workload with {Bs} concurrent streams per GPU ({keeps[3].bytes() / 1e9:.1f} GB of engine state)", "value": rs["value"], "unit": "frames/s",
                         "ms_per_step": rs["ms_per_step"], "lanes": Bs, "steps": Kst, "warmup": W, "repetitions": 3, "u1_frac_of_hbm_peak": rs["u1"]["achieved"] / HBM_PEAK_GBS,
                         "lanes_bit_identical": rs["parity"]["lanes_bit_identical"]}
                keeps[3].close()
                del keeps
                if Bs <= 64 and not args.graph:   # the few-lane regime is launch-bound: the same steps replayed as one hipGraph each
                    rg, keepg = run_config(ctx, dev, work, rows, cols, args.levels, iters, Bs, Kst, W, 3, args.streams, 1, args.fused, args.keyframes, K,
                                           {"use_dist": False, "world": 1}, check_streams=0, fast_numerics=args.fast, inputs=sub)
                    entry["hipgraph"] = {"value": rg["value"], "ms_per_step": rg["ms_per_step"], "lanes_bit_identical": rg["parity"]["lanes_bit_identical"]}
                    keepg[3].close()
                    del keepg
                del sub
                extras.append(entry)
            except Exception as e:
                extras.append({"config": f"lanes-{Bs}", "error": f"{type(e).__name__}: {e}"})
        del inputs
    if rank == 0 and world == 1 and not args.no_extras and headline_shape:
        del depth, rgb, keep, gathered
        torch.cuda.empty_cache()
        try:
            extras.append(extra_config1(ctx, dev, K))
        except Exception as e:
            extras.append({"config": "1", "error": f"{type(e).__name__}: {e}"})
        try:
            K5 = (1050.0, 1050.0, 639.5, 479.5)
            it5 = [10, 5, 3, 3]
            r5, keep5 = run_config(ctx, dev, work, 960, 1280, 4, it5, 128, min(Kst, 8), 1, 1, 8, args.graph, args.fused, args.keyframes, K5,
                                   {"use_dist": False, "world": 1}, check_streams=min(2, args.check_streams), fast_numerics=args.fast)
            keep5[3].close()
            del keep5
            torch.cuda.empty_cache()
            extras.append({"config": "5: 1280x960 upsampled synthetic stream, 4-level pyramid, GN iterations [10,5,3,3], 128 lanes (8 distinct streams), same engine switches as the headline run",
                           "value": r5["value"], "unit": "frames/s", "ms_per_step": r5["ms_per_step"], "steps": min(Kst, 8), "warmup": 1,
                           "tracked": r5["tracked"], "expected": r5["expected"], "lanes_bit_identical": r5["parity"]["lanes_bit_identical"], "parity": r5["parity"],
                           "u1_achieved_gbs": r5["u1"]["achieved"], "u1_frac_of_hbm_peak": r5["u1"]["achieved"] / HBM_PEAK_GBS,
                           "u1_avg_launch_us": r5["u1"]["avg_launch_us"], "u1_bytes_per_launch": r5["u1"]["bytes_per_launch"],
                           "u1_launches_timed": r5["u1"]["launches_timed"], "engine_hbm_bytes": r5["engine_hbm_bytes"]})
        except Exception as e:
            extras.append({"config": "5", "error": f"{type(e).__name__}: {e}"})
    if not args.no_extras and headline_shape and args.seq_frames > 0:
        if world > 1:
            del depth, rgb, keep, gathered
            torch.cuda.empty_cache()
        try:
            c4 = extra_config4(ctx, dev, K, world, rank, args.seq_frames, args.seq_chunks_per_gpu, args.fused, args.fast)
            if rank == 0:
                extras.append(c4)
        except Exception as e:
            if world > 1:       # a rank that drops out of a collective leaves the others hanging: end the job loudly
                sys.stderr.write(f"[bench] rank {rank}: extra config 4 failed ({type(e).__name__}: {e})\n"); sys.stderr.flush()
                os._exit(4)
            extras.append({"config": "4", "error": f"{type(e).__name__}: {e}"})
    if rank == 0 and world == 1 and not args.no_extras and headline_shape:
        try:
            extras.append(extra_kfalign(ctx, dev, K))
        except Exception as e:
            extras.append({"config": "f-1 (KeyframeAlign batched)", "error": f"{type(e).__name__}: {e}"})
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            ds = dataset_configs(K)
            if ds is not None:
                extras.append(ds)
        except Exception as e:
            extras.append({"config": "2-4 (RGBID_TUM_DIR)", "error": f"{type(e).__name__}: {e}"})
    if rank == 0 and extras:
        result["extra_configs"] = extras
        # north_star's own target kernel -- "the 640x480 residual + JTJ reduction" = unit U1 on stored W1 / I1 -- is what the reference's kernel sequence runs
        # (k_build_system<..., 0, 0>); it was timed above with HIP events in the `exact-unfused` configuration of the same workload: quote it next to the fused kernel
        for x in extras:
            if str(x.get("config", "")).startswith("exact-unfused") and x.get("u1_frac_of_hbm_peak"):
                result["roofline"]["u1_residual_jtj_kernel_unfused"] = {
                    "kernel": "rgbid::k_build_system<ByLane<SysParams>, true, 0, 0, 0> (residual + 27-term normal equations on stored W1 / I1: unit U1, 32 B/px)",
                    "achieved": x["u1_achieved_gbs"], "frac": x["u1_frac_of_hbm_peak"], "avg_launch_us": x["u1_avg_launch_us"], "unit": "GB/s",
                    "from": "extra_configs `exact-unfused` (the headline workload run as the reference's kernel sequence; HIP events over its timed level-0 launches)"}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(seqs[0], rows, cols, K)
        else:
            result["cpu_baseline"] = None
        flush_c_stdio()
        print(json.dumps(result), flush=True)   # before any teardown: nothing that happens at exit may cost the result line
    if dist_env.get("comm") is not None:
        dist_env["comm"].close()
    ctx.close()
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
