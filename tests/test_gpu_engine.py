"""GPU parity of the batched device-resident tracker (rgbid_engine) against the CPU oracle's VisodoTracker
restatement, lane by lane, on synthetic sequences with analytic ground truth.

Tolerance (north star): pose error vs the reference algorithm < 1e-4 rad / 1e-4 m per frame.  The engine and the
oracle share no code: the oracle is scalar C with double accumulation on the host, the engine runs fp32 partial
sums + double reductions + the 6x6 solve on the device.  Keyframe decisions (status bits) must agree exactly.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from rgbid import device, synth
from rgbid import engine as E
from tests.util import assert_bits

pytestmark = pytest.mark.gpu


def rot_angle(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1) / 2
    return float(np.arccos(np.clip(c, -1, 1)))


def make_lanes(n_lanes, n_frames, rows, cols, K, **kw):
    seqs = [synth.make_sequence(n_frames, seed=synth.SEED + 17 * l, K=K, rows=rows, cols=cols, device="cuda", **kw) for l in range(n_lanes)]
    depth = torch.stack([s["depth"] for s in seqs], 1).to(torch.int16).contiguous()   # [T, B, rows, cols]
    rgb = torch.stack([s["rgb"] for s in seqs], 1).contiguous()                       # [T, B, rows, cols, 3]
    return seqs, depth, rgb


def run_case(ctx, rows, cols, K, n_lanes, n_frames, cfg_kw, seq_kw, use_graph, map_outliers=5e-3, sigma_tol=1e-3, pose_tol=1e-4, chi_margin_tol=0.0):
    seqs, depth, rgb = make_lanes(n_lanes, n_frames, rows, cols, K, **seq_kw)
    eng = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=n_lanes, K=K, use_graph=use_graph, record_capacity=n_frames, **cfg_kw))
    for k in range(n_frames):
        eng.step(depth[k], rgb[k])
    rec = eng.records()
    okw = {k: v for k, v in cfg_kw.items() if k not in ("fused_gn", "fast_numerics")}
    okw.update(rows=rows, cols=cols, fx=K[0], fy=K[1], cx=K[2], cy=K[3])
    worst_r = worst_t = 0.0
    imposed = 0
    frames_checked = widened_flip = widened_stop = 0
    chi_margin = np.full((n_lanes, n_frames), 1e30)      # termination = CHI_SQUARED: how close each oracle frame came to the other branch of RMSE > RMSE_prev
    chi_stops = chi_on_threshold = 0
    th_o, th_i = cfg_kw.get("visratio_odo", 0.9), cfg_kw.get("visratio_integr", 0.7)
    fin = cfg_kw.get("finest_level", 0)
    n_lattice = device.error_lattice_size(rows >> fin, cols >> fin, cfg_kw.get("nsamples", 10000))[0]
    for l in range(n_lanes):
        trk = O.Tracker(O.default_config(**okw))
        d = depth[:, l].cpu().numpy().view(np.uint16)
        c = rgb[:, l].cpu().numpy()
        for k in range(n_frames):
            st = int(rec[k, l]["status"])
            if k:
                # EVERY frame of every lane is checked: where a covisibility ratio lands on its threshold the two implementations may decide
                # differently (their poses differ by ~1e-6); the oracle then continues with the engine's decision imposed (test hook,
                # oracle/rgbid_oracle.h) -- and the natural decision may only have differed because the ratio sat on the threshold
                trk.force_kf_decisions(bool(st & E.ST_ODO_KF), bool(st & E.ST_INTEGR_KF))
            ret = trk.track(d[k], c[k])
            info = trk.last_info()
            if k == 0:
                assert st & E.ST_FIRST
                continue
            chi_margin[l, k] = info.chi_stop_margin_frame; chi_stops += info.chi_stops_frame
            assert bool(st & E.ST_TRACKED) == ret, (l, k, st, ret)
            if not ret:
                continue
            # covisibility ratios are counts of gated pixels: ~1e-6 pose differences may flip a handful of them
            assert abs(rec[k, l]["vis_odo"] - info.visratio_odo) < 5e-4 and abs(rec[k, l]["vis_integr"] - info.visratio_integr) < 5e-4
            if bool(st & E.ST_ODO_KF) != bool(info.odo_kf_natural):
                assert abs(info.visratio_odo - th_o) < 5e-4, (l, k, st, info.visratio_odo)
                imposed += 1
            if bool(st & E.ST_INTEGR_KF) != bool(info.integr_kf_natural):
                assert abs(info.visratio_integr - th_i) < 5e-4, (l, k, st, info.visratio_integr)
                imposed += 1
            # FAST class x CHI_SQUARED (round 6): `RMSE > RMSE_prev` (visodo.cpp:1150) is a discontinuity -- where the two RMSEs agree to within the class's value
            # tolerance the engine may take the other branch, i.e. run a different number of iterations on that level: the frame's last-iteration statistics
            # and, below, its pose are then compared with the wider bound, under that condition only, counted and reported
            on_chi = chi_margin_tol > 0 and chi_margin[l, max(1, k - 1):k + 1].min() < chi_margin_tol
            if not on_chi:
                assert rec[k, l]["nu_depthinv"] == info.nu_depthinv and rec[k, l]["nu_int"] == info.nu_int, (l, k)
            # sigma is the scale of the residuals AT the current pose estimate, which itself agrees to ~1e-5: 1e-3 relative on a full-size lattice.
            # Two discontinuities of the reference algorithm widen that: (1) a pose difference of 1e-7 can move a point-sampled or range-checked
            # pixel across a pixel boundary; a Student-t sample's weighted square is bounded by (nu + 1) sigma^2, so ONE flipped sample moves sigma by
            # up to ~5.5 / n_samples relative -- on the 5 000-sample lattice of a 62x84 image that is 1e-3 per sample (seen at seed 14 of a fuzz
            # campaign: 2.3e-3, exact and fast engine alike): allow three; (2) the sigma iteration stops when its relative change drops below 0.1:
            # with the ratio ON that threshold (the oracle reports the distance, rgbid_oracle.h sigma_stop_margin_frame) the two may stop one iteration
            # apart, which moves sigma by a few 1e-3.
            # The TIGHT tolerance is the default; a frame may use one of the two widened bounds only under its stated condition, and the frames that
            # did are counted and reported (the gate must not loosen silently).
            ds = abs(rec[k, l]["sigma_int"] - info.sigma_int)
            frames_checked += 1
            if not ds < sigma_tol * info.sigma_int:
                if on_chi and ds < 5e-2 * info.sigma_int:
                    pass
                elif ds < 15.0 / n_lattice * info.sigma_int:
                    widened_flip += 1
                else:
                    assert info.sigma_stop_margin_frame < 1e-4 and ds < 2e-2 * info.sigma_int, (l, k, rec[k, l]["sigma_int"], info.sigma_int, info.sigma_stop_margin_frame, n_lattice)
                    widened_stop += 1
                    print(f"lane {l} frame {k}: sigma iteration on its stopping threshold (margin {info.sigma_stop_margin_frame:.1e}): sigma_int {rec[k, l]['sigma_int']} vs {info.sigma_int}")
        Rs, ts = trk.poses()
        oR, ot, ocov = trk.odometry()
        for k in range(1, n_frames):
            er, et = rot_angle(Rs[k], rec[k, l]["R"]), float(np.linalg.norm(ts[k] - rec[k, l]["t"]))
            worst_r, worst_t = max(worst_r, er), max(worst_t, et)
            tol_k, cov_tol = pose_tol, 1e-2
            if not (er < pose_tol and et < pose_tol) and chi_margin_tol > 0:
                # the oracle's own RMSE comparison of this frame (or of the frame before: the velocity prior carries one frame) sat inside the value tolerance
                assert chi_margin[l, max(1, k - 1):k + 1].min() < chi_margin_tol, (l, k, er, et, chi_margin[l])
                chi_on_threshold += 1; tol_k, cov_tol = 10 * pose_tol, 1e-1
            assert er < tol_k and et < tol_k, (l, k, er, et)
            assert rot_angle(oR[k], rec[k, l]["odo_R"]) < tol_k and np.linalg.norm(ot[k] - rec[k, l]["odo_t"]) < tol_k
            sc = np.sqrt(np.outer(np.diag(ocov[k]), np.diag(ocov[k]))) + 1e-30
            assert (np.abs(ocov[k] - rec[k, l]["odo_cov"]) / sc).max() < cov_tol, (l, k)
        # fused keyframe maps of the lane vs the oracle tracker's
        kd, kw, kv, kn, km = eng.keyframe_maps(l)
        od, ow = trk.kf_depthinv(), trk.kf_weight()
        nan_mismatch = np.count_nonzero(np.isnan(kd) != np.isnan(od))
        assert nan_mismatch <= 2e-3 * od.size, nan_mismatch
        both = ~np.isnan(kd) & ~np.isnan(od)
        # pose differences of ~1e-6 flip the fusion gate (|w_s - w_KF| < 0.0225) or the point-sampled source pixel for a handful of
        # pixels: all but a small fraction of the fused map must agree to 1e-4 (inverse depth) / 1e-3 (weight), the median much better
        rel = np.abs(kd[both] - od[both]) / np.abs(od[both])
        assert np.count_nonzero(rel > 1e-4) <= max(16, map_outliers * rel.size), (np.count_nonzero(rel > 1e-4), rel.size)
        assert np.median(rel) < 1e-5
        relw = np.abs(kw[both] - ow[both]) / np.abs(ow[both])
        assert np.count_nonzero(relw > 1e-3) <= max(16, 1e-2 * relw.size), (np.count_nonzero(relw > 1e-3), relw.size)
        assert np.count_nonzero(km != trk.kf_overlap_mask()) <= 2e-3 * km.size
        # ground truth sanity: the tracker follows the synthetic camera (sensor noise limits the accuracy)
        Rg, tg = seqs[l]["R_wc"].numpy(), seqs[l]["t_wc"].numpy()
        assert rot_angle(Rg[-1], rec[-1, l]["R"]) < 5e-3 and np.linalg.norm(tg[-1] - rec[-1, l]["t"]) < 2e-2
        trk.close()
    eng.close()
    if imposed:
        print(f"keyframe decisions on the threshold, imposed on the oracle: {imposed}")
    if cfg_kw.get("termination") == O.CHI_SQUARED:
        fin_ = chi_margin[:, 1:][chi_margin[:, 1:] < 1e29]
        print(f"chi-squared termination: {chi_stops} early level exits in the oracle over {n_lanes * (n_frames - 1)} frames; smallest relative RMSE margin {fin_.min() if fin_.size else float('nan'):.2e}; "
              f"{chi_on_threshold} frames beyond {pose_tol:g} with the comparison inside the value tolerance ({chi_margin_tol:g})")
        assert chi_stops > 0 or cfg_kw.get("sigma_estimator") == O.SIGMA_CONS or cfg_kw.get("levels", 3) < 3   # the early exit really fires on the plain sequences
        assert chi_on_threshold <= max(1, 0.15 * n_lanes * (n_frames - 1))
    if widened_flip or widened_stop:
        print(f"sigma: {widened_flip} of {frames_checked} frames used the flipped-sample bound (15 / n_lattice), {widened_stop} the stopping-threshold bound")
    assert widened_flip + widened_stop <= max(2, 0.15 * frames_checked), (widened_flip, widened_stop, frames_checked)
    return worst_r, worst_t


@pytest.mark.parametrize("use_graph", [0, 1])
def test_engine_vs_oracle_small(ctx, use_graph):
    K = tuple(v / 4 for v in synth.TUM_K)
    K = (K[0], K[1], (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
    wr, wt = run_case(ctx, 120, 160, K, n_lanes=3, n_frames=7, cfg_kw=dict(), seq_kw=dict(trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8)), use_graph=use_graph)
    print("worst pose deviation engine vs oracle:", wr, wt)


def test_engine_long_sequence(ctx):
    """40 frames per lane with keyframe switches and fusion: the engine must not drift away from the oracle tracker (every frame is
    still held to 1e-4 rad / 1e-4 m and to identical keyframe decisions)."""
    K = (synth.TUM_K[0] / 4, synth.TUM_K[1] / 4, (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
    wr, wt = run_case(ctx, 120, 160, K, n_lanes=2, n_frames=40, cfg_kw=dict(visratio_odo=0.97, visratio_integr=0.93),
                      seq_kw=dict(trans_step=(0.004, 0.012), rot_step_deg=(0.2, 0.8)), use_graph=0)
    print("worst pose deviation over 40 frames:", wr, wt)


def test_engine_warp_first(ctx):
    """warping = WARP_FIRST (visodo.cpp:1078-1105): warp the level-0 frame, pyrDown the warped maps to the working level."""
    K = (synth.TUM_K[0] / 4, synth.TUM_K[1] / 4, (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
    run_case(ctx, 120, 160, K, n_lanes=2, n_frames=5, cfg_kw=dict(warping=O.WARP_FIRST), seq_kw=dict(trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8)), use_graph=0)


def test_engine_skips_levels_without_iterations(ctx):
    """iterations {3, 0, 4}: the middle level runs no iteration, so the coarse level hands its pose straight to level 0."""
    K = (synth.TUM_K[0] / 4, synth.TUM_K[1] / 4, (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
    run_case(ctx, 120, 160, K, n_lanes=2, n_frames=4, cfg_kw=dict(iters=[3, 0, 4]), seq_kw=dict(trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8)), use_graph=0)


def test_engine_fused_gn_is_bit_identical(ctx):
    """The fused warp+residual+JTJ kernel computes W1/I1 with the same per-pixel device functions and accumulates in the
    same per-thread order as the unfused kernels: poses must be IDENTICAL, not merely close."""
    K = (131.25, 131.25, 79.5, 59.5)
    seqs, depth, rgb = make_lanes(3, 6, 120, 160, K, trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8))
    recs = []
    for fused in (0, 1):
        eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=3, K=K, use_graph=0, fused_gn=fused, record_capacity=6, fast_numerics=0))
        for k in range(6):
            eng.step(depth[k], rgb[k])
        recs.append(eng.records())
        eng.close()
    for name in ("R", "t", "odo_cov", "status", "nu_int", "sigma_depthinv", "vis_odo"):
        assert np.array_equal(recs[0][name], recs[1][name]), name


@pytest.mark.parametrize("rows,cols,levels", [(120, 160, 3), (122, 164, 2), (480, 640, 3)])
def test_engine_fused_fast_is_bit_identical_to_unfused_fast(ctx, rows, cols, levels):
    """REGRESSION test (the parity evidence for the fused kernel is tests/test_gpu_batched.py, kernel against oracle): the fused kernel and the
    stand-alone fast warp pair + normal equations evaluate the same device functions on the same pixels and accumulate the rows in the same
    order (same launch plan): records must be IDENTICAL.
    (122 x 164: rows that are not a multiple of the launch geometry, lanes past the end of the image inside a live wave.)"""
    s = cols / 640.0
    K = (525.0 * s, 525.0 * s, (319.5 + 0.5) * s - 0.5, (239.5 + 0.5) * rows / 480.0 - 0.5)
    T, B = (6, 3) if rows < 480 else (3, 2)
    seqs, depth, rgb = make_lanes(B, T, rows, cols, K, trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8))
    recs = []
    for fused in (0, 1):
        eng = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=B, K=K, levels=levels, iters=[6, 4, 3][:levels], use_graph=0, fused_gn=fused,
                                             record_capacity=T, fast_numerics=1))
        for k in range(T):
            eng.step(depth[k], rgb[k])
        recs.append(eng.records())
        eng.close()
    for name in ("R", "t", "odo_cov", "status", "nu_int", "sigma_depthinv", "vis_odo"):
        assert np.array_equal(recs[0][name], recs[1][name]), name


def test_engine_fused_vs_oracle_full_res(ctx):
    wr, wt = run_case(ctx, 480, 640, synth.TUM_K, n_lanes=2, n_frames=4, cfg_kw=dict(fused_gn=1), seq_kw=dict(), use_graph=1)
    print("worst pose deviation fused engine vs oracle (640x480):", wr, wt)


def test_engine_keyframe_switches(ctx):
    """Tight visibility thresholds force odometry / integration keyframe switches inside a short sequence."""
    K = (131.25, 131.25, 79.5, 59.5)
    run_case(ctx, 120, 160, K, n_lanes=2, n_frames=8, cfg_kw=dict(visratio_odo=0.985, visratio_integr=0.97),
             seq_kw=dict(trans_step=(0.01, 0.02), rot_step_deg=(0.5, 1.0)), use_graph=1)


def test_engine_vs_oracle_full_res(ctx):
    """BASELINE config 2 stand-in: 640x480, 3 levels, {10,5,3}, Student-t + sigmaML, pyrFirst, fusion on."""
    wr, wt = run_case(ctx, 480, 640, synth.TUM_K, n_lanes=2, n_frames=5, cfg_kw=dict(), seq_kw=dict(), use_graph=1)
    print("worst pose deviation engine vs oracle (640x480):", wr, wt)


@pytest.mark.parametrize("rows,cols", [(120, 160), (480, 640)])
def test_engine_fast_numerics_vs_exact(ctx, rows, cols):
    """The engine's default gather kernels (fast_numerics = 1: reference-build-class arithmetic) against its bit-exact ones on the same frames:
    same keyframe decisions, poses within 1e-5 rad / 1e-5 m (measured ~1e-7..1e-6), fused keyframe maps equal except boundary pixels --
    the same order the oracle's model of the reference's nvcc flags moves the results (tests/test_oracle_cuda_numerics.py)."""
    s = cols / 640.0
    K = (525.0 * s, 525.0 * s, (319.5 + 0.5) * s - 0.5, (239.5 + 0.5) * s - 0.5)
    T, B = (10, 3) if rows == 120 else (5, 2)
    kw = dict(trans_step=(0.004, 0.012), rot_step_deg=(0.2, 0.8)) if rows == 120 else dict()
    seqs, depth, rgb = make_lanes(B, T, rows, cols, K, **kw)
    out = []
    for fast in (0, 1):
        eng = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=B, K=K, use_graph=0, record_capacity=T, fast_numerics=fast))
        for k in range(T):
            eng.step(depth[k], rgb[k])
        out.append((eng.records().copy(), [eng.keyframe_maps(l) for l in range(B)]))
        eng.close()
    (ra, ma), (rb, mb) = out
    assert np.array_equal(ra["status"], rb["status"])
    assert np.array_equal(ra["nu_int"], rb["nu_int"]) and np.array_equal(ra["nu_depthinv"], rb["nu_depthinv"])
    wr = max(rot_angle(ra[k, l]["R"], rb[k, l]["R"]) for k in range(1, T) for l in range(B))
    wt = float(np.abs(ra["t"] - rb["t"]).max())
    assert np.abs(ra["vis_odo"] - rb["vis_odo"]).max() < 2e-4 and np.abs(ra["vis_integr"] - rb["vis_integr"]).max() < 2e-4
    print(f"fast vs exact engine {cols}x{rows}: worst pose difference {wr:.2e} rad / {wt:.2e} m")
    assert wr < 1e-5 and wt < 1e-5
    for l in range(B):
        kd_a, kw_a = ma[l][0], ma[l][1]; kd_b, kw_b = mb[l][0], mb[l][1]
        n = kd_a.size
        assert np.count_nonzero(np.isnan(kd_a) != np.isnan(kd_b)) <= 1e-3 * n
        m = ~np.isnan(kd_a) & ~np.isnan(kd_b)
        rel = np.abs(kd_a[m] - kd_b[m]) / np.abs(kd_a[m])
        assert np.count_nonzero(rel > 1e-4) <= max(16, 2e-3 * rel.size) and np.median(rel) < 1e-6, (np.count_nonzero(rel > 1e-4), rel.size)


def test_engine_vs_oracle_1280x960_four_levels(ctx):
    """BASELINE config 5: 1280x960 upsampled synthetic stream, 4-level pyramid (the reference's LEVELS is a compile-time constant,
    include/visodo.h:52; iterations {10,5,3,3} follow src/visodo.cpp:652-655), 2 lanes x 4 frames of the batched engine vs the oracle."""
    K = (1050.0, 1050.0, 639.5, 479.5)    # TUM factory intrinsics x2 with the half-pixel centre shift ((c + 0.5) * 2 - 0.5)
    wr, wt = run_case(ctx, 960, 1280, K, n_lanes=2, n_frames=4, cfg_kw=dict(levels=4, iters=[10, 5, 3, 3]), seq_kw=dict(), use_graph=0)
    print("worst pose deviation engine vs oracle (1280x960, 4 levels):", wr, wt)


def test_chunked_sequence_matches_sequential(ctx):
    """SURVEY 8e parity definition: a sequence tracked as 4 overlapping chunks (one engine lane each) must agree with the
    unsharded sequential run up to Gauss-Newton convergence tolerance at the chunk heads (fresh keyframe, no velocity prior),
    and both must follow the ground truth."""
    from rgbid import sequence
    K = (131.25, 131.25, 79.5, 59.5)
    T = 13
    seq = synth.make_sequence(T, K=K, rows=120, cols=160, device="cuda", trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8))
    depth = seq["depth"].to(torch.int16).contiguous(); rgb = seq["rgb"].contiguous()
    R1, t1, _ = sequence.track_chunked(ctx, depth, rgb, 1, K)
    R4, t4, ranges = sequence.track_chunked(ctx, depth, rgb, 4, K)
    assert ranges == [(0, 3), (3, 6), (6, 9), (9, 12)]
    Rg, tg = seq["R_wc"].numpy(), seq["t_wc"].numpy()
    d_rot = max(rot_angle(R1[k], R4[k]) for k in range(T)); d_tr = max(np.linalg.norm(t1[k] - t4[k]) for k in range(T))
    e1 = np.linalg.norm(t1[-1] - tg[-1]); e4 = np.linalg.norm(t4[-1] - tg[-1])
    print("chunked vs sequential:", d_rot, d_tr, " | end-point error vs ground truth:", e1, e4)
    assert d_rot < 2e-3 and d_tr < 5e-3 and e1 < 1e-2 and e4 < 1e-2
    # chunk 0 alone is bit-identical to the first frames of the sequential run (same keyframe, same prior)
    assert np.array_equal(R1[:4], R4[:4]) and np.array_equal(t1[:4], t4[:4])


def test_engine_reset_and_streamed_inputs_are_bit_identical():
    """reset() restores the exact initial state (a second run over the same frames reproduces every record bit for bit), and
    frames streamed from pinned host memory on a copy stream -- double-buffered against the engine's own stream with events, the
    way bench.py --h2d does -- give the same records as device-resident inputs."""
    from rgbid import device
    dev = torch.device("cuda", 0)
    K = (synth.TUM_K[0] / 4, synth.TUM_K[1] / 4, (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
    T, B = 6, 4
    seqs, depth, rgb = make_lanes(B, T, 120, 160, K)
    work = torch.cuda.Stream(dev)
    with torch.cuda.stream(work):
        ctx = device.Context(0)          # the context adopts the (non-null) torch stream that is current at creation
    ctx.set_async(1)
    eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=B, K=K, use_graph=0, record_capacity=T))
    torch.cuda.synchronize()

    def resident():
        for k in range(T):
            eng.step(depth[k], rgb[k])
        return eng.records(0, T).copy()

    a = resident(); eng.reset(); b = resident()
    assert a.tobytes() == b.tobytes()
    depth_h, rgb_h = depth.cpu().pin_memory(), rgb.cpu().pin_memory()
    bufs = [(torch.empty_like(depth[0]), torch.empty_like(rgb[0])) for _ in range(2)]
    copy_stream = torch.cuda.Stream(dev)
    ready = [torch.cuda.Event() for _ in range(2)]; free = [torch.cuda.Event() for _ in range(2)]

    def upload(k, slot):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(free[slot])
            bufs[slot][0].copy_(depth_h[k], non_blocking=True); bufs[slot][1].copy_(rgb_h[k], non_blocking=True)
            ready[slot].record(copy_stream)

    eng.reset()
    for ev in free:
        ev.record(work)
    upload(0, 0)
    for k in range(T):
        slot = k % 2
        if k + 1 < T:
            upload(k + 1, 1 - slot)
        work.wait_event(ready[slot])
        eng.step(bufs[slot][0], bufs[slot][1])
        free[slot].record(work)
    c = eng.records(0, T).copy()
    assert a.tobytes() == c.tobytes()
    eng.close(); ctx.close()


def test_engine_lost_and_recovery(ctx):
    """A lane whose sensor drops out for one frame (no valid depth): the normal equations are empty, the on-device solve yields NaN,
    the lane flags itself lost, re-keys and recovers exactly like the oracle tracker -- while its neighbour lanes are unaffected."""
    K = (synth.TUM_K[0] / 4, synth.TUM_K[1] / 4, (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
    n, B = 8, 3
    seqs, depth, rgb = make_lanes(B, n, 120, 160, K, trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8))
    depth[3, 1] = 0                                     # lane 1, frame 3
    eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=B, K=K, use_graph=0, record_capacity=n))
    for k in range(n):
        eng.step(depth[k], rgb[k])
    rec = eng.records()
    for l in range(B):
        trk = O.Tracker(O.default_config(rows=120, cols=160, fx=K[0], fy=K[1], cx=K[2], cy=K[3]))
        d = depth[:, l].cpu().numpy().view(np.uint16); c = rgb[:, l].cpu().numpy()
        pose_idx = []
        for k in range(n):
            before = len(trk.poses()[0])
            ret = trk.track(d[k], c[k])
            pose_idx.append(len(trk.poses()[0]) - 1 if len(trk.poses()[0]) > before or k == 0 else None)
            st = int(rec[k, l]["status"])
            if k:
                assert bool(st & E.ST_TRACKED) == ret, (l, k, st, ret)
                assert bool(st & E.ST_LOST) == bool(trk.last_info().lost), (l, k, st)
        Rs, ts = trk.poses()
        if l == 1:
            assert pose_idx[3] is not None and pose_idx[4] is None and pose_idx[5] is not None      # the frame tracked while lost adds no pose
        else:
            assert None not in pose_idx
        for k in range(1, n):
            if pose_idx[k] is None:
                continue
            i = pose_idx[k]
            er, et = rot_angle(Rs[i], rec[k, l]["R"]), float(np.linalg.norm(ts[i] - rec[k, l]["t"]))
            assert er < 1e-4 and et < 1e-4, (l, k, er, et)
        trk.close()
    eng.close()


def test_engine_step_accepts_torch_temporaries_async():
    """Engine.step orders the context's stream after torch's current stream and keeps its inputs alive until the next host sync, so
    per-step temporaries produced on torch's stream (gathers, casts) are safe with an asynchronous context."""
    from rgbid import device
    K = (synth.TUM_K[0] / 4, synth.TUM_K[1] / 4, (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
    T, B = 8, 4
    seqs, depth, rgb = make_lanes(B, T, 120, 160, K)
    ctx = device.Context(0)               # torch's current stream is the null stream here: the context owns a private stream
    ctx.set_async(1)
    eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=B, K=K, use_graph=0, record_capacity=T))
    for k in range(T):
        eng.step(depth[k], rgb[k])
    a = eng.records(0, T).copy()
    eng.reset()
    perm = torch.arange(B, device="cuda")
    for k in range(T):
        # fresh temporaries every step, dropped immediately; extra allocations in between invite the caching allocator to recycle them
        eng.step((depth[k].to(torch.int32) + 0)[perm].to(torch.int16).contiguous(), rgb[k][perm].clone())
        junk = torch.full((B, 120, 160, 3), 7, dtype=torch.uint8, device="cuda"); del junk
        junk = torch.full((B, 120, 160), 9, dtype=torch.int16, device="cuda"); del junk
    b = eng.records(0, T).copy()
    assert a.tobytes() == b.tobytes()
    eng.close(); ctx.close()


@pytest.mark.parametrize("name,rows,cols,cfg_kw", [
    ("odd size (scalar kernel paths), 2 levels", 122, 166, dict(levels=2, iters=[6, 4])),
    ("single level", 120, 160, dict(levels=1, iters=[8])),
    ("finest level 1", 120, 160, dict(finest_level=1, iters=[0, 6, 4])),
    ("Huber + sigma const + min weight", 120, 160, dict(mestimator=O.HUBER, sigma_estimator=O.SIGMA_CONS, weighting=O.MIN_WEIGHT)),
    ("Tukey + filtered gradients + no motion model", 120, 160, dict(mestimator=O.TUKEY, image_filtering=O.FILTER_GRADS, motion_model=O.NO_MM)),
    ("geometric only", 120, 160, dict(weighting=O.GEOM_ONLY)),
    ("photometric only, exact bilinear", 120, 160, dict(weighting=O.PHOT_ONLY, interp_mode=O.INTERP_EXACT)),
    ("keyframe counters", 120, 160, dict(max_odoKF_count=2, max_integrKF_count=3)),
    # round 5: termination = CHI_SQUARED (visodo.cpp:1134-1164) inside the engine -- a per-lane flag ends the level; with WARP_FIRST the level-0 warped maps the
    # test reads are fresh at every level; exact numerics class (an RMSE comparison is a discontinuity the value tolerance of the FAST class could sit on)
    ("chi-squared termination, warp first", 120, 160, dict(termination=O.CHI_SQUARED, warping=O.WARP_FIRST, fast_numerics=0)),
    # round 6: PYR_FIRST x CHI_SQUARED -- at levels > 0 the test reads whatever the LAST level-0 warp left in the level-0 warped maps
    # (the previous frame's final iteration; all-zero on the first tracked frame, where the reference reads uninitialised memory: both implementations
    # under test start from zero-filled maps, so frame 1 is compared too and the statement about the reference begins at frame 2)
    ("chi-squared termination, pyr first (stale level-0 maps)", 120, 160, dict(termination=O.CHI_SQUARED, warping=O.PYR_FIRST, fast_numerics=0)),
    # ... and both in the class the engine runs by default: a frame may leave the pose bar only where the oracle's RMSE comparison sat inside the class's
    # value tolerance (oracle diagnostic chi_stop_margin_frame, used like sigma_stop_margin_frame), counted and reported
    ("chi-squared termination, warp first, FAST class", 120, 160, dict(termination=O.CHI_SQUARED, warping=O.WARP_FIRST)),
    ("chi-squared termination, pyr first, FAST class", 120, 160, dict(termination=O.CHI_SQUARED, warping=O.PYR_FIRST)),
    ("chi-squared termination + Huber + sigma const + min weight, pyr first", 120, 160, dict(termination=O.CHI_SQUARED, warping=O.PYR_FIRST, fast_numerics=0, mestimator=O.HUBER,
                                                                                            sigma_estimator=O.SIGMA_CONS, weighting=O.MIN_WEIGHT)),
    ("chi-squared termination + filtered gradients, warp first, odd size", 122, 166, dict(termination=O.CHI_SQUARED, warping=O.WARP_FIRST, fast_numerics=0, image_filtering=O.FILTER_GRADS,
                                                                                         levels=2, iters=[6, 4])),
])
def test_engine_configurations(ctx, name, rows, cols, cfg_kw):
    """every run-time switch of the tracker through the batched engine, each held to the oracle (1e-4 rad / 1e-4 m, same keyframe decisions)"""
    s = cols / 640.0
    K = (525.0 * s, 525.0 * s, (319.5 + 0.5) * s - 0.5, (239.5 + 0.5) * rows / 480.0 - 0.5)
    # The geometric-only alignment is ill-conditioned on these scenes: ANY change of arithmetic class moves its poses by 1e-5 .. 1e-4 instead of
    # ~1e-6 -- the oracle rebuilt under the model of the reference's own nvcc flags moves them by up to 4.8e-5 rad / 4.0e-5 m, its intensity sigma
    # (the scale of residuals the pose was NOT optimised for: first-order in the pose difference) by 2.3e-3 and up to 1 % of the fused map beyond
    # 1e-4 (tests/test_oracle_cuda_numerics.py::test_geometric_only_is_ill_conditioned, same sequences); the engine's fast gather kernels: up to
    # 8.3e-5 / 8.7e-5, sigma 2.3e-3, 2 % of the map (tools/experiments/geom_only_diag.py; exact kernels: 5e-6 / 1.2e-5).  The pose tolerance stays
    # 1e-4; the derived quantities get the room that pose difference needs.
    geom = cfg_kw.get("weighting") == O.GEOM_ONLY
    chi_fast = cfg_kw.get("termination") == O.CHI_SQUARED and cfg_kw.get("fast_numerics", 1) != 0
    run_case(ctx, rows, cols, K, n_lanes=3 if chi_fast else 2, n_frames=8 if chi_fast else 5, cfg_kw=cfg_kw, seq_kw=dict(trans_step=(0.003, 0.01), rot_step_deg=(0.1, 0.6)), use_graph=0,
             map_outliers=4e-2 if geom else 5e-3, sigma_tol=5e-3 if geom else 1e-3, chi_margin_tol=1e-4 if chi_fast else 0.0)


@pytest.mark.parametrize("warping,fast", [(O.WARP_FIRST, 0), (O.PYR_FIRST, 0), (O.WARP_FIRST, 1), (O.PYR_FIRST, 1)])
def test_engine_chi_squared_stops_lanes_independently(ctx, warping, fast):
    """the CHI_SQUARED stop is taken PER LANE: a batch of different streams equals the same streams run one lane at a time, bit for bit, and the early exit
    really fires (the records differ from an ALL_ITERS run).  Round 6: also PYR_FIRST (the test reads the stale level-0 warped maps, per lane and per mask)
    and the FAST class."""
    K = (131.25, 131.25, 79.5, 59.5)
    T, B = 5, 3
    seqs, depth, rgb = make_lanes(B, T, 120, 160, K, trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8))

    def run(lanes, term, graph=0):
        eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=len(lanes), K=K, use_graph=graph, record_capacity=T, termination=term, warping=warping, fast_numerics=fast))
        for k in range(T):
            eng.step(depth[k][lanes].contiguous(), rgb[k][lanes].contiguous())
        r = eng.records().copy(); eng.close()
        return r
    batch = run([0, 1, 2], O.CHI_SQUARED)
    for l in range(B):
        single = run([l], O.CHI_SQUARED)
        assert batch[:, l].tobytes() == single[:, 0].tobytes(), l
    allit = run([0, 1, 2], O.ALL_ITERS)
    assert np.abs(batch["R"] - allit["R"]).max() > 1e-7
    assert run([0, 1, 2], O.CHI_SQUARED, graph=1).tobytes() == batch.tobytes()     # the per-lane stop is data in flag arrays: the step stays one replayable hipGraph


def test_engine_records_do_not_depend_on_the_map_placement(ctx):
    """round 6: the engine skews its maps against each other in memory (alloc_img: the k-th map starts k x 4 352 B into its allocation, so that the streams of the
    dominant kernel do not walk the HBM channels in lock-step) -- placement must not change a single bit of what the engine computes: no skew, the default, a skew that
    keeps only 256-byte alignment together with a lane pad, all give the same records, fused maps and exported keyframes"""
    import os
    K = (131.25, 131.25, 79.5, 59.5)
    T, B = 6, 3
    seqs, depth, rgb = make_lanes(B, T, 120, 160, K, trans_step=(0.01, 0.02), rot_step_deg=(0.5, 1.0))

    def run(skew, pad):
        old = {k: os.environ.get(k) for k in ("RGBID_ENGINE_MAP_SKEW", "RGBID_ENGINE_LANE_PAD")}
        if skew is not None:
            os.environ["RGBID_ENGINE_MAP_SKEW"] = str(skew); os.environ["RGBID_ENGINE_LANE_PAD"] = str(pad)
        try:
            eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=B, K=K, use_graph=0, record_capacity=T, keyframe_capacity=3, visratio_odo=0.985, visratio_integr=0.97))
            for k in range(T):
                eng.step(depth[k], rgb[k])
            rec = eng.records().copy()
            maps = [np.concatenate([m.reshape(-1).view(np.uint8) for m in eng.keyframe_maps(l)]) for l in range(B)]
            nkf = [int(v) for v in eng.keyframe_counts()]
            eng.close()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        return rec, maps, nkf
    ref = run(0, 0)
    for skew, pad in ((None, None), (256, 768), (69888, 0)):
        got = run(skew, pad)
        assert got[0].tobytes() == ref[0].tobytes(), (skew, pad)
        assert all(np.array_equal(a, b) for a, b in zip(got[1], ref[1])) and got[2] == ref[2], (skew, pad)
    assert sum(ref[2]) > 0    # keyframes were exported on these sequences


@pytest.mark.parametrize("lanes,use_graph", [(1, 0), (1, 1), (3, 0), (16, 0)])
def test_engine_update_prologue_is_bit_identical(ctx, lanes, use_graph):
    """round 6, few-lane plan: up to 16 lanes the Gauss-Newton update of an iteration runs as the prologue of the next iteration's lattice launch, redundantly in
    every workgroup (k_lattice_after_update), instead of as a launch of its own: the same doubles in the same order -- records, fused keyframe maps and launch
    count (17 launches fewer per tracked frame of the shipped schedule) must say so"""
    import os
    K = (131.25, 131.25, 79.5, 59.5)
    T = 6
    seqs, depth, rgb = make_lanes(lanes, T, 120, 160, K, trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8))
    if lanes == 3:
        depth[3, 1] = 0          # lane 1 loses frame 3: its Gauss-Newton fails inside a prologue (NaN) while the other lanes go on

    def run(prologue):
        old = os.environ.get("RGBID_ENGINE_UPDATE_PROLOGUE_LANES")
        os.environ["RGBID_ENGINE_UPDATE_PROLOGUE_LANES"] = "16" if prologue else "0"
        try:
            eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=lanes, K=K, use_graph=use_graph, record_capacity=T))
            for k in range(T):
                eng.step(depth[k], rgb[k])
            rec = eng.records().copy()
            maps = [np.concatenate([m.reshape(-1).view(np.uint8) for m in eng.keyframe_maps(l)]) for l in range(lanes)]
            n_launch = eng.launches_per_step()
            eng.close()
        finally:
            if old is None:
                del os.environ["RGBID_ENGINE_UPDATE_PROLOGUE_LANES"]
            else:
                os.environ["RGBID_ENGINE_UPDATE_PROLOGUE_LANES"] = old
        return rec, maps, n_launch
    a, ma, la = run(False)
    b, mb, lb = run(True)
    assert a.tobytes() == b.tobytes()
    for x, y in zip(ma, mb):
        assert np.array_equal(x, y)
    assert la - lb == 17, (la, lb)
    assert np.count_nonzero(a["status"] & E.ST_TRACKED) >= lanes * (T - 2)


@pytest.mark.parametrize("use_graph", [0, 1])
def test_engine_custom_calibration(ctx, use_graph):
    """cfg.custom_registration = 1 (round 5): prepareImagesCustomCalibration (visodo.cpp:775-824) as predicated prep-stage launches of the engine -- undistort,
    depth-distortion correction, depth -> colour registration -- per lane: the registered inverse depth and the undistorted intensity are the oracle's bit
    for bit, the poses within the north-star tolerance"""
    import ctypes as C
    from rgbid import _lib
    from rgbid.device import IntrK, depth_dist
    rows, cols, T, B = 480, 640, 3, 2
    K = synth.TUM_K
    seqs, depth, rgb = make_lanes(B, T, rows, cols, K)
    rgbk = (0.02, -0.04, 0.0005, -0.0004, 0.01)
    dk = (571.0, 572.5, 316.0, 241.5, -0.015, 0.03, 0.0003, 0.0002, -0.008)
    dd = dict(c1=1.01, c0=-0.002, q0=(0.001, -0.002, 0.001, 0.0, 0.0005, -0.0004, 0.0, 0.0, 0.0), q1=(0.005, 0.01, 0.0, 0.0, -0.002, 0.001, 0.0, 0.0, 0.0))
    dRc = [0.99995, -0.008, 0.006, 0.00803, 0.99995, -0.005, -0.00596, 0.00505, 0.99997]
    t_dc = (0.0251, -0.0012, 0.0031)
    cfg = E.default_config(rows=rows, cols=cols, lanes=B, K=K, use_graph=use_graph, record_capacity=T, custom_registration=1)
    for i, v in enumerate(rgbk):
        cfg.rgb_dist[i] = v
    cfg.depth_intr = IntrK(*dk)
    cfg.depth_dist = depth_dist(**dd)
    _lib.check(_lib.lib().rgbid_engine_config_set_stereo(C.byref(cfg), (C.c_float * 9)(*dRc), (C.c_float * 3)(*t_dc)))
    eng = E.Engine(ctx, cfg)
    orcs = []
    for l in range(B):
        o = O.Tracker(O.default_config(rows=rows, cols=cols, fx=K[0], fy=K[1], cx=K[2], cy=K[3]))
        o.set_custom_calibration((K[0], K[1], K[2], K[3]) + rgbk, dk, O.depth_dist(**dd), dRc, t_dc)
        orcs.append(o)
    for k in range(T):
        eng.step(depth[k], rgb[k])
        for l in range(B):
            orcs[l].track(depth[k, l].cpu().numpy().view(np.uint16), rgb[k, l].cpu().numpy())
            iD, I = eng.current_maps(l)
            assert_bits(iD, orcs[l].cur_depthinv(), 0, f"registered iD, lane {l} frame {k}")
            assert_bits(I, orcs[l].cur_intensity(), 0, f"undistorted intensity, lane {l} frame {k}")
    rec = eng.records()
    for l in range(B):
        Rs, ts = orcs[l].poses()
        for k in range(1, T):
            assert rec[k, l]["status"] & E.ST_TRACKED
            assert rot_angle(Rs[k], rec[k, l]["R"]) < 1e-4 and np.linalg.norm(ts[k] - rec[k, l]["t"]) < 1e-4, (l, k)
        assert np.isfinite(orcs[l].cur_depthinv()).mean() > 0.6
        orcs[l].close()
    eng.close()


def test_engine_negative_fy_icl_nuim_calibration(ctx):
    """ICL-NUIM (Handa) sequences use fy < 0 (config_data/calibration_syntheticHanda.ini: 481.2, -480): the whole path must accept it."""
    K = (481.2 / 4, -480.0 / 4, (319.5 + 0.5) / 4 - 0.5, (239.5 + 0.5) / 4 - 0.5)
    run_case(ctx, 120, 160, K, n_lanes=2, n_frames=5, cfg_kw=dict(), seq_kw=dict(trans_step=(0.003, 0.01), rot_step_deg=(0.1, 0.6)), use_graph=0)


def test_engine_create_destroy_does_not_leak(ctx):
    """40 create / step / destroy cycles (graph and eager) leave the device memory where it was"""
    import ctypes as C
    from rgbid import _lib
    L = _lib.lib()
    K = (131.25, 131.25, 79.5, 59.5)
    seqs, depth, rgb = make_lanes(2, 2, 120, 160, K)

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert L.rgbid_mem_info(C.byref(f), C.byref(t)) == 0
        return f.value

    def cycle(g):
        eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=2, K=K, use_graph=g, record_capacity=2))
        eng.step(depth[0], rgb[0]); eng.step(depth[1], rgb[1]); eng.records()
        eng.close()

    cycle(0); cycle(1)
    torch.cuda.synchronize()
    before = free_bytes()
    for i in range(40):
        cycle(i % 2)
    torch.cuda.synchronize()
    assert abs(free_bytes() - before) < (8 << 20), (before, free_bytes())


def test_engine_preview_and_chi_square_options(ctx):
    """cfg.preview renders getImage (visodo.cpp:559-580: Phong shading of the keyframe vertex / normal maps with the keyframe colours,
    light at the integration keyframe's global position) and cfg.chi_square_stats runs the reference's unused full-resolution
    chi-square; neither may change a single pose record."""
    K = (synth.TUM_K[0] / 4, synth.TUM_K[1] / 4, (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
    T, B = 6, 2
    seqs, depth, rgb = make_lanes(B, T, 120, 160, K, trans_step=(0.01, 0.02), rot_step_deg=(0.5, 1.0))
    recs = []
    for extra in (dict(), dict(preview=1, chi_square_stats=1)):
        eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=B, K=K, use_graph=0, record_capacity=T, visratio_integr=0.97, **extra))
        for k in range(T):
            eng.step(depth[k], rgb[k])
        recs.append(eng.records().copy())
        if extra:
            assert (recs[-1]["status"][1:] & E.ST_INTEGR_KF).any()           # the light source has moved away from the origin at least once
            for l in range(B):
                img, colors = eng.preview(l)
                kd, kw, vm, nm, km = eng.keyframe_maps(l)
                # global position of the lane's current integration keyframe = pose of the last frame that re-keyed it
                kf = max(k for k in range(T) if recs[-1]["status"][k, l] & (E.ST_INTEGR_KF | E.ST_FIRST))
                light = recs[-1]["t"][kf, l].astype(np.float32)
                ref = O.generate_image_rgb(vm, nm, colors, light)
                diff = np.abs(img.astype(np.int32) - ref.astype(np.int32))
                assert diff.max() <= 1 and np.count_nonzero(diff) <= 1e-3 * diff.size, (l, diff.max(), np.count_nonzero(diff))
        eng.close()
    assert recs[0].tobytes() == recs[1].tobytes()


def _cmp_keyframe(a, b, K, rows, cols, q=None):
    """an exported keyframe of the engine (a) against the oracle tracker's (b) [+ the oracle's SEQ_KF constraint q]"""
    assert a["id"] == b["id"]
    assert rot_angle(a["R"], b["R"]) < 1e-4 and np.linalg.norm(a["t"] - b["t"]) < 1e-4
    assert rot_angle(a["R_rel"], b["R_rel"]) < 1e-4 and np.linalg.norm(a["t_rel"] - b["t_rel"]) < 1e-4
    if q is not None:
        assert (a["id"], a["end_id"]) == (q["ini"], q["end"])
        sc = np.sqrt(np.outer(np.diag(q["cov"]), np.diag(q["cov"]))) + 1e-30
        assert (np.abs(a["cov_rel"] - q["cov"]) / sc).max() < 1e-2
    assert np.array_equal(a["colors"], b["colors"])
    assert np.count_nonzero(a["overlap_mask"] != b["overlap_mask"]) <= 2e-3 * rows * cols
    da, db = a["depthinv"], b["depthinv"]
    assert np.count_nonzero(np.isnan(da) != np.isnan(db)) <= 2e-3 * db.size
    m = ~np.isnan(da) & ~np.isnan(db)
    rel = np.abs(da[m] - db[m]) / db[m]
    assert np.count_nonzero(rel > 1e-4) <= max(16, 5e-3 * rel.size) and np.median(rel) < 1e-5
    # normals amplify map differences by f/w: exact against the oracle's normal-map function of the EXPORTED inverse depth (cf. test_gpu_tracker_cpp)
    na = a["normals"]
    gx, gy = O.gradient(da)
    own = O.nmap_gradients(da, gx, gy, K).reshape(3, rows, cols)
    va, vo = ~np.isnan(na[0]), ~np.isnan(own[0])
    assert np.count_nonzero(vo != va) <= 4 and np.abs(own[:, vo & va] - na[:, vo & va]).max() < 2e-6
    vb = ~np.isnan(b["normals"][0])
    assert np.count_nonzero(va != vb) <= 6e-3 * vb.size


@pytest.mark.parametrize("use_graph", [0, 1])
def test_engine_keyframe_export(ctx, use_graph):
    """SURVEY 8 f-3 in the batched engine: at every integration-keyframe switch (visodo.cpp:1610-1652) the lane's outgoing keyframe --
    header (ids, global pose, pose to the next keyframe, SEQ_KF covariance) and the packed overlap mask / colours / fused inverse depth /
    normals -- lands in the lane's export ring on the device; read back through the pinned staging buffer it equals the oracle's export.
    One lane loses a frame (the failure path exports too, :2084-2085); lanes switch at different frames."""
    K = (synth.TUM_K[0] / 4, synth.TUM_K[1] / 4, (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
    rows, cols, n, B = 120, 160, 9, 3
    seqs, depth, rgb = make_lanes(B, n, rows, cols, K, trans_step=(0.01, 0.02), rot_step_deg=(0.5, 1.0))
    depth[4, 2] = 0                                                        # lane 2 loses frame 4
    cfg_kw = dict(visratio_odo=0.985, visratio_integr=0.97)
    eng = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=B, K=K, use_graph=use_graph, record_capacity=n, keyframe_capacity=8, **cfg_kw))
    for k in range(n):
        eng.step(depth[k], rgb[k])
    rec = eng.records()
    counts = eng.keyframe_counts()
    total = 0
    for l in range(B):
        trk = O.Tracker(O.default_config(rows=rows, cols=cols, fx=K[0], fy=K[1], cx=K[2], cy=K[3], **cfg_kw))
        d = depth[:, l].cpu().numpy().view(np.uint16); c = rgb[:, l].cpu().numpy()
        exported = 0
        for k in range(n):
            trk.track(d[k], c[k])
            before = exported
            exported = trk.num_keyframes()
            assert bool(int(rec[k, l]["status"]) & E.ST_KF_EXPORTED) == (exported > before), (l, k)   # same frames export
        assert counts[l] == trk.num_keyframes() >= 1
        kfc = [q for q in trk.constraints() if q["type"] == O.SEQ_KF]
        for i in range(counts[l]):
            a = eng.read_keyframe(l, i)
            assert (a["lane"], a["seq"]) == (l, i)
            _cmp_keyframe(a, trk.keyframe(i), K, rows, cols, kfc[i])
            hdr = eng.read_keyframe(l, i, images=False)
            assert hdr["id"] == a["id"] and np.array_equal(hdr["cov_rel"], a["cov_rel"])
        total += counts[l]
        with pytest.raises(Exception):
            eng.read_keyframe(l, counts[l])                                  # not exported yet
        trk.close()
    assert total >= B + 1
    # zero-copy view for device-side consumers
    import ctypes as C
    hdr, blk, nbytes = C.c_void_p(), C.c_void_p(), C.c_size_t()
    assert eng.L.rgbid_engine_keyframes_dev(eng._h, C.byref(hdr), C.byref(blk), C.byref(nbytes)) == 0
    assert hdr.value and blk.value and nbytes.value == 20 * rows * cols
    eng.close()


def test_engine_keyframe_ring_wraps_and_off_by_default(ctx):
    """The export ring keeps the last `keyframe_capacity` keyframes of a lane: older ones are refused, not returned stale; an engine
    created without a ring exports nothing and says so."""
    K = (synth.TUM_K[0] / 4, synth.TUM_K[1] / 4, (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
    rows, cols, n = 120, 160, 7
    seqs, depth, rgb = make_lanes(1, n, rows, cols, K, trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8))
    eng = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=1, K=K, record_capacity=n, keyframe_capacity=2, max_integrKF_count=1))
    trk = O.Tracker(O.default_config(rows=rows, cols=cols, fx=K[0], fy=K[1], cx=K[2], cy=K[3], max_integrKF_count=1))
    d = depth[:, 0].cpu().numpy().view(np.uint16); c = rgb[:, 0].cpu().numpy()
    for k in range(n):
        eng.step(depth[k], rgb[k]); trk.track(d[k], c[k])
    cnt = int(eng.keyframe_counts()[0])
    assert cnt == trk.num_keyframes() == n - 1                                # max_integrKF_count = 1: a switch on every tracked frame
    for i in (cnt - 1, cnt - 2):
        _cmp_keyframe(eng.read_keyframe(0, i), trk.keyframe(i), K, rows, cols)
    with pytest.raises(Exception):
        eng.read_keyframe(0, cnt - 3)                                         # overwritten
    eng.reset()
    assert int(eng.keyframe_counts()[0]) == 0
    eng.close(); trk.close()
    off = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=1, K=K, record_capacity=n))
    for k in range(3):
        off.step(depth[k], rgb[k])
    assert int(off.keyframe_counts()[0]) == 0
    with pytest.raises(Exception):
        off.read_keyframe(0, 0)
    off.close()


@pytest.mark.parametrize("use_graph", [0, 1])
def test_engine_reset_single_lane(ctx, use_graph):
    """rgbid_engine_reset_lane: one lane starts a NEW sequence in the middle of a run (its stream ended) while its neighbours keep tracking;
    the restarted lane must behave like a fresh oracle tracker on the new sequence, the others like uninterrupted ones."""
    K = (synth.TUM_K[0] / 4, synth.TUM_K[1] / 4, (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
    rows, cols, n, B, cut = 120, 160, 9, 3, 4
    seqs, depth, rgb = make_lanes(B, n, rows, cols, K, trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8))
    new = synth.make_sequence(n - cut, seed=synth.SEED + 999, K=K, rows=rows, cols=cols, device="cuda", trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8))
    depth = depth.clone(); rgb = rgb.clone()
    depth[cut:, 1] = new["depth"].to(torch.int16); rgb[cut:, 1] = new["rgb"]          # lane 1 switches to another sequence at frame `cut`
    eng = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=B, K=K, use_graph=use_graph, record_capacity=n, keyframe_capacity=2))
    for k in range(n):
        if k == cut:
            eng.reset_lane(1)
        eng.step(depth[k], rgb[k])
    rec = eng.records()
    assert int(rec[cut, 1]["status"]) & E.ST_FIRST and not int(rec[cut, 0]["status"]) & E.ST_FIRST
    for l in range(B):
        spans = [(0, n)] if l != 1 else [(0, cut), (cut, n)]
        for a, b in spans:
            trk = O.Tracker(O.default_config(rows=rows, cols=cols, fx=K[0], fy=K[1], cx=K[2], cy=K[3]))
            d = depth[a:b, l].cpu().numpy().view(np.uint16); c = rgb[a:b, l].cpu().numpy()
            for k in range(b - a):
                trk.track(d[k], c[k])
            Rs, ts = trk.poses()
            for k in range(1, b - a):
                assert rot_angle(Rs[k], rec[a + k, l]["R"]) < 1e-4 and np.linalg.norm(ts[k] - rec[a + k, l]["t"]) < 1e-4, (l, a, k)
            if b == n:
                kd = eng.keyframe_maps(l)[0]; od = trk.kf_depthinv()
                m = ~np.isnan(kd) & ~np.isnan(od)
                assert np.count_nonzero(np.isnan(kd) != np.isnan(od)) <= 2e-3 * od.size
                assert np.count_nonzero(np.abs(kd[m] - od[m]) / od[m] > 1e-4) <= max(16, 5e-3 * m.sum())
            trk.close()
    eng.close()


@pytest.mark.parametrize("use_graph", [0, 1])
def test_engine_inactive_lanes_sit_steps_out(ctx, use_graph):
    """rgbid_engine_set_active: streams of different rates share an engine.  Lane 1 gets no frame on steps 0, 4 and 5 (whatever sits in its
    input slot is ignored): its records on those steps have status 0 and repeat the last pose, and the frames it does get are tracked exactly
    like an oracle tracker fed only those; lane 0 is fed every step and is unaffected."""
    K = (synth.TUM_K[0] / 4, synth.TUM_K[1] / 4, (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
    rows, cols, n, B = 120, 160, 9, 2
    seqs, depth, rgb = make_lanes(B, n, rows, cols, K, trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8))
    idle = {0, 4, 5}
    eng = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=B, K=K, use_graph=use_graph, record_capacity=n))
    fed = []                                                   # frame index of lane 1's own sequence consumed at each step (None = idle)
    nxt = 0
    garbage_d = torch.full_like(depth[0, 1], 1234); garbage_c = torch.zeros_like(rgb[0, 1])
    for k in range(n):
        d = depth[k].clone(); c = rgb[k].clone()
        if k in idle:
            eng.set_active([1, 0]); d[1] = garbage_d; c[1] = garbage_c; fed.append(None)
        else:
            eng.set_active(None); d[1] = depth[nxt, 1]; c[1] = rgb[nxt, 1]; fed.append(nxt); nxt += 1
        eng.step(d, c)
    rec = eng.records()
    trk = [O.Tracker(O.default_config(rows=rows, cols=cols, fx=K[0], fy=K[1], cx=K[2], cy=K[3])) for _ in range(B)]
    for k in range(n):
        trk[0].track(depth[k, 0].cpu().numpy().view(np.uint16), rgb[k, 0].cpu().numpy())
    R0, t0 = trk[0].poses()
    for k in range(1, n):
        assert rot_angle(R0[k], rec[k, 0]["R"]) < 1e-4 and np.linalg.norm(t0[k] - rec[k, 0]["t"]) < 1e-4, k
    for j in range(nxt):
        trk[1].track(depth[j, 1].cpu().numpy().view(np.uint16), rgb[j, 1].cpu().numpy())
    R1, t1 = trk[1].poses()
    last = None
    for k in range(n):
        st = int(rec[k, 1]["status"])
        if fed[k] is None:
            assert st == 0, (k, st)
            if last is not None:
                assert np.array_equal(rec[k, 1]["R"], rec[last, 1]["R"]) and np.array_equal(rec[k, 1]["t"], rec[last, 1]["t"])
            continue
        j = fed[k]
        assert bool(st & E.ST_FIRST) == (j == 0), (k, st)
        if j:
            assert st & E.ST_TRACKED
            assert rot_angle(R1[j], rec[k, 1]["R"]) < 1e-4 and np.linalg.norm(t1[j] - rec[k, 1]["t"]) < 1e-4, (k, j)
        last = k
    for t in trk:
        t.close()
    eng.close()


def test_delta_t_per_frame_reaches_captured_graphs(ctx):
    """rgbid_engine_set_delta_t (the per-frame computeInterframeTime of the compat tracker, visodo.cpp:1902-1965): the kernels read the value through a device
    pointer, so a step replayed as a captured hipGraph follows it.  Records with a different inter-frame time every frame are bit-identical between eager
    launches and graph replay, and differ from the run with the constant default (the constant-velocity prediction really uses it)."""
    K = (131.25, 131.25, 79.5, 59.5)
    n, B = 7, 2
    seqs = [synth.make_sequence(n, seed=synth.SEED + 5 * l, K=K, rows=120, cols=160, device="cuda", trans_step=(0.01, 0.02), rot_step_deg=(0.5, 1.0)) for l in range(B)]
    depth = torch.stack([s["depth"] for s in seqs], 1).to(torch.int16).contiguous(); rgb = torch.stack([s["rgb"] for s in seqs], 1).contiguous()
    dts = [0.0333, 0.05, 0.02, 0.0333, 0.1, 0.04, 0.03]
    out = []
    for use_graph, vary in ((0, True), (1, True), (1, False)):
        eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=B, K=K, use_graph=use_graph, record_capacity=n))
        for k in range(n):
            if vary:
                eng.set_delta_t(dts[k])
            eng.step(depth[k], rgb[k])
        out.append(eng.records().copy())
        eng.close()
    assert out[0].tobytes() == out[1].tobytes()
    assert not np.array_equal(out[1]["R"], out[2]["R"])


@pytest.mark.parametrize("use_graph", [0, 1])
def test_deferred_keyframe_maps_change_no_output(ctx, use_graph):
    """rgbid_engine_config.defer_keyframe_maps: the vertex / normal maps of the fused keyframe computed only where they are consumed (keyframe export, accessor)
    instead of after every fusion step (visodo.cpp:1758-1762).  Same fused map in, same kernel: pose records, every exported keyframe (mask, colours, inverse
    depth, NORMALS) and the accessor's maps are bit-identical to the per-frame schedule, on a sequence that switches keyframes."""
    K = (131.25, 131.25, 79.5, 59.5)
    n, B = 10, 3
    seqs = [synth.make_sequence(n, seed=synth.SEED + 7 * l, K=K, rows=120, cols=160, device="cuda", trans_step=(0.01, 0.02), rot_step_deg=(0.5, 1.0)) for l in range(B)]
    depth = torch.stack([s["depth"] for s in seqs], 1).to(torch.int16).contiguous(); rgb = torch.stack([s["rgb"] for s in seqs], 1).contiguous()
    out = []
    for defer in (0, 1):
        eng = E.Engine(ctx, E.default_config(rows=120, cols=160, lanes=B, K=K, use_graph=use_graph, record_capacity=n, keyframe_capacity=8, visratio_odo=0.985, visratio_integr=0.97,
                                             defer_keyframe_maps=defer))
        for k in range(n):
            eng.step(depth[k], rgb[k])
        rec = eng.records().copy()
        counts = eng.keyframe_counts().copy()
        kfs = []
        for l in range(B):
            for q in range(max(0, int(counts[l]) - 8), int(counts[l])):     # what the 8-slot ring still holds
                kf = eng.read_keyframe(l, q)
                nrm = kf["normals"].copy(); nrm[:, np.isnan(nrm[0])] = np.nan
                kfs.append((kf["id"], kf["end_id"], kf["R"].tobytes(), kf["t"].tobytes(), kf["R_rel"].tobytes(), kf["t_rel"].tobytes(), kf["cov_rel"].tobytes(),
                            kf["overlap_mask"].tobytes(), kf["colors"].tobytes(), kf["depthinv"].tobytes(), nrm.tobytes()))
        maps = [eng.keyframe_maps(l) for l in range(B)]
        out.append((rec, counts, kfs, maps, eng.launches_per_step()))
        eng.close()
    a, b = out
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and int(a[1].sum()) >= 3
    assert a[2] == b[2]
    for ma, mb in zip(a[3], b[3]):
        for x, y in zip(ma, mb):
            assert np.array_equal(x, y, equal_nan=True)
    assert b[4] == a[4]      # the export-side launch replaces the end-of-step one


def test_engine_one_wave_scalar_kernels(ctx):
    """More lanes than compute units: the per-lane reduce-and-solve kernels (k_solve_update, k_frame_finish) run as ONE wave per lane instead of four
    (engine_device.h reduce_partials<64>: four slices per thread, the same doubles in the same order as the 256-thread form).  320 lanes carrying 4 distinct
    streams: duplicates are bit-identical, and every stream agrees with a 4-lane engine (256-thread workgroups; another launch plan of the normal equations,
    so another rounding of the fp32 partial sums) far inside the pose bar, with the same keyframe decisions."""
    rows, cols, K = 120, 160, (131.25, 131.25, 79.875, 59.875)
    n_frames, n_streams, B = 6, 4, 320
    seqs, depth, rgb = make_lanes(n_streams, n_frames, rows, cols, K)
    idx = torch.arange(B, device="cuda") % n_streams
    few = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=n_streams, K=K, record_capacity=n_frames))
    many = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=B, K=K, record_capacity=n_frames))
    for k in range(n_frames):
        few.step(depth[k], rgb[k])
        many.step(depth[k][idx].contiguous(), rgb[k][idx].contiguous())
    a, b = few.records(), many.records()
    for l in range(n_streams, B):
        assert a.dtype == b.dtype and b[:, l].tobytes() == b[:, l % n_streams].tobytes(), l     # duplicates: every byte
    for l in range(n_streams):
        assert np.array_equal(a["status"][:, l], b["status"][:, l])
        for k in range(n_frames):
            assert rot_angle(a["R"][k, l], b["R"][k, l]) < 2e-5 and np.abs(a["t"][k, l] - b["t"][k, l]).max() < 2e-5, (k, l)
    few.close(); many.close()
