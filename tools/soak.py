import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/rgbid-slam_amd")
import numpy as np, torch
from oracle import oracle as O
from rgbid import synth, engine as E, device
def rot_angle(Ra, Rb): return float(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))
full = len(sys.argv) > 1 and sys.argv[1] == "full"      # `python tools/soak.py full`: 2 lanes x 60 frames at 640x480 instead of 4 x 300 at 160x120
if full:
    K, n, B, rows, cols = synth.TUM_K, (int(sys.argv[2]) if len(sys.argv) > 2 else 60), 2, 480, 640   # `python tools/soak.py full 200`: longer
else:
    K = (synth.TUM_K[0] / 4, synth.TUM_K[1] / 4, (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
    n, B, rows, cols = 300, 4, 120, 160
seqs = [synth.make_sequence(n, seed=synth.SEED + 31 * l, K=K, rows=rows, cols=cols, device="cuda", trans_step=(0.003, 0.012), rot_step_deg=(0.1, 0.8)) for l in range(B)]
depth = torch.stack([s["depth"] for s in seqs], 1).to(torch.int16).contiguous(); rgb = torch.stack([s["rgb"] for s in seqs], 1).contiguous()
ctx = device.Context(0)
eng = E.Engine(ctx, E.default_config(rows=rows, cols=cols, lanes=B, K=K, record_capacity=n, keyframe_capacity=4))
for k in range(n): eng.step(depth[k], rgb[k])
rec = eng.records()
for l in range(B):
    trk = O.Tracker(O.default_config(rows=rows, cols=cols, fx=K[0], fy=K[1], cx=K[2], cy=K[3]))
    d = depth[:, l].cpu().numpy().view(np.uint16); c = rgb[:, l].cpu().numpy()
    worst = (0, 0, -1); div = []
    for k in range(n):
        st = int(rec[k, l]["status"])
        if k:
            trk.force_kf_decisions(bool(st & E.ST_ODO_KF), bool(st & E.ST_INTEGR_KF))   # every frame is compared: a decision on the threshold is imposed, and reported
        trk.track(d[k], c[k])
        if k:
            info = trk.last_info()
            if bool(st & E.ST_ODO_KF) != bool(info.odo_kf_natural) or bool(st & E.ST_INTEGR_KF) != bool(info.integr_kf_natural):
                div.append((k, round(info.visratio_odo, 5), round(info.visratio_integr, 5)))
    Rs, ts = trk.poses()
    m = len(Rs)
    for k in range(1, m):
        er, et = rot_angle(Rs[k], rec[k, l]["R"]), float(np.linalg.norm(ts[k] - rec[k, l]["t"]))
        if max(er, et) > max(worst[0], worst[1]): worst = (er, et, k)
    gt_R, gt_t = seqs[l]["R_wc"].numpy(), seqs[l]["t_wc"].numpy()
    print("lane", l, "frames compared", m, "worst |dR| %.2e rad |dt| %.2e m at frame %d" % worst, "decisions imposed (ratio on its threshold):", div,
          "| drift vs ground truth at end: %.2e rad %.2e m" % (rot_angle(gt_R[m - 1], rec[m - 1, l]["R"]), np.linalg.norm(gt_t[m - 1] - rec[m - 1, l]["t"])),
          "| keyframes exported", int(eng.keyframe_counts()[l]), "oracle", trk.num_keyframes())
