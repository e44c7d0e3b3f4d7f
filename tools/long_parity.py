import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/rgbid-slam_amd")
import numpy as np, torch
from rgbid import device, synth
from tests import test_gpu_engine as T
ctx = device.Context(0)
K = (synth.TUM_K[0] / 4, synth.TUM_K[1] / 4, (synth.TUM_K[2] + 0.5) / 4 - 0.5, (synth.TUM_K[3] + 0.5) / 4 - 0.5)
for kw in (dict(), dict(visratio_odo=0.97, visratio_integr=0.93)):
    try:
        print(kw, T.run_case(ctx, 120, 160, K, n_lanes=3, n_frames=40, cfg_kw=kw, seq_kw=dict(trans_step=(0.004, 0.012), rot_step_deg=(0.2, 0.8)), use_graph=0))
    except AssertionError as e:
        import traceback; traceback.print_exc()
