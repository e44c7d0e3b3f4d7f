import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/rgbid-slam_amd')
from tests import test_gpu_fuzz as F
from rgbid import device
from oracle import oracle as O
ctx=device.Context(0)
for seed in [6]:
    rows, cols, K, grid, grid_dom, src, inten, Rp, tp = F._fast_case(seed)
    new = lambda: torch.full((rows, cols), float("nan"), device="cuda")
    W1, I1 = new(), new()
    ctx.warpPair(F.dev(src), F.dev(inten), F.dev(grid), W1, I1, Rp, tp, fast=True)
    gW1=W1.cpu().numpy(); gI1=I1.cpu().numpy()
    w1_dom = np.where((gW1 >= F.W_LO) & (gW1 <= F.W_HI), gW1, np.float32(np.nan)).astype(np.float32)
    oI1 = O.warp_intensity(inten, w1_dom, Rp, tp, O.INTERP_TEX8)
    ok=~np.isnan(oI1)
    d=np.abs(np.where(ok,gI1-oI1,0))
    ys,xs=np.nonzero(d>2)
    print(rows,cols,len(ys))
    for y,x in list(zip(ys,xs))[:8]:
        print(y,x,gI1[y,x],oI1[y,x],gW1[y,x])
        # oracle coords
    print(np.isnan(inten).sum())
