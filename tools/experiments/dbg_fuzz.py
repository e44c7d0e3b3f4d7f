import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/rgbid-slam_amd')
from tests import test_gpu_fuzz as F
from rgbid import device
from oracle import oracle as O
ctx=device.Context(0)
f32=np.float32
def exact_xy(x,y,w,R,t):
    zd=f32(1)/w; X=f32(x)*zd; Y=f32(y)*zd
    r=lambda a,b,c,tt: f32(f32(f32(f32(a*X)+f32(b*Y))+f32(c*zd))+tt)
    X0=r(R[0],R[1],R[2],t[0]); X1=r(R[3],R[4],R[5],t[1]); X2=r(R[6],R[7],R[8],t[2])
    wc=f32(1)/X2
    return f32(f32(X0*wc)+f32(.5)), f32(f32(X1*wc)+f32(.5)), X2
for seed in [int(a) for a in sys.argv[1:]] or [8]:
    rows, cols, K, grid, grid_dom, src, inten, Rp, tp = F._fast_case(seed)
    new = lambda: torch.full((rows, cols), float("nan"), device="cuda")
    W1, I1 = new(), new()
    ctx.warpPair(F.dev(src), F.dev(inten), F.dev(grid), W1, I1, Rp, tp, fast=True)
    gW1=W1.cpu().numpy()
    with np.errstate(all="ignore"):
        oW1 = O.warp_invdepth(src, grid_dom, Rp, tp)
    both=~np.isnan(oW1)&~np.isnan(gW1)
    with np.errstate(all="ignore"):
        bad=both&(np.abs(gW1-oW1)>1e-3*np.abs(oW1))
    print('seed',seed,rows,cols,'bad',bad.sum(),'nanmis',(np.isnan(oW1)!=np.isnan(gW1)).sum())
    R=np.asarray(Rp,f32).reshape(-1); t=np.asarray(tp,f32)
    print('R',R,'t',t)
    for y,x in zip(*np.nonzero(bad)):
        w=grid[y,x]
        xs,ys,X2=exact_xy(x,y,w,R,t)
        q=[np.float64(R[3*i])*x+np.float64(R[3*i+1])*y+np.float64(R[3*i+2]) for i in range(3)]
        Y=[q[i]+np.float64(w)*np.float64(t[i]) for i in range(3)]
        print('px',y,x,'w',w,'got',gW1[y,x],'ora',oW1[y,x],'exact xs,ys',xs,ys,'X2',X2,'true xs,ys',Y[0]/Y[2]+.5,Y[1]/Y[2]+.5,'Y2',Y[2],'q2',q[2])
        ix,iy=int(np.floor(xs)),int(np.floor(ys))
        print('   src at oracle px',src[iy,ix],' neighbours',src[max(iy-1,0):iy+2,max(ix-1,0):ix+2])
