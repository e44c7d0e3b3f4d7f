"""probe: does stdout survive process exit after the C-ABI RCCL communicator was used? (python tools/experiments/rccl_exit_probe.py MODE > out.txt)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rgbid-slam_amd"))
import torch
from rgbid import device, dist as D
mode = sys.argv[1]
ctx = device.Context(0)
comm = D.Comm(ctx, 1, 0)
x = torch.zeros(392 * 4, dtype=torch.uint8, device="cuda")
y = comm.gather(x, 4); comm.barrier()
print("RESULT-LINE", mode, int(y.sum().item()))
if mode == "close":
    comm.close(); ctx.close()
elif mode == "noclose":
    pass
elif mode == "torchdist":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    dist.barrier()
    comm.close(); ctx.close()
    dist.destroy_process_group()
sys.stderr.write("reached end of script\n")
