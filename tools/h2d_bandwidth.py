"""Sustained pinned-host -> device copy rate of the box (what bounds bench.py's `pcie_inclusive` leg): python tools/h2d_bandwidth.py"""
import time, torch
n = 3 * 1024**3
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(2): d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"pinned H2D: {n / dt / 1e9:.1f} GB/s ({n / 1e9:.2f} GB in {dt * 1e3:.1f} ms)")
