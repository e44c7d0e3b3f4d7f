"""Idle time between consecutive kernels of the engine's stream, from a rocprofv3 --kernel-trace CSV (run on the GPU box).

  python tools/gap_analysis.py <dir with *kernel_trace.csv>

For every library kernel: the average gap between the end of the previous dispatch (any kernel on the device) and its own start, over the
dispatches of the last timed steps -- i.e. what a step spends NOT inside kernels, attributed to the kernel that was waited for."""
import collections, csv, glob, os, sys

d = sys.argv[1]
for tr in glob.glob(os.path.join(d, "*kernel_trace.csv")):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(tr))]
    rows.sort()
    # the timed region = the last stretch of dispatches made of library kernels only (the input generator's torch kernels come first)
    last_torch = max((i for i, r in enumerate(rows) if "at::native" in r[2]), default=-1)
    seg = rows[last_torch + 1:]
    if len(seg) < 100:
        seg = rows
    gaps = collections.defaultdict(list)
    busy = 0
    for prev, cur in zip(seg, seg[1:]):
        gaps[cur[2]].append(cur[0] - prev[1])
        busy += cur[1] - cur[0]
    span = seg[-1][1] - seg[0][0]
    tot_gap = sum(sum(v) for v in gaps.values())
    print(f"{os.path.basename(tr)}: {len(seg)} dispatches, span {span/1e6:.2f} ms, in kernels {busy/1e6:.2f} ms, gaps {tot_gap/1e6:.2f} ms ({tot_gap/span*100:.1f} %)")
    for n, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:16]:
        print(f"  {sum(v)/1e3:9.1f} us total  {len(v):5d} x {sum(v)/len(v)/1e3:7.2f} us  before {n.replace('void rgbid::', '')[:80]}")
