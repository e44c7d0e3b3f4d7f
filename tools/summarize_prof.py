"""Condense rocprofv3 CSV output (run on the GPU box) into small per-kernel summaries that fit in profiles/.

  python tools/summarize_prof.py <dir>   -> <dir>/kernels_rgbid.csv (+ pmc_rgbid.csv when counter CSVs exist)

Only the library's own kernels are kept (the synthetic-scene generator's torch kernels are dropped); launches whose lanes
were all predicated off finish in a few microseconds and are reported separately (`calls_active` counts launches lasting
more than 10% of the kernel's median-of-the-top-half duration)."""
import collections, csv, glob, os, re, sys
import numpy as np

d = sys.argv[1]
keep = lambda n: ("rgbid" in n) or ("anonymous namespace" in n) or ("rocclr" in n)

# rocprofv3's VGPR_Count column is HALF the allocation of a wave64 kernel on gfx950 (60 for the fused Gauss-Newton kernel's 118): the
# register figures come from the compiler's own resource remarks (tools/kernel_resources.py -> profiles/*kernel_resources.csv) when present
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
norm = lambda n: re.sub(r"\(anonymous namespace\)::", "", n.split("(")[0].replace("void ", "")).strip()
RES = {}
for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*kernel_resources.csv"))):
    for r in csv.DictReader(open(f)):
        RES[r["name"]] = r

for tr in glob.glob(os.path.join(d, "*kernel_trace.csv")):
    agg = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(tr)):
        n = r["Kernel_Name"]
        if keep(n):
            agg[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[n] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size_X"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    out = tr.replace("kernel_trace.csv", "kernels_rgbid.csv")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "CallsActive", "AverageActiveNs", "MinNs", "MaxNs", "VGPRs", "SGPRs", "LDS", "WG_X", "LastGrid", "VGPRsFrom"])
        for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            v = np.array(v, dtype=np.float64)
            top = np.sort(v)[len(v) // 2:]
            thr = 0.1 * np.median(top)
            act = v[v > thr]
            m = meta[n]
            res = RES.get(norm(n))
            vg, sg, src = (res["vgprs"], res["sgprs"], "compiler remarks") if res else (m[0], m[1], "rocprofv3 column (half the wave64 allocation on gfx950)")
            w.writerow([n, len(v), int(v.sum()), v.mean(), len(act), act.mean() if len(act) else 0.0, int(v.min()), int(v.max()), vg, sg, m[2], m[3], "x".join(m[4:]), src])
    if "pmc" in os.path.basename(tr):
        os.remove(tr)

for cc in glob.glob(os.path.join(d, "*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    bygrid = collections.defaultdict(list)     # (kernel, grid, counter) -> values: a kernel that runs at several pyramid levels, one row per level
    for r in csv.DictReader(open(cc)):
        n = r["Kernel_Name"]
        if keep(n):
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            grid = "x".join(str(r.get(k, "")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else r.get("Grid_Size", "")
            bygrid[(n, grid, r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(cc.replace("counter_collection.csv", "pmc_bygrid_rgbid.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Grid", "Counter", "Dispatches", "Mean", "Max"])
        for (n, g, c), v in sorted(bygrid.items()):
            w.writerow([n, g, c, len(v), float(np.mean(v)), float(np.max(v))])
    out = cc.replace("counter_collection.csv", "pmc_rgbid.csv")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Counter", "Dispatches", "Mean", "MeanOfTopHalf", "Max"])
        for n, cs in sorted(agg.items()):
            for c, v in cs.items():
                v = np.array(v)
                top = np.sort(v)[len(v) // 2:]
                w.writerow([n, c, len(v), v.mean(), top.mean(), v.max()])
    os.remove(cc)
for big in glob.glob(os.path.join(d, "*kernel_trace.csv")):
    if os.path.getsize(big) > 4 << 20:
        os.remove(big)
