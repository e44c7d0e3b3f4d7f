"""Issue-slot accounting per kernel from the SQ / GRBM counter pass of tools/pmc_sq.sh (gpurun_out/pmc_sq/ -> markdown table).

VALU busy = SQ_ACTIVE_INST_VALU (quad-cycles, summed over all SIMDs) x 4 / 1024 SIMDs / kernel cycles, kernel cycles =
GRBM_GUI_ACTIVE / 8 XCDs; clock = kernel cycles / kernel duration of the same pass (counters slow the launch a little, the ratio holds).
Only full-batch launches are used (mean of the top half of the dispatches)."""
import collections, csv, sys

d = sys.argv[1]
ctr = collections.defaultdict(dict)
for r in csv.DictReader(open(f"{d}/sq_pmc_rgbid.csv")):
    ctr[r["Name"]][r["Counter"]] = float(r["MeanOfTopHalf"])
dur, uniform = {}, {}
for r in csv.DictReader(open(f"{d}/sq_kernels_rgbid.csv")):
    dur[r["Name"]] = float(r["MaxNs"])
    uniform[r["Name"]] = float(r["MaxNs"]) < 1.6 * float(r["AverageActiveNs"])   # (nearly) all active launches have the size of the largest
    n = int(r["CallsActive"])
    if not uniform[r["Name"]] and n > 2:
        # ONE slow launch among equal ones (the launch that first touches freshly allocated memory: tens of ms): take the mean of the others
        rest = (float(r["AverageActiveNs"]) * n - float(r["MaxNs"])) / (n - 1)
        if float(r["MinNs"]) > 0.7 * rest:
            dur[r["Name"]] = rest
            uniform[r["Name"]] = True
rows = []
for k, c in ctr.items():
    if "rgbid::" not in k or c.get("SQ_WAVES", 0) < 1000 or not uniform.get(k, False):
        continue   # kernels launched at several sizes (pyramid levels, keyframe-switch subsets) cannot be paired with one duration
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    name = k.split("(")[0].replace("void ", "")[:70]
    if not name:
        continue
    rows.append((dur[k], name, c["SQ_WAVES"], c["SQ_INSTS_VALU"] / c["SQ_WAVES"], cyc / dur[k],
                 c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
print("| kernel (largest launch) | µs (counter pass) | waves | VALU instr / wave | clock GHz | VALU busy | waves parked (WAIT_ANY) | issue-stalled (WAIT_INST_ANY) |")
print("|---|---|---|---|---|---|---|---|")
for r in sorted(rows, reverse=True):
    if r[5] > 1.6:
        continue   # duration and counters of different launch subsets (keyframe-switch kernels)
    # (values a little above 100 %: SQ_ACTIVE_INST_VALU counts 4 cycles per instruction, full-rate fp32 multiplies / adds / FMAs issue in 2)
    print(f"| `{r[1]}` | {r[0] / 1e3:.0f} | {r[2]:.0f} | {r[3]:.0f} | {r[4]:.2f} | {100 * r[5]:.0f} % | {100 * r[6]:.0f} % | {100 * r[7]:.0f} % |")
