#!/usr/bin/env python
"""Register / LDS / scratch allocation of every kernel of csrc/*.hip, from hipcc's own resource-usage remarks (-Rpass-analysis=kernel-resource-usage),
demangled.  rocprofv3's VGPR_Count column reports half the allocation of a wave64 kernel on gfx950 (e.g. 60 for the fused Gauss-Newton kernel's
118), so the per-round profile summaries take the register figures from here.  Runs without a GPU (hipcc cross-compiles).

    python tools/kernel_resources.py [out.csv]        (default: profiles/kernel_resources.csv)
"""
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rgbid-slam_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -Wno-unused-value".split()


def resources(src):
    extra = ["-fno-slp-vectorize"] if src in ("kernels_system.hip", "kernels_bilateral.hip", "kernels_warp.hip") else []   # csrc/Makefile
    p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=CSRC)
    rows, cur = [], None
    for line in p.stdout.splitlines():
        m = re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*(Function Name|Name): (\S+)", line) or re.search(r"Function Name: (\S+)()", line)
        if "Name:" in line and "remark" in line:
            name = line.split("Name:")[1].split("[")[0].strip()
            cur = {"mangled": name, "file": src}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("sgprs", r"TotalSGPRs: (\d+)"), ("vgprs", r"\bVGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("scratch_bytes_per_lane", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("waves_per_simd", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds_bytes_per_block", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return rows


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "kernel_resources.csv")
    rows = []
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        rows += resources(src)
    names = subprocess.run(["c++filt"], input="\n".join(r["mangled"] for r in rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["name"] = re.sub(r"\(anonymous namespace\)::", "", n.split("(")[0].replace("void ", ""))
    cols = ["name", "file", "vgprs", "agprs", "sgprs", "scratch_bytes_per_lane", "waves_per_simd", "lds_bytes_per_block"]
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(cols)
        for r in sorted(rows, key=lambda r: (r["file"], r["name"])):
            w.writerow([r.get(c, "") for c in cols])
    print(f"{len(rows)} kernels -> {out}")


if __name__ == "__main__":
    main()
