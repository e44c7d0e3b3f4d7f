"""profiles/<tag>_pmc_traffic.json from the condensed PMC summaries (tools/summarize_prof.py output of the two --pmc passes).

  python tools/make_pmc_traffic.py <dir with pmc_fetch_pmc_rgbid.csv / pmc_write_pmc_rgbid.csv> <out.json> <lanes> [rows cols [fused]]

HBM bytes per launch of the dominant kernel (level-0 normal equations) = 2 x FETCH_SIZE[KB] x 1024 + WRITE_SIZE[KB] x 1024:
gfx950 counts the 128-byte requests of wide coalesced reads as 64 B, hence the x2 on the read side (MI355X_MICROARCH.md)."""
import csv, glob, json, os, sys

d, out, lanes = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows, cols = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (480, 640)
fused = int(sys.argv[6]) if len(sys.argv) > 6 else 1
# level-0 Gauss-Newton evaluation: the fused kernel (warp + residual + normal equations, template tag 2 = fast numerics) or the unfused one
# (last template argument 1 = the branch-free variant the Gauss-Newton iterations of the shipped configuration run)
KEY = "k_build_system<rgbid::ByLane<rgbid::SysParams>, true, 0, 2, 1>" if fused else "k_build_system<rgbid::ByLane<rgbid::SysParams>, true, 0, 0, 0>"


def counter(pattern, name):
    for f in glob.glob(os.path.join(d, pattern)):
        for r in csv.DictReader(open(f)):
            if KEY in r["Name"] and r["Counter"] == name:
                return float(r["Mean"]), int(r["Dispatches"])
    raise SystemExit(f"{name} of {KEY} not found in {d}/{pattern}")


fetch, n = counter("*pmc_fetch*pmc_rgbid.csv", "FETCH_SIZE")
write, _ = counter("*pmc_write*pmc_rgbid.csv", "WRITE_SIZE")
traffic = 2.0 * fetch * 1024.0 + write * 1024.0
alg = 32.0 * rows * cols * lanes
json.dump({
    "command": f"python bench.py --lanes {lanes} (tools/profile_bench.sh: separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes)",
    "kernel": "rgbid::" + KEY, "lanes": lanes, "rows": rows, "cols": cols, "fused_gn": bool(fused),
    "dispatches": n, "FETCH_SIZE_KB_raw_mean_per_launch": fetch, "WRITE_SIZE_KB_raw_mean_per_launch": write,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section; calibrated here on the 4-byte gathers too: the stand-alone warp pair reads 2 x 5.97 = 11.9 B/px raw against 12 algorithmic once its tiles share an L2); WRITE_SIZE uncorrected; KB = 1024 B",
    "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": traffic / alg}, open(out, "w"), indent=1)
print(open(out).read())
