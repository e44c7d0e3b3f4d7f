"""profiles/<round>_pmc_traffic_all.json: counted HBM traffic against algorithmic bytes for every kernel that takes >= 2 % of a step.

  python tools/make_pmc_traffic_all.py <dir with *pmc_fetch*pmc_bygrid_rgbid.csv, *pmc_write*..., trace_kernels_rgbid.csv> <out.json> <lanes> [rows cols nsamples]

Counted bytes of a launch = 2 x FETCH_SIZE[KB] x 1024 + WRITE_SIZE[KB] x 1024 (gfx950 counts the 128-byte requests of wide reads as 64 B:
MI355X_MICROARCH.md, HBM section; as profiles/*_pmc_traffic.json).  A kernel that runs at several pyramid levels is taken at its LARGEST grid
(level 0) and at the launch with the most traffic (all lanes active: the first step of the profiled run switches every lane's keyframes).
Algorithmic bytes: what the kernel must read and write once, per level-0 pixel (or lattice sample) -- the figures of DESIGN.md section 5."""
import csv, glob, json, os, sys

d, out, lanes = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows, cols, nsamples = (int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (480, 640, 10000)
px = rows * cols


def lattice_samples(r, c, n):   # csrc/kernels_sigma.hip lattice_geometry: halve both axes while they stay even and the lattice keeps >= n samples
    pr, pc = r, c
    while n < pr * pc:
        cr, cc = pr // 2, pc // 2
        if 2 * cr != pr or 2 * cc != pc or n > cr * cc:
            break
        pr, pc = cr, cc
    return pr * pc


ns = lattice_samples(rows, cols, nsamples)
# kernel-name fragment -> (algorithmic bytes per lane at level 0, what they are)
TABLE = [
    ("k_build_system<rgbid::ByLane<rgbid::SysParams>, true, 0, 2, 1>", 32.0 * px, "fused GN iteration L0: keyframe iD, I, 4 gradients (24 B) + gathered current iD, I (8 B) per pixel"),
    ("k_build_system<rgbid::ByLane<rgbid::SysParams>, true, 0, 2, 2>", 32.0 * px, "covariance pass L0: as the GN iteration"),
    ("k_build_system<rgbid::ByLane<rgbid::SysParams>, true, 1, 2, 1>", 32.0 * px / 4, "fused GN iteration L1"),
    ("k_lattice_residuals_fused", 36.0 * ns, "per lattice sample: packed keyframe pair 8 B, gathered current iD 4 B + 4 intensity texels 16 B, two residuals written 8 B"),
    ("k_sigma_pair_arrays", 8.0 * ns, "two residual arrays read once"),
    ("k_kf_maps4", 28.0 * px, "iD read 4 B, vertex + normal maps written 24 B"),
    ("k_bilateral<2>", 2 * 8.0 * px, "two maps per launch (keyframe inverse depth + intensity): map read + filtered map written each"),
    ("k_visibility_pair", 2 * 16.0 * px, "two covisibility checks per launch (frame vs odometry and vs integration keyframe): per check two inverse-depth maps, each read as grid and as gather source (the frame's map is common to both checks: 8 of the 32 B/px can come from L2)"),
    ("k_prep_frame4", 25.0 * px, "u16 depth 2 B + rgb 3 B read; iD, luma, r, g, b planes written 20 B"),
    ("k_pyr_down_dpp", 2 * 5.0 * px, "two maps per launch (intensity + inverse depth), L0 -> L1: source read 4 B/px, quarter-size destination written"),
    ("k_fuse_frame4", 20.0 * px, "keyframe iD + weight read and written (16 B), current iD gathered (4 B)"),
    ("k_gradient4<true>", 2 * 16.0 * px, "two maps per launch: map read, two gradients + the keyframe copy written"),
    ("k_gradient4<false>", 2 * 12.0 * px, "two maps per launch: map read, two gradients written"),
]


def load(pattern):
    m = {}
    for f in glob.glob(os.path.join(d, pattern)):
        for r in csv.DictReader(open(f)):
            m.setdefault(r["Name"], []).append(r)
    return m


def grid_size(g):
    n = 1
    for t in g.split("x"):
        n *= int(t) if t.strip().isdigit() else 1
    return n


fetch, write = load("*pmc_fetch*pmc_bygrid_rgbid.csv"), load("*pmc_write*pmc_bygrid_rgbid.csv")
share = {}
tr = glob.glob(os.path.join(d, "trace_kernels_rgbid.csv"))
if tr:
    rs = list(csv.DictReader(open(tr[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rs)
    share = {r["Name"]: (float(r["TotalDurationNs"]) / tot, float(r["MaxNs"])) for r in rs}
res = []
for frag, alg_lane, what in TABLE:
    names = [n for n in fetch if frag in n]
    if not names:
        continue
    n = names[0]
    fr = [r for r in fetch[n] if r["Counter"] == "FETCH_SIZE"]
    gmax = max(grid_size(r["Grid"]) for r in fr)
    f0 = max(float(r["Max"]) for r in fr if grid_size(r["Grid"]) == gmax)
    wr = [r for r in write.get(n, []) if r["Counter"] == "WRITE_SIZE" and grid_size(r["Grid"]) == gmax]
    w0 = max(float(r["Max"]) for r in wr) if wr else 0.0
    counted = 2.0 * f0 * 1024 + w0 * 1024
    alg = alg_lane * lanes
    e = {"kernel": n.split("(")[0].replace("void ", ""), "algorithmic_bytes_per_launch": alg, "algorithmic": what,
         "FETCH_SIZE_KB_raw_max": f0, "WRITE_SIZE_KB_raw_max": w0, "counted_bytes_per_launch": counted, "counted_over_algorithmic": counted / alg}
    if n in share:
        e["share_of_kernel_time"] = share[n][0]
        e["max_launch_us"] = share[n][1] / 1e3
        e["algorithmic_TBps_at_max_launch"] = alg / share[n][1] / 1e3
    res.append(e)
json.dump({"command": f"python bench.py --lanes {lanes} (tools/profile_bench.sh: separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, largest grid of each kernel)",
           "lanes": lanes, "rows": rows, "cols": cols, "lattice_samples_per_lane": ns,
           "correction": "counted = 2 x FETCH_SIZE KB x 1024 + WRITE_SIZE KB x 1024 (gfx950 counts 128-B read requests as 64 B; MI355X_MICROARCH.md)",
           "kernels": res}, open(out, "w"), indent=1)
for e in res:
    print(f'{e["kernel"][:70]:70s} counted/alg {e["counted_over_algorithmic"]:.2f}  share {100 * e.get("share_of_kernel_time", 0):.1f} %')
