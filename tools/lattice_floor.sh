#!/bin/bash
# HBM bytes per lattice sample of k_lattice_residuals_fused at level 0 (640x480, stride-4 lattice), counted (separate --pmc FETCH_SIZE / WRITE_SIZE
# passes), with the sequence's relative poses and with the identity warp: the second is the line-touch floor of an axis-aligned lattice.
#   LANES=1024 bash tools/lattice_floor.sh     (through gpurun; writes gpurun_out/lattice_floor.txt)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
LANES=${LANES:-1024}
OUTF=$ROOT/gpurun_out/lattice_floor.txt; mkdir -p $ROOT/gpurun_out; : > $OUTF
for mode in "" "--identity-pose"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    OUT=/tmp/latfloor; rm -rf $OUT; mkdir -p $OUT
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -o p -- python $ROOT/tools/kernel_bench.py --lanes $LANES --only lattice --reps 6 $mode > /dev/null 2>&1
    python - "$OUT" "$c" "$LANES" "${mode:-sequence-poses}" <<'PY' | tee -a $OUTF
import csv, glob, sys
d, c, lanes, mode = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
v = [float(r["Counter_Value"]) for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))
     if "k_lattice_residuals_fused" in r["Kernel_Name"] and r["Counter_Name"] == c]
ns = 19200
mult = 2.0 if c == "FETCH_SIZE" else 1.0     # gfx950: FETCH_SIZE counts 128-B requests as 64 B
print(f"{mode:16s} {c:10s} launches {len(v):3d}  max {max(v):12.1f} KB raw  -> {mult * max(v) * 1024 / (lanes * ns):7.2f} B per sample")
PY
  done
done
