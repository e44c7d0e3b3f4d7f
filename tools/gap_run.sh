ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/gap; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for G in 0 1; do
rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace_g$G -- python $ROOT/bench.py --steps 6 --warmup 2 --reps 1 --graph $G --no-cpu-baseline --no-extras --check-streams 0 > $OUT/b$G.json 2>/dev/null
done
python $ROOT/tools/gap_analysis.py $OUT
find $OUT -name "*kernel_trace.csv" -delete
