# kernel time vs idle time between dependent launches at a given lane count: LANES=64 bash tools/gap_run.sh  (through gpurun)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/gap; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for G in ${GRAPHS:-0 1}; do
rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace_g$G -- python $ROOT/bench.py --lanes ${LANES:-2048} --streams ${STREAMS:-32} --steps ${STEPS:-6} --warmup 2 --reps 1 --graph $G --no-cpu-baseline --no-extras --check-streams 0 > $OUT/b$G.json 2>/dev/null
done
python $ROOT/tools/gap_analysis.py $OUT | tee $OUT/gap_${LANES:-2048}.txt
find $OUT -name "*kernel_trace.csv" -delete
