#!/bin/bash
# round 6, call 5: the whole GPU suite on the final tree, then the profile set (ROUND=r06) + per-kernel bench + few-lane traces
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c5; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "all rc $?" > $O/rc.txt
tail -3 $O/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/rc.txt
ROUND=r06 bash tools/profile_bench.sh > $O/profile.log 2>&1
python tools/kernel_bench.py --lanes 1024 --json gpurun_out/profiles_r06/kernel_bench_1024.json > $O/kernel_bench.log 2>&1
tail -30 $O/kernel_bench.log
cat $O/rc.txt
tail -c 3000 gpurun_out/profiles_r06/bench.json
