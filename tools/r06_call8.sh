#!/bin/bash
# round 6, call 8: the long fuzz campaign and the soak runs on the final tree
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c8; mkdir -p $O
RGBID_FUZZ_N=600 RGBID_FUZZ_ENGINE_N=120 RGBID_FUZZ_CPP_N=40 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider > $O/fuzz.log 2>&1; echo "fuzz rc $?" > $O/rc.txt
tail -3 $O/fuzz.log
timeout 1200 python tools/soak.py full 200 > $O/soak_full.log 2>&1; echo "soak full rc $?" >> $O/rc.txt
timeout 1200 python tools/soak.py > $O/soak_small.log 2>&1; echo "soak small rc $?" >> $O/rc.txt
timeout 900 python -m pytest tests/test_gpu_engine.py -q -m gpu -s -k "chi" 2>&1 | grep -i "chi-squared termination" > $O/chi_stats.txt
cat $O/rc.txt; cat $O/soak_full.log $O/soak_small.log | cut -c1-400; cat $O/chi_stats.txt
