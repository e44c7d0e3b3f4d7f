#!/bin/bash
# round 6, call 11: what the guard band's cold paths cost in the bench (timing-only variant in which no pixel is ever recomputed) and two narrower lane-constant bands (WCM)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c11; mkdir -p $O
for v in nofire wcm150 wcm125; do
  echo "== $v" >> $O/ab.txt
  VARIANT=$v STEPS=20 WARMUP=5 REPS=3 PAIRS=2 bash tools/ab_bench.sh >> $O/ab.txt 2>&1
done
cat $O/ab.txt
