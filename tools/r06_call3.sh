#!/bin/bash
# round 6, call 3: the whole GPU suite on the new tree, few-lane latency with / without the update prologue, config-5 placement probe, the split TA counter passes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c3; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "all rc $?" > $O/rc.txt
tail -3 $O/t_all.log
# few-lane latency: prologue on (default) / off
for P in 16 0; do
  echo "== RGBID_ENGINE_UPDATE_PROLOGUE_LANES=$P" >> $O/lanes.txt
  RGBID_ENGINE_UPDATE_PROLOGUE_LANES=$P LANES_LIST="1 8 16" GRAPHS="0 1" bash tools/lane_latency.sh >> $O/lanes.txt 2>&1
done
for P in 16 0; do
  echo "== RGBID_ENGINE_UPDATE_PROLOGUE_LANES=$P (second pass)" >> $O/lanes.txt
  RGBID_ENGINE_UPDATE_PROLOGUE_LANES=$P LANES_LIST="1 8" GRAPHS="0" bash tools/lane_latency.sh >> $O/lanes.txt 2>&1
done
RGBID_ENGINE_UPDATE_PROLOGUE_LANES=32 LANES_LIST="32" GRAPHS="0" bash tools/lane_latency.sh >> $O/lanes.txt 2>&1
RGBID_ENGINE_UPDATE_PROLOGUE_LANES=0 LANES_LIST="32" GRAPHS="0" bash tools/lane_latency.sh >> $O/lanes.txt 2>&1
cat $O/lanes.txt
# placement probe: config 5 in fresh processes, standalone and after the headline engine, with the placement switches
for rep in 1 2; do
  for cfg in "0 0" "4352 0" "0 69888" "4352 69888"; do
    set -- $cfg
    for sc in "" "--after-big"; do
      RGBID_ENGINE_LANE_PAD=$1 RGBID_ENGINE_MAP_SKEW=$2 RGBID_ENGINE_DEBUG_ALLOC=1 timeout 600 python tools/placement_probe.py $sc 2>> $O/placement.err | grep "^{" >> $O/placement.txt
    done
  done
done
cat $O/placement.txt
# TA counters, two per pass
P=$ROOT/gpurun_out/pmc_ta2; rm -rf $P; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
BASE="--steps 2 --warmup 1 --reps 1 --lanes 2048 --no-cpu-baseline --no-extras --check-streams 0"
for cfg in fused unfused; do
  EX=""; [ $cfg = unfused ] && EX="--fused 0 --fast 0"
  timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --kernel-trace --output-format csv -d $P -o ${cfg}_ta -- python $ROOT/bench.py $BASE $EX > /dev/null 2> $P/${cfg}_ta.err; echo "pass $cfg ta rc $?"
  timeout 300 rocprofv3 --pmc TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum --kernel-trace --output-format csv -d $P -o ${cfg}_ta3 -- python $ROOT/bench.py $BASE $EX > /dev/null 2> $P/${cfg}_ta3.err; echo "pass $cfg ta3 rc $?"
done
rm -f $P/*agent_info.csv
python $ROOT/tools/summarize_prof.py $P > /dev/null
find $P -name "*kernel_trace.csv" -delete; find $P -name "*counter_collection.csv" -size +2M -delete
grep -h "k_build_system" $P/*_pmc_rgbid.csv | grep "0, 2, 1>\|0, 0, 0>" | cut -c1-40,200-
