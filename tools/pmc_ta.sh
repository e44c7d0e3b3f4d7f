#!/bin/bash
# Texture-address / L1 / L2 counters of the level-0 Gauss-Newton kernel in the bench (round 6): what bounds the fused kernel in situ.
# Own --pmc passes (kernel-trace only, never combined with the hip / hsa / sys trace domains), the fused FAST kernel <..., 0, 2, 1> beside the
# unfused 8-stream kernel <..., 0, 0, 0> of the same workload -> gpurun_out/pmc_ta/{fused,unfused}_*_pmc_rgbid.csv
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_ta
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BASE="--steps ${STEPS:-2} --warmup 1 --reps 1 --lanes ${LANES:-2048} --no-cpu-baseline --no-extras --check-streams 0"
declare -A PASS
# a TA instance has two counters: three in one pass is refused ("exceeds the capabilities of the hardware")
PASS[ta]="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
PASS[ta3]="TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
PASS[ta2]="TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
PASS[tcp]="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
PASS[tcp2]="TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
PASS[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
PASS[sq]="SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
for cfg in fused unfused; do
  EX=""; [ $cfg = unfused ] && EX="--fused 0 --fast 0"
  for p in ta ta3 ta2 tcp tcp2 tcc sq; do
    timeout 420 rocprofv3 --pmc ${PASS[$p]} --kernel-trace --output-format csv -d $OUT -o ${cfg}_$p -- python $ROOT/bench.py $BASE $EX > /dev/null 2> $OUT/${cfg}_$p.err
    echo "pass $cfg $p rc $?"
  done
done
rm -f $OUT/*agent_info.csv
python $ROOT/tools/summarize_prof.py $OUT > /dev/null
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
grep -h "k_build_system" $OUT/*_pmc_rgbid.csv | head -120
ls -la $OUT
