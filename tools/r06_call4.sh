#!/bin/bash
# round 6, call 4: map-skew sweep (config 5 + headline), no-SLP variants of the warp / prep translation units, pyrDown fast path again, the driver's command with / without the skew
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c4; mkdir -p $O
for rep in 1 2; do
  for sk in 0 4352 20736 69888 266496 1118464; do
    RGBID_ENGINE_MAP_SKEW=$sk timeout 600 python tools/placement_probe.py --after-big 2>/dev/null | grep "^{" >> $O/skew.txt
  done
done
cat $O/skew.txt
L=$ROOT/rgbid-slam_amd/lib
for rep in 1 2; do
  for v in base warpnoslp; do
    lib=$L/librgbid_hip.so; [ $v != base ] && lib=$L/librgbid_hip_$v.so
    echo "== $v rep $rep" >> $O/ab_warp.txt
    RGBID_HIP_LIB=$lib python tools/kernel_bench.py --lanes 1024 --only vis,fuse,lattice,unfused 2>&1 | grep -E "us/lane" >> $O/ab_warp.txt
  done
  for v in base prepnoslp; do
    lib=$L/librgbid_hip.so; [ $v != base ] && lib=$L/librgbid_hip_$v.so
    echo "== $v rep $rep" >> $O/ab_prep.txt
    RGBID_HIP_LIB=$lib python tools/kernel_bench.py --lanes 1024 --only pyr,sobel,maps,prep 2>&1 | grep -E "us/lane" >> $O/ab_prep.txt
  done
done
for rep in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then export RGBID_PYRDOWN_NO_FASTPATH=1; else unset RGBID_PYRDOWN_NO_FASTPATH; fi
    echo "== $v rep $rep" >> $O/ab_pyr.txt
    python tools/kernel_bench.py --lanes 1024 --only pyr --reps 40 2>&1 | grep -E "us/lane" >> $O/ab_pyr.txt
  done
done
unset RGBID_PYRDOWN_NO_FASTPATH
cat $O/ab_warp.txt $O/ab_prep.txt $O/ab_pyr.txt
# is the fused kernel faster when a launch's maps fit the 256 MB memory-side cache? (9.8 MB of level-0 maps per lane; repeated launches on the same lanes)
for L in 8 16 24 48 128 1024; do
  echo "== lanes $L" >> $O/mall.txt
  python tools/kernel_bench.py --lanes $L --only gn,unfused --reps 30 2>&1 | grep -E "us/lane" >> $O/mall.txt
done
cat $O/mall.txt
for rep in 1 2; do
  for sk in 0 69888; do
    RGBID_ENGINE_MAP_SKEW=$sk python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_skew${sk}_$rep.json 2>/dev/null
    python - <<PY >> $O/bench_skew.txt
import json
d=json.loads(open("$O/bench_skew${sk}_$rep.json").read().strip().splitlines()[-1])
x5=[x for x in d.get("extra_configs",[]) if str(x.get("config","")).startswith("5:")]
print("skew $sk rep $rep frames/s", round(d["value"]), "frac", round(d["roofline"]["frac"],4), "config5", [(round(x["value"]), round(x["u1_frac_of_hbm_peak"],4)) for x in x5],
      "lanes1/8/64 ms", [round(x["ms_per_step"],3) for x in d.get("extra_configs",[]) if str(x.get("config","")).startswith("lanes-")])
PY
  done
done
cat $O/bench_skew.txt
