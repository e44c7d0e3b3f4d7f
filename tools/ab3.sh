#!/bin/bash
# same-box A/B/C of product libraries on the per-kernel bench: VARIANTS="old new nofire" (lib/librgbid_hip_<v>.so; "new" = the product library)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
L=$ROOT/rgbid-slam_amd/lib
for rep in 1 2; do for v in ${VARIANTS:-old new}; do
  lib=$L/librgbid_hip.so; [ $v != new ] && lib=$L/librgbid_hip_$v.so
  echo "== $v rep $rep"
  RGBID_HIP_LIB=$lib python $ROOT/tools/kernel_bench.py --lanes ${LANES:-1024} --only ${ONLY:-gn} 2>&1 | grep -E "us/lane"
done; done
