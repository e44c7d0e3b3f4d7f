#!/bin/bash
# same-box A/B of two product libraries on the per-kernel bench: lib/librgbid_hip_old.so against lib/librgbid_hip.so, alternating
#   usage (through gpurun): LANES=1024 ONLY=gn,vis,fuse,lattice bash tools/ab_kernels.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
L=$ROOT/rgbid-slam_amd/lib
for rep in 1 2; do for v in old new; do
  lib=$L/librgbid_hip.so; [ $v = old ] && lib=$L/librgbid_hip_old.so
  echo "== $v rep $rep"
  RGBID_HIP_LIB=$lib python $ROOT/tools/kernel_bench.py --lanes ${LANES:-1024} --only ${ONLY:-gn,vis,fuse,lattice} 2>&1 | grep -E "us/lane"
done; done
