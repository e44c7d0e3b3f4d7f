#!/bin/bash
# Builds a variant of librgbid_hip.so with extra -D flags for EVERY source file (constants of shared headers, e.g. -DRGBID_WCM=1.5f), next to the product library:
#   tools/build_variant_all.sh <name> "<flags>"   ->  rgbid-slam_amd/lib/librgbid_hip_<name>.so   (use with RGBID_HIP_LIB=... / tools/ab_bench.sh VARIANT=<name>)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/rgbid-slam_amd/csrc
name=$1; flags=$2
base="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value"
T=/tmp/variant_all_$name; mkdir -p $T
objs=""
for f in c_api c_api_batched kernels_prep kernels_bilateral kernels_warp kernels_sigma kernels_system kernels_calib engine kfalign; do
  extra=""; case $f in kernels_system|kernels_bilateral|kernels_warp) extra="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc $base $extra $flags -c $C/$f.hip -o $T/$f.o 2>/dev/null &
  objs="$objs $T/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/rgbid-slam_amd/lib/librgbid_hip_$name.so $objs
echo built $ROOT/rgbid-slam_amd/lib/librgbid_hip_$name.so
