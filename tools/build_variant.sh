#!/bin/bash
# Builds a variant of librgbid_hip.so with extra -D flags for ONE source file, next to the product library (which is left untouched):
#   tools/build_variant.sh <name> <file.hip> "<flags>"   ->  rgbid-slam_amd/lib/librgbid_hip_<name>.so   (use with RGBID_HIP_LIB=...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/rgbid-slam_amd/csrc
name=$1; file=$2; flags=$3
base="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value"
{ [ "$file" = kernels_system.hip ] || [ "$file" = kernels_bilateral.hip ] || [ "$file" = kernels_warp.hip ]; } && base="$base -fno-slp-vectorize"
make -C $C -j8 > /dev/null
/opt/rocm/bin/hipcc $base $flags -c $C/$file -o /tmp/variant_$name.o
objs=""
for f in c_api c_api_batched kernels_prep kernels_bilateral kernels_warp kernels_sigma kernels_system kernels_calib engine kfalign; do
  if [ "$f.hip" = "$file" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $C/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/rgbid-slam_amd/lib/librgbid_hip_$name.so $objs
echo built $ROOT/rgbid-slam_amd/lib/librgbid_hip_$name.so
