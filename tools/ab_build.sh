#!/bin/bash
# A/B of compile-time variants ON the GPU box: each argument is a set of extra -D flags for kernels_system.hip; rebuilds the library and
# prints the per-kernel table of a short bench run.   usage: FILTER=... EXTRA="--fused 1" bash tools/ab_build.sh "-DX=1" "-DX=2"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for flags in "$@"; do
  i=$((i+1))
  echo "=== [$i] $flags"
  rm -f $ROOT/rgbid-slam_amd/csrc/kernels_system.o $ROOT/rgbid-slam_amd/csrc/kernels_sigma.o $ROOT/rgbid-slam_amd/csrc/kernels_prep.o
  make -C $ROOT/rgbid-slam_amd/csrc -j8 SYSFLAGS="-fno-slp-vectorize -DRGBID_SYS_NT_LOADS -DRGBID_ROW_PTR_MUL64 $flags" SIGFLAGS="$flags" > /dev/null 2>&1 || { echo build failed; continue; }
  TAG=_abb$i STEPS=${STEPS:-3} WARMUP=1 EXTRA="${EXTRA:-}" bash $ROOT/tools/quick_prof.sh 2>&1 | grep -E "${FILTER:-rgbid::}" | head -${HEAD:-8}
  grep -o '"value": [0-9.]*' $ROOT/gpurun_out/quick_abb$i/bench_under_rocprof.json | head -1
done
