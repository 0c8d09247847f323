#!/bin/bash
# Build a variant of libgpbo.so with extra -D flags on ONE translation unit: tools/build_variant.sh NAME FILE "-DX=1 ..."
# -> ab_libs/NAME.so (git-ignored, travels to the GPU box).  For same-box A/B runs (tools/ab_i8.sh).
set -euo pipefail
NAME=$1; FILE=$2; EXTRA=${3:-}
cd "$(dirname "$0")/../bayesian_optimisation_amd/csrc"
mkdir -p ../../ab_libs build
X=""; [ $FILE = cholinv ] && X="-mllvm -amdgpu-kernarg-preload-count=16"   # as build.sh does
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $X $EXTRA -c $FILE.hip -o ../../ab_libs/${FILE}_$NAME.o
objs=""
for f in api kernel_build kstar_mfma gemm_f64 factor cholinv subset update sigma_acq ard ard_wave posterior_f32 rescore ozaki host_api; do
  if [ $f = $FILE ]; then objs="$objs ../../ab_libs/${FILE}_$NAME.o"; else objs="$objs build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab_libs/$NAME.so $objs
rm -f ../../ab_libs/${FILE}_$NAME.o   # (the variant object is not kept: csrc/build/ holds the 14 production objects only)
echo "built ab_libs/$NAME.so"
