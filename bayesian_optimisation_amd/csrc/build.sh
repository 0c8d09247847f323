#!/bin/bash
# Build libgpbo.so for gfx950 in-tree (travels to the GPU box with the snapshot; git-ignored).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
# GPBO_DIAG=1: also compile the timing-only kernel variants that tools/tile_stamps.py and the GPBO_*_VARIANT
# environment switches select (wrong results by design; never in the shipped library)
if [ "${GPBO_DIAG:-0}" = "1" ]; then FLAGS="$FLAGS -DGPBO_DIAGNOSTICS"; fi
mkdir -p build
pids=()
for f in api kernel_build kstar_mfma gemm_f64 factor cholinv subset update sigma_acq ard ard_wave posterior_f32 rescore ozaki host_api; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ gpbo_internal.h -nt build/$f.o ] || [ cholinv_plan.h -nt build/$f.o ] || [ exp_neg.h -nt build/$f.o ] || [ potrf_diag64.h -nt build/$f.o ] || [ ../../include/gpbo.h -nt build/$f.o ]; then
    X=""; [ $f = cholinv ] && X="-mllvm -amdgpu-kernarg-preload-count=16"   # cholinv.hip: see cholinv_kernel
    $HIPCC $FLAGS $X -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libgpbo.so build/api.o build/kernel_build.o build/kstar_mfma.o build/gemm_f64.o build/factor.o build/cholinv.o build/subset.o build/update.o build/sigma_acq.o build/ard.o build/ard_wave.o build/posterior_f32.o build/rescore.o build/ozaki.o build/host_api.o
echo "built $(cd .. && pwd)/libgpbo.so"
