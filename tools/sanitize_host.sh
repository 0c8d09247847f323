#!/bin/bash
# Host-side sanitizer job (CPU only - no GPU AddressSanitizer on the pool): the HOST half of every translation
# unit of bayesian_optimisation_amd/csrc (planner cholinv_plan.h, argument validation, workspace arithmetic, launch set-up)
# compiled with -fsanitize=address,undefined -fno-gpu-sanitize (the device half is compiled as usual, uninstrumented: a
# --cuda-host-only object does not link, it still refers to its fat binary) and driven by tools/sanitize_host.cpp, which
# never reaches a kernel launch.  Usage: bash tools/sanitize_host.sh [build dir]   (tests/test_host_sanitizers_cpu.py runs it)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=${1:-/tmp/gpbo_sanitize}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SAN="-fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer"
FLAGS="--offload-arch=gfx950 -fno-gpu-sanitize -O1 -gline-tables-only -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $SAN"
mkdir -p "$OUT"
cd "$ROOT/bayesian_optimisation_amd/csrc"
pids=()
objs=""
for f in api kernel_build kstar_mfma gemm_f64 factor cholinv subset update sigma_acq ard ard_wave posterior_f32 rescore ozaki host_api; do
  $HIPCC $FLAGS -c $f.hip -o "$OUT/$f.o" &
  pids+=($!)
  objs="$objs $OUT/$f.o"
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC $FLAGS -x hip -c "$ROOT/tools/sanitize_host.cpp" -o "$OUT/sanitize_host.o"
$HIPCC --offload-arch=gfx950 $SAN -o "$OUT/sanitize_host" "$OUT/sanitize_host.o" $objs
# detect_leaks stays on: the planner's vectors and the device-plan cache must not leak on the refused paths
ASAN_OPTIONS=${ASAN_OPTIONS:-abort_on_error=0:halt_on_error=1} UBSAN_OPTIONS=${UBSAN_OPTIONS:-print_stacktrace=1:halt_on_error=1} "$OUT/sanitize_host"
