#!/usr/bin/env bash
# Builds libomfs_splat.so for gfx950 in-tree (cross-compiles without a GPU).
# -ffp-contract=off: FMAs are written explicitly so the geometric stage is bit-exact vs oracle/.
# -fno-slp-vectorize: the SLP pass pairs scalar fp32 operations into v_pk_* and pays for every pair with register moves to make
#   its operands adjacent -- the one-wave forward carried 22 v_mov per 4 list entries for 18 packed operations (measured: composite
#   forward 0.233 -> 0.213 ms, loss, binning and the FLAME backward a few us each; same bits either way).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libomfs_splat.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function"
objs=()
pids=()
# machine-scheduler strategy per translation unit (measured on the bench workload, tools/_ab/flags.sh: the loss kernels 0.104 ->
# 0.099 ms and composite_bwd 0.237 -> 0.234 ms with max-ilp, project_bwd 0.056 -> 0.052 ms with max-memory-clause; the rest do not care)
declare -A SCHED=([loss_adam]="-mllvm -amdgpu-sched-strategy=max-ilp" [composite]="-mllvm -amdgpu-sched-strategy=max-ilp"
                  [project_bwd]="-mllvm -amdgpu-sched-strategy=max-memory-clause")
for src in flame project binning composite project_bwd loss_adam simple_flame densify image_io; do
  "$HIPCC" $FLAGS ${SCHED[$src]:-} ${EXTRA_HIPCC_FLAGS:-} -c "$here/$src.hip" -o "$here/$src.o" &
  pids+=($!)
  objs+=("$here/$src.o")
done
"$HIPCC" $FLAGS -x hip -c "$here/api.cpp" -o "$here/api.o" &
pids+=($!)
objs+=("$here/api.o")
"$HIPCC" $FLAGS -x hip -c "$here/collectives.cpp" -o "$here/collectives.o" &      # RCCL by dlopen: no link-time dependency
pids+=($!)
objs+=("$here/collectives.o")
for p in "${pids[@]}"; do wait "$p"; done          # a failed translation unit fails the build (a bare `wait` returns 0)
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}" -ldl
echo "built $out"
# Second implementations of the composite backward pass (matrix-core reduction, lanes = list entries): a library of their own, loaded
# only by tests/ and tools/ -- the shipped library holds ONE backward pass and no switch that changes gradients.
"$HIPCC" $FLAGS -mllvm -amdgpu-sched-strategy=max-ilp ${EXTRA_HIPCC_FLAGS:-} -shared -o "$here/../libomfs_experiments.so" "$here/composite_experiments.hip"
echo "built $here/../libomfs_experiments.so"
