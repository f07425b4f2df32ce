#!/bin/bash
# A/B builds of the same ABI with compile-time knobs: tools/build_variant.sh NAME [-DKNOB=V ...] -> build/libptgpu_NAME.so
# (load with PTGPU_LIB=build/libptgpu_NAME.so).  csrc/ptgpu.hip and csrc/pt_raycast_step.hip are recompiled with the knobs (each with
# its own scheduling strategy, as __graft_entry__.build() does); pt_multigpu.o comes from the normal build.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value"
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-sched-strategy=iterative-ilp "$@" -c -o build/ptgpu_$name.o pathtracer_amd/csrc/ptgpu.hip &
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c -o build/pt_raycast_step_$name.o pathtracer_amd/csrc/pt_raycast_step.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build/libptgpu_$name.so build/ptgpu_$name.o build/pt_raycast_step_$name.o build/pt_multigpu.o -ldl
echo build/libptgpu_$name.so
