#!/bin/bash
# Build a variant of librgcn_mi355x.so with extra -D flags into scaling_rgcn_training_amd/_build/variants/<name>.so
# (travels to the GPU box with the snapshot; tools/debug/variant_timing.py times it).  Usage: build_variant.sh NAME [-DFOO=1 ...]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
name="$1"; shift
out="$ROOT/scaling_rgcn_training_amd/_build/variants"
mkdir -p "$out/obj_$name"
cd "$out/obj_$name"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$ROOT/scaling_rgcn_training_amd/csrc/rgcn_kernels.hip" -o k.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$ROOT/scaling_rgcn_training_amd/csrc/rgcn_dw_root.hip" -o r.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$ROOT/scaling_rgcn_training_amd/csrc/rgcn_plan.hip" -o p.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$ROOT/scaling_rgcn_training_amd/csrc/rgcn_tile3p.hip" -o t.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared k.o r.o p.o t.o -o "$out/$name.so"
echo "built $out/$name.so"
