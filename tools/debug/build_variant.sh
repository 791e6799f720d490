#!/bin/bash
# Build a variant of librgcn_mi355x.so with extra -D flags into scaling_rgcn_training_amd/_build/variants/<name>.so
# (travels to the GPU box with the snapshot; tools/debug/variant_timing.py times it).  Usage: build_variant.sh NAME [-DFOO=1 ...]
# Every kernel file is rebuilt with the flags (the plan builder without); for a variant of ONE file use build_variant_p3.sh's scheme.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
name="$1"; shift
out="$ROOT/scaling_rgcn_training_amd/_build/variants"
mkdir -p "$out/obj_$name"
cd "$out/obj_$name"
C="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC"
for f in rgcn_tile_fp32 rgcn_tile_fp32_narrow rgcn_tile_fp32_wide rgcn_tile3p rgcn_dw_relmajor rgcn_dw_tile rgcn_dw_root rgcn_ep rgcn_abi; do
  $C "$@" -c "$ROOT/scaling_rgcn_training_amd/csrc/$f.hip" -o $f.o &
done
$C -c "$ROOT/scaling_rgcn_training_amd/csrc/rgcn_plan.hip" -o rgcn_plan.o
wait
$C -shared *.o -o "$out/$name.so"
echo "built $out/$name.so"
