#!/bin/bash
# Variant of the library in which only csrc/rgcn_tile3p.hip is rebuilt with extra -D flags (seconds instead of minutes);
# the other objects come from the product build (scaling_rgcn_training_amd/_build/*.o).  Usage: build_variant_p3.sh NAME [-DFOO=1 ...]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
name="$1"; shift
B="$ROOT/scaling_rgcn_training_amd/_build"
out="$B/variants"
mkdir -p "$out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$ROOT/scaling_rgcn_training_amd/csrc/rgcn_tile3p.hip" -o "$out/p3_$name.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$B/rgcn_tile_fp32.o" "$B/rgcn_tile_fp32_narrow.o" "$B/rgcn_tile_fp32_wide.o" "$B/rgcn_dw_relmajor.o" "$B/rgcn_dw_tile.o" \
    "$B/rgcn_dw_root.o" "$B/rgcn_ep.o" "$B/rgcn_abi.o" "$B/rgcn_plan.o" "$out/p3_$name.o" -o "$out/$name.so"
echo "built $out/$name.so"
