#!/bin/bash
# Build librgcn_mi355x.so for gfx950 in-tree (cross-compiles without a GPU).  Usage: tools/build_lib.sh [--temps DIR]
#   --temps DIR : also keep the .s files and print the register / LDS use of every kernel (tools/kernel_resources.py)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
if [[ "${1:-}" == "--temps" ]]; then
  mkdir -p "$2"; cd "$2"
  for f in rgcn_tile_fp32 rgcn_tile_fp32_narrow rgcn_tile_fp32_wide rgcn_tile3p rgcn_dw_relmajor rgcn_dw_tile rgcn_dw_root rgcn_ep rgcn_abi rgcn_plan; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -save-temps -Rpass-analysis=kernel-resource-usage \
        "$ROOT/scaling_rgcn_training_amd/csrc/$f.hip" -o "$f.o" 2> "$f.resources.txt"
  done
fi
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build(force=True)"
echo "built $ROOT/scaling_rgcn_training_amd/librgcn_mi355x.so"
