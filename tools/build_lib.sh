#!/bin/bash
# Build librgcn_mi355x.so for gfx950 in-tree (cross-compiles without a GPU).  Usage: tools/build_lib.sh [--temps DIR]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/scaling_rgcn_training_amd/csrc/rgcn_kernels.hip"
OUT="$ROOT/scaling_rgcn_training_amd/librgcn_mi355x.so"
EXTRA=()
if [[ "${1:-}" == "--temps" ]]; then
  mkdir -p "$2"; cd "$2"; EXTRA=(-save-temps -Rpass-analysis=kernel-resource-usage)
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "${EXTRA[@]}" "$SRC" -o "$OUT"
echo "built $OUT"
