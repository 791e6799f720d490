#!/bin/bash
set -uo pipefail
O=gpurun_out/r04x_tile_boundary_timing.txt
for v in epifast epinovm noepi epifast epinovm; do
  VT_WHICH=fwd,dx VT_FLAGS=32 VT_SPLIT=3 VT_TILE=272 VT_CHUNK=112 timeout -k 10 300 python tools/debug/variant_timing.py $v >> $O 2>&1
done
grep -v "amdgpu.ids" $O
