#!/bin/bash
set -uo pipefail
O=gpurun_out/r04z2_weight_sets_timing.txt
for v in base wsets2 base wsets2; do
  VT_WHICH=fwd,dx VT_FLAGS=32 VT_SPLIT=3 VT_TILE=272 VT_CHUNK=112 timeout -k 10 300 python tools/debug/variant_timing.py $v >> $O 2>&1
done
grep -v "amdgpu.ids" $O
