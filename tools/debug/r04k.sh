#!/bin/bash
set -uo pipefail
for t in 224 256 288 304 320; do
  echo "== tile $t (6-tile ring slots, chunks clamped: TIMING ONLY)" >> gpurun_out/r04k_slot6_timing.txt
  VT_WHICH=fwd,dx VT_FLAGS=32 VT_SPLIT=3 VT_TILE=$t timeout -k 10 300 python tools/debug/variant_timing.py p3s6 >> gpurun_out/r04k_slot6_timing.txt 2>&1
done
echo "== tile 224, product" >> gpurun_out/r04k_slot6_timing.txt
VT_WHICH=fwd,dx VT_FLAGS=32 VT_SPLIT=3 VT_TILE=224 timeout -k 10 300 python tools/debug/variant_timing.py prod >> gpurun_out/r04k_slot6_timing.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r04k_slot6_timing.txt
