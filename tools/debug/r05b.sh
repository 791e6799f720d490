#!/bin/bash
set -uo pipefail
O=gpurun_out/r05c_flags_timing.txt
for v in base flags2 base flags2; do
  VT_WHICH=fwd,dx VT_FLAGS=32 VT_SPLIT=3 VT_TILE=272 VT_CHUNK=112 timeout -k 10 120 python tools/debug/variant_timing.py $v >> $O 2>&1 || { echo "variant $v failed or timed out" >> $O; break; }
done
grep -v "amdgpu.ids" $O
