#!/bin/bash
set -uo pipefail
VT_WHICH=dw VT_FLAGS=32 timeout -k 10 600 python tools/debug/variant_timing.py prod dwpairs prod dwpairs > gpurun_out/r04s_dw_pairs_prototype.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r04s_dw_pairs_prototype.txt | grep "variant\|dw tiles"
VT_WHICH=dw VT_FLAGS=0 timeout -k 10 600 python tools/debug/variant_timing.py prod dwpairs >> gpurun_out/r04s_dw_pairs_prototype.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r04s_dw_pairs_prototype.txt | grep "variant\|dw tiles" | tail -4
