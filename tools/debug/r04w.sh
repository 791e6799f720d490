#!/bin/bash
set -uo pipefail
O=gpurun_out/r04w_tile_boundary_timing.txt
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_merge_runs.py tests/test_gpu_models.py tests/test_gpu_decomposed.py -x -q > gpurun_out/r04w_tests.log 2>&1 || { tail -40 gpurun_out/r04w_tests.log; exit 1; }
tail -3 gpurun_out/r04w_tests.log
for v in epifast epiall epifast; do
  VT_WHICH=fwd,dx VT_FLAGS=32 VT_SPLIT=3 VT_TILE=272 VT_CHUNK=112 timeout -k 10 300 python tools/debug/variant_timing.py $v >> $O 2>&1
done
echo "== exact fp32 kernel, layout 3 at T = 352" >> $O
VT_WHICH=fwd,dx VT_FLAGS=0 VT_SPLIT=3 VT_TILE=352 VT_CHUNK=128 timeout -k 10 300 python tools/debug/variant_timing.py epifast >> $O 2>&1
grep -v "amdgpu.ids" $O
