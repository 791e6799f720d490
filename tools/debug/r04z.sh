#!/bin/bash
set -uo pipefail
O=gpurun_out/r04z_weight_sets_timing.txt
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_merge_runs.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r04z_tests.log 2>&1 || { tail -40 gpurun_out/r04z_tests.log; exit 1; }
tail -3 gpurun_out/r04z_tests.log
for v in base wsets base wsets; do
  VT_WHICH=fwd,dx VT_FLAGS=32 VT_SPLIT=3 VT_TILE=272 VT_CHUNK=112 timeout -k 10 300 python tools/debug/variant_timing.py $v >> $O 2>&1
done
grep -v "amdgpu.ids" $O
