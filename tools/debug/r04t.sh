#!/bin/bash
set -uo pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_plan_build.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "dw or pair or full_size or module or tile_major" > gpurun_out/r04t_tests.log 2>&1 || { tail -40 gpurun_out/r04t_tests.log; exit 1; }
tail -3 gpurun_out/r04t_tests.log
for l in 0 5; do
  echo "== dW plan layout $l" >> gpurun_out/r04t_dw_pairs_timing.txt
  RGCN_DW_PLAN_LAYOUT=$l VT_WHICH=dw VT_FLAGS=32 timeout -k 10 300 python tools/debug/variant_timing.py prod >> gpurun_out/r04t_dw_pairs_timing.txt 2>&1
  RGCN_DW_PLAN_LAYOUT=$l VT_WHICH=dw VT_FLAGS=0 timeout -k 10 300 python tools/debug/variant_timing.py prod >> gpurun_out/r04t_dw_pairs_timing.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/r04t_dw_pairs_timing.txt | grep "==\|dw tiles"
