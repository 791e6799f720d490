#!/bin/bash
set -uo pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_plan_build.py tests/test_gpu_merge_runs.py -x -q > gpurun_out/r04g_tests.log 2>&1 || { tail -30 gpurun_out/r04g_tests.log; exit 1; }
tail -3 gpurun_out/r04g_tests.log
timeout -k 10 600 python tools/debug/exact_merge_timing.py > gpurun_out/r04g_exact_merge_timing.txt 2>&1 || { tail gpurun_out/r04g_exact_merge_timing.txt; exit 1; }
cat gpurun_out/r04g_exact_merge_timing.txt
VT_WHICH=fwd,dx VT_FLAGS=32 VT_SPLIT=3 VT_TILE=224 timeout -k 10 600 python tools/debug/variant_timing.py sbuf > gpurun_out/r04g_p3_check.txt 2>&1
grep -v amdgpu gpurun_out/r04g_p3_check.txt | head -5
