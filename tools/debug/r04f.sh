#!/bin/bash
set -uo pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_merge_runs.py -x -q > gpurun_out/r04f_tests.log 2>&1 || { tail -30 gpurun_out/r04f_tests.log; exit 1; }
tail -3 gpurun_out/r04f_tests.log
timeout -k 10 600 python tools/debug/exact_merge_timing.py > gpurun_out/r04f_exact_merge_timing.txt 2>&1 || { tail gpurun_out/r04f_exact_merge_timing.txt; exit 1; }
cat gpurun_out/r04f_exact_merge_timing.txt
timeout -k 10 600 python tools/debug/small_rung_layouts.py > gpurun_out/r04f_small_rung_layouts.txt 2>&1 || { tail gpurun_out/r04f_small_rung_layouts.txt; exit 1; }
cat gpurun_out/r04f_small_rung_layouts.txt
