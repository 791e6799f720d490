#!/bin/bash
set -uo pipefail
VT_WHICH=fwd,dx,dw VT_FLAGS=32 VT_SPLIT=3 VT_TILE=224 timeout -k 10 900 python tools/debug/variant_timing.py base_r04c sbuf base_r04c sbuf > gpurun_out/r04d_sbuf_timing.txt 2>&1 || exit 1
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_merge_runs.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py tests/test_gpu_nccl_smoke.py -q > gpurun_out/r04d_tests.log 2>&1; tail -40 gpurun_out/r04d_tests.log
