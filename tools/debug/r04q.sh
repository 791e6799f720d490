#!/bin/bash
set -uo pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_nccl_smoke.py -x -q > gpurun_out/r04q_tests.log 2>&1 || { tail -40 gpurun_out/r04q_tests.log; exit 1; }
tail -3 gpurun_out/r04q_tests.log
