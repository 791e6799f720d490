#!/bin/bash
set -uo pipefail
timeout -k 10 1000 python -m pytest tests/test_gpu_fullsize.py -x -q > gpurun_out/r05a_tests.log 2>&1 || { tail -40 gpurun_out/r05a_tests.log; exit 1; }
tail -5 gpurun_out/r05a_tests.log
