#!/bin/bash
set -uo pipefail
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r04h_gpu_tests.log 2>&1; rc=$?
tail -25 gpurun_out/r04h_gpu_tests.log
exit $rc
