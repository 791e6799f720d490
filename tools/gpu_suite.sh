#!/bin/bash
# The whole `pytest -m gpu` suite on the GPU box, as the driver runs it:  gpurun --timeout 1200 -- bash tools/gpu_suite.sh
set -uo pipefail
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -25 gpurun_out/gpu_tests.log
exit $rc
