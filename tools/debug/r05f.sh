#!/bin/bash
set -uo pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r05f_tests.log 2>&1 || { tail -40 gpurun_out/r05f_tests.log; exit 1; }
tail -3 gpurun_out/r05f_tests.log
timeout -k 10 600 python bench.py --emulate-only --no-ladder --no-cpu-baseline --steps 10 --warmup 5 > gpurun_out/r05f_bench.json 2> gpurun_out/r05f_bench.err || { tail gpurun_out/r05f_bench.err; exit 1; }
grep "timed steps\|emulated rank" gpurun_out/r05f_bench.err
