#!/bin/bash
set -uo pipefail
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_merge_runs.py tests/test_gpu_models.py tests/test_gpu_decomposed.py tests/test_gpu_binding_example.py -x -q > gpurun_out/r04y_tests.log 2>&1 || { tail -40 gpurun_out/r04y_tests.log; exit 1; }
tail -3 gpurun_out/r04y_tests.log
timeout -k 10 600 python bench.py --emulate-world 0 --no-cpu-baseline --steps 20 --warmup 8 > gpurun_out/r04y_bench.json 2> gpurun_out/r04y_bench.err || { tail gpurun_out/r04y_bench.err; exit 1; }
grep "timed steps\|launch \|alt\|ladder" gpurun_out/r04y_bench.err
