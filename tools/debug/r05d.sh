#!/bin/bash
set -uo pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_plan_build.py tests/test_gpu_dist.py tests/test_gpu_ep.py -x -q > gpurun_out/r05d_tests.log 2>&1 || { tail -40 gpurun_out/r05d_tests.log; exit 1; }
tail -3 gpurun_out/r05d_tests.log
timeout -k 10 300 python bench.py --emulate-world 0 --no-ladder --no-cpu-baseline --steps 10 --warmup 5 > gpurun_out/r05d_bench.json 2> gpurun_out/r05d_bench.err || { tail gpurun_out/r05d_bench.err; exit 1; }
grep "timed steps\|plan built\|plans" gpurun_out/r05d_bench.err
python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -2
