#!/bin/bash
set -uo pipefail
V=scaling_rgcn_training_amd/_build/variants
for v in dwa64 dwa32 dwa16; do
  echo "== $v" >> gpurun_out/r04b_dw_fold_error.txt
  RGCN_LIB=$PWD/$V/$v.so timeout -k 10 300 python tools/debug/dw_split_error_probe.py >> gpurun_out/r04b_dw_fold_error.txt 2>&1 || exit 1
done
VT_WHICH=dw VT_FLAGS=32 timeout -k 10 600 python tools/debug/variant_timing.py dwf0 dwa64 dwa32 dwa16 dwf128 dwf0 dwa64 > gpurun_out/r04b_dw_fold_timing.txt 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_ep.py tests/test_gpu_merge_runs.py -x -q > gpurun_out/r04b_tests.log 2>&1 || { tail -30 gpurun_out/r04b_tests.log; exit 1; }
timeout -k 10 900 python bench.py --emulate-only --steps 10 --warmup 5 > gpurun_out/r04b_bench_emulate.json 2> gpurun_out/r04b_bench_emulate.err || { tail -30 gpurun_out/r04b_bench_emulate.err; exit 1; }
