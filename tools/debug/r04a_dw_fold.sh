#!/bin/bash
# round 4: the periodic fold of the split dW kernel's accumulator -- error against float64 and launch time per variant
set -uo pipefail
V=scaling_rgcn_training_amd/_build/variants
for v in dwf0 dwf64 dwf16 dwf64ns dwf256; do
  echo "== $v" >> gpurun_out/r04a_dw_fold_error.txt
  RGCN_LIB=$PWD/$V/$v.so timeout -k 10 300 python tools/debug/dw_split_error_probe.py >> gpurun_out/r04a_dw_fold_error.txt 2>&1 || exit 1
done
VT_WHICH=dw VT_FLAGS=32 timeout -k 10 600 python tools/debug/variant_timing.py dwf0 dwf64 dwf16 dwf64ns dwf256 dwf0 dwf64 > gpurun_out/r04a_dw_fold_timing.txt 2>&1
