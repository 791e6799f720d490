#!/bin/bash
set -uo pipefail
export TMPDIR=/tmp
timeout -k 10 120 ./tools/probes/mfma_bf16_dependent_chain > gpurun_out/r04e_probe_mfma_bf16_dependent_chain.txt 2>&1 || exit 1
cat gpurun_out/r04e_probe_mfma_bf16_dependent_chain.txt
for rung in 100k AM-like; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$rung -- python3 tools/debug/rung_profile.py $rung > gpurun_out/r04e_rung_$rung.log 2>&1 || exit 1
  cp $(find gpurun_out/prof_$rung -name "*kernel_stats.csv" | head -1) gpurun_out/r04e_${rung}_kernel_stats.csv
  rm -rf gpurun_out/prof_$rung
done
