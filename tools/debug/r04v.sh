#!/bin/bash
# kernel traces of the two small rungs that miss their targets (100k / 1M and AM-like)
set -uo pipefail
export TMPDIR=/tmp
O=gpurun_out
for r in 100k AM-like; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rung_$r -- python3 tools/debug/rung_profile.py $r > $O/r04v_rung_$r.log 2>&1 || { tail -20 $O/r04v_rung_$r.log; exit 1; }
  cp $(find $O/prof_rung_$r -name "*kernel_stats.csv" | head -1) $O/r04v_rung_${r}_kernel_stats.csv
  rm -rf $O/prof_rung_$r
done
echo ok
