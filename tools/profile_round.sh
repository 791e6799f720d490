#!/bin/bash
# Collect the artefacts profiles/ holds for one build, on the GPU box:  tools/profile_round.sh TAG
#   gpurun_out/TAG_bench.json                 python bench.py   (defaults: 20 warm-up + 50 timed steps, ladder, CPU baseline)
#   gpurun_out/TAG_rocprofv3_kernel_stats.csv rocprofv3 --kernel-trace --stats of bench.py --steps 5 --warmup 2
#   gpurun_out/TAG_pmc_traffic.json           FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 correction)
#   gpurun_out/TAG_pmc_sq.txt                 SQ counters (MFMA busy, LDS conflicts, instruction mix), one pass per set
# The program itself follows `--` (no env / bash -c hop under rocprofv3); every step is bounded by timeout.
set -eo pipefail
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 500 python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ladder --emulate-world 0 > $O/prof_stats.log 2>&1
cp $(find $O/prof_stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_rocprofv3_kernel_stats.csv
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/prof_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ladder --emulate-world 0 > $O/prof_$c.log 2>&1
    echo "$c done"
done
python3 tools/pmc_summary.py --traffic-json $O/prof_FETCH_SIZE $O/prof_WRITE_SIZE > $O/${TAG}_pmc_traffic.json
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
    i=$((i + 1))
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/prof_sq$i -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ladder --emulate-world 0 > $O/prof_sq$i.log 2>&1
    echo "sq set $i done"
done
python3 tools/pmc_summary.py $O/prof_sq1 $O/prof_sq2 $O/prof_sq3 > $O/${TAG}_pmc_sq.txt
rm -rf $O/prof_stats $O/prof_FETCH_SIZE $O/prof_WRITE_SIZE $O/prof_sq1 $O/prof_sq2 $O/prof_sq3
cat $O/${TAG}_bench.json
