#!/bin/bash
# quick SQ counter passes for the current build: tools/pmc_quick.sh TAG
set -eo pipefail
TAG=${1:?tag}
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; do
    i=$((i + 1))
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/q_sq$i -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ladder --emulate-world 0 > $O/q_sq$i.log 2>&1
done
python3 tools/pmc_summary.py $O/q_sq1 $O/q_sq2 $O/q_sq3 $O/q_sq4 > $O/${TAG}_pmc_sq.txt
rm -rf $O/q_sq1 $O/q_sq2 $O/q_sq3 $O/q_sq4
grep -A18 "rgcn_tile" $O/${TAG}_pmc_sq.txt
