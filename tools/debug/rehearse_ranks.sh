#!/bin/bash
# N-rank rehearsal of bench.py's multi-GPU code path on a ONE-GPU box: gloo collectives (slow: seconds per step), the ranks share
# the card (<= 4 ranks), the 1M-node / 10M-edge rung; RGCN_CU_ROUND=64 makes dist.piece_tiles cut pieces of whole "rounds" there.
#   gpurun --timeout 900 -- bash tools/debug/rehearse_ranks.sh 4
set -uo pipefail
N=${1:-4}
RGCN_CU_ROUND=${RGCN_CU_ROUND:-64} RGCN_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus $N --steps 3 --warmup 1 --nodes 1000000 --edges 10000000 > gpurun_out/rehearsal_${N}rank.json 2> gpurun_out/rehearsal_${N}rank.err || { tail -30 gpurun_out/rehearsal_${N}rank.err; exit 1; }
grep "bench " gpurun_out/rehearsal_${N}rank.err | tail -12
tail -c 1200 gpurun_out/rehearsal_${N}rank.json
