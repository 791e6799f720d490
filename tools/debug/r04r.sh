#!/bin/bash
# N > 1 rehearsal of bench.py on the one GPU of the box: gloo backend, ranks share the card (the driver's SCALE run uses RCCL)
set -uo pipefail
export RGCN_BENCH_BACKEND=gloo
for cfg in "2 full" "4 full" "2 needed"; do
  set -- $cfg
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 2951$1 bench.py --gpus $1 --steps 3 --warmup 2 --nodes 1000000 --edges 10000000 --exchange $2 > gpurun_out/r04r_bench_$1rank_gloo_$2.json 2> gpurun_out/r04r_bench_$1rank_gloo_$2.err || { tail -30 gpurun_out/r04r_bench_$1rank_gloo_$2.err; exit 1; }
  tail -c 1500 gpurun_out/r04r_bench_$1rank_gloo_$2.json; echo
done
# hub graph, balanced cut + hubs split across ranks: skewed destinations via a small script
