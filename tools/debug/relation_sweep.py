"""What relation-packed ring slots could buy the forward / dX kernel at most: the headline graph's nodes and edges with 1, 2, 4,
8, 16, 32 relations -- the same rows through the same kernel, but a (tile, relation) group spans 19 ... 0.6 chunks, so the
per-chunk fixed cost (barrier, drain, weight swap, pipeline fill) is paid per 8 full row tiles at R = 1 and per ~4.3 at R = 32.
    python tools/debug/relation_sweep.py [layout]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from scaling_rgcn_training_amd import _lib, plan as P
import bench

n, e = 10_000_000, 100_000_000
layout = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
for r in (1, 2, 4, 8, 16, 32):
    ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, r, 64, 64, dev)
    plans = P.build_graph_plans_device(ei, et, n, r, 224, chunk=128, split=layout)
    del ei, et
    pk = _lib.pack_weights(w, root, False)
    ps = _lib.plan_struct(plans.fwd)
    out = torch.empty(n, 64, device=dev)
    bias = torch.zeros(64, device=dev)
    fn = lambda: _lib.fwd(ps, x, 64, pk, bias, out, 64, 0, _lib.FLAG_SPLIT_PRODUCERS)
    fn(); fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    pf = plans.fwd
    rt = int(((pf.chunk_cnt + 15) // 16).sum()) if layout == 0 else -1
    print(f"R {r:3d} layout {layout}: fwd {ts[len(ts) // 2]:.3f} ms (min {ts[0]:.3f})  chunks {pf.n_chunks}  row tiles {rt}  "
          f"row tiles per chunk {rt / pf.n_chunks:.2f}", flush=True)
    del plans, ps, pk, out, x, dg
    P.clear_plan_cache()
    torch.cuda.empty_cache()
