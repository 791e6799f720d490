"""Times rgcn_bwd_dw_tiles (relations) and the root-only walk of rgcn_bwd_dw at the headline size (debug aid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scaling_rgcn_training_amd import _lib, plan as P
import bench
n, e, r = 10_000_000, 100_000_000, 32
dev = torch.device("cuda:0")
ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, r, 64, 64, dev)
tile, chunk = P.choose_layout(n, e, r, 64, 64)
plans = P.build_graph_plans_device(ei, et, n, r, tile, chunk=chunk, dw_tiles=True)
del ei, et
dw, dr, db = torch.empty_like(w), torch.empty_like(root), torch.empty(64, device=dev)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]
psd, psf = _lib.plan_struct(plans.dw), _lib.plan_struct(plans.fwd)
print("dw plan: tiles", plans.dw.n_tiles, "units", plans.dw.n_units, "chunks", plans.dw.n_chunks, flush=True)
print("tile-major d_weight       %.3f ms" % t(lambda: _lib.bwd_dw_tiles(psd, plans.dw_walk, x, 64, dg, 64, dw)), flush=True)
print("root-only d_root, d_bias  %.3f ms" % t(lambda: _lib.bwd_dw(psf, x, 64, dg, 64, None, dr, db, _lib.FLAG_DW_ROOT_ONLY)), flush=True)
print("relation-major, all       %.3f ms" % t(lambda: _lib.bwd_dw(psf, x, 64, dg, 64, dw, dr, db)), flush=True)
