"""Exact-fp32 forward / dX kernel at the headline size on layout-0 and layout-3 plans at several tiles (round 4).
    python tools/debug/exact_merge_timing.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from scaling_rgcn_training_amd import _lib, plan as P
import bench
n, e, r = 10_000_000, 100_000_000, 32
dev = torch.device("cuda:0")
ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, r, 64, 64, dev)
pk, pkt = _lib.pack_weights(w, root, False), _lib.pack_weights(w, root, True)
bias = torch.zeros(64, device=dev)
out = torch.empty(n, 64, device=dev)
def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]
ref = None
for tile, layout in ((352, 0), (352, 3), (320, 3), (288, 0), (288, 3), (272, 3), (256, 3), (224, 3)):
    plans = P.build_graph_plans_device(ei, et, n, r, tile, chunk=128, split=layout)
    pf, pb = plans.fwd, plans.bwd
    psf, psb = _lib.plan_struct(pf), _lib.plan_struct(pb)
    f = t(lambda: _lib.fwd(psf, x, 64, pk, bias, out, 64, 0, 0))
    cs = out.double().abs().sum().item()
    if ref is None:
        ref = out.clone()
    diff = float((out - ref).abs().max())
    b = t(lambda: _lib.bwd_dx(psb, dg, 64, pkt, out, 64, None, 0))
    merged = int((((pf.chunk_flags >> 16) & 7) != 0).sum()) if layout == 3 else 0
    print(f"tile {tile} layout {layout}: fwd {f:.3f} ms  dx {b:.3f} ms  chunks {pf.n_chunks}  head row tiles {int(pf.chunk_cnt.sum()) // 16}  "
          f"compacted chunks {merged} ({100.0 * merged / pf.n_chunks:.0f} %)  max |out - out(352, 0)| {diff:.2e}  checksum {cs:.9e}", flush=True)
    del plans, pf, pb, psf, psb
    P.clear_plan_cache()
