"""Producer-split bf16 x 3 kernel (RGCN_FLAG_SPLIT_PRODUCERS) against the exact-fp32 kernel on the same plan (tile 224):
max |difference| of the forward and dX outputs, then timing.  Usage: tile3p_check.py N E [skew]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scaling_rgcn_training_amd import _lib, plan as P
import bench
n, e, r = int(sys.argv[1]), int(sys.argv[2]), 32
dev = torch.device("cuda:0")
ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, r, 64, 64, dev)
if len(sys.argv) > 3:      # hub graph: a tenth of the edges point at 16 nodes (repeated destinations inside row tiles)
    m = torch.rand(e, device=dev) < 0.1
    ei[1, m] = torch.randint(0, 16, (int(m.sum()),), device=dev)
tile = int(os.environ.get("VT_TILE", 224))
plans = P.build_graph_plans_device(ei, et, n, r, tile, chunk=128, dw_tiles=False)
print("tile", tile, "chunks", plans.fwd.n_chunks, "flagged", int((plans.fwd.chunk_flags & 0xFF != 0).sum()), flush=True)
bias = torch.randn(64, device=dev)
F = _lib.FLAG_SPLIT_PRODUCERS
for name, plan, src, transpose in (("fwd", plans.fwd, x, False), ("dx", plans.bwd, dg, True)):
    pk = _lib.pack_weights(w, root, transpose)
    ps = _lib.plan_struct(plan)
    o0, o1 = torch.empty(n, 64, device=dev), torch.full((n, 64), float("nan"), device=dev)
    if name == "fwd":
        _lib.fwd(ps, src, 64, pk, bias, o0, 64, 0, 0)
        _lib.fwd(ps, src, 64, pk, bias, o1, 64, 0, F)
    else:
        _lib.bwd_dx(ps, src, 64, pk, o0, 64, None, 0)
        _lib.bwd_dx(ps, src, 64, pk, o1, 64, None, F)
    torch.cuda.synchronize()
    d = (o0 - o1).abs()
    print(f"{name}: max |fp32 - split| {d.max().item():.3e}  mean {d.mean().item():.3e}  max |ref| {o0.abs().max().item():.3f}  nan {int(torch.isnan(o1).sum())}", flush=True)

def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]
pk = _lib.pack_weights(w, root, False)
ps = _lib.plan_struct(plans.fwd)
out = torch.empty(n, 64, device=dev)
print("fwd exact fp32 (this tile)  %.3f ms" % t(lambda: _lib.fwd(ps, x, 64, pk, bias, out, 64, 0, 0)), flush=True)
print("fwd producer-split bf16x3   %.3f ms" % t(lambda: _lib.fwd(ps, x, 64, pk, bias, out, 64, 0, F)), flush=True)
