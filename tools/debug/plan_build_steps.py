"""Step-by-step run of the device plan builder with a synchronise after every C-ABI call (debug aid)."""
import ctypes as C, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scaling_rgcn_training_amd import _lib, plan as P
from oracle import rgcn_oracle as O
n, e, r, tile, chunk = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (300, 4000, 5, 16, 64)
dev = torch.device("cuda:0")
ei, et = O.synthetic_graph(n, e, r, seed=1)
ei, et = ei.to(dev), et.to(dev)
lib = _lib.load()
g, keep = _lib.graph_struct(ei, et, n, r)
ws = _lib.plan_workspace(e, n, r, tile, dev)
print("workspace", ws.numel(), flush=True)
w = _lib.edge_weights(g, "mean", ws)
torch.cuda.synchronize()
print("edge_weights done", flush=True)
wr = P.edge_weights(ei[0], ei[1], et, r)
print("weights equal:", torch.equal(w, wr), float((w - wr).abs().max()), flush=True)
for tr in (False, True):
    sizes = _lib.RgcnPlanSizes()
    st = lib.rgcn_plan_build_begin(C.byref(g), w.data_ptr(), int(tr), 0, n, tile, chunk, ws.data_ptr(), ws.numel(), C.byref(sizes),
                                   torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("begin", tr, st, sizes.n_tiles, sizes.n_chunks, sizes.n_units, sizes.n_slots, sizes.n_edges, flush=True)
    ref = P.build_plan(ei[1] if tr else ei[0], ei[0] if tr else ei[1], et, wr, n, r, tile, chunk=chunk)
    print("ref  ", ref.n_tiles, ref.n_chunks, ref.n_units, ref.n_chunks * chunk, ref.n_edges, flush=True)
    ps, arr, ne = _lib.plan_build(g, w, tr, 0, n, tile, chunk, ws)
    torch.cuda.synchronize()
    print("finish done", flush=True)
    for k, v in arr.items():
        rv = getattr(ref, k)
        ok = v.shape == rv.shape and torch.equal(v, rv)
        print(f"  {k:12s} {'ok' if ok else 'DIFF'} {tuple(v.shape)} {tuple(rv.shape)}", flush=True)
        if not ok and v.shape == rv.shape:
            bad = torch.nonzero(v != rv).flatten()
            print("     first diffs at", bad[:8].tolist(), v[bad[:8]].tolist(), rv[bad[:8]].tolist(), flush=True)
