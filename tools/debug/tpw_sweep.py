"""Forward launch time against the tiles a workgroup walks (debug aid; needs a library built with -DRGCN_TPW_ENV)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scaling_rgcn_training_amd import _lib, plan as P
import bench
dev = torch.device("cuda:0")
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]
for n, e in ((1_000_000, 10_000_000), (10_000_000, 100_000_000)):
    r = 32
    ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, r, 64, 64, dev)
    tile, chunk = P.choose_layout(n, e, r, 64, 64)
    plans = P.build_graph_plans_device(ei, et, n, r, tile, chunk=chunk, dw_tiles=False)
    del ei, et
    packed = _lib.pack_weights(w, root, False)
    out = torch.empty(n, 64, device=dev)
    bias = torch.zeros(64, device=dev)
    ps = _lib.plan_struct(plans.fwd)
    print("nodes", n, "tiles", plans.fwd.n_tiles, flush=True)
    for tpw in sys.argv[1:]:
        os.environ["RGCN_TPW"] = tpw
        print("  tiles per workgroup %3s: %.3f ms" % (tpw, t(lambda: _lib.fwd(ps, x, 64, packed, bias, out, 64))), flush=True)
    del plans, x, dg, out
