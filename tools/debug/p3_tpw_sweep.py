"""Forward launch time of rgcn_tile3p_kernel against the tiles a workgroup walks (needs a library built with -DRGCN_TPW_ENV:
tools/debug/build_variant.sh tpwenv -DRGCN_TPW_ENV; RGCN_LIB=.../tpwenv.so).  Usage: p3_tpw_sweep.py 8 12 16 24 32"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scaling_rgcn_training_amd import _lib, plan as P
import bench
dev = torch.device("cuda:0")
def t(fn, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]
n, e, r = 10_000_000, 100_000_000, 32
ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, r, 64, 64, dev)
plans = P.build_graph_plans_device(ei, et, n, r, 224, chunk=128, dw_tiles=False)
del ei, et
pk = _lib.pack_weights(w, root, False)
out = torch.empty(n, 64, device=dev)
ps = _lib.plan_struct(plans.fwd)
print("tiles", plans.fwd.n_tiles, flush=True)
for tpw in sys.argv[1:]:
    os.environ["RGCN_TPW"] = tpw
    print("  tiles per workgroup %3s: %.3f ms" % (tpw, t(lambda: _lib.fwd(ps, x, 64, pk, None, out, 64, 0, _lib.FLAG_SPLIT_PRODUCERS))), flush=True)
