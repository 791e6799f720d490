"""Does rgcn_bwd_dw_root hide under the dX launch?  Times, at the headline size: the root kernel alone, dX alone, and both
forked onto two streams and joined (what conv.py's backward enqueues)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scaling_rgcn_training_amd import _lib, plan as P
import bench
n, e, r = 10_000_000, 100_000_000, 32
dev = torch.device("cuda:0")
ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, r, 64, 64, dev)
tile, chunk = P.choose_layout(n, e, r, 64, 64)
KF = int(os.environ.get("VT_FLAGS", 0))
if KF & _lib.FLAG_SPLIT_PRODUCERS:
    tile = min(tile, 224)
plans = P.build_graph_plans_device(ei, et, n, r, tile, chunk=chunk, dw_tiles=False)
del ei, et
pkt = _lib.pack_weights(w, root, True)
pst = _lib.plan_struct(plans.bwd)
dx = torch.empty(n, 64, device=dev)
dr, db = torch.empty_like(root), torch.empty(64, device=dev)
side = torch.cuda.Stream(device=dev)
prio = torch.cuda.Stream(device=dev, priority=-1)

def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]

def run_dx():
    _lib.bwd_dx(pst, dg, 64, pkt, dx, 64, None, KF)

def run_root():
    _lib.bwd_dw_root(x, 64, dg, 64, dr, db)

def both(stream, root_first=True):
    cur = torch.cuda.current_stream(dev)
    stream.wait_stream(cur)
    if root_first:
        with torch.cuda.stream(stream):
            run_root()
        run_dx()
    else:
        run_dx()
        with torch.cuda.stream(stream):
            run_root()
    cur.wait_stream(stream)

print("root alone            %.3f ms" % t(run_root), flush=True)
print("dX alone              %.3f ms" % t(run_dx), flush=True)
print("sequential            %.3f ms" % t(lambda: (run_dx(), run_root())), flush=True)
print("forked, root first    %.3f ms" % t(lambda: both(side)), flush=True)
print("forked, dX first      %.3f ms" % t(lambda: both(side, False)), flush=True)
print("forked (high-priority side stream), root first %.3f ms" % t(lambda: both(prio)), flush=True)
print("forked (high-priority side stream), dX first   %.3f ms" % t(lambda: both(prio, False)), flush=True)
