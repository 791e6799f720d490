"""Can the weight gradients hide under the dX launch?  The tile-major dW kernel cannot share a CU with rgcn_tile3p_kernel
(152 + 160 KiB of LDS); the relation-major direct kernel uses no LDS.  Times, at the headline size: dX, dW (tile-major, bf16 x 3),
dW (direct, exact fp32) alone; dX then tile-major dW (what a step runs); dX with the direct kernel forked onto a second stream."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scaling_rgcn_training_amd import _lib, plan as P
import bench
n, e, r = 10_000_000, 100_000_000, 32
dev = torch.device("cuda:0")
ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, r, 64, 64, dev)
KF = _lib.FLAG_SPLIT_PRODUCERS
plans = P.build_graph_plans_device(ei, et, n, r, 224, chunk=128, dw_tiles=True)
del ei, et
pkt = _lib.pack_weights(w, root, True)
pst, psf, psd = _lib.plan_struct(plans.bwd), _lib.plan_struct(plans.fwd), _lib.plan_struct(plans.dw)
dx = torch.empty(n, 64, device=dev)
dw, dr, db = torch.empty_like(w), torch.empty_like(root), torch.empty(64, device=dev)
side = torch.cuda.Stream(device=dev)

def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]

run_dx = lambda: _lib.bwd_dx(pst, dg, 64, pkt, dx, 64, None, KF)
run_tiles = lambda: (_lib.bwd_dw_tiles(psd, plans.dw_walk, x, 64, dg, 64, dw, KF), _lib.bwd_dw_root(x, 64, dg, 64, dr, db))
run_direct = lambda: _lib.bwd_dw(psf, x, 64, dg, 64, dw, dr, db, _lib.FLAG_DW_DIRECT)

def forked(first_dw):
    cur = torch.cuda.current_stream(dev)
    side.wait_stream(cur)
    if first_dw:
        with torch.cuda.stream(side):
            run_direct()
        run_dx()
    else:
        run_dx()
        with torch.cuda.stream(side):
            run_direct()
    cur.wait_stream(side)

print("dX alone                         %.3f ms" % t(run_dx), flush=True)
print("dW tile-major + root alone       %.3f ms" % t(run_tiles), flush=True)
print("dW direct (relation-major) alone %.3f ms" % t(run_direct), flush=True)
print("dX then dW tile-major + root     %.3f ms" % t(lambda: (run_dx(), run_tiles())), flush=True)
print("dX then dW direct                %.3f ms" % t(lambda: (run_dx(), run_direct())), flush=True)
print("dX || dW direct (dW enqueued first) %.3f ms" % t(lambda: forked(True)), flush=True)
print("dX || dW direct (dX enqueued first) %.3f ms" % t(lambda: forked(False)), flush=True)
