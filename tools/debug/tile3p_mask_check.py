"""Debug: rgcn_bwd_dx with the ReLU mask / rgcn_fwd with a fused activation, producer-split kernel against the exact-fp32 one."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rgcn_oracle as O
from scaling_rgcn_training_amd import _lib, plan as P
from scaling_rgcn_training_amd.conv import _rows16
dev = torch.device("cuda:0")
n, e, r = 6000, 200000, 5
ei, et = O.synthetic_graph(n, e, r, seed=4)
g = torch.Generator().manual_seed(3)
x = torch.relu(torch.randn(n, 63, generator=g))
dg = torch.randn(n, 64, generator=g)
w = torch.randn(r, 63, 64, generator=g) * 0.1
root = torch.randn(63, 64, generator=g) * 0.1
plans = P.build_graph_plans_device(ei.to(dev), et.to(dev), n, r, 224, chunk=128, dw_tiles=False)
xd, gd = _rows16(x.to(dev), 63), dg.to(dev)
wd, rd = w.to(dev), root.to(dev)
F = _lib.FLAG_SPLIT_PRODUCERS
pkt = _lib.pack_weights(wd, rd, True)
pst = _lib.plan_struct(plans.bwd)
for mask in (None, xd):
    o0, o1 = torch.zeros(n, 64, device=dev), torch.zeros(n, 64, device=dev)
    _lib.bwd_dx(pst, gd, 64, pkt, o0, 63, mask, 0)
    _lib.bwd_dx(pst, gd, 64, pkt, o1, 63, mask, F)
    torch.cuda.synchronize()
    d = (o0 - o1).abs()
    bad = (d > 1e-4).any(1).nonzero().flatten()
    print("dx mask" if mask is not None else "dx", "max diff %.3e" % d.max().item(), "bad rows", bad.numel(), bad[:20].tolist())
    if bad.numel():
        rr = bad[0].item()
        print(" row", rr, "fp32", o0[rr, :8].tolist(), "split", o1[rr, :8].tolist())
pk = _lib.pack_weights(wd, rd, False)
ps = _lib.plan_struct(plans.fwd)
for act in (0, 1, 2):
    o0, o1 = torch.zeros(n, 64, device=dev), torch.zeros(n, 64, device=dev)
    _lib.fwd(ps, xd, 63, pk, None, o0, 64, act, 0)
    _lib.fwd(ps, xd, 63, pk, None, o1, 64, act, F)
    torch.cuda.synchronize()
    d = (o0 - o1).abs()
    print("fwd act", act, "max diff %.3e" % d.max().item(), "bad rows", int((d > 1e-4).any(1).sum()))
