import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import rgcn_oracle as O
from scaling_rgcn_training_amd import _lib, plan as P
from scaling_rgcn_training_amd.conv import _rows16
dev = torch.device("cuda:0")
n, e, r, d = 5000, 90000, 7, 64
ei, et = O.synthetic_graph(n, e, r, seed=21)
g = torch.Generator().manual_seed(13)
x = torch.randn(n, d, generator=g).to(dev); dg = torch.randn(n, d, generator=g).to(dev)
plans = P.build_graph_plans(ei.to(dev), et.to(dev), n, r, 128, chunk=64)
res = {}
for mode in ("0", "2"):
    os.environ["RGCN_DW_DIRECT"] = mode
    dw = torch.full((r, d, d), float("nan"), device=dev); dr = torch.full((d, d), float("nan"), device=dev); db = torch.full((d,), float("nan"), device=dev)
    _lib.bwd_dw(_lib.plan_struct(plans.fwd), x, d, dg, d, dw, dr, db)
    torch.cuda.synchronize()
    res[mode] = (dw.cpu().numpy(), dr.cpu().numpy(), db.cpu().numpy())
a, b = res["0"][0], res["2"][0]
bad = ~np.isfinite(b) | (np.abs(a - b) > 1e-3 * (1 + np.abs(a)))
print("bad", bad.sum(), "of", bad.size, "n_units", plans.fwd.n_units)
for rr in range(r):
    idx = np.argwhere(bad[rr])
    if len(idx):
        print("rel", rr, "count", len(idx), "rows mod 4", sorted(set(idx[:, 0] % 4)), "cols mod 4", sorted(set(idx[:, 1] % 4)), "rows", sorted(set(idx[:,0]))[:20], "cols", sorted(set(idx[:,1]))[:20])
        print("  sample", b[rr][idx[0][0], idx[0][1]], a[rr][idx[0][0], idx[0][1]])
print("root bad", (~np.isfinite(res["2"][1]) | (np.abs(res["0"][1]-res["2"][1]) > 1e-3*(1+np.abs(res["0"][1])))).sum(), "bias bad", (np.abs(res["0"][2]-res["2"][2]) > 1e-3*(1+np.abs(res["0"][2]))).sum())
