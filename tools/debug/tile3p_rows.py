"""Debug: which rows of the producer-split kernel's forward output differ from the exact-fp32 kernel's (tile / workgroup position)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scaling_rgcn_training_amd import _lib, plan as P
import bench
n, e, r = int(sys.argv[1]), int(sys.argv[2]), 32
dev = torch.device("cuda:0")
ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, r, 64, 64, dev)
if len(sys.argv) > 3:
    m = torch.rand(e, device=dev) < 0.1
    ei[1, m] = torch.randint(0, 16, (int(m.sum()),), device=dev)
plans = P.build_graph_plans_device(ei, et, n, r, 224, chunk=128, dw_tiles=False)
bias = torch.randn(64, device=dev)
which = os.environ.get("WHICH", "fwd")
pk = _lib.pack_weights(w, root, which == "dx")
plan = plans.fwd if which == "fwd" else plans.bwd
ps = _lib.plan_struct(plan)
o0, o1 = torch.empty(n, 64, device=dev), torch.full((n, 64), float("nan"), device=dev)
if which == "fwd":
    _lib.fwd(ps, x, 64, pk, bias, o0, 64, 0, 0)
    _lib.fwd(ps, x, 64, pk, bias, o1, 64, 0, _lib.FLAG_SPLIT_PRODUCERS)
else:
    _lib.bwd_dx(ps, dg, 64, pk, o0, 64, None, 0)
    _lib.bwd_dx(ps, dg, 64, pk, o1, 64, None, _lib.FLAG_SPLIT_PRODUCERS)
torch.cuda.synchronize()
tp = plan.tile_ptr.cpu()
print(which, "chunks per tile min/max", int((tp[1:] - tp[:-1]).min()), int((tp[1:] - tp[:-1]).max()), "flagged", int((plan.chunk_flags & 0xFF != 0).sum()))
d = (o0 - o1).abs()
bad = ((d > 1e-4) | torch.isnan(d)).any(1).nonzero().flatten()
print("tiles", plan.n_tiles, "bad rows", bad.numel())
if bad.numel():
    t = bad // 224
    ut = torch.unique(t)
    print("bad tiles", ut[:40].tolist(), "... tile %% 16:", torch.unique(ut % 16).tolist())
    print("rows within tile:", torch.unique(bad % 224)[:40].tolist())
    rr = bad[0].item()
    print("row", rr, o0[rr, :6].tolist(), o1[rr, :6].tolist())
