"""Error of the tile-major d_weight kernel in its two forms (exact fp32 MFMA / bf16 x 3 split of both operands) against
float64, as a function of the number of edges per relation: separates a per-product error (ratio independent of n) from an
accumulation effect (ratio growing with n).  VERDICT r2 item 5a.   python tools/debug/dw_split_error_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from scaling_rgcn_training_amd import _lib, plan as P

dev = torch.device("cuda:0")
R = 4
print("# edges/relation   max|d_w|   err exact   err split   split/exact   rms err exact   rms err split   (relation 0; float64 reference)")
for n, e in ((20_000, 40_000), (100_000, 400_000), (500_000, 4_000_000), (2_000_000, 16_000_000), (5_000_000, 48_000_000)):
    g = torch.Generator(device=dev).manual_seed(1)
    src = torch.randint(0, n, (e,), generator=g, device=dev)
    dst = torch.randint(0, n, (e,), generator=g, device=dev)
    typ = torch.randint(0, R, (e,), generator=g, device=dev)
    x = torch.randn(n, 64, generator=g, device=dev)
    dg = torch.randn(n, 64, generator=g, device=dev)
    ei = torch.stack([src, dst])
    plans = P.build_graph_plans_device(ei, typ, n, R, 224, chunk=128, dw_tiles=True)
    psd = _lib.plan_struct(plans.dw)
    dw_s, dw_e = torch.empty(R, 64, 64, device=dev), torch.empty(R, 64, 64, device=dev)
    _lib.bwd_dw_tiles(psd, plans.dw_walk, x, 64, dg, 64, dw_s, _lib.FLAG_SPLIT_PRODUCERS)
    _lib.bwd_dw_tiles(psd, plans.dw_walk, x, 64, dg, 64, dw_e, 0)
    cnt = torch.bincount(dst * R + typ, minlength=n * R)
    idx = torch.nonzero(typ == 0).squeeze(1)
    s, d = src[idx], dst[idx]
    we = 1.0 / cnt[d * R].double()
    ref = torch.zeros(64, 64, dtype=torch.float64, device=dev)
    for lo in range(0, idx.numel(), 1 << 22):
        ref += (x[s[lo:lo + (1 << 22)]].double() * we[lo:lo + (1 << 22), None]).T @ dg[d[lo:lo + (1 << 22)]].double()
    ee, es = (dw_e[0].double() - ref).abs(), (dw_s[0].double() - ref).abs()
    print(f"{idx.numel():12d}   {float(ref.abs().max()):9.2f}   {float(ee.max()):.3e}   {float(es.max()):.3e}   {float(es.max() / ee.max()):6.2f}"
          f"   {float(ee.pow(2).mean().sqrt()):.3e}   {float(es.pow(2).mean().sqrt()):.3e}   mean signed err exact {float((dw_e[0].double() - ref).mean()):+.2e} split {float((dw_s[0].double() - ref).mean()):+.2e}", flush=True)
    del plans, psd
    P.clear_plan_cache()
