"""Experiment (DESIGN 'dW bytes'): does the ORDER of the weight-gradient walk let the caches serve the gathered
upstream-gradient rows?  Times rgcn_bwd_dw at the headline size for walk orders sorted | rr | phase (RGCN_WALK, read
at import) with the default and the nt cache policy on the x gathers (two builds).  Usage, on the GPU box:
    RGCN_WALK=phase python tools/debug/dw_walk_experiment.py [nt]"""
import os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
nt = len(sys.argv) > 1 and sys.argv[1] == "nt"
from scaling_rgcn_training_amd import _lib
if nt:
    so = os.path.join(ROOT, "gpurun_out", "librgcn_dwnt.so")
    if not os.path.exists(so):
        csrc = os.path.join(ROOT, "scaling_rgcn_training_amd", "csrc")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DRGCN_DW_X_AUX=2"] +
                       [os.path.join(csrc, f + ".hip") for f in ("rgcn_tile_fp32", "rgcn_tile_fp32_narrow", "rgcn_tile_fp32_wide", "rgcn_tile3p", "rgcn_dw_relmajor", "rgcn_dw_tile",
                                                                  "rgcn_dw_root", "rgcn_ep", "rgcn_abi", "rgcn_plan")] + ["-o", so], check=True)
    _lib.LIB_PATH = so
_lib.load()
from scaling_rgcn_training_amd import plan as P
import bench
n, e = 10_000_000, 100_000_000
dev = torch.device("cuda:0")
ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, 32, 64, 64, dev)
tile, chunk = P.choose_layout(n, e, 32, 64, 64)
w_e = P.edge_weights(ei[0], ei[1], et, 32)
fp = P.build_plan(ei[0], ei[1], et, w_e, n, 32, tile, chunk=chunk)
del ei, et, w_e
gmod = int(os.environ.get("RGCN_EXP_GMOD", "0"))
if gmod:   # every upstream-gradient gather hits one of `gmod` rows: the g side is served by L2 whatever the order
    fp.slot_row = torch.where(fp.slot_row < fp.n_owned, fp.slot_row % gmod, fp.slot_row)
xmod = int(os.environ.get("RGCN_EXP_XMOD", "0"))
if xmod:
    fp.slot_src = torch.where(fp.slot_src < fp.n_nodes, fp.slot_src % xmod, fp.slot_src)
dwt, drt, dbt = torch.empty_like(w), torch.empty_like(root), torch.empty(64, device=dev)
ps = _lib.plan_struct(fp)
for _ in range(3):
    _lib.bwd_dw(ps, x, 64, dg, 64, dwt, drt, dbt)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in evs:
    a.record(); _lib.bwd_dw(ps, x, 64, dg, 64, dwt, drt, dbt); b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in evs)
print(f"walk={P._WALK_MODE} nt={nt} gmod={gmod} xmod={xmod}: dW median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}  (units {fp.n_units})", flush=True)
