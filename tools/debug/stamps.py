"""Diagnostic: per-segment cycle shares of the tile kernel's consumer / producer loops (stamp build).
Builds librgcn_stamps.so with -DRGCN_STAMPS and runs one forward launch at the given size."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "gpurun_out", "librgcn_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
abl = os.environ.get("RGCN_ABL", "0")
CSRC = os.path.join(ROOT, "scaling_rgcn_training_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DRGCN_STAMPS", "-DRGCN_ABL=" + abl] +
               [os.path.join(CSRC, f + ".hip") for f in ("rgcn_tile_fp32", "rgcn_tile_fp32_narrow", "rgcn_tile_fp32_wide", "rgcn_tile3p", "rgcn_dw_relmajor", "rgcn_dw_tile", "rgcn_dw_root",
                                                          "rgcn_ep", "rgcn_abi", "rgcn_plan")] + ["-o", so], check=True)
from scaling_rgcn_training_amd import _lib
_lib.LIB_PATH = so
lib = _lib.load()
from scaling_rgcn_training_amd import plan as P
import bench
n, e = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, 32, 64, 64, dev)
_t, _c = P.choose_layout(n, e, 32, 64, 64)
tile = int(os.environ.get("RGCN_TILE", _t))
chunk = int(os.environ.get('RGCN_CHUNK', _c))
split = os.environ.get('RGCN_SPLIT', '0') == '1'
plans = P.build_graph_plans(ei, et, n, 32, tile, chunk=chunk, split=split)
print("tile", tile)
fp = plans.fwd
stamps = torch.zeros(max(fp.n_tiles, 1024) * 32, dtype=torch.int64, device=dev)
lib.rgcn_debug_set_stamps.argtypes = [ctypes.c_void_p]
assert lib.rgcn_debug_set_stamps(stamps.data_ptr()) == 0
out = torch.empty(n, 64, device=dev)
pk = _lib.pack_weights(w, root, False)
which = os.environ.get("RGCN_WHICH", "fwd")
if which == "fwd":
    for _ in range(2):
        _lib.fwd(_lib.plan_struct(fp), x, 64, pk, None, out, 64)
else:
    dwt, drt, dbt = torch.empty_like(w), torch.empty_like(root), torch.empty(64, device=dev)
    for _ in range(2):
        _lib.bwd_dw(_lib.plan_struct(fp), x, 64, dg, 64, dwt, drt, dbt)
torch.cuda.synchronize()
print("kernel:", which)
s = stamps.cpu().numpy().reshape(-1, 32).astype(np.float64)
if which == 'fwd':
    s = s[:fp.n_tiles]
nch = s[:, 7]
tot_c = s[:, 0:4].sum(1); tot_p = s[:, 4:7].sum(1)
print("ABL", abl, "tiles", len(s), "chunks/tile mean", nch.mean())
names = ["cons scalar-loads", "cons compute", "cons B-wait", "cons barrier", "prod issue", "prod dma-wait", "prod barrier"]
for i, nm in enumerate(names):
    print(f"{nm:20s} {s[:, i].sum() / nch.sum():9.1f} cycles/chunk")
print(f"consumer loop total {tot_c.sum() / nch.sum():9.1f} cycles/chunk; producer wave0 loop total {tot_p.sum() / nch.sum():9.1f}")
if split:
    print("pair 1 (wave 6): scalar %.1f compute %.1f B-wait %.1f barrier %.1f cycles/chunk" % tuple(s[:, 8 + i].sum() / nch.sum() for i in range(4)))
if split:
    print("wave 4 fast path per chunk: prologue (first loads + split) %.1f | per-chunk sums over its tiles: top (acc addr, reads issued) %.1f  mid (split + MFMA) %.1f  accumulate (wait, FMA, store) %.1f" % tuple(s[:, 16 + i].sum() / nch.sum() for i in range(4)))
if which == "fwd" and not split:
    for i in range(8):
        c = s[:, 16 + i].sum()
        print(f"chunks with {i + 1} row tiles: {int(c):9d}  compute {s[:, 8 + i].sum() / max(c, 1):8.1f} cycles/chunk")
