"""Per-phase cycle shares of rgcn_dw_tile_kernel<true> (stamp build of csrc/rgcn_dw_tile.hip only, linked against the product
objects).  Usage: dw_stamps.py [N E]      DW_DEFS="-DRGCN_DW_PIPE=0 ..." adds build flags."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
B = os.path.join(ROOT, "scaling_rgcn_training_amd", "_build")
so = os.path.join(ROOT, "gpurun_out", "librgcn_dwstamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
H = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
DEFS = os.environ.get("DW_DEFS", "").split()
subprocess.run(H + ["-DRGCN_DW_STAMPS"] + DEFS + ["-c", os.path.join(ROOT, "scaling_rgcn_training_amd/csrc/rgcn_dw_tile.hip"), "-o", so + ".o"], check=True)
subprocess.run(H + ["-shared"] + [os.path.join(B, f"rgcn_{n}.o") for n in ("tile_fp32", "tile_fp32_narrow", "tile_fp32_wide", "tile3p", "dw_relmajor", "dw_root", "ep", "abi", "plan")] + [so + ".o", "-o", so], check=True)
from scaling_rgcn_training_amd import _lib
_lib.LIB_PATH = so
lib = _lib.load()
from scaling_rgcn_training_amd import plan as P
import bench
n, e = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10_000_000, 100_000_000)
dev = torch.device("cuda:0")
ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, 32, 64, 64, dev)
tile, chunk = P.choose_layout(n, e, 32, 64, 64)
plans = P.build_graph_plans_device(ei, et, n, 32, tile, chunk=chunk, dw_tiles=True)
stamps = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
lib.rgcn_debug_set_dw_stamps.argtypes = [ctypes.c_void_p]
assert lib.rgcn_debug_set_dw_stamps(stamps.data_ptr()) == 0
dw = torch.empty_like(w)
psd = _lib.plan_struct(plans.dw)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(3):
    stamps.zero_()
    ev[0].record()
    _lib.bwd_dw_tiles(psd, plans.dw_walk, x, 64, dg, 64, dw, _lib.FLAG_SPLIT_PRODUCERS)
    ev[1].record()
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(256, 8, 16).astype(np.float64)
units = s[:, :, 11].sum()
names = ["loop top", "issue h1 rows", "h0: LDS g/w", "h0: rows + A cut", "h0: B cut 0", "h0: MFMA block", "idx + issue next h0 rows",
         "h1: LDS g/w", "h1: rows + A cut", "h1: B cut 0", "h1: MFMA block", None, "tile: loop", "tile: DMA wait", "tile: barrier", "tile: DMA issue"]
print("defs", DEFS, "launch with stamps %.3f ms; units walked %d (%.1f per wave)" % (ev[0].elapsed_time(ev[1]), units, units / 2048))
tot = 0.0
per_wave_total = np.delete(s, 11, axis=2).sum(2)
print("cycles per wave, whole launch: mean %.0f  min %.0f  max %.0f (100 MHz-independent: s_memtime ticks)" % (per_wave_total.mean(), per_wave_total.min(), per_wave_total.max()))
for i, nm in enumerate(names):
    if nm is None:
        continue
    v = s[:, :, i].sum() / units
    tot += v
    print("  %-28s %8.1f ticks per unit" % (nm, v))
print("  %-28s %8.1f" % ("sum", tot))
