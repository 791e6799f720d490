"""Per-phase cycle shares of rgcn_tile3p_kernel's producer / consumer loops (stamp build of csrc/rgcn_tile3p.hip only,
linked against the product objects).  Usage: p3_stamps.py [N E]"""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
B = os.path.join(ROOT, "scaling_rgcn_training_amd", "_build")
so = os.path.join(ROOT, "gpurun_out", "librgcn_p3stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
H = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
DEFS = os.environ.get("P3_DEFS", "").split()
subprocess.run(H + ["-DRGCN_P3_STAMPS"] + DEFS + ["-c", os.path.join(ROOT, "scaling_rgcn_training_amd/csrc/rgcn_tile3p.hip"), "-o", so + ".o"], check=True)
subprocess.run(H + ["-shared"] + [os.path.join(B, f"rgcn_{n}.o") for n in ("tile_fp32", "tile_fp32_narrow", "tile_fp32_wide", "dw_relmajor", "dw_tile", "dw_root", "ep", "abi", "plan")] + [so + ".o", "-o", so], check=True)
from scaling_rgcn_training_amd import _lib
_lib.LIB_PATH = so
lib = _lib.load()
from scaling_rgcn_training_amd import plan as P
import bench
n, e = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10_000_000, 100_000_000)
dev = torch.device("cuda:0")
ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, 32, 64, 64, dev)
if os.environ.get("VT_UNIQ") == "1":
    key = torch.randperm(n * 32, device=dev)[:e]
    ei = torch.stack([ei[0], key // 32]); et = key % 32
plans = P.build_graph_plans_device(ei, et, n, 32, 224, chunk=128, dw_tiles=False, split=os.environ.get("VT_SPLIT", "0") == "1")
fp = plans.fwd
stamps = torch.zeros(fp.n_tiles * 12 * 8, dtype=torch.int64, device=dev)
lib.rgcn_debug_set_p3_stamps.argtypes = [ctypes.c_void_p]
assert lib.rgcn_debug_set_p3_stamps(stamps.data_ptr()) == 0
out = torch.empty(n, 64, device=dev)
pk = _lib.pack_weights(w, root, False)
for _ in range(2):
    stamps.zero_()
    _lib.fwd(_lib.plan_struct(fp), x, 64, pk, None, out, 64, 0, _lib.FLAG_SPLIT_PRODUCERS)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 12, 8).astype(np.float64)
nch = s[:, 0, 4].sum()
print("defs", DEFS, "layout", fp.layout, "tiles", fp.n_tiles, "chunks", int(nch))
print("cycles per chunk, every wave of the workgroup (producers: wait batch / issue batch / split + store / barrier; consumers: metadata / compute / W swap / barrier)")
for wv in range(12):
    if s[:, wv, 4].sum() == 0:
        continue
    role = "producer %d" % wv if wv < 4 else "consumer %d" % (wv - 4)
    v = s[:, wv, :4].sum(0) / nch
    print(f"{role:12s} {v[0]:8.1f} {v[1]:8.1f} {v[2]:8.1f} {v[3]:8.1f}   total {v.sum():8.1f}")
