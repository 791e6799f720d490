"""Probe: does the dW kernel speed up when the G rows it gathers come from a band that fits the 256 MiB Infinity
Cache while the H rows stream from all of HBM?  Builds librgcn_probe.so with RGCN_NT_H = $RGCN_NT_H."""
import os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
nt = os.environ.get("RGCN_NT_H", "0")
so = os.path.join(ROOT, "gpurun_out", f"librgcn_probe_nt{nt}.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-DRGCN_NT_H={nt}",
                os.path.join(ROOT, "scaling_rgcn_training_amd/csrc/rgcn_kernels.hip"), "-o", so], check=True)
from scaling_rgcn_training_amd import _lib
_lib.LIB_PATH = so
_lib.load()
from scaling_rgcn_training_amd import plan as P
dev = torch.device("cuda:0")
N, R, D = 10_000_000, 32, 64
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, D, device=dev, generator=g)
dg = torch.randn(N, D, device=dev, generator=g)
w = torch.randn(R, D, D, device=dev) * 0.1
root = torch.randn(D, D, device=dev) * 0.1
for band in (N, 1_000_000, 300_000):
    e = int(0.3125 * R * band)            # same edges per (dst, relation) as the headline graph
    src = torch.randint(0, N, (e,), device=dev, generator=g)
    dst = torch.randint(0, band, (e,), device=dev, generator=g)
    typ = torch.randint(0, R, (e,), device=dev, generator=g)
    wts = P.edge_weights(src, dst, typ, R)
    plan = P.build_plan(src, dst, typ, wts, N, R, 384, 0, (band + 383) // 384 * 384 if band < N else N)
    ps = _lib.plan_struct(plan)
    dw, dr, db = torch.empty_like(w), torch.empty_like(root), torch.empty(D, device=dev)
    gown = dg[:plan.n_owned]
    for _ in range(2):
        _lib.bwd_dw(ps, x, D, gown, D, dw, dr, db)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        _lib.bwd_dw(ps, x, D, gown, D, dw, dr, db)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    rows = e + plan.n_owned
    print(f"nt={nt} band {band:>9d} nodes ({band * 256 / 2**20:7.0f} MiB of G rows): {dt*1e3:7.3f} ms, {dt / rows * 1e9:.3f} ns per row")
