"""AIFB-shaped layer (8,243 nodes, 49,838 edges, 89 relations, 63 -> 16), fwd + bwd through the drop-in module: host time
per step against GPU time per step (debug aid; run under `rocprofv3 --kernel-trace --stats` for the launch list)."""
import os, sys, time, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from scaling_rgcn_training_amd.conv import RGCNConv
dev = torch.device("cuda:0")
n, e, r, din, dout = 8_243, 49_838, 89, 63, 16
if len(sys.argv) > 1:
    n, e, r, din, dout = (int(v) for v in sys.argv[1:6])
ei, et, x, dg, weight, root = bench.synthetic_on_device(n, e, r, din, dout, dev, seed=1)
conv = RGCNConv(din, dout, r).to(dev)
x.requires_grad_(True)
conv._plans(x, ei, et)
def step():
    x.grad = None
    conv.zero_grad(set_to_none=True)
    conv(x, ei, et).backward(dg)
for _ in range(20):
    step()
torch.cuda.synchronize()
steps = 200
t0 = time.perf_counter()
evs = []
for _ in range(steps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); step(); b.record()
    evs.append((a, b))
t_submit = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host submit %.3f ms/step, wall %.3f ms/step, GPU events median %.3f ms/step" %
      (1e3 * t_submit / steps, 1e3 * t_all / steps, statistics.median(a.elapsed_time(b) for a, b in evs)))
