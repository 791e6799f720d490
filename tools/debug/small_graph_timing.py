"""AIFB-shaped layer timing (N=8243, E=49838, R'=89; 63->16 and 16->4) through the drop-in module."""
import sys, time, torch
sys.path.insert(0, '.')
from oracle import rgcn_oracle as O
from scaling_rgcn_training_amd.conv import RGCNConv
dev = torch.device('cuda:0')
n, e, r = 8243, 49838, 89
ei, et = O.synthetic_graph(n, e, r, seed=0, skew=True)
ei, et = ei.to(dev), et.to(dev)
for din, dout in ((63, 16), (16, 4)):
    conv = RGCNConv(din, dout, r).to(dev)
    x = torch.randn(n, din, device=dev, requires_grad=True)
    dg = torch.randn(n, dout, device=dev)
    for _ in range(5):
        conv(x, ei, et).backward(dg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        x.grad = None
        conv(x, ei, et).backward(dg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{din}->{dout}: {dt*1e3:.3f} ms per fwd+bwd  ({e/dt:.3g} edges/s)")
