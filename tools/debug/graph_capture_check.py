import os, sys, time, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from scaling_rgcn_training_amd.conv import RGCNConv
dev = torch.device("cuda:0")
n, e, r, din, dout = 8_243, 49_838, 89, 63, 16
ei, et, x, dg, weight, root = bench.synthetic_on_device(n, e, r, din, dout, dev, seed=1)
conv = RGCNConv(din, dout, r).to(dev)
x.requires_grad_(True)
def step():
    x.grad = None
    conv.zero_grad(set_to_none=True)
    out = conv(x, ei, et)
    out.backward(dg)
    return out
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
ref_out = step().detach().clone(); ref_gx = x.grad.clone(); ref_gw = conv.weight.grad.clone()
g = torch.cuda.CUDAGraph()
x.grad = None; conv.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    out = conv(x, ei, et)
    out.backward(dg)
g.replay(); torch.cuda.synchronize()
print("replay equals eager:", torch.equal(out, ref_out), torch.equal(x.grad, ref_gx), torch.equal(conv.weight.grad, ref_gw))
evs = []
for _ in range(200):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); evs.append((a, b))
torch.cuda.synchronize()
print("graph replay median %.3f ms/step" % statistics.median(a.elapsed_time(b) for a, b in evs))
