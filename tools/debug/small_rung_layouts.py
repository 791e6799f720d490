"""The 100k-node / 1M-edge rung (32 relations, 64 -> 64) under other plan layouts: what the layer picks (cost model of the
exact-fp32 kernel: chunk 64) against forced (tile, 128-slot chunk) pairs that put forward / dX on rgcn_tile3p_kernel, with and
without the tile-major d_weight kernel.  Median HIP-event ms of an eager step and of a hipGraph replay.
    python tools/debug/small_rung_layouts.py [nodes edges]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from scaling_rgcn_training_amd import conv as C
from scaling_rgcn_training_amd.conv import RGCNConv
from scaling_rgcn_training_amd.plan import clear_plan_cache

n, e = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100_000, 1_000_000)
r, d = 32, 64
dev = torch.device("cuda:0")
ei, et, x, dg, weight, root = bench.synthetic_on_device(n, e, r, d, d, dev, seed=1)


def run(label, layout=None, dw_min=None, merge=True):
    old = C.DW_TILES_MIN_EDGES
    if dw_min is not None:
        C.DW_TILES_MIN_EDGES = dw_min
    conv = RGCNConv(d, d, r).to(dev)
    conv.path = "ring"
    conv.merge_runs = merge
    with torch.no_grad():
        conv.weight.copy_(weight)
        conv.root.copy_(root)
    if layout is not None:
        conv.layout = lambda nn, ee: layout
    xx = x.clone().requires_grad_(True)

    def step():
        xx.grad = None
        conv.zero_grad(set_to_none=True)
        conv(xx, ei, et).backward(dg)
    for _ in range(5):
        step()
    plans = conv._plans(xx, ei, et)
    evs = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); step(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    eager = statistics.median(a.elapsed_time(b) for a, b in evs)
    g = torch.cuda.CUDAGraph()
    xx.grad = None
    conv.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        conv(xx, ei, et).backward(dg)
    evs = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    rep = statistics.median(a.elapsed_time(b) for a, b in evs)
    print(f"{label:58s} tile {plans.fwd.tile:4d} chunk {plans.fwd.chunk:4d} layout {plans.fwd.layout} tiles {plans.fwd.n_tiles:5d} chunks {plans.fwd.n_chunks:7d} "
          f"dw plan {'yes' if plans.dw is not None else 'no '}  eager {eager:.3f} ms  replay {rep:.3f} ms", flush=True)
    del g, plans, conv
    clear_plan_cache()
    C.DW_TILES_MIN_EDGES = old


run("what the layer picks")
for t in (224, 208, 196, 176, 160, 128, 112):
    t = t // 16 * 16
    run(f"forced ({t}, 128)", (t, 128))
    run(f"forced ({t}, 128) + tile-major d_weight (layout 3)", (t, 128), dw_min=1)
