#!/usr/bin/env python3
"""bench.py -- edges/s per R-GCN layer (forward + backward) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nodes .. --edges .. --relations .. --width ..]

Workload (BASELINE.json configs[3], the configuration the metric is quoted on; fits one GPU):
synthetic graph, 10M nodes / 100M edges / 32 relations, features 64 -> 64, fp32, inputs of
SURVEY.md 8d (uniform src/dst/type, x ~ N(0,1), dOut ~ N(0,1)), generated on the device.

A "step" is ONE pass of the hot path over the whole graph: RGCNConv.forward (weight pack + tile
kernel) and its autograd backward (W^T pack + dX tile kernel on the transposed plan + dW kernel +
slab reduction), called through the drop-in nn.Module exactly as reference model/layers.py does.
The graph plan (device-side sort / chunk layout, rgcn_plan_build_*) is built once before the timed region and
reported separately (plan_build_s), as SURVEY.md 8d prescribes.  Inputs are resident in HBM when the clock starts.

Timing: W untimed warm-up steps, then exactly K steps between barrier + synchronize pairs (value = E * K / that
time, max over ranks); every timed step is also bracketed by HIP events on the launch stream, whose MEDIAN is
reported beside it (20 warm-up + 50 timed iterations, median -- the defaults).

N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL): the SAME graph is
edge-partitioned by destination range (strong scaling); every step includes the per-layer
all-gathers and the weight-gradient all-reduce; `comm` reports what they moved and how long the step waited on them.

The JSON line also carries
  roofline       the dominant kernel (HIP-event timed per launch; algorithmic bytes and flops of SURVEY.md 8d /
                 DESIGN.md; the binding roof is the one that needs more time at its peak -- HBM for the default
                 forward / dX kernel, whose contraction runs on bf16 MFMAs over 3-way split fp32 operands; the fp32
                 MFMA peak for the exact-fp32 kernel (--no-split-producers), with the HBM figures beside it)
  alt_forward_kernel  the same step with the OTHER kernels (exact-fp32 forward / dX / d_weight when the bf16x3 split forms ran,
                 and vice versa), timed on the same box right after the main run, + the element-wise differences of their results
  roofline_step  the whole step against HBM: algorithmic bytes of fwd + bwd (92.1 GB at the headline config) / ms_per_step
                 -- the number north_star's ">= 40 % of HBM roofline" refers to
  ladder         GPU edges/s on the smaller rungs of SURVEY.md 8d ((100k, 1M), (1M, 10M)) and on the shapes of the reference's
                 datasets (AIFB, MUTAG, AM-like with basis decomposition), each with its plan's slot fill
  cpu_baseline   the oracle's PyG-loop restatement timed on this host's cores on the (1M, 10M) rung, beside the GPU
                 figure of the SAME rung (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32-input MFMA (v_mfma_f32_16x16x4_f32), 155 measured
# HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE, then WRITE_SIZE; gfx950 correction): counters
# cannot be read inside this process, so `roofline.traffic` quotes the newest committed pass and names it
PMC_TRAFFIC_FILE = os.environ.get("RGCN_PMC_TRAFFIC_FILE", "r04x_pmc_traffic.json")
HEADLINE = (10_000_000, 100_000_000, 32, 64)
# (name, nodes, edges, relations R', in, out, num_bases): the smaller rungs of SURVEY.md 8d and the shapes of the reference's
# own datasets (model/modelTrainer.py:78,92: R' = 2R + 1 = 89 / 45 / ~267; MUTAG's edge count is a guess -- the file is not
# shipped; AM: BASELINE.json configs[2], hidden 32, basis decomposition B = 30)
LADDER = (("100k/1M", 100_000, 1_000_000, 32, 64, 64, None), ("1M/10M", 1_000_000, 10_000_000, 32, 64, 64, None),
          ("AIFB shape 63->16", 8_243, 49_838, 89, 63, 16, None), ("MUTAG shape 63->16", 23_644, 148_000, 45, 63, 16, None),
          ("AM-like 32->32 B=30", 1_500_000, 6_000_000, 267, 32, 32, 30),
          # SURVEY.md 8d "skew" variant at the headline size: dst ~ Zipf(1.2)-tailed mod N (KG hubs: node 0 receives 13 % of the
          # edges, the first 224 nodes two thirds); the forward takes the edge-parallel path, dX the tile kernel
          ("10M/100M skew", 10_000_000, 100_000_000, 32, 64, 64, None))
CPU_RUNG = ("1M/10M", 1_000_000, 10_000_000)


def algorithmic_bytes(e, n, r, din, dout):
    """SURVEY.md 8d, per launch.  int32 index + fp32 weight per edge, fp32 features."""
    w = 4 * (r + 1) * din * dout
    fwd = e * (8 + 4 * din) + n * (4 + 4 * din + 4 * dout) + w
    dx = e * (8 + 4 * dout) + n * (4 + 4 * dout + 4 * din) + w
    dw = e * (8 + 4 * din) + n * (4 + 4 * din) + w          # dOut counted once, in dx
    return {"fwd": fwd, "dx": dx, "dw": dw}


def algorithmic_flops(e, n, r, din, dout):
    """SURVEY.md 8d, per launch: one in x out contraction per distinct (dst, relation) pair S (uniform graph:
    S = N R' (1 - exp(-E / (N R')))) and per node (root), plus the E x in adds of the aggregation.  The kernels
    execute 2 in out (E + N) -- they contract per edge -- so the achieved figure is the conservative one."""
    s_pairs = n * r * (1.0 - math.exp(-e / (n * r)))
    fwd = 2.0 * din * dout * (s_pairs + n) + e * din
    return {"fwd": fwd, "dx": 2.0 * din * dout * (s_pairs + n) + e * dout, "dw": fwd}


def synthetic_on_device(n, e, r, din, dout, dev, seed=0, skew=False):
    """SURVEY.md 8d inputs, generated on the device.  skew: dst ~ Zipf(1.2)-tailed, mod N (the same law as
    oracle.synthetic_graph(skew=True): P(dst >= k) = (k + 1)^-0.2 before the fold) -- KG hubs: node 0 receives 13 % of the
    edges, the first 224 nodes two thirds."""
    g = torch.Generator(device=dev).manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g, device=dev)
    if skew:
        u = torch.rand(e, generator=g, device=dev, dtype=torch.float64)
        dst = (torch.floor(u.pow(-5.0)).clamp_(max=2.0 ** 62).to(torch.int64) - 1) % n
        del u
    else:
        dst = torch.randint(0, n, (e,), generator=g, device=dev)
    typ = torch.randint(0, r, (e,), generator=g, device=dev)
    x = torch.randn(n, din, generator=g, device=dev)
    dout_grad = torch.randn(n, dout, generator=g, device=dev)
    bw, br = math.sqrt(6.0 / (din * dout)), math.sqrt(6.0 / (din + dout))
    weight = torch.empty(r, din, dout, device=dev).uniform_(-bw, bw, generator=g)
    root = torch.empty(din, dout, device=dev).uniform_(-br, br, generator=g)
    return torch.stack([src, dst]), typ, x, dout_grad, weight, root


def host_threads():
    # the GPU box gives one GPU's share of the host: 16 cores (os.cpu_count() reports the whole host and
    # oversubscribing ATen's OpenMP pool makes the loop crawl)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))


def cpu_baseline(n, e, r, din, dout):
    """The oracle's restatement of the PyG loop (the ops the reference executes on CPU), fwd + bwd, ONE pass
    (about 10-30 s of CPU work at (1M, 10M): R' live [N, in] temporaries under autograd)."""
    from oracle import rgcn_oracle as O
    threads = host_threads()
    torch.set_num_threads(threads)
    ei, et = O.synthetic_graph(n, e, r, seed=0)
    w, root, bias = O.synthetic_params(r, din, dout, seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, din, generator=g).requires_grad_(True)
    dg = torch.randn(n, dout, generator=g)
    for t in (w, root, bias):
        t.requires_grad_(True)
    t0 = time.perf_counter()
    out = O.rgcn_conv_loop(x, ei, et, w, root, bias)
    out.backward(dg)
    dt = time.perf_counter() - t0
    return {"value": e / dt, "unit": "edges/s", "cores": threads, "kind": "port", "seconds": dt,
            "sample": f"synthetic {n} nodes / {e} edges / {r} relations, {din}->{dout}, fwd+bwd, one pass "
                      f"(PyG-loop restatement oracle/rgcn_oracle.py under autograd, torch {torch.__version__} CPU)"}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def plan_stats(plan):
    """how full the plan's slots are: the forward / dX kernels pay per chunk and per 16-row tile, whatever they hold"""
    real = int((plan.slot_w != 0).sum())
    rows16 = int(plan.chunk_cnt.sum())
    return {"tile": plan.tile, "chunk": plan.chunk, "layout": plan.layout, "tiles": plan.n_tiles, "chunks": plan.n_chunks,
            "chunks_per_tile": plan.n_chunks / max(1, plan.n_tiles),
            "slot_fill": real / max(1, plan.n_chunks * plan.chunk),          # real slots / allocated slots
            "row_tile_fill": real / max(1, rows16),                          # real slots / slots of the used 16-row tiles
            "rows_per_chunk": real / max(1, plan.n_chunks)}


def gpu_rung(n, e, r, din, dout, dev, steps=20, warmup=5, graph=False, num_bases=None, skew=False):
    """fwd + bwd of one layer through the drop-in module on a fresh synthetic graph: (median ms per step, plan s)"""
    from scaling_rgcn_training_amd.conv import RGCNConv
    from scaling_rgcn_training_amd.plan import clear_plan_cache
    ei, et, x, dg, weight, root = synthetic_on_device(n, e, r, din, dout, dev, seed=1, skew=skew)
    conv = RGCNConv(din, dout, r, num_bases=num_bases).to(dev)
    with torch.no_grad():
        if num_bases is None:
            conv.weight.copy_(weight)
        conv.root.copy_(root)
    x.requires_grad_(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plans = conv._plans(x, ei, et)
    torch.cuda.synchronize()
    plan_s = time.perf_counter() - t0
    if plans.fwd is not None:
        stats = plan_stats(plans.fwd)
    else:       # edge-parallel forward: dense relation-major units
        ep = plans.ep_fwd
        stats = {"units": ep.n_units, "slot_fill": ep.n_rows / max(1, ep.n_units * 64), "max_rows_per_dst": ep.max_rows_per_dst,
                 "sum_levels": len(ep.levels)}
    stats["path"] = {"fwd": "ep" if plans.ep_fwd is not None else "ring", "dx": "ep" if plans.ep_bwd is not None else "ring"}
    del plans
    evs = []
    for i in range(warmup + steps):
        x.grad = None
        conv.zero_grad(set_to_none=True)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        conv(x, ei, et).backward(dg)
        b.record()
        if i >= warmup:
            evs.append((a, b))
    torch.cuda.synchronize()
    ms = statistics.median(a.elapsed_time(b) for a, b in evs)
    ms_graph = None
    if graph:
        # the same step captured once in a hipGraph and replayed: what a launch-bound small graph costs without the
        # host between its ~12 launches (the library never synchronises or allocates on the layer path, so capture works)
        x.grad = None
        conv.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            conv(x, ei, et).backward(dg)
        evs = []
        for i in range(warmup + steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            if i >= warmup:
                evs.append((a, b))
        torch.cuda.synchronize()
        ms_graph = statistics.median(a.elapsed_time(b) for a, b in evs)
        del g
    clear_plan_cache()
    return ms, plan_s, ms_graph, stats


def emulate_world(weight, root, ei, et, x, dg, dev, world, single_step_ms, steps=12, warmup=3, pieces_list=(1, 2, 4),
                  exchanges=("full", "needed"), ranks=None):
    """ONE GPU standing in for one rank of a ``world``-rank job (no process group, no collectives): the rank's plans of the
    world-``world`` cut built as dist.rank_plans builds them, its forward / dX / d_weight launches over its pieces into the
    full-size buffers, the packing and scattering of the rows an exchange = "needed" run would send and receive -- everything a
    rank does per step except the bytes on the wire.  Per (pieces, exchange, rank): median HIP-event ms of the forward part and
    of the backward part of a step (``rank_share_ms`` = their sum) next to ``single_step_ms / world``, the rows the rank would
    receive, and a step time PREDICTED from those measurements and a link rate: forward = max(K_f + c / p, K_f / p + c) (the
    exchange of piece s runs under the kernels of piece s + 1; the last piece's is exposed), backward = max(K_b, K_f / p + c)
    (the dX exchange stays in flight under the weight-gradient kernels, conv.py), c = bytes per link and gather / rate with
    all world - 1 links of a GPU busy at once (xGMI is point to point)."""
    from scaling_rgcn_training_amd import dist as rdist
    from scaling_rgcn_training_amd.conv import RGCNConv
    from scaling_rgcn_training_amd.plan import clear_plan_cache
    n, d = x.shape
    e = int(et.shape[0])
    r = int(weight.shape[0])
    rows = []
    for pieces in pieces_list:
        for exch in exchanges:
            conv = RGCNConv(d, d, r).to(dev)
            conv.path = "ring"
            with torch.no_grad():
                conv.weight.copy_(weight)
                conv.root.copy_(root)
            rdist.attach(conv, n, e, edge_index=ei, pieces=pieces, exchange=exch, emulate=(world, 0))
            bc = getattr(conv.dist, "block_costs", None)
            heaviest = int(bc.sum(0).argmax()) if bc is not None else 0
            todo = ranks if ranks is not None else sorted({0, heaviest})
            for rk in todo:
                rdist.attach(conv, n, e, edge_index=ei, pieces=pieces, exchange=exch, emulate=(world, rk))
                dctx = conv.dist
                xx = x.detach().requires_grad_(True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                plans = conv._plans(xx, ei, et)
                torch.cuda.synchronize()
                plan_s = time.perf_counter() - t0
                evs = []
                for i in range(warmup + steps):
                    xx.grad = None
                    conv.zero_grad(set_to_none=True)
                    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                    a.record()
                    out = conv(xx, ei, et)
                    b.record()
                    out.backward(dg)
                    c.record()
                    if i >= warmup:
                        evs.append((a, b, c))
                torch.cuda.synchronize()
                f_ms = statistics.median(a.elapsed_time(b) for a, b, c in evs)
                b_ms = statistics.median(b.elapsed_time(c) for a, b, c in evs)
                own_rows = sum(dctx.node_range(s_, n)[1] - dctx.node_range(s_, n)[0] for s_ in range(dctx.pieces))
                row = {"world": world, "pieces": dctx.pieces, "exchange": exch, "rank": rk, "heaviest_rank": heaviest,
                       "piece_tiles": [(dctx.bounds[s_ * world + 1] - dctx.bounds[s_ * world]) // max(1, plans.pieces[0].fwd.tile if plans.pieces[0].fwd is not None else 1)
                                       for s_ in range(dctx.pieces)],
                       "rank_share_ms": f_ms + b_ms, "forward_ms": f_ms, "backward_ms": b_ms,
                       "single_gpu_step_over_world_ms": single_step_ms / world,
                       "share_over_ideal": (f_ms + b_ms) / (single_step_ms / world),
                       "owned_rows": own_rows, "owned_edges_in": sum(p.fwd.n_edges for p in plans.pieces if p.fwd is not None),
                       "plan_build_s": plan_s}
                remote = n - own_rows
                if exch == "needed":
                    nf, nb = plans.needed_fwd, plans.needed_bwd
                    row["rows_needed_fraction"] = {"forward_output": nf.rows_needed / max(1, nf.rows_remote),
                                                   "dx_output": nb.rows_needed / max(1, nb.rows_remote)}
                    recv_rows = (nf.rows_needed + nb.rows_needed) / 2
                else:
                    recv_rows = remote
                ld = (d + 3) // 4 * 4
                per_link = recv_rows * ld * 4 / max(1, world - 1)
                pred = {}
                # the pieces' shares of the rank's rows (dist.piece_tiles: pieces of whole launch rounds are not equal)
                fr = [(dctx.node_range(s_, n)[1] - dctx.node_range(s_, n)[0]) / max(1, own_rows) for s_ in range(dctx.pieces)]

                def wire_end(k_ms, c_ms):
                    # kernels of piece s, then its exchange on the collective's stream (one exchange at a time), under the kernels
                    # of the pieces behind it: when the last exchange ends.  Equal pieces: max(K + c / p, K / p + c).
                    t_k = t_c = 0.0
                    for f_ in fr:
                        t_k += k_ms * f_
                        t_c = max(t_c, t_k) + c_ms * f_
                    return t_c

                for name, rate in (("153_GBps", 153e9), ("76_GBps", 76e9)):
                    c_ms = per_link / rate * 1e3
                    t_f = wire_end(f_ms, c_ms)
                    t_b = max(b_ms, wire_end(f_ms, c_ms))      # (the dX launches take about what the forward ones take)
                    pred[name] = {"ms_per_gather_on_the_wire": c_ms, "step_ms": t_f + t_b, "speedup_over_1_gpu": single_step_ms / (t_f + t_b)}
                row["bytes_per_link_per_gather"] = per_link
                row["predicted"] = pred
                rows.append(row)
                log(f"emulated rank {rk}/{world}, pieces {dctx.pieces}, exchange {exch}: fwd {f_ms:.3f} + bwd {b_ms:.3f} = {f_ms + b_ms:.3f} ms "
                    f"(step / {world} = {single_step_ms / world:.3f}); predicted {pred['153_GBps']['speedup_over_1_gpu']:.2f}x at 153 GB/s per link, "
                    f"{pred['76_GBps']['speedup_over_1_gpu']:.2f}x at 76")
                del plans, out, xx
                clear_plan_cache()
            del conv
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nodes", type=int, default=HEADLINE[0])
    ap.add_argument("--edges", type=int, default=HEADLINE[1])
    ap.add_argument("--relations", type=int, default=HEADLINE[2])
    ap.add_argument("--width", type=int, default=HEADLINE[3])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ladder", action="store_true")
    ap.add_argument("--split-producers", dest="split_producers", action="store_true", default=None,
                    help="forward / dX on the bf16x3 kernel whose producer waves split the rows (fp32-equivalent; tile 224)")
    ap.add_argument("--no-split-producers", dest="split_producers", action="store_false")
    ap.add_argument("--pieces", type=int, default=None,
                    help="N > 1: gather pipeline depth (blocks per rank; default dist.PIECES = 4): the collective of piece s runs "
                         "under the kernels of piece s + 1, the exposed tail is 1 / pieces of a gather")
    ap.add_argument("--balance", choices=["auto", "on", "off"], default="auto",
                    help="N > 1: cut the node ranges by edge count (auto: only where equal node blocks differ by more than 5 %%)")
    ap.add_argument("--exchange", choices=["full", "needed"], default="full",
                    help="N > 1: 'needed' = a rank receives only the rows its plans read (all_to_all_single with split sizes, opt-in: "
                         "unread rows of the layer's output are not written); 'full' = the in-place all-gather")
    ap.add_argument("--emulate-world", type=int, default=8,
                    help="N = 1: after the headline run, ONE GPU stands in for a rank of a world of this size (no collectives): "
                         "rank_share_ms per pieces / exchange, 0 = skip")
    ap.add_argument("--emulate-rank", type=int, default=None, help="which rank to emulate (default: rank 0 and the heaviest rank of the cut)")
    ap.add_argument("--emulate-only", action="store_true", help="skip the ladder, the CPU baseline and the alt-kernel leg")
    args = ap.parse_args()

    import __graft_entry__ as ge
    ge.build()          # before anything touches the GPU or the process group (hipcc is a child process)
    if os.environ.get("RGCN_BENCH_WATCHDOG"):      # seconds: every rank dumps its Python stack to stderr that often (a hung multi-rank run says where)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RGCN_BENCH_WATCHDOG"]), repeat=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # RGCN_BENCH_BACKEND=gloo + fewer GPUs than ranks: rehearsal of the N > 1 code path on a 1-GPU box
    backend = os.environ.get("RGCN_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from scaling_rgcn_training_amd import _lib, dist as rdist
    from scaling_rgcn_training_amd.conv import RGCNConv

    n, e, r, d = args.nodes, args.edges, args.relations, args.width
    log(f"rank {rank}/{world}: generating {n} nodes / {e} edges / {r} relations on {torch.cuda.get_device_name(dev)}")
    ei, et, x, dg, weight, root = synthetic_on_device(n, e, r, d, d, dev)
    conv = RGCNConv(d, d, r).to(dev)
    conv.path = "ring"        # the headline leg measures the tile kernels (eplan.choose_path picks them at this shape anyway)
    if args.split_producers is not None:
        conv.split_producers = bool(args.split_producers)
    with torch.no_grad():
        conv.weight.copy_(weight)
        conv.root.copy_(root)
    del weight, root
    if world > 1:
        rdist.attach(conv, n, e, edge_index=ei, pieces=args.pieces or rdist.PIECES,
                     balance={"auto": None, "on": True, "off": False}[args.balance], exchange=args.exchange, edge_type=et)
    x.requires_grad_(True)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plans = conv._plans(x, ei, et)          # one-time graph plan (excluded from the metric)
    torch.cuda.synchronize()
    plan_s = time.perf_counter() - t0
    _pl = [plans] if world == 1 else plans.pieces
    log(f"plan built in {plan_s:.3f}s (device-side builder): fwd {sum(p.fwd.n_chunks for p in _pl)} chunks / "
        f"{sum(p.fwd.n_tiles for p in _pl)} tiles, bwd {sum(p.bwd.n_chunks for p in _pl)} chunks, "
        f"{sum(p.fwd.nbytes() + p.bwd.nbytes() for p in _pl) / 1e9:.2f} GB, tile {_pl[0].fwd.tile}")

    def step():
        x.grad = None
        conv.weight.grad = conv.root.grad = conv.bias.grad = None
        out = conv(x, ei, et)
        out.backward(dg)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step()
        if i < 3 or i == args.warmup - 1:
            torch.cuda.synchronize()
            log(f"warmup step {i} done")
    dctx = conv.dist
    if dctx is not None:
        for k in ("all_gather", "all_gather_bytes", "all_reduce", "all_reduce_bytes"):
            dctx.stats[k] = 0
        dctx.stats["wait_events"] = []
        dctx.stats["dw_tiles_pieces"] = dctx.stats["dw_tiles_rank"] = 0
        dctx.time_waits = True
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        step()
        b.record()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    step_ms = sorted(a.elapsed_time(b) for a, b in evs)
    log(f"{args.steps} timed steps: {dt / args.steps * 1e3:.2f} ms/step (per-step HIP events: median "
        f"{statistics.median(step_ms):.2f}, min {step_ms[0]:.2f}, max {step_ms[-1]:.2f})")
    comm = None
    if dctx is not None:
        dctx.time_waits = False
        st = dctx.stats
        # one list of (start, end) HIP-event pairs per gather (forward and dX of every step), one pair per piece
        per_piece = [0.0] * dctx.pieces
        for evs_ in st["wait_events"]:
            for i, (a, b) in enumerate(evs_):
                per_piece[i] += a.elapsed_time(b)
        n_gathers = max(1, len(st["wait_events"]))
        wait_ms = sum(per_piece) / args.steps
        bc = getattr(dctx, "block_costs", None)
        # rows (in- + out-edges + root rows) every rank walks per step: its blocks' + its share of the heavy segments' rows
        rank_rows = bc.sum(0) + float(getattr(dctx, "shared_rows_per_rank", 0.0)) if bc is not None else None
        gather_bytes = n * d * 4 * (world - 1) / world             # what one gather moves INTO a rank
        nf_, nb_ = getattr(plans, "needed_fwd", None), getattr(plans, "needed_bwd", None)
        comm = {"backend": backend, "pieces": dctx.pieces, "cut": "uniform" if dctx.uniform else "balanced by edge count",
                "exchange": dctx.exchange,
                # exchange = "needed": the share of the rows other ranks own that this rank's plans read at all (rank 0)
                "rows_needed_fraction": None if nf_ is None else {"forward_output": nf_.rows_needed / max(1, nf_.rows_remote),
                                                                  "dx_output": nb_.rows_needed / max(1, nb_.rows_remote)},
                "rows_walked_per_rank": None if rank_rows is None else {
                    "max": float(rank_rows.max()), "mean": float(rank_rows.mean()), "max_over_mean": float(rank_rows.max() / rank_rows.mean()),
                    "per_block_max_over_mean": float(bc.max() / bc.mean()),
                    # rows of heavy (node, relation) segments dealt over all ranks (edge-parallel directions: eplan.SharedHeavy)
                    "shared_heavy_rows_per_rank": float(getattr(dctx, "shared_rows_per_rank", 0.0))},
                "d_weight_pieces_on_tile_major_kernel_per_step": st.get("dw_tiles_pieces", 0) / args.steps,
                # full exchange: one tile-major launch per rank over a contiguous node range of its own (dist.dw_range)
                "d_weight_rank_launches_on_tile_major_kernel_per_step": st.get("dw_tiles_rank", 0) / args.steps,
                # which piece's gather the launch stream had to wait for (mean ms per gather; the last piece has no kernels
                # left to hide under: its share is the exposed tail the pipeline depth trades against launch count)
                "exposed_ms_per_piece": [v / n_gathers for v in per_piece],
                # xGMI is point to point: a gather moves gather_bytes / (world - 1) over each of a rank's world - 1 links
                "link_budget": {"bytes_per_link_per_gather": gather_bytes / max(1, world - 1),
                                "ms_per_gather_at_153_GBps": gather_bytes / max(1, world - 1) / 153e9 * 1e3,
                                "ms_per_gather_at_76_GBps": gather_bytes / max(1, world - 1) / 76e9 * 1e3,
                                "gathers_per_step": 2,
                                "measured_wait_ms_per_gather": sum(per_piece) / n_gathers},
                "all_gather_per_step": st["all_gather"] / args.steps,
                "all_gather_recv_bytes_per_step_per_rank": st["all_gather_bytes"] / args.steps,
                "all_reduce_per_step": st["all_reduce"] / args.steps,
                "all_reduce_bytes_per_step": st["all_reduce_bytes"] / args.steps,
                "comm_bytes_per_step_per_rank": (st["all_gather_bytes"] + 2 * st["all_reduce_bytes"]) / args.steps,
                "wait_ms_per_step": wait_ms,
                "compute_ms_per_step": statistics.median(step_ms) - wait_ms,
                "note": "wait_ms = HIP-event time the launch stream spent blocked on the collectives after the last "
                        "piece's kernels were enqueued (rank 0); the collectives of earlier pieces overlap the kernels"}

    # ---- per-launch timing of the three hot kernels (HIP events on the launch stream) --------------
    fps = [plans.fwd] if world == 1 else [p.fwd for p in plans.pieces]
    bps = [plans.bwd] if world == 1 else [p.bwd for p in plans.pieces]
    fps = [p for p in fps if p.n_owned > 0]
    bps = [p for p in bps if p.n_owned > 0]
    xd = x.detach()
    wf, rt, bs = conv.weight.detach(), conv.root.detach(), conv.bias.detach()
    rows_max = max([p.n_owned for p in fps + bps] + [1])
    out = torch.empty(rows_max, d, device=dev)
    dxb = torch.empty(rows_max, d, device=dev)
    dw, dr, db = torch.empty_like(wf), torch.empty_like(rt), torch.empty_like(bs)
    pk, pkt = _lib.pack_weights(wf, rt, False), _lib.pack_weights(wf, rt, True)
    psf = [(_lib.plan_struct(p), p) for p in fps]
    psb = [(_lib.plan_struct(p), p) for p in bps]

    kf = conv.kernel_flags
    split_producers = bool(fps) and conv._use_split_producers(fps[0].chunk)
    if split_producers:
        kf |= _lib.FLAG_SPLIT_PRODUCERS

    def run_fwd():
        for ps, p in psf:
            _lib.fwd(ps, xd, d, pk, bs, out[:p.n_owned], d, 0, kf)

    def run_dx():
        for ps, p in psb:
            _lib.bwd_dx(ps, dg, d, pkt, dxb[:p.n_owned], d, None, kf)

    dwp = getattr(plans, "dw", None) if world == 1 else None
    psd = _lib.plan_struct(dwp) if dwp is not None else None
    # a rank's pieces: each on the tile-major kernel with its own plan where the module built one (conv.py backward)
    piece_dw = [] if world == 1 else [(pc.fwd, getattr(pc, "dw", None), getattr(pc, "dw_walk", None)) for pc in plans.pieces
                                      if pc.fwd is not None and pc.fwd.n_owned > 0]

    def run_dw():      # what the module's backward launches (conv.py): tile-major kernel + root part, or the relation-major walk
        if psd is not None:
            _lib.bwd_dw_tiles(psd, plans.dw_walk, xd, d, dg, d, dw, kf)
            _lib.bwd_dw_root(xd, d, dg, d, dr, db)       # (the module enqueues it on a side stream beside dX: conv.py)
            return
        if world > 1 and getattr(plans, "dw_rank", None) is not None and plans.dw_rank[0] is not None:
            dwp_, walk_ = plans.dw_rank       # full exchange: one launch over the rank's own contiguous range (conv.py backward)
            b_, e_ = dwp_.node_begin, dwp_.node_end
            _lib.bwd_dw_tiles(_lib.plan_struct(dwp_), walk_, xd, d, dg[b_:e_], d, dw, kf)
            _lib.bwd_dw_root(xd[b_:e_], d, dg[b_:e_], d, dr, db)
            return
        if world > 1:
            for fp_, dwp_, walk_ in piece_dw:
                b_, e_ = fp_.node_begin, fp_.node_end
                if dwp_ is not None:
                    _lib.bwd_dw_tiles(_lib.plan_struct(dwp_), walk_, xd, d, dg[b_:e_], d, dw, kf)
                    _lib.bwd_dw_root(xd[b_:e_], d, dg[b_:e_], d, dr, db)
                else:
                    _lib.bwd_dw(_lib.plan_struct(fp_), xd, d, dg[b_:e_], d, dw, dr, db, kf)
            return
        for ps, p in psf:
            _lib.bwd_dw(ps, xd, d, dg[p.node_begin:p.node_end], d, dw, dr, db, kf)

    launches = {"fwd": run_fwd, "dx": run_dx, "dw": run_dw}
    kernel_ms = {}
    reps = max(5, min(args.steps, 20))
    for name, fn in launches.items():
        fn()
        torch.cuda.synchronize()
        kevs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in kevs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        kernel_ms[name] = statistics.median(a.elapsed_time(b) for a, b in kevs)
        log(f"launch {name}: {kernel_ms[name]:.3f} ms (median of {reps})")
    # the forward and dX launches run the same kernel (rgcn_tile_kernel); dw = rgcn_dw_tile_kernel + reduce + the root part
    # (rgcn_dw_root_kernel, timed here BEHIND the tile-major kernel; in a step it runs beside dX), or memsets +
    # rgcn_dw_direct_kernel + reduce without a dW plan
    alg = algorithmic_bytes(e / world, n / world, r, d, d)
    flops = algorithmic_flops(e / world, n / world, r, d, d)
    tile_ms = kernel_ms["fwd"] + kernel_ms["dx"]
    if tile_ms >= kernel_ms["dw"]:
        kname, prof_name = "rgcn_tile_kernel (fwd + dX launches)", "rgcn::rgcn_tile_kernel<64, 64"
        if split_producers:
            kname, prof_name = "rgcn_tile3p_kernel (fwd + dX launches)", "rgcn::rgcn_tile3p_kernel"
        kbytes, kflops, kms = (alg["fwd"] + alg["dx"]) / 2, (flops["fwd"] + flops["dx"]) / 2, tile_ms / 2
    else:
        kname, prof_name = "dW launches", "rgcn::rgcn_dw_tile_kernel" if psd is not None else "rgcn::rgcn_dw_direct_kernel"
        kbytes, kflops, kms = alg["dw"], flops["dw"], kernel_ms["dw"]
    hbm_achieved = kbytes / (kms * 1e-3) / 1e9
    mfma_achieved = kflops / (kms * 1e-3) / 1e12
    # Which roof binds: the kernels contract in exact fp32 on the matrix cores (tolerance 1e-5 rules out bf16), whose
    # dense peak is 1/16 of bf16's -- at 64 -> 64 the contraction needs more time at its peak than the gather at HBM's
    t_hbm, t_mfma = kbytes / (HBM_PEAK_GBS * 1e9), kflops / (MFMA_F32_PEAK_TFLOPS * 1e12)
    if split_producers and tile_ms >= kernel_ms["dw"]:
        # six bf16 products per fp32 product at the dense bf16 MFMA peak (2.5 PFLOP/s): the gathers bind, not the contraction
        t_mfma = 6.0 * kflops / (MFMA_BF16_PEAK_TFLOPS * 1e12)
    traffic, traffic_source = None, None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
        if world == 1 and (n, e, r, d) == HEADLINE:
            traffic = next(v["hbm_bytes_per_launch"] for k, v in pm["kernels"].items() if k.startswith(prof_name))
            traffic_source = (f"profiles/{PMC_TRAFFIC_FILE}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, "
                              f"gfx950 FETCH_SIZE correction) of this command on an earlier run; not measured in this process")
    except Exception:
        traffic, traffic_source = None, None
    if t_mfma >= t_hbm:
        roofline = {"bound": "mfma", "kernel": kname, "achieved": mfma_achieved, "peak": MFMA_F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": mfma_achieved / MFMA_F32_PEAK_TFLOPS, "traffic": traffic,
                    "algorithmic_flops_per_launch": kflops}
    else:
        roofline = {"bound": "hbm", "kernel": kname, "achieved": hbm_achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS, "traffic": traffic}
    roofline.update({"traffic_source": traffic_source, "algorithmic_bytes_per_launch": kbytes, "avg_launch_ms": kms,
                     "hbm_achieved_GBs": hbm_achieved, "hbm_frac": hbm_achieved / HBM_PEAK_GBS,
                     "t_at_peak_ms": {"hbm": t_hbm * 1e3,
                                      ("mfma_bf16x6" if split_producers and tile_ms >= kernel_ms["dw"] else "mfma_f32"): t_mfma * 1e3}})
    ms_per_step = dt / args.steps * 1e3
    step_bytes = sum(algorithmic_bytes(e, n, r, d, d).values())
    roofline_step = {"bound": "hbm", "algorithmic_bytes_per_step": step_bytes,
                     "achieved": step_bytes / (ms_per_step * 1e-3) / 1e9 / world, "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                     "frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / world / HBM_PEAK_GBS,
                     "target_frac": 0.40, "ms_per_step_at_target": step_bytes / (0.40 * HBM_PEAK_GBS * 1e9) * 1e3 / world}

    rec = None
    # Secondary measurement, never the headline `value`: the same step with the OTHER forward / dX kernel -- exact-fp32 MFMA
    # (rgcn_tile_kernel) when the default producer-split bf16 x 3 kernel ran above, and vice versa -- on the same box right
    # after the main run, with its own plans
    alt = None
    if args.emulate_only:
        args.no_ladder = args.no_cpu_baseline = True
    if world == 1 and (n, e, r, d) == HEADLINE and not args.no_ladder:
        main_mode = conv.split_producers
        try:
            with torch.no_grad():
                out_main = conv(x, ei, et)
            conv.split_producers = not main_mode
            with torch.no_grad():
                out_alt = conv(x, ei, et)
            fwd_diff = float((out_main - out_alt).abs().max())
            fwd_max = float(out_alt.abs().max())
            del out_main, out_alt
            # the split-precision and the exact-fp32 kernels must agree far inside the 1e-5 tolerance on this very input
            assert fwd_diff <= 2e-5 * max(1.0, fwd_max), f"forward kernels differ by {fwd_diff:.3e} (max |out| {fwd_max:.3e})"
            dw_diff = dw_max = None
            if psd is not None:      # d_weight by both forms of the tile-major kernel on the same operands
                dw_b = torch.empty_like(dw)
                _lib.bwd_dw_tiles(psd, plans.dw_walk, xd, d, dg, d, dw, conv.kernel_flags | _lib.FLAG_SPLIT_PRODUCERS)
                _lib.bwd_dw_tiles(psd, plans.dw_walk, xd, d, dg, d, dw_b, conv.kernel_flags)
                dw_diff, dw_max = float((dw - dw_b).abs().max()), float(dw_b.abs().max())
                del dw_b
                assert dw_diff <= 2e-5 * max(1.0, dw_max), f"d_weight kernels differ by {dw_diff:.3e} (max {dw_max:.3e})"
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            aevs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
            for a, b in aevs:
                a.record()
                step()
                b.record()
            torch.cuda.synchronize()
            ams = statistics.median(a.elapsed_time(b) for a, b in aevs)
            alt = {"what": ("forward / dX on rgcn_tile3p_kernel (operand split by the producer waves, bf16 x 3 MFMAs, fp32-equivalent)"
                            if conv.split_producers else
                            "forward / dX on rgcn_tile_kernel and d_weight on rgcn_dw_tile_kernel<false> (exact-fp32 MFMA everywhere; RGCNConv.split_producers = False / RGCN_SPLIT_PRODUCERS=0)"),
                   "ms_per_step_median": ams, "steps": 15, "edges_per_s": e / (ams * 1e-3),
                   # the two kernels' forward outputs on this very input, element by element (fp32-equivalence in the record)
                   "forward_max_abs_diff_between_kernels": fwd_diff, "forward_max_abs": fwd_max,
                   "d_weight_max_abs_diff_between_kernels": dw_diff, "d_weight_max_abs": dw_max}
            log(f"alt ({'bf16x3 split' if conv.split_producers else 'exact-fp32'} forward / dX / dW): {ams:.2f} ms/step")
        except AssertionError:
            raise                      # two kernels of the product disagree: that IS a failed run
        except Exception as err:      # (an allocation failure etc. in the secondary leg must not take the headline record down)
            log(f"alt leg skipped: {err!r}")
        finally:
            conv.split_producers = main_mode
    emu = None
    if world == 1 and args.emulate_world > 1:
        try:
            emu = emulate_world(conv.weight.detach(), conv.root.detach(), ei, et, x.detach(), dg, dev, args.emulate_world,
                                statistics.median(step_ms), ranks=None if args.emulate_rank is None else [args.emulate_rank],
                                pieces_list=(args.pieces,) if args.pieces else (1, 2, 4))
        except Exception as err:      # (a secondary leg must not take the headline record down)
            log(f"emulate-world leg skipped: {err!r}")
    if rank == 0:
        rec = {
            "metric": "edges/s per RGCN layer (fwd+bwd)",
            "value": e * args.steps / dt,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_median": statistics.median(step_ms),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            # fp32 in, fp32 out, fp32 accumulation; with the producer-split kernel the contraction's fp32 operands are split
            # into three bf16 pieces each (24 significant bits kept) and multiplied on bf16 MFMAs: fp32-equivalent
            "dtype": "f32 (contraction: bf16x3 split of fp32 operands, fp32 accumulate)" if split_producers else "f32",
            "data": "synthetic",
            "config": {"workload": f"synthetic {n} nodes / {e} edges / {r} relations, {d}->{d} fp32, "
                                   f"full-graph layer fwd+bwd (BASELINE.json configs[3])",
                       "nodes": n, "edges": e, "relations": r, "in": d, "out": d,
                       "partition": f"dst-range x{world}" if world > 1 else "none"},
            "roofline": roofline,
            "roofline_step": roofline_step,
            "kernel_ms": kernel_ms,
            "alt_forward_kernel": alt,
            "plan_build_s": plan_s,
            "plan_bytes": sum(p.nbytes() for p in fps + bps),
        }
        if emu is not None:
            # one GPU standing in for one rank of the 8-GPU job: what a rank's share of the step costs (measured), what its
            # exchanges would move, and the step time / speed-up those two predict at a given link rate (see emulate_world)
            rec["emulated_world"] = emu
        if comm is not None:
            rec["comm"] = comm
    # ---- the smaller rungs and the CPU baseline beside its rung (N = 1 only) ----------------------------------------
    if world == 1 and rank == 0:
        from scaling_rgcn_training_amd.plan import clear_plan_cache
        del plans, psf, psb, psd, dwp, fps, bps, out, dxb, x, xd, dg, ei, et, conv, _pl
        clear_plan_cache()
        torch.cuda.empty_cache()
        if not args.no_ladder:
            ladder = [{"rung": "10M/100M", "nodes": n, "edges": e, "relations": r, "in": d, "out": d,
                       "gpu_ms_per_step": rec["ms_per_step_median"], "gpu_edges_per_s": e / (rec["ms_per_step_median"] * 1e-3),
                       "roofline_step": {k: rec["roofline_step"][k] for k in ("bound", "algorithmic_bytes_per_step", "achieved", "peak", "unit", "frac")}}]
            for name, ln, le, lr, lin, lout, nb in LADDER:
                skew = "skew" in name
                ms, ps_, msg, st = gpu_rung(ln, le, lr, lin, lout, dev, graph=le <= 1_000_000, num_bases=nb, skew=skew,
                                            steps=10 if skew else 20, warmup=3 if skew else 5)
                ladder.append({"rung": name, "nodes": ln, "edges": le, "relations": lr, "in": lin, "out": lout,
                               "gpu_ms_per_step": ms, "gpu_edges_per_s": le / (ms * 1e-3), "plan_build_s": ps_, "plan": st})
                # every rung's own step-level roofline: SURVEY.md 8d bytes for its N / E / R' / widths (the skew rung: the same
                # algorithmic bytes as the uniform graph of that size) over the eager step and, where one was taken, the replayed one
                sb = sum(algorithmic_bytes(le, ln, lr, lin, lout).values())
                ladder[-1]["roofline_step"] = {"bound": "hbm", "algorithmic_bytes_per_step": sb, "achieved": sb / (ms * 1e-3) / 1e9,
                                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                if msg is not None:
                    ladder[-1]["roofline_step"]["frac_hipgraph_replay"] = sb / (msg * 1e-3) / 1e9 / HBM_PEAK_GBS
                if nb is not None:
                    ladder[-1]["num_bases"] = nb
                if msg is not None:
                    ladder[-1]["gpu_ms_per_step_hipgraph_replay"] = msg
                log(f"ladder {name}: {ms:.3f} ms/step = {le / (ms * 1e-3):.3e} edges/s" +
                    (f" (captured in a hipGraph: {msg:.3f} ms/step)" if msg is not None else ""))
            rec["ladder"] = ladder
        if not args.no_cpu_baseline:
            log(f"timing the CPU baseline on the {CPU_RUNG[0]} rung (one pass, {host_threads()} threads)")
            cb = cpu_baseline(CPU_RUNG[1], CPU_RUNG[2], r, d, d)
            if not args.no_ladder:
                same = next(l for l in rec["ladder"] if l["rung"] == CPU_RUNG[0])
                cb["gpu_same_rung_edges_per_s"] = same["gpu_edges_per_s"]
                cb["gpu_over_cpu_same_rung"] = same["gpu_edges_per_s"] / cb["value"]
            rec["cpu_baseline"] = cb
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
