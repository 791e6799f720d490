"""The real edge-partitioned HIP path with 2 ranks sharing the one GPU of the test box (gloo backend
moving CUDA tensors; RCCL refuses two ranks on one device).  The driver measures true multi-GPU scaling
at round end with bench.py; this checks that the partitioned layer equals the single-rank layer."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret, dw_direct):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import rgcn_oracle as O
    from scaling_rgcn_training_amd import _lib, dist as rdist
    from scaling_rgcn_training_amd.conv import RGCNConv
    dev = torch.device("cuda:0")
    # "0" / "2": pin the relation-major dW kernel (ring / direct) on every piece; "tiles": the default of 64 x 64 layers on
    # large graphs -- every piece on the tile-major kernel with its own T = 320 plan -- reached here by lowering the
    # edge-count threshold; "skew": the same on a hub graph, whose cut follows the edge counts (unequal blocks, broadcasts)
    from scaling_rgcn_training_amd import conv as C
    flags = {"0": _lib.FLAG_DW_RING, "2": _lib.FLAG_DW_DIRECT}.get(dw_direct, 0)
    # "skew-ep": the hub graph with the path choice left to the layer -- every piece's forward on the edge-parallel kernels
    # "needed": the opt-in exchange in which a rank receives only the rows its plans read (all_to_all_single with split sizes);
    # unread rows are poisoned with NaN here and the read ones must be bit-identical to the single-rank layer
    if dw_direct in ("tiles", "skew", "skew-ep", "needed", "needed-skew"):
        C.DW_TILES_MIN_EDGES = 1
    skew = "skew" in dw_direct
    needed = dw_direct.startswith("needed")
    n, e, r, din, dout = 3000, 40000, 6, 64, 64
    ei, et = O.synthetic_graph(n, e, r, seed=2, skew=skew)
    w, root, bias = O.synthetic_params(r, din, dout, seed=2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, din, generator=g)
    dg = torch.randn(n, dout, generator=g)

    def run(partitioned):
        conv = RGCNConv(din, dout, r).to(dev)
        conv.kernel_flags = flags
        if dw_direct != "skew-ep":
            conv.path = "ring"      # tile kernels in both runs: a rank's tiles are the single-rank tiles, bit for bit
        with torch.no_grad():
            conv.weight.copy_(w)
            conv.root.copy_(root)
            conv.bias.copy_(bias + 0.25)
        if partitioned:
            rdist.attach(conv, n, e, edge_index=ei, exchange="needed" if needed else "full", edge_type=et)
            assert conv.dist is not None and conv.dist.world == world
            assert conv.dist.uniform == (not skew)
            conv.dist.poison_unread = needed
        xd = x.to(dev).requires_grad_(True)
        out = conv(xd, ei.to(dev), et.to(dev))
        out.backward(dg.to(dev))
        torch.cuda.synchronize()
        if partitioned and dw_direct in ("tiles", "skew"):
            # full exchange: x and g are replicated, so a rank's d_weight is ONE tile-major launch over a contiguous range of its own
            assert conv.dist.stats.get("dw_tiles_rank", 0) == 1 and conv.dist.stats.get("dw_tiles_pieces", 0) == 0
        if partitioned and needed:
            owned = sum(1 for pc in conv._plans(xd, ei.to(dev), et.to(dev)).pieces if pc.fwd.n_owned > 0)
            assert conv.dist.stats.get("dw_tiles_pieces", 0) == owned > 0, "needed rows: every piece's d_weight on the tile-major kernel with its own plan"
        if partitioned and needed:
            pl = conv._plans(xd, ei.to(dev), et.to(dev))
            dctx = conv.dist
            masks = []
            for need in (pl.needed_fwd, pl.needed_bwd):
                read = torch.zeros(n, dtype=torch.bool)
                for s_ in range(dctx.pieces):
                    b_, e_ = dctx.node_range(s_, n)
                    read[b_:e_] = True
                    read[need.recv_idx[s_].cpu()] = True
                masks.append(read.numpy())
            o, gx = out.detach().cpu().numpy(), xd.grad.cpu().numpy()
            assert np.isnan(o[~masks[0]]).all() and np.isnan(gx[~masks[1]]).all(), "unread rows are not written"
            assert 0 < pl.needed_fwd.rows_needed <= pl.needed_fwd.rows_remote
            return (o, gx, conv.weight.grad.cpu().numpy(), conv.root.grad.cpu().numpy(), conv.bias.grad.cpu().numpy(), masks)
        if dw_direct == "skew-ep":
            pl = conv._plans(xd, ei.to(dev), et.to(dev))
            pcs = pl.pieces if partitioned else [pl]
            assert all(pc.ep_fwd is not None for pc in pcs if pc.fwd is None) and any(pc.ep_fwd is not None for pc in pcs)
            if partitioned:      # hubs split across ranks: the heavy segments of the whole graph, an equal share of their rows per rank
                sh = pl.shared_fwd
                assert sh is not None and sh.n_seg > 0 and 0 < sh.row_hi - sh.row_lo <= sh.n_rows // world + 1
                assert all(pc.ep_fwd.heavy is None or pc.ep_fwd.heavy.shared is sh for pc in pcs)
                assert conv.dist.stats.get("shared_heavy_rows", 0) >= sh.row_hi - sh.row_lo
                bc = conv.dist.block_costs.sum(0) + conv.dist.shared_rows_per_rank
                assert float(bc.max() / bc.mean()) <= 1.3, "rows walked per rank max / mean with the hubs split across the ranks"
        return (out.detach().cpu().numpy(), xd.grad.cpu().numpy(), conv.weight.grad.cpu().numpy(),
                conv.root.grad.cpu().numpy(), conv.bias.grad.cpu().numpy())

    single = run(False)
    part = run(True)
    if rank == 0:
        if dw_direct == "skew-ep":      # per-destination sums in slot order: a piece's units pack differently from the whole graph's
            np.testing.assert_allclose(part[0], single[0], rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(single[0]).max())))
            np.testing.assert_allclose(part[1], single[1], rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(single[1]).max())))
        elif needed:
            mf, mb = part[5]
            assert np.array_equal(single[0][mf], part[0][mf]), "owned + read rows of the forward output bit-identical"
            assert np.array_equal(single[1][mb], part[1][mb]), "owned + read rows of dX bit-identical"
        else:
            assert np.array_equal(single[0], part[0]), "partitioned forward must be bit-identical (tile-aligned ranges)"
            assert np.array_equal(single[1], part[1]), "partitioned dX must be bit-identical"
        # weight grads: per-rank partial sums all-reduced -> summation order differs; both must meet the
        # parity criterion against the float64 oracle
        from oracle.tolerance import abs_condition, assert_close
        _, gr = O.rgcn_conv_segments(x.numpy(), ei.numpy(), et.numpy(), w.numpy(), root.numpy(),
                                     (bias + 0.25).numpy(), dg.numpy())
        _, c = abs_condition(x, ei, et, w, root, bias + 0.25, dg)
        for res in (single, part):
            assert_close(res[2], gr["weight"], c["weight"], "d_weight")
            assert_close(res[3], gr["root"], c["root"], "d_root")
            assert_close(res[4], gr["bias"], c["bias"], "d_bias")
        ret.put("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("dw_direct", ["0", "2", "tiles", "skew", "skew-ep", "needed", "needed-skew"])
def test_two_ranks_one_gpu_partitioned_layer_equals_single_rank(dw_direct):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret, dw_direct)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ret.get(timeout=5) == "ok"


@pytest.mark.parametrize("exchange", ["full", "needed"])
def test_emulated_world_8_ranks_stitch_to_the_single_rank_layer(exchange):
    """``dist.attach(..., emulate=(world, rank))`` (bench.py --emulate-world): ONE process builds rank r's plans of the world-8
    cut and launches rank r's kernels with the collectives skipped.  The eight ranks' owned rows, stitched, are the
    single-GPU layer bit for bit (forward and dX); the eight partial weight gradients add up to the single-GPU ones."""
    from oracle import rgcn_oracle as O
    from scaling_rgcn_training_amd import conv as C, dist as rdist
    from scaling_rgcn_training_amd.conv import RGCNConv
    from scaling_rgcn_training_amd.plan import clear_plan_cache
    dev = torch.device("cuda:0")
    old = C.DW_TILES_MIN_EDGES
    C.DW_TILES_MIN_EDGES = 1
    try:
        n, e, r, d, world = 20000, 300000, 6, 64, 8
        ei, et = O.synthetic_graph(n, e, r, seed=9)
        w, root, bias = O.synthetic_params(r, d, d, seed=9)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(n, d, generator=g).to(dev)
        dg = torch.randn(n, d, generator=g).to(dev)
        eid, etd = ei.to(dev), et.to(dev)

        def run(emulate):
            conv = RGCNConv(d, d, r).to(dev)
            conv.path = "ring"
            with torch.no_grad():
                conv.weight.copy_(w)
                conv.root.copy_(root)
                conv.bias.copy_(bias + 0.5)
            if emulate is not None:
                rdist.attach(conv, n, e, edge_index=eid, pieces=2, exchange=exchange, emulate=emulate)
                assert conv.dist.emulate and conv.dist.world == world and conv.dist.rank == emulate[1]
            xd = x.clone().requires_grad_(True)
            out = conv(xd, eid, etd)
            out.backward(dg)
            torch.cuda.synchronize()
            return conv, out.detach(), xd.grad, conv.weight.grad, conv.root.grad, conv.bias.grad

        _, o1, gx1, gw1, gr1, gb1 = run(None)
        o8, gx8 = torch.full_like(o1, float("nan")), torch.full_like(gx1, float("nan"))
        gw8, gr8, gb8 = torch.zeros_like(gw1), torch.zeros_like(gr1), torch.zeros_like(gb1)
        for rk in range(world):
            conv, o, gx, gw, gr, gb = run((world, rk))
            for s in range(conv.dist.pieces):
                b, e_ = conv.dist.node_range(s, n)
                o8[b:e_], gx8[b:e_] = o[b:e_], gx[b:e_]
            gw8 += gw
            gr8 += gr
            gb8 += gb
        assert torch.equal(o8, o1) and torch.equal(gx8, gx1), "the ranks' owned rows are the single-GPU rows"
        for a, b_, what in ((gw8, gw1, "d_weight"), (gr8, gr1, "d_root"), (gb8, gb1, "d_bias")):
            tol = 2e-5 * max(1.0, float(b_.abs().max()))
            assert float((a - b_).abs().max()) <= tol, what
    finally:
        C.DW_TILES_MIN_EDGES = old
        clear_plan_cache()
