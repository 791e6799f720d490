"""N > 1 path on CPU: world_size-2 gloo processes run the edge partition, the per-layer all-gather and
the weight-gradient all-reduce around the numpy plan walk (tests/plan_emulator.py stands in for the
HIP kernels, which need a GPU), and must reproduce the single-rank result.  CPU only."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret, n, e, tile, pieces, skew=False, balance=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import rgcn_oracle as O
    from scaling_rgcn_training_amd import dist as rdist, plan as P
    from scaling_rgcn_training_amd.conv import _gather_pieces
    from tests.plan_emulator import emulate_dw, emulate_spmm
    r, din, dout = 5, 8, 6
    ei, et = O.synthetic_graph(n, e, r, seed=4, skew=skew)
    w, root, bias = O.synthetic_params(r, din, dout, seed=4)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, din, generator=g).double()
    dg = torch.randn(n, dout, generator=g).double()
    w_all = np.concatenate([w.numpy(), root.numpy()[None]], 0).astype(np.float64)
    ctx = rdist.make_context(n, tile, pieces=pieces, edge_index=ei, balance=balance)
    assert ctx is not None and ctx.world == world and ctx.rank == rank
    if balance or skew:
        assert not ctx.uniform, "edge counts of equal node blocks differ by more than 5 %: the cut follows the edges"
        assert all(b % tile == 0 for b in ctx.bounds[:-1]) and ctx.bounds[-1] == n and ctx.bounds == sorted(ctx.bounds)
        uni = rdist.block_costs(rdist.tile_costs(ei, n, tile), [i * rdist.piece_rows(n, tile, world, ctx.pieces) for i in range(world * ctx.pieces + 1)], tile)
        assert float(ctx.block_costs.max()) <= float(uni.max()), "never worse than the uniform cut"
    plans = rdist.rank_plans(ei, et, n, r, tile, "mean", ctx)
    assert len(plans.pieces) == ctx.pieces
    assert sum(p.fwd.n_edges for p in plans.pieces) <= e

    def gather(plan_list, feat, w_mats, b_vec, width):
        """the PRODUCT's pipeline (conv._gather_pieces: own block written in place, asynchronous in-place all-gather
        of its super-block) with the numpy plan walk standing in for the kernel launch"""
        def launch(pl, rows):
            rows[:pl.n_owned] = torch.from_numpy(emulate_spmm(pl, feat, w_mats, b_vec))
        return _gather_pieces(ctx, plan_list, launch, width, n, torch.device("cpu"), dtype=torch.float64)

    out = gather([p.fwd for p in plans.pieces], x.numpy(), w_all, bias.numpy(), dout)
    dx = gather([p.bwd for p in plans.pieces], dg.numpy(), np.transpose(w_all, (0, 2, 1)), None, din)
    assert ctx.stats["all_gather"] == 2 * ctx.pieces
    # weight gradients: partial over the own blocks, all-reduced
    dw = torch.zeros(r + 1, din, dout, dtype=torch.float64)
    for p in plans.pieces:
        if p.fwd.n_owned:
            b, e_ = p.fwd.node_begin, p.fwd.node_end
            dw += torch.from_numpy(emulate_dw(p.fwd, x.numpy(), dg.numpy()[b:e_], r + 1, din, dout))
    dist.all_reduce(dw)
    # ---- exchange = "needed" (opt-in): a rank receives only the rows its plans read.  Owned and read rows bit-identical to the
    # full exchange; every other row is never written (poisoned with NaN here) and never read: a SECOND layer walked over the
    # exchanged matrix (forward over the forward output, transposed over the dX output) must come out NaN-free and equal
    ctx2 = rdist.make_context(n, tile, pieces=pieces, edge_index=ei, balance=balance, exchange="needed")
    ctx2.poison_unread = True
    assert ctx2.bounds == ctx.bounds
    plans2 = rdist.rank_plans(ei, et, n, r, tile, "mean", ctx2)
    nf, nb = plans2.needed_fwd, plans2.needed_bwd
    assert nf is not None and nb is not None

    def gather2(plan_list, feat, w_mats, b_vec, width, needed):
        def launch(pl, rows):
            rows[:pl.n_owned] = torch.from_numpy(emulate_spmm(pl, feat, w_mats, b_vec))
        return _gather_pieces(ctx2, plan_list, launch, width, n, torch.device("cpu"), dtype=torch.float64, needed=needed)

    w_t = np.transpose(w_all, (0, 2, 1))
    for plist, feat, wm, bv, width, need, full_res in (([p.fwd for p in plans2.pieces], x.numpy(), w_all, bias.numpy(), dout, nf, out),
                                                       ([p.bwd for p in plans2.pieces], dg.numpy(), w_t, None, din, nb, dx)):
        got = gather2(plist, feat, wm, bv, width, need)
        read = torch.zeros(n, dtype=torch.bool)
        for s_ in range(ctx2.pieces):
            b_, e_ = ctx2.node_range(s_, n)
            read[b_:e_] = True
            read[need.recv_idx[s_]] = True
            assert len(need.recv_splits[s_]) == world and need.recv_splits[s_][rank] == 0 and need.send_splits[s_][rank] == 0
        assert torch.equal(got[read], full_res[read]), "owned + read rows bit-identical to the full exchange"
        assert torch.isnan(got[~read]).all(), "rows no plan of this rank reads are not written"
        assert int(read.sum()) - sum(ctx2.node_range(s_, n)[1] - ctx2.node_range(s_, n)[0] for s_ in range(ctx2.pieces)) == need.rows_needed
        # the next layer in that direction, over this rank's own plans: it must not touch an unwritten row
        w_next = w_t if need is nf else w_all
        for pl in plist:
            if pl.n_owned:
                a = emulate_spmm(pl, got.numpy(), w_next, None)
                assert np.isfinite(a).all() and np.array_equal(a, emulate_spmm(pl, full_res.numpy(), w_next, None))
    # what the peers' lists say about the cut: every remote row a rank reads is sent by exactly its owner
    tot = torch.tensor([sum(sum(sp) for sp in nf.send_splits), sum(sum(sp) for sp in nf.recv_splits)], dtype=torch.int64)
    dist.all_reduce(tot)
    assert int(tot[0]) == int(tot[1]), "rows sent == rows received over all ranks"
    if rank == 0:
        ref, gr = O.rgcn_conv_segments(x.numpy(), ei.numpy(), et.numpy(), w.numpy(), root.numpy(), bias.numpy(),
                                       dg.numpy())
        np.testing.assert_allclose(out.numpy(), ref, rtol=1e-6, atol=1e-6)   # w is fp32 1/c: 6e-8 relative
        np.testing.assert_allclose(dx.numpy(), gr["x"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(dw[:-1].numpy(), gr["weight"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(dw[-1].numpy(), gr["root"], rtol=1e-6, atol=1e-6)
        # a rank's slots are exactly the single-rank slots of its tiles (bit-identical per-row results)
        single = P.build_graph_plans(ei, et, n, r, tile)
        one = emulate_spmm(single.fwd, x.numpy(), w_all, bias.numpy())
        assert np.array_equal(one, out.numpy())
        ret.put("ok")
    dist.destroy_process_group()


def _run(world, n, e, tile, pieces, skew=False, balance=None):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret, n, e, tile, pieces, skew, balance)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ret.get(timeout=5) == "ok"


def test_two_rank_gloo_partition_matches_single_rank():
    _run(2, 1000, 9000, 64, 3)


def test_two_rank_gloo_empty_trailing_blocks():
    """n_nodes not a tile multiple and fewer tiles than world * pieces blocks can hold: 637 nodes / tile 64 = 10 tiles
    in 2 x 4 blocks of 2 tiles -> blocks 5..7 lie (partly or wholly) past the last node (the case that raised
    'node_begin must be a multiple of the tile size' in round 1)."""
    _run(2, 637, 5000, 64, 4)


def test_four_ranks_on_an_aifb_sized_graph():
    """fewer tiles than world x pieces blocks want (ADVICE r1's failing case: AIFB at world 4): 8,243 nodes at tile 512 = 17
    tiles -> the piece count drops to 4 tiles-per-rank's worth and trailing blocks are empty"""
    _run(4, 8243, 49838, 512, 4)


def test_eight_ranks_on_an_aifb_sized_graph():
    _run(8, 8243, 49838, 512, 4)


def test_four_ranks_edge_balanced_cut_on_a_hub_graph():
    """dst ~ Zipf(1.2)-tailed (SURVEY.md 8d 'skew'): equal node blocks differ several-fold in edge count, so the cut follows the
    prefix sum of the tiles' edge counts; unequal blocks are gathered by one broadcast per rank and piece; results still
    bit-identical to the single-rank layer"""
    _run(4, 6000, 60000, 64, 4, skew=True)


def test_eight_ranks_edge_balanced_cut_on_a_hub_graph():
    _run(8, 6000, 60000, 64, 3, skew=True)


def test_two_ranks_balanced_cut_pinned_on_a_uniform_graph():
    _run(2, 1000, 9000, 64, 3, balance=True)
