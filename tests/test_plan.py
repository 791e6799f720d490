"""Host logic: the graph plan (tile / chunk layout) reproduces the layer when walked the way
the kernels walk it.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import rgcn_oracle as O
from scaling_rgcn_training_amd import plan as P
from tests.plan_emulator import emulate_dw, emulate_spmm


def _plans_from_golden(g, tile, chunk=64):
    ei = torch.from_numpy(g["edge_index"]).long()
    et = torch.from_numpy(g["edge_type"]).long()
    return P.build_graph_plans(ei, et, int(g["num_nodes"]), int(g["num_relations"]), tile, chunk=chunk)


def _distinct(ei, et, n, r):
    ei, et = torch.as_tensor(ei).long(), torch.as_tensor(et).long()
    return int(torch.unique((ei[0] * n + ei[1]) * r + et).numel())


def _check_invariants(plan, n_real_edges):
    c = plan.chunk
    assert plan.slot_src.numel() == plan.n_chunks * c
    src = plan.slot_src.view(-1, 16)
    valid = src < plan.n_nodes
    nvalid = valid.sum(1)
    col16 = torch.arange(16)[None, :]
    assert torch.all(valid == (col16 < nvalid[:, None]))           # valid slots are a prefix of every 16-slot row tile
    assert torch.all(plan.slot_w.view(-1, 16)[~valid] == 0)
    cnt = plan.chunk_cnt.long()
    assert torch.all(cnt % 16 == 0) and torch.all(cnt >= 16) and torch.all(cnt <= c)
    used = (torch.arange(c // 16)[None, :] * 16 < cnt[:, None]).reshape(-1)   # row tiles inside chunk_cnt
    assert torch.all(nvalid[~used] == 0)
    assert torch.all(nvalid[used] > 0)
    assert int(nvalid.sum()) == n_real_edges + plan.n_owned         # + one root pseudo edge per node
    col = torch.arange(c)[None, :]
    dl = plan.slot_dstl.view(-1, 16)
    assert torch.all(dl[valid] < plan.tile) and torch.all(dl[~valid] == plan.tile)
    assert torch.all(dl[:, 1:] >= dl[:, :-1])                      # sorted by destination inside a row tile
    dup = torch.zeros(dl.shape[0], dtype=torch.bool)
    dup[(valid[:, 1:] & (dl[:, 1:] == dl[:, :-1])).any(1)] = True
    flags = ((plan.chunk_flags.long()[:, None] >> torch.arange(c // 16)[None, :]) & 1).bool().reshape(-1)
    assert torch.equal(flags, dup)
    # run metadata: inside each 16-slot row tile, exactly the LAST slot of a run of equal destinations
    # carries that destination, all others the dummy row; every slot points at its run's last slot
    acc = (plan.slot_acc & 0xFFFFFF).view(-1, 16).long()
    rend = (plan.slot_acc >> 24).view(-1, 16).long()
    d16 = plan.slot_dstl.view(-1, 16).long()
    for t in range(min(acc.shape[0], 400)):
        for i in range(16):
            j = int(rend[t, i])
            assert i <= j < 16 and torch.all(d16[t, i:j + 1] == d16[t, i]) and (j == 15 or d16[t, j + 1] != d16[t, i])
            assert acc[t, i] == (d16[t, i] if i == j else plan.tile)
    tile_of_slot = torch.repeat_interleave(plan.chunk_tile.long(), c).view(-1, 16)
    row = plan.slot_row.view(-1, 16).long()
    assert torch.all(row[valid] == (tile_of_slot * plan.tile + dl)[valid]) and torch.all(row[~valid] == plan.n_owned)
    tp = plan.tile_ptr.long()
    assert tp[0] == 0 and tp[-1] == plan.n_chunks and torch.all(tp[1:] > tp[:-1])
    for t in range(plan.n_tiles):                                   # tile-major, rel ascending, root last
        rels = plan.chunk_rel[tp[t]:tp[t + 1]]
        assert torch.all(plan.chunk_tile[tp[t]:tp[t + 1]] == t)
        assert torch.all(rels[1:] >= rels[:-1]) and rels[-1] == plan.num_relations
    # the dW walk: every 64-row unit that holds a row tile exactly once, relation-major
    ro = plan.rel_order.long()
    upc = c // 64
    want = [ch * upc + h for ch in range(plan.n_chunks) for h in range(upc) if int(cnt[ch]) > 64 * h]
    assert sorted(ro.tolist()) == want and plan.n_units == len(want)
    assert torch.all(plan.chunk_rel[ro // upc][1:] >= plan.chunk_rel[ro // upc][:-1])


@pytest.mark.parametrize("tile,chunk", [(4, 64), (64, 64), (256, 64), (64, 128), (256, 128)])
def test_plan_walk_matches_golden(golden, tile, chunk):
    if str(golden["mode"]) != "full":
        pytest.skip("plan is weight-mode independent")
    if tile == 4 and golden["edge_index"].shape[1] > 2000:
        pytest.skip("tiny tiles only on small graphs")
    plans = _plans_from_golden(golden, tile, chunk)
    e = _distinct(golden["edge_index"], golden["edge_type"], int(golden["num_nodes"]), int(golden["num_relations"]))
    _check_invariants(plans.fwd, e)
    _check_invariants(plans.bwd, e)
    w_all = np.concatenate([golden["weight"], golden["root"][None]], 0).astype(np.float64)
    out = emulate_spmm(plans.fwd, golden["x"], w_all, golden["bias"])
    np.testing.assert_allclose(out, golden["out"], rtol=1e-6, atol=1e-6)
    dx = emulate_spmm(plans.bwd, golden["dout"], np.transpose(w_all, (0, 2, 1)))
    np.testing.assert_allclose(dx, golden["d_x"], rtol=1e-6, atol=1e-6)
    dw = emulate_dw(plans.fwd, golden["x"], golden["dout"], w_all.shape[0], w_all.shape[1], w_all.shape[2])
    np.testing.assert_allclose(dw[:-1], golden["d_wfull"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dw[-1], golden["d_root"], rtol=1e-6, atol=1e-6)


def test_edge_weights_mean_and_sum():
    src = torch.tensor([0, 1, 1, 2, 3, 3])
    dst = torch.tensor([2, 2, 2, 0, 2, 2])
    rel = torch.tensor([0, 0, 0, 1, 1, 0])
    w = P.edge_weights(src, dst, rel, 2, "mean")
    assert torch.allclose(w, torch.tensor([0.25, 0.25, 0.25, 1.0, 1.0, 0.25]))
    assert torch.all(P.edge_weights(src, dst, rel, 2, "sum") == 1)


def test_empty_graph_and_isolated_nodes():
    ei = torch.zeros(2, 0, dtype=torch.long)
    et = torch.zeros(0, dtype=torch.long)
    plans = P.build_graph_plans(ei, et, 7, 3, 4)
    _check_invariants(plans.fwd, 0)
    assert plans.fwd.n_tiles == 2 and plans.fwd.n_chunks == 2
    x = np.random.default_rng(0).normal(size=(7, 5))
    w_all = np.random.default_rng(1).normal(size=(4, 5, 3))
    out = emulate_spmm(plans.fwd, x, w_all, np.zeros(3))
    np.testing.assert_allclose(out, x @ w_all[-1], rtol=1e-12)


def test_hub_spans_many_chunks():
    n, e = 500, 1000
    g = torch.Generator().manual_seed(0)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.full((e,), 7)
    typ = torch.randint(0, 2, (e,), generator=g)
    ei = torch.stack([src, dst])
    plans = P.build_graph_plans(ei, typ, n, 3, 16)
    _check_invariants(plans.fwd, _distinct(ei, typ, n, 3))
    assert plans.fwd.n_edges == e
    w, root, bias = O.synthetic_params(3, 6, 4)
    x = torch.randn(n, 6, generator=g)
    ref = O.rgcn_conv_dense(x.numpy(), ei.numpy(), typ.numpy(), w.numpy(), root.numpy(), bias.numpy())
    w_all = np.concatenate([w.numpy(), root.numpy()[None]], 0)
    np.testing.assert_allclose(emulate_spmm(plans.fwd, x.numpy(), w_all, bias.numpy()), ref, rtol=1e-6, atol=1e-6)


def test_duplicate_triples_merge_into_one_weighted_slot():
    # 3 copies of 0->2 (rel 0) and one 1->2 (rel 0): c[2,0] = 4, merged weights 3/4 and 1/4
    ei = torch.tensor([[0, 0, 1, 0], [2, 2, 2, 2]])
    et = torch.tensor([0, 0, 0, 0])
    p = P.build_graph_plans(ei, et, 3, 1, 16).fwd
    valid = p.slot_src < p.n_nodes
    rel = torch.repeat_interleave(p.chunk_rel, P.CHUNK)
    real = valid & (rel == 0)
    assert real.sum() == 2
    assert sorted(zip(p.slot_src[real].tolist(), p.slot_w[real].tolist())) == [(0, 0.75), (1, 0.25)]
    assert torch.all(p.slot_dstl[real] == 2)


def test_out_of_range_inputs_raise():
    ei = torch.tensor([[0, 5], [1, 2]])
    with pytest.raises(ValueError):
        P.build_graph_plans(ei, torch.tensor([0, 0]), 4, 2, 4)
    with pytest.raises(ValueError):
        P.build_graph_plans(torch.tensor([[0, 1], [1, 2]]), torch.tensor([0, 2]), 4, 2, 4)


def test_node_range_plans_tile_aligned_partition():
    ei, et = O.synthetic_graph(1000, 8000, 5, seed=2)
    tile = 64
    w = P.edge_weights(ei[0], ei[1], et, 5)
    full = P.build_plan(ei[0], ei[1], et, w, 1000, 5, tile)
    counts = torch.bincount(ei[1] // tile, minlength=full.n_tiles)
    ranges = P.balanced_ranges(counts, 3, tile, 1000)
    assert ranges[0][0] == 0 and ranges[-1][1] == 1000
    assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
    assert all(b % tile == 0 for b, _ in ranges)
    x = np.random.default_rng(0).normal(size=(1000, 8))
    w_all = np.random.default_rng(1).normal(size=(6, 8, 4))
    ref = emulate_spmm(full, x, w_all)
    parts = []
    for b, e_ in ranges:
        p = P.build_plan(ei[0], ei[1], et, w, 1000, 5, tile, b, e_)
        parts.append(emulate_spmm(p, x, w_all))
        # a rank's chunks are exactly the single-rank chunks of its tiles
        t0, t1 = b // tile, (e_ + tile - 1) // tile
        c0, c1 = int(full.tile_ptr[t0]), int(full.tile_ptr[t1])
        assert torch.equal(p.slot_src, full.slot_src[c0 * P.CHUNK:c1 * P.CHUNK])
        assert torch.equal(p.slot_w, full.slot_w[c0 * P.CHUNK:c1 * P.CHUNK])
    np.testing.assert_allclose(np.concatenate(parts, 0), ref, rtol=1e-12, atol=1e-12)


def test_plan_cache_identity():
    ei, et = O.synthetic_graph(100, 500, 3, seed=1)
    P.clear_plan_cache()
    a = P.cached_graph_plans(ei, et, 100, 3, 16, "mean")
    b = P.cached_graph_plans(ei, et, 100, 3, 16, "mean")
    assert a is b
    ei2 = ei.clone()
    c = P.cached_graph_plans(ei2, et, 100, 3, 16, "mean")
    assert c is not a


def test_plan_walk_random_graphs_property():
    """hypothesis: any small multigraph (duplicates, self loops, empty relations, isolated nodes), any tile / chunk
    size: the plans walked the way the kernels walk them reproduce the dense float64 evaluation of the layer
    (to the fp32 rounding of the plan's edge weights)."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None)
    @given(n=st.integers(1, 70), e=st.integers(0, 400), r=st.integers(1, 6), tile=st.sampled_from([16, 32, 48, 64]),
           chunk=st.sampled_from([64, 128]), seed=st.integers(0, 10_000), hub=st.booleans())
    def run(n, e, r, tile, chunk, seed, hub):
        g = torch.Generator().manual_seed(seed)
        ei = torch.randint(0, n, (2, e), generator=g)
        et = torch.randint(0, r, (e,), generator=g)
        if hub and e > 8:
            ei[1, : e // 2] = ei[1, 0]          # half the edges into one node
            et[: e // 4] = et[0]                # many of them in one relation (runs longer than a row tile)
        din, dout = 5, 3
        x = torch.randn(n, din, generator=g, dtype=torch.float64)
        w = torch.randn(r, din, dout, generator=g, dtype=torch.float64)
        root = torch.randn(din, dout, generator=g, dtype=torch.float64)
        bias = torch.randn(dout, generator=g, dtype=torch.float64)
        dg = torch.randn(n, dout, generator=g, dtype=torch.float64)
        plans = P.build_graph_plans(ei, et, n, r, tile, chunk=chunk)
        _check_invariants(plans.fwd, _distinct(ei, et, n, r))
        _check_invariants(plans.bwd, _distinct(ei, et, n, r))
        ref = O.rgcn_conv_dense(x.numpy(), ei.numpy(), et.numpy(), w.numpy(), root.numpy(), bias.numpy())
        grads = O.rgcn_conv_grads_dense(x.numpy(), ei.numpy(), et.numpy(), w.numpy(), root.numpy(), dg.numpy())
        w_all = np.concatenate([w.numpy(), root.numpy()[None]], 0)
        np.testing.assert_allclose(emulate_spmm(plans.fwd, x.numpy(), w_all, bias.numpy()), ref, rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(emulate_spmm(plans.bwd, dg.numpy(), np.transpose(w_all, (0, 2, 1))), grads["x"],
                                   rtol=2e-6, atol=2e-6)
        dw = emulate_dw(plans.fwd, x.numpy(), dg.numpy(), r + 1, din, dout)
        np.testing.assert_allclose(dw[:-1], grads["weight"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(dw[-1], grads["root"], rtol=2e-6, atol=2e-6)

    run()


def test_dw_walk_interleave():
    """rel_order: relations contiguous and ascending, every unit once, and inside a relation the units are dealt
    round-robin over J_r pieces so that every piece sweeps the tile sequence (plan.interleave_walk)."""
    from scaling_rgcn_training_amd.plan import interleave_walk
    r1 = 6
    cnt = torch.tensor([1000, 3, 0, 517, 2000, 1])
    rel = torch.repeat_interleave(torch.arange(r1), cnt)
    units = torch.arange(rel.numel())                  # unit id == rank in the (relation, tile) order
    walkers = 64
    out = interleave_walk(units, rel, r1, walkers=walkers, mode="rr")
    assert torch.equal(interleave_walk(units, rel, r1, walkers=walkers, mode="sorted"), units)
    ph = interleave_walk(units, rel, r1, walkers=walkers, mode="phase")
    assert sorted(ph.tolist()) == units.tolist() and torch.all(rel[ph][1:] >= rel[ph][:-1])
    assert sorted(out.tolist()) == units.tolist()
    assert torch.all(rel[out][1:] >= rel[out][:-1])
    n = int(rel.numel())
    start = torch.cumsum(cnt, 0) - cnt
    for r in range(r1):
        u = int(cnt[r])
        if u == 0:
            continue
        j_r = max(1, (u * walkers + n // 2) // n)
        seg = (out[int(start[r]):int(start[r]) + u] - int(start[r])).tolist()      # ranks q in walk order
        # piece j holds q = j, j + J, j + 2J, ... ascending; pieces follow each other
        want = [q for j in range(j_r) for q in range(j, u, j_r)]
        assert seg == want
    # few walkers or few units: identity
    assert torch.equal(interleave_walk(units[:40], rel[:40], r1, walkers=2048, mode="rr"), units[:40])


@pytest.mark.parametrize("tile", [64, 224, 352])
def test_team_placement_walks_and_keeps_parts_disjoint(golden, tile):
    """Layout 1 (plan.team_placement): same multiset of slots and the same sums as layout 0 (the emulated walk of every
    kernel still reproduces the golden results), a chunk takes exactly the ceil(rows / 16) row tiles of layout 0, every one
    of them non-empty, and part A (the first ceil(nt / 2) row tiles) and part B (the others) hold disjoint destinations
    unless chunk_flags bit 8 says otherwise -- what lets two consumer teams of rgcn_tile3p_kernel accumulate them at once."""
    if str(golden["mode"]) != "full":
        pytest.skip("plan is weight-mode independent")
    f = lambda k: torch.from_numpy(golden[k])
    n, r = int(golden["num_nodes"]), int(golden["num_relations"])
    ei, et = f("edge_index").long(), f("edge_type").long()
    plans = P.build_graph_plans_torch(ei, et, n, r, tile, chunk=128, split=True)
    base = P.build_graph_plans_torch(ei, et, n, r, tile, chunk=128, split=False)
    e = _distinct(golden["edge_index"], golden["edge_type"], n, r)
    for plan, plan0 in ((plans.fwd, base.fwd), (plans.bwd, base.bwd)):
        assert plan.layout == 1
        _check_invariants(plan, e)
        assert plan.n_chunks == plan0.n_chunks and torch.equal(plan.chunk_rel, plan0.chunk_rel)
        assert torch.equal(plan.tile_ptr, plan0.tile_ptr)
        assert int(plan.chunk_cnt.sum()) == int(plan0.chunk_cnt.sum()), "the team placement costs no row tile"
        dl = plan.slot_dstl.view(-1, 8, 16).long()
        nt = (plan.chunk_cnt.long() // 16)
        fl = plan.chunk_flags.long()
        used_tiles = (dl < plan.tile).any(2)                    # [chunks, 8]
        assert torch.equal(used_tiles, torch.arange(8)[None, :] < nt[:, None]), "row tiles 0 .. nt - 1, none of them empty"
        n_straddle = 0
        for c in range(plan.n_chunks):
            na = (int(nt[c]) + 1) // 2
            a = set(dl[c, :na][dl[c, :na] < plan.tile].tolist())
            b = set(dl[c, na:][dl[c, na:] < plan.tile].tolist())
            if fl[c] & 256:
                n_straddle += 1
                assert a & b, f"chunk {c}: flagged although its parts are disjoint"
            else:
                assert not (a & b), f"chunk {c}: parts share destinations {sorted(a & b)[:4]} without the flag"
        assert n_straddle == 0 or n < 3000, "only the hub graphs (thousands of rows on a few destinations) have such chunks"
    w_all = np.concatenate([golden["weight"], golden["root"][None]], 0).astype(np.float64)
    out = emulate_spmm(plans.fwd, golden["x"], w_all, golden["bias"])
    np.testing.assert_allclose(out, golden["out"], rtol=1e-6, atol=1e-6)
    dx = emulate_spmm(plans.bwd, golden["dout"], np.transpose(w_all, (0, 2, 1)))
    np.testing.assert_allclose(dx, golden["d_x"], rtol=1e-6, atol=1e-6)
    dw = emulate_dw(plans.fwd, golden["x"], golden["dout"], w_all.shape[0], w_all.shape[1], w_all.shape[2])
    np.testing.assert_allclose(dw[:-1], golden["d_wfull"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("n,e,r,tile", [(20000, 60000, 32, 224), (3000, 30000, 32, 224), (5000, 100000, 32, 224), (300, 2500, 8, 64),
                                        (9000, 90000, 32, 288), (7000, 70000, 32, 352)])
def test_compact_runs_keeps_the_layer_and_its_own_invariants(n, e, r, tile):
    """plan.compact_runs (layout 3; twin of compact_runs_kernel): the walk over all slots still sums the layer (forward and
    transposed plan, whose runs differ in weight); a compacted chunk holds pairwise distinct destinations on its head slots, every
    shadow sits in its head's lane position (second rows: row tile 7 - h // 16; third rows: the row tile below those, 7 - ns1;
    place h % 16) with its head's output row, the flags'
    row-tile count equals chunk_cnt / 16 on EVERY chunk, and no edge row is lost."""
    import numpy as np
    from oracle import rgcn_oracle as O
    from scaling_rgcn_training_amd import plan as P
    from tests.plan_emulator import emulate_spmm
    ei, et = O.synthetic_graph(n, e, r, seed=n + e)
    ei[:, 40:70] = ei[:, 5:35]              # duplicate triples: merged slots of weight 2 / c inside runs of weight 1 / c
    et[40:70] = et[5:35]
    w, root, bias = O.synthetic_params(r, 8, 6, seed=2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 8, generator=g).double().numpy()
    w_all = np.concatenate([w.numpy(), root.numpy()[None]], 0).astype(np.float64)
    ew = P.edge_weights(ei[0], ei[1], et, r, "mean")
    some = 0
    for gather, scatter in ((ei[0], ei[1]), (ei[1], ei[0])):
        ch = 112 if tile > 224 else 128          # tiles above 224: chunks of at most seven row tiles of rows
        p0 = P.build_plan(gather, scatter, et, ew, n, r, tile, chunk=ch)
        p3 = P.build_plan(gather, scatter, et, ew, n, r, tile, chunk=ch, split=3)
        assert p0.chunk == 128 and p0.chunk_rows == ch and int(p0.chunk_cnt.max()) <= ch
        assert p3.layout == 3 and p3.n_chunks == p0.n_chunks
        np.testing.assert_allclose(emulate_spmm(p3, x, w_all, bias.numpy()), emulate_spmm(p0, x, w_all, bias.numpy()), rtol=0, atol=1e-12)
        fl = p3.chunk_flags.numpy().astype(np.int64)
        cnt = p3.chunk_cnt.numpy().astype(np.int64)
        assert np.array_equal((fl >> 20) & 15, cnt // 16)
        assert int((p3.slot_w != 0).sum()) == int((p0.slot_w != 0).sum())
        assert np.array_equal(np.sort(p3.slot_src.numpy()[p3.slot_src.numpy() < n]), np.sort(p0.slot_src.numpy()[p0.slot_src.numpy() < n]))
        src = p3.slot_src.numpy().reshape(-1, 128)
        row = p3.slot_row.numpy().reshape(-1, 128)
        for c in np.nonzero((fl >> 16) & 7)[0]:
            some += 1
            ns1, ns2, nh = (fl[c] >> 16) & 3, (fl[c] >> 18) & 1, cnt[c] // 16
            heads = row[c, :nh * 16][src[c, :nh * 16] < n]
            assert len(set(heads.tolist())) == len(heads), "a compacted chunk's heads scatter into pairwise distinct rows"
            assert (fl[c] & 0xFFFF) == 0, "so none of its row tiles needs the run-sum"
            third = 7 - ns1        # third rows: right below the tiles of second rows (round 4; round 3: always row tile 5)
            layout = [(7, 0, ns1 >= 1)] + ([(6, 1, True), (5, 0, bool(ns2))] if ns1 >= 2 else [(6, 0, bool(ns2)), (5, 0, False)])
            assert not ns2 or third == (5 if ns1 >= 2 else 6)
            for t, head_tile, shadow in layout:
                used = src[c, 16 * t:16 * t + 16] < n
                if shadow:      # behind the head tiles, every row the output row of the head in the same place
                    assert t >= nh and used.any()
                    assert np.array_equal(row[c, 16 * t:16 * t + 16][used], row[c, 16 * head_tile:16 * head_tile + 16][used])
                elif t >= nh:
                    assert not used.any()
            assert nh + ns1 + ns2 <= 8
    assert some > 0 or e > 64 * n, "no chunk was compacted: the case tests nothing"
