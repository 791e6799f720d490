"""Parity of the edge-parallel path (csrc/rgcn_ep.hip: rgcn_ep_transform + rgcn_ep_segment_sum) against the float64 oracle,
through the raw C ABI and through the drop-in module (path selection, weight gradients on the dense relation-major units)."""
import numpy as np
import pytest
import torch

from oracle import rgcn_oracle as O
from oracle.tolerance import abs_condition, assert_close, cpu32_reference

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the GPU box"
    from scaling_rgcn_training_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _ep_layer(dev, ei, et, n, r, x, w, root, bias, dg, piece=None, flags=0, heavy=0):
    """forward, dX (edge-parallel kernels) and dW / dRoot / db (relation-major kernels on the plan's dense units)"""
    from scaling_rgcn_training_amd import _lib, eplan as E, plan as P
    from scaling_rgcn_training_amd.conv import _rows16, _round4
    din, dout = w.shape[1], w.shape[2]
    eid, etd = ei.to(dev), et.to(dev)
    wgt = P.edge_weights(eid[0], eid[1], etd, r)
    kw = {"heavy": heavy} if piece is None else {"piece": piece, "heavy": heavy}
    fwd = E.build_edge_plan(eid[0], eid[1], etd, wgt, n, r, **kw)
    bwd = E.build_edge_plan(eid[1], eid[0], etd, wgt, n, r, **kw)
    xd, gd = _rows16(x.to(dev), din), _rows16(dg.to(dev), dout)
    wd, rd = w.to(dev).contiguous(), None if root is None else root.to(dev).contiguous()
    bd = None if bias is None else bias.to(dev).contiguous()
    out = torch.full((n, _round4(dout)), float("nan"), device=dev)
    _lib.ep_layer(fwd, xd, din, _lib.pack_weights(wd, rd, False), bd, out, dout, 0, None, flags)
    dx = torch.full((n, _round4(din)), float("nan"), device=dev)
    _lib.ep_layer(bwd, gd, dout, _lib.pack_weights(wd, rd, True), None, dx, din, 0, None, flags)
    dw = torch.full((r, din, dout), float("nan"), device=dev)
    dr = torch.full((din, dout), float("nan"), device=dev)
    db = torch.full((dout,), float("nan"), device=dev)
    _lib.bwd_dw(_lib.plan_struct(fwd.as_tile_plan()), xd, din, gd, dout, dw, dr, db, flags)
    if fwd.heavy is not None:       # the heavy segments' pseudo rows over the aggregated matrix H
        dw2 = torch.empty_like(dw)
        _lib.bwd_dw(_lib.plan_struct(fwd.heavy_tile_plan()), _lib.ep_aggregate_heavy(fwd, xd, din), din, gd, dout, dw2, None, None, flags)
        dw += dw2
    torch.cuda.synchronize()
    return (out[:, :dout].cpu().numpy(), dx[:, :din].cpu().numpy(), dw.cpu().numpy(), dr.cpu().numpy(), db.cpu().numpy()), fwd


def _check(res, ref, gr, x, ei, et, w, root, bias, dg, tag):
    out, dx, dw, dr, db = res
    c_out, c = abs_condition(x, ei, et, w, root, bias, dg)
    o32, g32 = cpu32_reference(x, ei, et, w, root, bias, dg)
    assert_close(out, ref, c_out, "ep out" + tag, cpu32=o32)
    assert_close(dx, gr["x"], c["x"], "ep d_x" + tag, cpu32=g32["x"])
    assert_close(dw, gr["weight"], c["weight"], "ep d_weight" + tag, cpu32=g32["weight"])
    assert_close(dr, gr["root"], c["root"], "ep d_root" + tag, cpu32=g32["root"])
    assert_close(db, gr["bias"], c["bias"], "ep d_bias" + tag, cpu32=g32["bias"])


def test_ep_matches_golden(dev, golden):
    """the topologies the reference ships (TEST graph, AIFB / MUTAG summaries: hubs of in-degree up to 11,825)"""
    if str(golden["mode"]) != "full":
        pytest.skip("weight modes go through the module (test_gpu_shapes)")
    f = lambda k: torch.from_numpy(golden[k])
    n, r = int(golden["num_nodes"]), int(golden["num_relations"])
    res, ep = _ep_layer(dev, f("edge_index").long(), f("edge_type").long(), n, r, f("x"), f("weight"), f("root"), f("bias"), f("dout"), heavy=16)
    x, dg = f("x"), f("dout")
    ei, et = f("edge_index").long(), f("edge_type").long()
    ref, gr = O.rgcn_conv_segments(x.numpy(), ei.numpy(), et.numpy(), golden["weight"], golden["root"], golden["bias"], dg.numpy())
    _check(res, ref, gr, x, ei, et, f("weight"), f("root"), f("bias"), dg, f" [golden n{n}]")


@pytest.mark.parametrize("heavy", [0, 16], ids=["per-row", "heavy-segments-aggregated"])
@pytest.mark.parametrize("n,e,r,din,dout,skew,piece,flags", [
    (8243, 49838, 89, 63, 16, False, None, 0),          # AIFB shape
    (3000, 40000, 45, 16, 7, True, 32, 0),              # second layer of the reference's models (hidden 16 -> classes), hubs in levels
    (5000, 120000, 267, 32, 32, False, None, 0),        # AM-like: more relations than rows per tile
    (2000, 150000, 5, 64, 64, True, 64, 0),             # hubs of thousands of rows: three levels
    (2000, 150000, 5, 64, 64, True, 64, 32),            # RGCN_FLAG_SPLIT_PRODUCERS: the bf16 x 3 transform kernel of 64 x 64 layers
    (5000, 90000, 40, 50, 64, False, None, 32),         # ... with padded input columns and many relation changes per wave
    (700, 9000, 3, 100, 128, False, None, 0),           # 128-wide: weight fragments reloaded per row tile
    (700, 9000, 3, 128, 33, True, 16, 1),               # RGCN_FLAG_POINTER_GATHER: 64-bit pointer gathers
    (40, 0, 2, 8, 8, False, None, 0),                   # no edges: the root pseudo edges alone
])
def test_ep_matches_oracle(dev, n, e, r, din, dout, skew, piece, flags, heavy):
    ei, et = O.synthetic_graph(n, max(e, 1), r, seed=n + r, skew=skew)
    if e == 0:
        ei, et = ei[:, :0], et[:0]
    else:
        ei[:, 10:40] = ei[:, 50:80]            # duplicate triples
        et[10:40] = et[50:80]
    w, root, bias = O.synthetic_params(r, din, dout, seed=3)
    g = torch.Generator().manual_seed(5)
    bias = torch.randn(dout, generator=g) * 0.1
    x = torch.randn(n, din, generator=g)
    dg = torch.randn(n, dout, generator=g)
    ref, gr = O.rgcn_conv_segments(x.numpy(), ei.numpy(), et.numpy(), w.numpy(), root.numpy(), bias.numpy(), dg.numpy())
    res, ep = _ep_layer(dev, ei, et, n, r, x, w, root, bias, dg, piece=piece, flags=flags, heavy=heavy)
    if piece is not None and ep.max_rows_per_dst > piece:
        assert len(ep.levels) >= 2
    if heavy and skew and e > 0:
        assert ep.heavy is not None and ep.heavy.n_seg > 0, "a hub graph has segments of 16 rows and more"
    _check(res, ref, gr, x, ei, et, w, root, bias, dg, f" [n{n} r{r} {din}->{dout} heavy {heavy}]")
    again, _ = _ep_layer(dev, ei, et, n, r, x, w, root, bias, dg, piece=piece, flags=flags, heavy=heavy)
    assert all(np.array_equal(a, b) for a, b in zip(res, again)), "bit-reproducible"


@pytest.mark.parametrize("act", [None, "relu", "sigmoid"])
def test_ep_through_the_module_matches_the_tile_kernels(dev, act):
    """``RGCNConv.path``: 'auto' picks the edge-parallel path on an AIFB-shaped graph; pinned 'ep' and pinned 'ring' agree
    on the output and all four gradients (fused activation in the store, ReLU mask of the input in the dX store)."""
    from scaling_rgcn_training_amd.conv import RGCNConv
    from scaling_rgcn_training_amd.plan import clear_plan_cache
    n, e, r = 8243, 49838, 89
    ei, et = O.synthetic_graph(n, e, r, seed=11, skew=True)
    g = torch.Generator().manual_seed(2)
    x = torch.relu(torch.randn(n, 63, generator=g))
    dg = torch.randn(n, 16, generator=g)
    eid, etd = ei.to(dev), et.to(dev)
    res = {}
    for path in ("auto", "ep", "ring"):
        torch.manual_seed(0)
        conv = RGCNConv(63, 16, r).to(dev)
        conv.path = path
        with torch.no_grad():
            conv.bias.uniform_(-0.1, 0.1)
        xd = x.to(dev).requires_grad_(True)
        out = conv(xd, eid, etd, _activation=act, _input_relu=True)
        plans = conv._plans(xd, eid, etd)
        assert (plans.ep_fwd is not None) == (path != "ring") and (plans.ep_bwd is not None) == (path != "ring")
        out.backward(dg.to(dev))
        res[path] = [t.detach().cpu().numpy() for t in (out, xd.grad, conv.weight.grad, conv.root.grad, conv.bias.grad)]
        clear_plan_cache()
    for a, b in zip(res["auto"], res["ep"]):
        assert np.array_equal(a, b)
    for name, a, b in zip(("out", "d_x", "d_weight", "d_root", "d_bias"), res["ep"], res["ring"]):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(b).max())), err_msg=name)


def test_ep_layer_step_is_hipgraph_capturable(dev):
    """the edge-parallel path allocates through torch only and never synchronises: a forward + backward captures as it is and
    the replay reproduces the eager step bit for bit"""
    from scaling_rgcn_training_amd.conv import RGCNConv
    n, e, r = 3000, 20000, 45
    ei, et = O.synthetic_graph(n, e, r, seed=5)
    eid, etd = ei.to(dev), et.to(dev)
    torch.manual_seed(1)
    conv = RGCNConv(63, 16, r).to(dev)
    conv.path = "ep"
    x = torch.randn(n, 63, device=dev).requires_grad_(True)
    dg = torch.randn(n, 16, device=dev)
    conv(x, eid, etd).backward(dg)                    # plans, allocator pools
    eager = [t.detach().clone() for t in (x.grad, conv.weight.grad, conv.root.grad)]
    x.grad = None
    conv.zero_grad(set_to_none=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            out = conv(x, eid, etd)
            out.backward(dg)
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(eager, (x.grad, conv.weight.grad, conv.root.grad)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,e,r,skew,rng", [(8243, 49838, 89, False, None), (3000, 40000, 45, True, None), (700, 9000, 3, True, (128, 448)),
                                            (40, 0, 2, False, None), (100000, 2000000, 7, True, None)])
def test_device_edge_plan_is_bit_identical_to_the_torch_twin(dev, n, e, r, skew, rng):
    """The edge plan behind the C ABI -- rgcn_plan_build_begin / _finish with layout 2 (one relation-major 'tile') +
    rgcn_eplan_segments -- against eplan.build_edge_plan (torch sorts): units, slots, destination-major index, sum levels, and
    the plan struct the weight-gradient kernels walk; both directions, an owned sub-range, hubs, an edgeless graph."""
    from scaling_rgcn_training_amd import _lib, eplan as E, plan as P
    ei, et = O.synthetic_graph(n, max(e, 1), r, seed=n + 1, skew=skew)
    if e == 0:
        ei, et = ei[:, :0], et[:0]
    else:
        ei[:, 5:25] = ei[:, 30:50]             # duplicate triples
        et[5:25] = et[30:50]
    eid, etd = ei.to(dev), et.to(dev)
    b, en = rng if rng is not None else (0, n)
    graph, keep = _lib.graph_struct(eid, etd, n, r)
    ws = _lib.plan_workspace(int(etd.shape[0]), en - b, r, 16, dev)
    w = _lib.edge_weights(graph, "mean", ws)
    for transposed, heavy in ((False, 0), (True, 0), (False, 16), (True, 16)):
        got = E.build_edge_plan_device(graph, w, transposed, n, r, ws, b, en, heavy=heavy, edge_index=eid, edge_type=etd)
        g_, s_ = (eid[1], eid[0]) if transposed else (eid[0], eid[1])
        want = E.build_edge_plan(g_, s_, etd, w, n, r, b, en, heavy=heavy)
        assert (got.heavy is None) == (want.heavy is None)
        if got.heavy is not None:
            assert (got.heavy.n_seg, got.heavy.n_units) == (want.heavy.n_seg, want.heavy.n_units)
            for name in ("unit_rel", "unit_cnt", "slot_src", "slot_w", "slot_row"):
                assert torch.equal(getattr(got.heavy, name), getattr(want.heavy, name)), ("heavy " + name, transposed)
            for l1, l2 in zip(got.heavy.levels, want.heavy.levels):
                assert all((a is None and b_ is None) or (torch.is_tensor(a) and torch.equal(a, b_)) or a == b_ for a, b_ in zip(l1, l2))
        assert (got.n_units, got.n_rows, got.max_rows_per_dst) == (want.n_units, want.n_rows, want.max_rows_per_dst)
        for name in ("unit_rel", "unit_cnt", "slot_src", "slot_w", "slot_row"):
            assert torch.equal(getattr(got, name), getattr(want, name)), (name, transposed)
        assert len(got.levels) == len(want.levels)
        for (p1, i1, n1), (p2, i2, n2) in zip(got.levels, want.levels):
            assert n1 == n2 and torch.equal(p1, p2) and ((i1 is None and i2 is None) or torch.equal(i1, i2))
        tp = got.as_tile_plan()
        assert tp.layout == 2 and tp.n_chunks == got.n_units and torch.equal(tp.rel_order, torch.arange(got.n_units, device=dev, dtype=torch.int32))
    del keep


def test_heavy_part_of_a_few_units_on_a_graph_of_several_pseudo_tiles(dev):
    """more than 32,768 owned nodes (the heavy pseudo plan's tile bound) and ONE hub whose heavy (dst, relation)
    segments fill fewer 64-slot units than the pseudo plan has tiles -- n_chunks < n_tiles, which check_plan refused for every
    layout.  Forward on the edge-parallel path, backward with weight.requires_grad, through the module; against the float64
    oracle."""
    from scaling_rgcn_training_amd.conv import RGCNConv
    from scaling_rgcn_training_amd.plan import clear_plan_cache
    n, r, din, dout = 200_000, 4, 16, 8
    ei, et = O.synthetic_graph(n, 300_000, r, seed=21)
    g = torch.Generator().manual_seed(4)
    hub_src = torch.randint(0, n, (3 * 300,), generator=g)
    hub = torch.stack([hub_src, torch.full((900,), 777)])
    ei = torch.cat([ei, hub], 1)
    et = torch.cat([et, torch.arange(3).repeat_interleave(300)])
    w, root, bias = O.synthetic_params(r, din, dout, seed=6)
    x = torch.randn(n, din, generator=g)
    dg = torch.randn(n, dout, generator=g)
    conv = RGCNConv(din, dout, r).to(dev)
    conv.path = ("ep", "ring")
    with torch.no_grad():
        conv.weight.copy_(w)
        conv.root.copy_(root)
        conv.bias.copy_(bias)
    xd = x.to(dev).requires_grad_(True)
    eid, etd = ei.to(dev), et.to(dev)
    out = conv(xd, eid, etd)
    plans = conv._plans(xd, eid, etd)
    hv = plans.ep_fwd.heavy
    assert hv is not None and hv.n_seg == 3, "the hub's three segments are aggregated before the transform"
    tp = plans.ep_fwd.heavy_tile_plan()
    assert tp.n_chunks < tp.n_tiles, "the case check_plan used to refuse"
    out.backward(dg.to(dev))
    torch.cuda.synchronize()
    ref, gr = O.rgcn_conv_segments(x.numpy(), ei.numpy(), et.numpy(), w.numpy(), root.numpy(), bias.numpy(), dg.numpy())
    c_out, c = abs_condition(x, ei, et, w, root, bias, dg)
    assert_close(out.detach().cpu().numpy(), ref, c_out, "hub out")
    assert_close(xd.grad.cpu().numpy(), gr["x"], c["x"], "hub d_x")
    assert_close(conv.weight.grad.cpu().numpy(), gr["weight"], c["weight"], "hub d_weight")
    assert_close(conv.root.grad.cpu().numpy(), gr["root"], c["root"], "hub d_root")
    clear_plan_cache()
