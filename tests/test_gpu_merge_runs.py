"""Plan layout 3 (plan.compact_runs / compact_runs_kernel): the rows of a (destination, relation) run on ONE slot, added by the
producer waves of rgcn_tile3p_kernel before the cut -- aggregate, then transform, the reference's own order
(torch_geometric RGCNConv: mean over the relation's in-edges, then W_r).  Through the C ABI against the float64 oracle and
against the same kernel on the layout-0 plan; what must not walk such a plan refuses it; through the module at a size where
the layer picks the layout by itself."""
import numpy as np
import pytest
import torch

from oracle import rgcn_oracle as O
from oracle.tolerance import abs_condition, assert_close, cpu32_reference

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _fwd_dx(dev, ei, et, n, r, x, w, root, bias, dg, tile, layout, flags=None):
    """flags None: the bf16 x 3 kernel (RGCN_FLAG_SPLIT_PRODUCERS); 0: the exact-fp32 kernel, whose producers add the shadow rows
    of a layout-3 chunk to their heads in LDS"""
    from scaling_rgcn_training_amd import _lib, plan as P
    flags = _lib.FLAG_SPLIT_PRODUCERS if flags is None else flags
    # tiles above 224: chunks of at most 112 rows (seven row tiles) -- the bf16 x 3 kernel's 42 KiB ring slots
    plans = P.build_graph_plans(ei.to(dev), et.to(dev), n, r, tile, "mean", chunk=112 if tile > 224 and tile <= 272 else 128, split=layout)
    xd, gd = x.to(dev).contiguous(), dg.to(dev).contiguous()
    wd, rd, bd = w.to(dev).contiguous(), root.to(dev).contiguous(), bias.to(dev).contiguous()
    out = torch.full((n, 64), float("nan"), device=dev)
    _lib.fwd(_lib.plan_struct(plans.fwd), xd, 64, _lib.pack_weights(wd, rd, False), bd, out, 64, 0, flags)
    dx = torch.full((n, 64), float("nan"), device=dev)
    _lib.bwd_dx(_lib.plan_struct(plans.bwd), gd, 64, _lib.pack_weights(wd, rd, True), dx, 64, None, flags)
    torch.cuda.synchronize()
    return out.cpu().numpy(), dx.cpu().numpy(), plans


@pytest.mark.parametrize("kernel", ["bf16x3", "fp32"])
@pytest.mark.parametrize("n,e,r,tile,skew,some", [(20000, 60000, 32, 224, False, True), (3000, 30000, 32, 224, False, True),
                                                  (6000, 60000, 32, 224, False, True), (6000, 200000, 32, 224, False, False),
                                                  (4000, 40000, 16, 128, True, None), (300, 2500, 8, 64, False, None),
                                                  (9000, 90000, 32, 288, False, True), (9000, 90000, 32, 272, False, True),
                                                  (6000, 250000, 32, 272, False, None)])
def test_merged_runs_match_the_oracle_and_layout_0(dev, n, e, r, tile, skew, some, kernel):
    """graphs whose (tile, relation) groups fit one chunk (compacted: runs of 2 and 3, second and third rows in the shadow row
    tiles), groups of several chunks and hub rows (left in layout 0), duplicate triples (unequal weights inside a run: left
    alone): every chunk is right either way"""
    ei, et = O.synthetic_graph(n, e, r, seed=n + r, skew=skew)
    ei[:, 100:160] = ei[:, 20:80]           # duplicate triples: merged slots with weight 2 / c inside runs of weight 1 / c
    et[100:160] = et[20:80]
    w, root, bias = O.synthetic_params(r, 64, 64, seed=5)
    g = torch.Generator().manual_seed(23)
    bias = torch.randn(64, generator=g) * 0.1
    x = torch.randn(n, 64, generator=g)
    dg = torch.randn(n, 64, generator=g)
    ref, gr = O.rgcn_conv_segments(x.numpy(), ei.numpy(), et.numpy(), w.numpy(), root.numpy(), bias.numpy(), dg.numpy())
    kf = None if kernel == "bf16x3" else 0
    if kernel == "bf16x3" and tile > 272:
        pytest.skip("the bf16 x 3 kernel's ring leaves room for tiles up to 272 (chunks of at most 112 rows)")
    out3, dx3, p3 = _fwd_dx(dev, ei, et, n, r, x, w, root, bias, dg, tile, 3, kf)
    out0, dx0, p0 = _fwd_dx(dev, ei, et, n, r, x, w, root, bias, dg, tile, 0, kf)
    assert p3.fwd.layout == 3 and p3.bwd.layout == 3
    for a, b in ((p3.fwd, p0.fwd), (p3.bwd, p0.bwd)):
        merged = int((((a.chunk_flags >> 16) & 7) != 0).sum())
        if some is True:
            assert merged > 0 and int(a.chunk_cnt.sum()) < int(b.chunk_cnt.sum())
        elif some is False:       # every group spans several chunks: nothing is compacted, the plan is layout 0's
            assert merged == 0 and torch.equal(a.slot_src, b.slot_src)
        assert int((a.slot_w != 0).sum()) == int((b.slot_w != 0).sum()), "every edge row keeps a slot (head or shadow)"
    c_out, c = abs_condition(x, ei, et, w, root, bias, dg)
    o32, g32 = cpu32_reference(x, ei, et, w, root, bias, dg)
    assert_close(out3, ref, c_out, f"out [layout 3 T{tile} n{n}]", cpu32=o32)
    assert_close(dx3, gr["x"], c["x"], f"d_x [layout 3 T{tile} n{n}]", cpu32=g32["x"])
    for a, b in ((out3, out0), (dx3, dx0)):
        assert np.max(np.abs(a - b)) <= 2e-5 * max(1.0, float(np.abs(b).max()))
    again = _fwd_dx(dev, ei, et, n, r, x, w, root, bias, dg, tile, 3, kf)
    assert np.array_equal(out3, again[0]) and np.array_equal(dx3, again[1])


def test_what_does_not_add_the_shadow_rows_refuses_a_layout_3_plan(dev):
    from scaling_rgcn_training_amd import _lib, plan as P
    n, e, r = 3000, 30000, 32
    ei, et = O.synthetic_graph(n, e, r, seed=1)
    w, root, bias = O.synthetic_params(r, 64, 64, seed=1)
    plans = P.build_graph_plans(ei.to(dev), et.to(dev), n, r, 224, "mean", chunk=128, split=3)
    x = torch.randn(n, 64, device=dev)
    out = torch.empty(n, 64, device=dev)
    pk = _lib.pack_weights(w.to(dev), root.to(dev), False)
    # (the 64 x 64 forms of both forward / dX kernels walk it: test above)  A narrower layer's kernel, and the 64-bit pointer
    # gathers of the exact-fp32 kernel, do not add the shadow rows
    w32, root32 = O.synthetic_params(r, 32, 64, seed=1)[:2]
    with pytest.raises(_lib.RgcnLibraryError) as err:
        _lib.fwd(_lib.plan_struct(plans.fwd), x[:, :32].contiguous(), 32, _lib.pack_weights(w32.to(dev), root32.to(dev), False), None, out, 64, 0, 0)
    assert err.value.status == _lib.ERR_PLAN
    with pytest.raises(_lib.RgcnLibraryError) as err:
        _lib.fwd(_lib.plan_struct(plans.fwd), x, 64, pk, None, out, 64, 0, _lib.FLAG_POINTER_GATHER)
    assert err.value.status == _lib.ERR_PLAN
    dw, dr, db = torch.empty(r, 64, 64, device=dev), torch.empty(64, 64, device=dev), torch.empty(64, device=dev)
    with pytest.raises(_lib.RgcnLibraryError) as err:          # nor do the relation-major weight-gradient kernels
        _lib.bwd_dw(_lib.plan_struct(plans.fwd), x, 64, x, 64, dw, dr, db, 0)
    assert err.value.status == _lib.ERR_PLAN
    with pytest.raises(_lib.RgcnLibraryError) as err:          # nor the tile-major one (its walk table and its launch: layout 0 only)
        _lib.dw_tiles_walk(_lib.plan_struct(plans.fwd), dev)
    assert err.value.status == _lib.ERR_PLAN
    walk = torch.zeros(r, _lib.dw_tiles_geometry()[1] + 1, dtype=torch.int32, device=dev)
    with pytest.raises(_lib.RgcnLibraryError) as err:
        _lib.bwd_dw_tiles(_lib.plan_struct(plans.fwd), walk, x, 64, x, 64, dw, 0)
    assert err.value.status == _lib.ERR_PLAN


@pytest.mark.parametrize("split_producers", [True, False], ids=["bf16x3", "fp32"])
def test_the_module_picks_layout_3_and_agrees_with_layout_0(dev, split_producers):
    """a graph large enough for the tile-major d_weight kernel (the condition under which nothing but the 64 x 64 forward / dX
    kernels walks the forward / transposed plans): merge_runs on (default) and off through autograd, frozen weights included,
    on the bf16 x 3 kernel and on the exact-fp32 one"""
    from scaling_rgcn_training_amd import conv as C
    from scaling_rgcn_training_amd.conv import RGCNConv
    n, e, r = 300_000, 4_200_000, 32
    ei, et = O.synthetic_graph(n, e, r, seed=8)
    ei, et = ei.to(dev), et.to(dev)
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(n, 64, generator=g).to(dev)
    dg = torch.randn(n, 64, generator=g).to(dev)
    res = {}
    for merge in (True, False):
        torch.manual_seed(0)
        conv = RGCNConv(64, 64, r).to(dev)
        conv.merge_runs = merge
        conv.split_producers = split_producers
        x = x0.clone().requires_grad_(True)
        plans = conv._plans(x, ei, et)
        if not split_producers:       # the exact-fp32 kernel takes the tile that leaves its chunks room for the shadow row tiles
            assert plans.fwd.tile == conv.layout(n, e)[0] and plans.fwd.chunk == 128 and (plans.fwd.tile <= C.EXACT_MERGE_TILE or not merge)
        assert plans.fwd.layout == (3 if merge else 0) and plans.bwd.layout == plans.fwd.layout and plans.dw is not None
        out = conv(x, ei, et)
        out.backward(dg)
        res[merge] = [out.detach(), x.grad, conv.weight.grad, conv.root.grad, conv.bias.grad]
        if merge:       # frozen relation weights: d_root / d_bias must not fall back to a walk over the layout-3 plan
            conv.weight.requires_grad_(False)
            conv.root.grad = conv.bias.grad = None
            conv(x0, ei, et).backward(dg)
            assert torch.equal(conv.root.grad, res[True][3]) and torch.equal(conv.bias.grad, res[True][4])
    for a, b in zip(res[True][:2], res[False][:2]):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
    for a, b in zip(res[True][2:], res[False][2:]):
        assert torch.equal(a, b)          # the weight gradients never see the forward / transposed plans


def test_layout_3_needs_128_slot_chunks(dev):
    from scaling_rgcn_training_amd import _lib, plan as P
    ei, et = O.synthetic_graph(500, 4000, 4, seed=2)
    with pytest.raises(_lib.RgcnLibraryError) as err:
        P.build_graph_plans_device(ei.to(dev), et.to(dev), 500, 4, 64, "mean", None, None, 64, split=3)
    assert err.value.status == _lib.ERR_PLAN
    with pytest.raises(ValueError):
        P.build_graph_plans_torch(ei, et, 500, 4, 64, "mean", None, None, 64, split=3)
