"""The oracle against its independent pins: dense formula, analytic grads,
gradcheck, and the committed golden vectors.  CPU only."""
import numpy as np
import torch

from oracle import rgcn_oracle as O


def _t(g, k, dtype=torch.float64):
    return torch.from_numpy(g[k]).to(dtype)


def _loop_from_golden(g, dtype):
    mode = str(g["mode"])
    ei = torch.from_numpy(g["edge_index"]).long()
    et = torch.from_numpy(g["edge_type"]).long()
    x = _t(g, "x", dtype).requires_grad_(True)
    w = _t(g, "weight", dtype).requires_grad_(True)
    root = _t(g, "root", dtype).requires_grad_(True)
    bias = _t(g, "bias", dtype).requires_grad_(True)
    comp = _t(g, "comp", dtype).requires_grad_(True) if mode == "basis" else None
    nb = int(g["nb"]) if mode == "block" else None
    out = O.rgcn_conv_loop(x, ei, et, w, root, bias, comp=comp, num_blocks=nb)
    return out, dict(x=x, weight=w, root=root, bias=bias, comp=comp)


def test_loop_matches_golden_fp64(golden):
    out, leaves = _loop_from_golden(golden, torch.float64)
    np.testing.assert_allclose(out.detach().numpy(), golden["out"], rtol=1e-10, atol=1e-10)
    out.backward(_t(golden, "dout"))
    np.testing.assert_allclose(leaves["x"].grad.numpy(), golden["d_x"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(leaves["root"].grad.numpy(), golden["d_root"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(leaves["bias"].grad.numpy(), golden["d_bias"], rtol=1e-9, atol=1e-9)
    if str(golden["mode"]) == "full":
        np.testing.assert_allclose(leaves["weight"].grad.numpy(), golden["d_wfull"], rtol=1e-9, atol=1e-9)


def test_loop_fp32_within_north_star_tolerance(golden):
    # the tolerance BASELINE.json states for this path: 1e-5 in fp32
    out, _ = _loop_from_golden(golden, torch.float32)
    np.testing.assert_allclose(out.detach().numpy(), golden["out"], rtol=1e-5, atol=1e-5)


def test_dead_relation_gets_zero_grad(golden):
    # relation id 2R never occurs on an edge (model/modelTrainer.py:78, graphs/graph.py:62-63)
    if str(golden["mode"]) != "full":
        return
    r_dead = int(golden["num_relations"]) - 1
    assert not (golden["edge_type"] == r_dead).any()
    assert np.all(golden["d_wfull"][r_dead] == 0.0)


def test_segments_form_matches_dense():
    ei, et = O.synthetic_graph(300, 3000, 7, seed=3)
    w, root, bias = O.synthetic_params(7, 12, 10, seed=3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(300, 12, generator=g)
    dout = torch.randn(300, 10, generator=g)
    dense = O.rgcn_conv_dense(x.numpy(), ei.numpy(), et.numpy(), w.numpy(), root.numpy(), bias.numpy())
    gd = O.rgcn_conv_grads_dense(x.numpy(), ei.numpy(), et.numpy(), w.numpy(), root.numpy(), dout.numpy())
    seg, gs = O.rgcn_conv_segments(x.numpy(), ei.numpy(), et.numpy(), w.numpy(), root.numpy(), bias.numpy(),
                                   dout.numpy())
    np.testing.assert_allclose(seg, dense, rtol=1e-12, atol=1e-12)
    for k in ("x", "weight", "root", "bias"):
        np.testing.assert_allclose(gs[k], gd[k], rtol=1e-11, atol=1e-11)


def test_gradcheck_all_modes():
    torch.manual_seed(0)
    n, r = 9, 4
    ei = torch.randint(0, n, (2, 30))
    ei[:, 5] = ei[:, 4]  # duplicate edge
    ei[1, 6] = ei[0, 6]  # self loop
    et = torch.randint(0, r - 1, (30,))  # relation r-1 left empty
    x = torch.randn(n, 6, dtype=torch.float64, requires_grad=True)
    root = torch.randn(6, 4, dtype=torch.float64, requires_grad=True)
    bias = torch.randn(4, dtype=torch.float64, requires_grad=True)
    w = torch.randn(r, 6, 4, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda *a: O.rgcn_conv_loop(a[0], ei, et, a[1], a[2], a[3]), (x, w, root, bias))
    wb = torch.randn(3, 6, 4, dtype=torch.float64, requires_grad=True)
    comp = torch.randn(r, 3, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda *a: O.rgcn_conv_loop(a[0], ei, et, a[1], a[2], a[3], comp=a[4]),
                                    (x, wb, root, bias, comp))
    wk = torch.randn(r, 2, 3, 2, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda *a: O.rgcn_conv_loop(a[0], ei, et, a[1], a[2], a[3], num_blocks=2),
                                    (x, wk, root, bias))


def test_reference_init_bounds():
    # [89,63,16] -> fan_in 1008, bound 0.07715
    g = torch.Generator().manual_seed(0)
    w, root, bias = O.reference_layer_params(89, 63, 16, g)
    assert abs(w.abs().max().item() - 0.07715) < 2e-4
    assert root.abs().max().item() <= (6.0 / (63 + 16)) ** 0.5
    assert torch.all(bias == 0)
