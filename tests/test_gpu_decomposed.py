"""PyG's two weight decompositions inside the library (BASELINE.json configs[2]:
basis decomposition B = 30): rgcn_pack_weights_basis / _block compose W_r inside the packer, rgcn_basis_backward /
rgcn_block_backward turn the dense d_W scratch into the gradients of the layer's own parameters -- against torch autograd of
``RGCNConv.effective_weight()`` (the differentiable torch form, which is no longer on the forward path)."""
import numpy as np
import pytest
import torch

from oracle import rgcn_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the GPU box"
    from scaling_rgcn_training_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.mark.parametrize("mode,din,dout", [("basis", 32, 32), ("basis", 63, 16), ("basis", 64, 64), ("block", 32, 32), ("block", 64, 64),
                                           ("block", 24, 12)])
@pytest.mark.parametrize("transpose", [False, True])
def test_decomposed_pack_equals_pack_of_the_composed_weights(dev, mode, din, dout, transpose):
    from scaling_rgcn_training_amd import _lib
    from scaling_rgcn_training_amd.conv import RGCNConv
    torch.manual_seed(3)
    r = 11
    conv = RGCNConv(din, dout, r, **({"num_bases": 5} if mode == "basis" else {"num_blocks": 4})).to(dev)
    dense = conv.effective_weight().detach().contiguous()
    a = _lib.pack_weights(dense, conv.root.detach(), transpose)
    b = _lib.pack_weights_decomposed(conv.weight.detach(), None if conv.comp is None else conv.comp.detach(), conv.root.detach(), r, din, dout, transpose)
    torch.cuda.synchronize()
    n32 = (r + 1) * _lib.padded_width(din) * _lib.padded_width(dout)
    # fp32 fragments: the packer sums b = 0 .. B - 1 with fused multiply-adds where torch runs a GEMM (block: exact copies)
    np.testing.assert_allclose(b[:n32].cpu().numpy(), a[:n32].cpu().numpy(), rtol=2e-6, atol=2e-7)
    if b.numel() > n32:      # bf16 x 3 planes of 64 x 64 layers: pieces of values that may differ in the last bit
        pa, pb = a[n32:].view(torch.bfloat16).float(), b[n32:].view(torch.bfloat16).float()      # (-0.0 == +0.0: torch's block form multiplies by zeros)
        assert float((pa != pb).float().mean()) < (0.5 if mode == "basis" else 1e-9)


@pytest.mark.parametrize("mode", ["basis", "block"])
@pytest.mark.parametrize("path", ["ring", "ep"])
def test_decomposed_layer_gradients_match_autograd_of_the_composed_weights(dev, mode, path):
    """forward + all gradients of a basis / block layer through the module == the same layer with the dense weights that
    ``effective_weight()`` composes by torch ops, differentiated by autograd down to weight / comp"""
    from scaling_rgcn_training_amd.conv import RGCNConv, rgcn_conv_function
    n, e, r, din, dout = 3000, 40000, 23, 32, 32
    ei, et = O.synthetic_graph(n, e, r, seed=9)
    eid, etd = ei.to(dev), et.to(dev)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, din, generator=g).to(dev)
    dg = torch.randn(n, dout, generator=g).to(dev)
    torch.manual_seed(4)
    conv = RGCNConv(din, dout, r, **({"num_bases": 6} if mode == "basis" else {"num_blocks": 4})).to(dev)
    conv.path = path
    with torch.no_grad():
        conv.bias.uniform_(-0.2, 0.2)
    xa = x.clone().requires_grad_(True)
    out = conv(xa, eid, etd)
    out.backward(dg)
    got = [out.detach(), xa.grad, conv.weight.grad.clone(), None if conv.comp is None else conv.comp.grad.clone(), conv.root.grad.clone(),
           conv.bias.grad.clone()]
    conv.zero_grad()
    xb = x.clone().requires_grad_(True)
    ref_out = rgcn_conv_function(xb, conv.effective_weight(), conv.root, conv.bias, conv._plans(xb, eid, etd))
    ref_out.backward(dg)
    ref = [ref_out.detach(), xb.grad, conv.weight.grad, None if conv.comp is None else conv.comp.grad, conv.root.grad, conv.bias.grad]
    for name, a, b in zip(("out", "d_x", "d_weight", "d_comp", "d_root", "d_bias"), got, ref):
        if b is None:
            assert a is None
            continue
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=2e-5 * max(1.0, float(b.abs().max())), err_msg=name)
    # frozen decomposition parameters (model/layers.py:33-46 freezes transferred weights): no gradient, no crash
    conv.zero_grad()
    conv.weight.requires_grad_(False)
    xc = x.clone().requires_grad_(True)
    conv(xc, eid, etd).backward(dg)
    assert conv.weight.grad is None and torch.equal(xc.grad, xa.grad)
