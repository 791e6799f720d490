"""BASELINE.json's headline configuration (10M nodes / 100M edges / 32 relations, 64 -> 64) on the GPU:
too big for the oracle as a whole, so parity is checked (a) exactly, on a random sample of output rows whose
complete neighbourhoods are extracted and handed to the float64 oracle, and (b) through size-independent
properties: linearity in x, the bias shift, linearity of the weight gradients in dOut, run-to-run determinism."""
import numpy as np
import pytest
import torch

from oracle import rgcn_oracle as O
from oracle.tolerance import assert_close

pytestmark = pytest.mark.gpu
N, E, R, D = 10_000_000, 100_000_000, 32, 64


@pytest.fixture(scope="module")
def big():
    import bench
    from scaling_rgcn_training_amd.conv import RGCNConv
    dev = torch.device("cuda:0")
    ei, et, x, dg, weight, root = bench.synthetic_on_device(N, E, R, D, D, dev)
    conv = RGCNConv(D, D, R).to(dev)
    with torch.no_grad():
        conv.weight.copy_(weight)
        conv.root.copy_(root)
        conv.bias.copy_(torch.linspace(-0.5, 0.5, D, device=dev))
    return dict(dev=dev, ei=ei, et=et, x=x, dg=dg, conv=conv)


def _sample_reference(big, rows, direction):
    """float64 oracle restricted to `rows`: forward (direction='out': all edges INTO the rows) or dX
    (direction='dx': all edges OUT OF the rows)."""
    ei, et, conv = big["ei"], big["et"], big["conv"]
    w = conv.weight.detach().double().cpu().numpy()
    root = conv.root.detach().double().cpu().numpy()
    rows_d = torch.as_tensor(rows, device=ei.device)
    key = ei[1] if direction == "out" else ei[0]
    sel = torch.isin(key, rows_d)
    src, dst, typ = ei[0][sel], ei[1][sel], et[sel]
    # mean normaliser c[dst, rel] over ALL edges into dst
    if direction == "out":
        cnt_nodes = rows_d
    else:
        cnt_nodes = torch.unique(dst)
    in_sel = torch.isin(ei[1], cnt_nodes)
    ckey = ei[1][in_sel] * R + et[in_sel]
    uk, uc = torch.unique(ckey, return_counts=True)
    cnt = dict(zip(uk.cpu().tolist(), uc.cpu().tolist()))
    src, dst, typ = src.cpu().numpy(), dst.cpu().numpy(), typ.cpu().numpy()
    feat = big["x"] if direction == "out" else big["dg"]
    gather = src if direction == "out" else dst
    scatter = dst if direction == "out" else src
    fg = feat[torch.as_tensor(gather, device=feat.device)].double().cpu().numpy()
    out = {int(r): np.zeros(D) for r in rows}
    cond = {int(r): np.zeros(D) for r in rows}
    for k in range(len(src)):
        wk = 1.0 / cnt[int(dst[k]) * R + int(typ[k])]
        m = w[typ[k]] if direction == "out" else w[typ[k]].T
        out[int(scatter[k])] += wk * (fg[k] @ m)
        cond[int(scatter[k])] += wk * (np.abs(fg[k]) @ np.abs(m))
    own = feat[rows_d].double().cpu().numpy()
    rm = root if direction == "out" else root.T
    res = np.stack([out[int(r)] for r in rows]) + own @ rm
    cnd = np.stack([cond[int(r)] for r in rows]) + np.abs(own) @ np.abs(rm)
    if direction == "out":
        b = conv.bias.detach().double().cpu().numpy()
        res += b
        cnd += np.abs(b)
    return res, cnd


def test_full_size_sampled_rows_match_oracle(big):
    conv, x = big["conv"], big["x"]
    xg = x.clone().requires_grad_(True)
    out = conv(xg, big["ei"], big["et"])
    out.backward(big["dg"])
    torch.cuda.synchronize()
    g = torch.Generator().manual_seed(7)
    rows = torch.randint(0, N, (96,), generator=g).unique().tolist() + [0, N - 1, 383, 384]
    ref, cond = _sample_reference(big, rows, "out")
    assert_close(out[rows].detach().cpu().numpy(), ref, cond, "out rows")
    refx, condx = _sample_reference(big, rows, "dx")
    assert_close(xg.grad[rows].cpu().numpy(), refx, condx, "dX rows")
    big["out"], big["dx"] = out.detach(), xg.grad
    big["dw"], big["droot"], big["dbias"] = conv.weight.grad.clone(), conv.root.grad.clone(), conv.bias.grad.clone()
    # d_bias is the column sum of dOut; d_root = X^T dOut (float64 on the device)
    np.testing.assert_allclose(big["dbias"].cpu().numpy(), big["dg"].double().sum(0).cpu().numpy(), rtol=2e-4, atol=0.5)
    droot_ref = (x.double().T @ big["dg"].double()).cpu().numpy()
    np.testing.assert_allclose(big["droot"].cpu().numpy(), droot_ref, rtol=2e-4, atol=1.0)
    assert torch.all(torch.isfinite(big["dw"]))


def test_full_size_linearity_and_determinism(big):
    conv, x, ei, et = big["conv"], big["x"], big["ei"], big["et"]
    with torch.no_grad():
        base = conv(x, ei, et)
        assert torch.equal(base, big["out"]), "run-to-run determinism (no float atomics anywhere)"
        # f(x) - b is linear in x: f(2x) - b == 2 (f(x) - b) exactly in binary floating point
        doubled = conv(2.0 * x, ei, et)
        lhs = (doubled - conv.bias).cpu()
        rhs = (2.0 * (base - conv.bias)).cpu()
        assert torch.allclose(lhs, rhs, rtol=1e-5, atol=1e-5)
        # superposition with a second input
        y = torch.roll(x, 1, 0)
        fy = conv(y, ei, et)
        fxy = conv(x + y, ei, et)
        err = (fxy - (base + fy - conv.bias)).abs().max().item()
        assert err < 2e-4, err
