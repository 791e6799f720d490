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


@pytest.fixture(scope="module", params=["uniform", "skew"])
def big(request):
    """uniform: SURVEY.md 8d's headline inputs; skew: its 'skew' variant (dst ~ Zipf(1.2)-tailed mod N: node 0 receives 13 % of
    the edges, the first 224 nodes two thirds) -- the module's forward then takes the edge-parallel path, dX the tile kernel."""
    import bench
    from scaling_rgcn_training_amd.conv import RGCNConv
    from scaling_rgcn_training_amd.plan import clear_plan_cache
    clear_plan_cache()
    torch.cuda.empty_cache()
    dev = torch.device("cuda:0")
    ei, et, x, dg, weight, root = bench.synthetic_on_device(N, E, R, D, D, dev, skew=request.param == "skew")
    conv = RGCNConv(D, D, R).to(dev)
    with torch.no_grad():
        conv.weight.copy_(weight)
        conv.root.copy_(root)
        conv.bias.copy_(torch.linspace(-0.5, 0.5, D, device=dev))
    return dict(dev=dev, ei=ei, et=et, x=x, dg=dg, conv=conv, kind=request.param)


def _row_reference_on_device(big, row):
    """float64 forward value (and its condition number) of ONE output row with any number of in-edges, by plain torch ops on
    the device: per relation the sum of the gathered rows / count, times W_r; + x[row] @ root + bias."""
    ei, et, conv, x = big["ei"], big["et"], big["conv"], big["x"]
    sel = torch.nonzero(ei[1] == row).squeeze(1)
    s, t = ei[0][sel], et[sel]
    c = torch.bincount(t, minlength=R).double().clamp(min=1.0)
    h = torch.zeros(R, D, dtype=torch.float64, device=x.device)
    ha = torch.zeros_like(h)
    for lo in range(0, int(sel.numel()), 1 << 22):                     # 4M rows at a time: 2 GiB of float64 rows
        xs = x[s[lo:lo + (1 << 22)]].double()
        h.index_add_(0, t[lo:lo + (1 << 22)], xs)
        ha.index_add_(0, t[lo:lo + (1 << 22)], xs.abs())
    w, root, b = conv.weight.detach().double(), conv.root.detach().double(), conv.bias.detach().double()
    own = x[row].double()
    val = torch.einsum("rk,rkn->n", h / c[:, None], w) + own @ root + b
    cnd = torch.einsum("rk,rkn->n", ha / c[:, None], w.abs()) + own.abs() @ root.abs() + b.abs()
    return val.cpu().numpy(), cnd.cpu().numpy(), int(sel.numel())


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
    if big["kind"] == "skew":      # the row-by-row reference loop takes rows of ordinary in-degree; the hubs are checked below
        deg_rows = torch.bincount(big["ei"][1], minlength=N)[torch.as_tensor(rows, device=big["dev"])].tolist()
        rows = [r for r, dgr in zip(rows, deg_rows) if dgr <= 5000]
    ref, cond = _sample_reference(big, rows, "out")
    assert_close(out[rows].detach().cpu().numpy(), ref, cond, "out rows")
    refx, condx = _sample_reference(big, rows, "dx")
    assert_close(xg.grad[rows].cpu().numpy(), refx, condx, "dX rows")
    if big["kind"] == "skew":
        # the ten destinations with the most in-edges (millions of rows each: their (destination, relation) segments are summed in
        # levels BEFORE the transform, one pseudo row per segment goes through it)
        from scaling_rgcn_training_amd.plan import cached_graph_plans
        plans = conv._plans(xg, big["ei"], big["et"])
        epf = plans.ep_fwd
        assert epf is not None and epf.heavy is not None, "the hub graph's forward runs the edge-parallel path, its hubs pre-aggregated"
        assert len(epf.heavy.levels) >= 3, "segments of hundreds of thousands of rows are summed in levels"
        deg = torch.bincount(big["ei"][1], minlength=N)
        hubs = torch.topk(deg, 10).indices.tolist()
        for h in hubs:
            val, cnd, n_in = _row_reference_on_device(big, h)
            assert_close(out[h].detach().cpu().numpy()[None], val[None], cnd[None], f"hub row {h} ({n_in} in-edges)")
        del deg
    big["out"], big["dx"] = out.detach(), xg.grad
    big["dw"], big["droot"], big["dbias"] = conv.weight.grad.clone(), conv.root.grad.clone(), conv.bias.grad.clone()
    # ---- weight gradients at full size, against float64 on the device with plain torch ops (independent of the
    # plan): d_bias = column sums of dOut, d_root = X^T dOut, d_weight[r] = H_r^T dOut for the first relation, the
    # last one and a random one (~3.1M edges each).  Criterion: the a-priori bound of oracle/tolerance.py (flat 1e-5
    # + 4 u cond) AND no worse than 2.5 x the STOCK fp32 path on the same sums (measured: d_weight 0.6 x, d_bias 2.0 x) -- rocBLAS / ATen fp32 evaluations of the
    # same products stand in for the reference's CPU loop, which cannot run at this size.
    from oracle.tolerance import SLACK_LOG
    dg, ei, et = big["dg"], big["ei"], big["et"]
    U = 2.0 ** -24

    def check(name, got, ref64, cond64, stock32):
        err = (got.double() - ref64).abs()
        flat = 1e-5 + 1e-5 * ref64.abs()
        assert torch.all(err <= flat + 4 * U * cond64), (name, float((err - flat - 4 * U * cond64).max()))
        excess, stock_err = float((err - flat).max()), float((stock32.double() - ref64).abs().max())
        assert excess <= 2.5 * stock_err, f"{name}: excess over flat 1e-5 {excess:.3e} > 2.5 x the stock fp32 path's error {stock_err:.3e}"
        SLACK_LOG.append((f"full-size {name}", excess, stock_err))

    check("d_bias", big["dbias"], dg.double().sum(0), dg.double().abs().sum(0), dg.sum(0))
    check("d_root", big["droot"], x.double().T @ dg.double(), x.double().abs().T @ dg.double().abs(), x.T @ dg)
    cnt = torch.bincount(ei[1] * R + et, minlength=N * R)
    refs = {}
    for r in (0, R - 1, 13):
        idx = torch.nonzero(et == r).squeeze(1)
        s, d = ei[0][idx], ei[1][idx]
        we = 1.0 / cnt[d * R + r].double()
        h64 = x[s].double() * we[:, None]
        g64 = dg[d].double()
        ref_r = h64.T @ g64
        check(f"d_weight[{r}]", big["dw"][r], ref_r, h64.abs().T @ g64.abs(), (x[s] * we.float()[:, None]).T @ dg[d])
        refs[r] = ref_r
        del h64, g64
    plans = conv._plans(xg, ei, et)
    if getattr(plans, "dw", None) is not None:
        # The tile-major kernel in BOTH forms on the same plan, against float64.  Round 3 found the bf16 x 3 form the less accurate
        # one at this size (worst error 2.1 x the exact form's, a mean signed error growing with the rows per slab:
        # v_mfma_f32_16x16x32_bf16 TRUNCATES its aligned addends next to a large accumulator).  Round 4: both forms fold the
        # accumulator into the wave's slab every 128 units, the split form with alternating signs (csrc/rgcn_dw_tile.hip) -- the
        # bias is gone (tools/debug/dw_split_error_probe.py, profiles/r04a_*), both forms are blocked sums, and the split form
        # has to be as accurate as the exact one: worst error within 1.5 x, |mean signed error| within 2 x + the noise floor.
        from scaling_rgcn_training_amd import _lib
        dw_s, dw_e = torch.empty_like(big["dw"]), torch.empty_like(big["dw"])
        _lib.bwd_dw_tiles(_lib.plan_struct(plans.dw), plans.dw_walk, x, D, dg, D, dw_s, _lib.FLAG_SPLIT_PRODUCERS)
        _lib.bwd_dw_tiles(_lib.plan_struct(plans.dw), plans.dw_walk, x, D, dg, D, dw_e, 0)
        for r, ref_r in refs.items():
            es, ee = float((dw_s[r].double() - ref_r).abs().max()), float((dw_e[r].double() - ref_r).abs().max())
            print(f"full-size d_weight[{r}]: worst error against float64, bf16 x 3 form {es:.3e}, exact-fp32 form {ee:.3e}, ratio {es / max(ee, 1e-30):.2f}")
            ms, me = float((dw_s[r].double() - ref_r).mean()), float((dw_e[r].double() - ref_r).mean())
            print(f"full-size d_weight[{r}]: mean signed error, bf16 x 3 form {ms:+.3e}, exact-fp32 form {me:+.3e}")
            assert es <= 1.5 * ee, f"d_weight[{r}]: bf16 x 3 form {es:.3e} vs exact-fp32 form {ee:.3e} against float64"
            assert abs(ms) <= 2.0 * abs(me) + 0.1 * ee, f"d_weight[{r}]: mean signed error {ms:+.3e} (bf16 x 3) vs {me:+.3e} (exact fp32): a bias"


def test_full_size_every_row_by_a_second_evaluation_and_column_checksums(big):
    """The sampled rows above leave ~10M rows to the properties below ("what's weak").  Two whole-output checks:
    (a) EVERY element of the forward and of dX against a second evaluation that shares nothing with the first but the input:
        the exact-fp32 tile kernel (v_mfma_f32_16x16x4_f32) on layout-0 plans of its own tile size, where the module's default is
        the bf16 x 3 kernel on layout-3 plans with 112-row chunks (uniform graph) or the edge-parallel path (hub graph) -- an
        error would have to be made twice, by different kernels walking different plans, to pass;
    (b) the column sums of the whole output in float64 against their closed form from the edge list by plain torch ops:
        sum_dst out = sum_r (sum_{e in r} x[src_e] / c[dst_e, r]) W_r + (sum_i x_i) root + N bias, and the same for dX with dOut and
        W_r^T -- every edge exactly once, with the reference's mean normaliser (duplicates counted)."""
    from scaling_rgcn_training_amd.conv import RGCNConv
    conv, x, ei, et, dg, dev = big["conv"], big["x"], big["ei"], big["et"], big["dg"], big["dev"]
    out, dx = big["out"], big["dx"]
    # ---- (a)
    other = RGCNConv(D, D, R).to(dev)
    other.split_producers, other.merge_runs, other.path = False, False, "ring"
    with torch.no_grad():
        for q, v in zip(other.parameters(), conv.parameters()):
            q.copy_(v)
    xg = x.clone().requires_grad_(True)
    out2 = other(xg, ei, et)
    out2.backward(dg)
    plans = other._plans(xg, ei, et)
    assert plans.fwd is not None and plans.fwd.layout == 0 and plans.bwd.layout == 0, "the second evaluation walks layout-0 plans"
    for name, got, ref in (("out", out, out2.detach()), ("d_x", dx, xg.grad)):
        diff = (got - ref).abs()
        bound = 1e-5 + 1e-5 * ref.abs()
        if big["kind"] == "skew":      # rows that sum millions of terms in fp32, in two different orders
            bound = bound * 10.0
        worst = float((diff - bound).max())
        assert worst <= 0.0, f"{name}: {int((diff > bound).sum())} of {diff.numel()} elements differ by more than the bound (worst excess {worst:.3e})"
        del diff, bound
    del out2, xg, other, plans
    torch.cuda.empty_cache()
    # ---- (b)
    key = ei[1] * R + et
    cnt = torch.bincount(key, minlength=N * R)
    w_e = 1.0 / cnt[key].double()
    del key, cnt
    W, root, bias = conv.weight.detach().double(), conv.root.detach().double(), conv.bias.detach().double()
    s_f = torch.zeros(R, D, dtype=torch.float64, device=dev)
    s_b = torch.zeros(R, D, dtype=torch.float64, device=dev)
    for lo in range(0, E, 1 << 22):
        sl = slice(lo, lo + (1 << 22))
        s_f.index_add_(0, et[sl], x[ei[0][sl]].double() * w_e[sl, None])
        s_b.index_add_(0, et[sl], dg[ei[1][sl]].double() * w_e[sl, None])
    want_f = torch.einsum("rk,rkn->n", s_f, W) + x.double().sum(0) @ root + N * bias
    want_b = torch.einsum("rn,rkn->k", s_b, W) + dg.double().sum(0) @ root.t()
    for name, got, want in (("out", out, want_f), ("d_x", dx, want_b)):
        col = got.double().sum(0)
        tol = 1e-6 * got.double().abs().sum(0) + 1e-3       # ~ sqrt(N) roundings of size u |value| each would be far below this
        err = (col - want).abs()
        assert bool((err <= tol).all()), f"column sums of {name}: worst {float((err / tol).max()):.3f} x the tolerance"


def test_full_size_weight_gradients_linear_in_dout(big):
    """d_weight / d_root / d_bias are linear in dOut: grads(2 g) == 2 grads(g) bit for bit (scaling by two is exact in
    binary floating point and every kernel sums in a fixed order), and grads(g1 + g2) == grads(g1) + grads(g2) up to
    fp32 rounding of the sums."""
    conv, x, ei, et = big["conv"], big["x"], big["ei"], big["et"]

    def grads(g):
        xg = x.detach().requires_grad_(False)
        conv.zero_grad()
        conv(xg, ei, et).backward(g)
        return [conv.weight.grad.clone(), conv.root.grad.clone(), conv.bias.grad.clone()]

    g1 = big["dg"]
    g2 = torch.roll(g1, 7, 0)
    a, b, two, both = grads(g1), grads(g2), grads(2.0 * g1), grads(g1 + g2)
    for base, ref in zip(a, (big["dw"], big["droot"], big["dbias"])):
        assert torch.equal(base, ref), "run-to-run determinism of the weight gradients"
    for u, v in zip(two, a):
        assert torch.equal(u, 2.0 * v)
    for u, v, w in zip(both, a, b):
        scale = float(torch.maximum(v.abs(), w.abs()).max())
        assert float((u - (v + w)).abs().max()) <= 2e-5 * scale + 1e-3, float((u - (v + w)).abs().max())


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
