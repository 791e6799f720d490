"""numpy walk over a TilePlan exactly as the HIP kernels walk it (test helper, CPU only).

fwd / dX kernel: for every chunk, rows = slots; out[tile*T + dstl] += w * (x[src] @ Wp[rel])
dW kernel     : dWp[rel] += (w * x[src])^T @ g[tile*T + dstl]
where Wp = concat(weight[R'], root[None]) -- the root is relation id R'."""
import numpy as np



def _slots(plan):
    src = plan.slot_src.cpu().numpy().astype(np.int64)
    w = plan.slot_w.cpu().numpy().astype(np.float64)
    rel = np.repeat(plan.chunk_rel.cpu().numpy().astype(np.int64), plan.chunk)
    valid = src < plan.n_nodes
    node = plan.slot_row.cpu().numpy().astype(np.int64)  # row of the owned range (local to node_begin)
    return src, w, rel, node, valid


def emulate_spmm(plan, x, w_all, bias=None):
    src, w, rel, node, valid = _slots(plan)
    x = np.asarray(x, np.float64)
    w_all = np.asarray(w_all, np.float64)
    out = np.zeros((plan.n_owned, w_all.shape[2]))
    if bias is not None:
        out += np.asarray(bias, np.float64)[None, :]
    for r in range(w_all.shape[0]):
        sel = valid & (rel == r)
        if sel.any():
            np.add.at(out, node[sel], (x[src[sel]] * w[sel, None]) @ w_all[r])
    return out


def emulate_dw(plan, x, g_owned, n_rel_all, din, dout):
    """The weight-gradient kernels' walk: 64-row units in ``rel_order`` (relation-major); a unit's relation and
    row count come from its chunk."""
    src, w, rel, node, valid = _slots(plan)
    x = np.asarray(x, np.float64)
    g = np.asarray(g_owned, np.float64)
    dw = np.zeros((n_rel_all, din, dout))
    upc = plan.chunk // 64
    cnt = plan.chunk_cnt.cpu().numpy().astype(np.int64)
    crel = plan.chunk_rel.cpu().numpy().astype(np.int64)
    seen = np.zeros(src.shape[0], bool)
    for u in plan.rel_order.cpu().numpy().astype(np.int64):
        chunk, h = u // upc, u % upc
        n = min(max(cnt[chunk] - 64 * h, 0), 64)
        assert n > 0, "empty unit in rel_order"
        rows = np.arange(64 * u, 64 * u + n)
        seen[rows] = True
        sel = rows[valid[rows]]
        xs = x[src[sel]].copy()
        if getattr(plan, "slot_src2", None) is not None:      # plan layout 5: the second row of a pair, added before the product
            s2 = plan.slot_src2.cpu().numpy().astype(np.int64)[8 * u:8 * u + 8]
            for k_, pos in enumerate((0, 1, 2, 3, 32, 33, 34, 35)):
                if s2[k_] < plan.n_nodes:
                    xs[np.nonzero(sel == 64 * u + pos)[0][0]] += x[s2[k_]]
        dw[crel[chunk]] += (xs * w[sel, None]).T @ g[node[sel]]
    assert not (valid & ~seen).any(), "rel_order misses slots"
    return dw
