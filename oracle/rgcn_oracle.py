"""CPU oracle for the R-GCN layer hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package never does (it fails loudly when
the HIP library is missing instead of falling back to this code).

PARITY UNPINNED.  The arithmetic of the reference's hot path is not in the
reference repo: it is ``torch_geometric.nn.RGCNConv`` (torch_geometric==2.3.1,
/root/reference/requirements.txt:7) called from /root/reference/model/layers.py:15-16,
21-23 (and :54-55, 62-64, 98-99, 108-110).  PyG is neither vendored under
/root/reference nor installed here and the reference ships no tests or golden
vectors.  This file restates the published algorithm of
the non-``pyg_lib`` branch of PyG 2.3.1 ``rgcn_conv.py``:

    out = 0
    for r in range(num_relations):
        mask = edge_type == r ; e = edge_index[:, mask]
        h = scatter_mean(x[e[0]], e[1], dim_size=N)      # sum / clamp(count, 1)
        out = out + h @ W_r
    out = out + x @ root + bias

with W_r = weight[r]                         (full mode, the only one the reference uses,
                                              model/layers.py:15 ``num_bases=None``)
     W_r = (comp @ weight.view(B,-1))[r]     (basis mode)
     per-block einsum                        (block-diagonal mode)

What pins it instead of reference vectors: an INDEPENDENT dense evaluation of
``sum_r D_r^-1 A_r X W_r + X root + b`` (``rgcn_conv_dense`` below, float64, no
shared code with the loop form), analytic gradients (``rgcn_conv_grads_dense``)
checked against autograd of the loop form, and ``torch.autograd.gradcheck``; see
tests/test_oracle.py.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
from torch import Tensor


# --------------------------------------------------------------------------------------
# Loop form (the ops the reference executes through PyG) -- differentiable, any dtype
# --------------------------------------------------------------------------------------
def effective_weight(weight: Tensor, comp: Optional[Tensor], num_relations: int,
                     num_blocks: Optional[int], in_channels: int, out_channels: int) -> Tensor:
    """Dense [R, in, out] relation weights for the three PyG weight modes (Appendix A)."""
    if comp is not None:  # basis: weight [B, in, out], comp [R, B]
        nb = weight.shape[0]
        return (comp @ weight.reshape(nb, -1)).reshape(num_relations, in_channels, out_channels)
    if num_blocks is not None:  # block diagonal: weight [R, nb, in/nb, out/nb]
        r, nb, bi, bo = weight.shape
        w = weight.new_zeros(r, in_channels, out_channels)
        for b in range(nb):
            w[:, b * bi:(b + 1) * bi, b * bo:(b + 1) * bo] = weight[:, b]
        return w
    return weight


def rgcn_conv_loop(x: Tensor, edge_index: Tensor, edge_type: Tensor, weight: Tensor,
                   root: Optional[Tensor], bias: Optional[Tensor], comp: Optional[Tensor] = None,
                   num_blocks: Optional[int] = None, aggr: str = "mean") -> Tensor:
    """PyG 2.3.1 RGCNConv.forward, per-relation loop (model/layers.py:21,23 call sites)."""
    n, in_c = x.shape
    num_rel = comp.shape[0] if comp is not None else weight.shape[0]
    if num_blocks is not None:
        out_c = weight.shape[1] * weight.shape[3]
    else:
        out_c = weight.shape[-1]
    src, dst = edge_index[0], edge_index[1]
    out = x.new_zeros(n, out_c)
    w_full = None
    if comp is not None:
        w_full = effective_weight(weight, comp, num_rel, None, in_c, out_c)
    for r in range(num_rel):
        mask = edge_type == r
        s, d = src[mask], dst[mask]
        xj = x.index_select(0, s)
        h = x.new_zeros(n, in_c).index_add_(0, d, xj)
        if aggr == "mean":
            cnt = x.new_zeros(n).index_add_(0, d, x.new_ones(d.shape[0])).clamp_(min=1)
            h = h / cnt.unsqueeze(1)
        elif aggr not in ("sum", "add"):
            raise ValueError(aggr)
        if num_blocks is not None:
            nb = weight.shape[1]
            hb = h.view(n, nb, in_c // nb)
            out = out + torch.einsum("abc,bcd->abd", hb, weight[r]).reshape(n, out_c)
        elif w_full is not None:
            out = out + h @ w_full[r]
        else:
            out = out + h @ weight[r]
    if root is not None:
        out = out + x @ root
    if bias is not None:
        out = out + bias
    return out


# --------------------------------------------------------------------------------------
# Independent dense formula, float64 numpy (ground truth for small N)
# --------------------------------------------------------------------------------------
def _dense_norm_adj(n: int, src: np.ndarray, dst: np.ndarray, typ: np.ndarray, r: int,
                    aggr: str) -> np.ndarray:
    a = np.zeros((n, n), dtype=np.float64)
    sel = typ == r
    np.add.at(a, (dst[sel], src[sel]), 1.0)  # multiplicities count (duplicate edges)
    if aggr == "mean":
        deg = a.sum(axis=1)
        a = a / np.maximum(deg, 1.0)[:, None]
    return a


def rgcn_conv_dense(x, edge_index, edge_type, w_full, root, bias, aggr: str = "mean") -> np.ndarray:
    """out = sum_r D_r^-1 A_r X W_r + X root + b with dense N x N matrices, float64."""
    x = np.asarray(x, dtype=np.float64)
    w_full = np.asarray(w_full, dtype=np.float64)
    src = np.asarray(edge_index[0]).astype(np.int64)
    dst = np.asarray(edge_index[1]).astype(np.int64)
    typ = np.asarray(edge_type).astype(np.int64)
    n = x.shape[0]
    out = np.zeros((n, w_full.shape[2]), dtype=np.float64)
    for r in range(w_full.shape[0]):
        out += _dense_norm_adj(n, src, dst, typ, r, aggr) @ x @ w_full[r]
    if root is not None:
        out += x @ np.asarray(root, dtype=np.float64)
    if bias is not None:
        out += np.asarray(bias, dtype=np.float64)[None, :]
    return out


def rgcn_conv_grads_dense(x, edge_index, edge_type, w_full, root, dout, aggr: str = "mean") -> Dict[str, np.ndarray]:
    """Analytic gradients, dense float64, independent of autograd."""
    x = np.asarray(x, dtype=np.float64)
    w_full = np.asarray(w_full, dtype=np.float64)
    dout = np.asarray(dout, dtype=np.float64)
    src = np.asarray(edge_index[0]).astype(np.int64)
    dst = np.asarray(edge_index[1]).astype(np.int64)
    typ = np.asarray(edge_type).astype(np.int64)
    n = x.shape[0]
    dw = np.zeros_like(w_full)
    dx = np.zeros_like(x)
    for r in range(w_full.shape[0]):
        a = _dense_norm_adj(n, src, dst, typ, r, aggr)
        h = a @ x
        dw[r] = h.T @ dout
        dx += a.T @ (dout @ w_full[r].T)
    g = {"weight": dw, "x": dx, "bias": dout.sum(axis=0)}
    if root is not None:
        root = np.asarray(root, dtype=np.float64)
        g["root"] = x.T @ dout
        g["x"] = g["x"] + dout @ root.T
    return g


# --------------------------------------------------------------------------------------
# Sparse float64 evaluation for graphs too large for the dense form (segment form)
# --------------------------------------------------------------------------------------
def rgcn_conv_segments(x, edge_index, edge_type, w_full, root, bias, dout=None, aggr: str = "mean"):
    """Edge-wise float64 evaluation: out[i] += (x[j] / c[i,r]) W_r per edge, then the same
    chain rule backwards.  O(E * in * out / chunk) memory; used for mid-sized cases."""
    x = np.asarray(x, dtype=np.float64)
    w_full = np.asarray(w_full, dtype=np.float64)
    src = np.asarray(edge_index[0]).astype(np.int64)
    dst = np.asarray(edge_index[1]).astype(np.int64)
    typ = np.asarray(edge_type).astype(np.int64)
    n = x.shape[0]
    num_rel = w_full.shape[0]
    cnt = np.zeros(n * num_rel, dtype=np.float64)
    np.add.at(cnt, dst * num_rel + typ, 1.0)
    ew = 1.0 / np.maximum(cnt[dst * num_rel + typ], 1.0) if aggr == "mean" else np.ones(len(dst))
    out = np.zeros((n, w_full.shape[2]), dtype=np.float64)
    grads = None
    if dout is not None:
        dout = np.asarray(dout, dtype=np.float64)
        grads = {"weight": np.zeros_like(w_full), "x": np.zeros_like(x), "bias": dout.sum(axis=0)}
    for r in range(num_rel):
        sel = np.nonzero(typ == r)[0]
        if sel.size == 0:
            continue
        h = np.zeros((n, x.shape[1]), dtype=np.float64)
        np.add.at(h, dst[sel], x[src[sel]] * ew[sel, None])
        out += h @ w_full[r]
        if grads is not None:
            grads["weight"][r] = h.T @ dout
            gh = dout @ w_full[r].T
            np.add.at(grads["x"], src[sel], gh[dst[sel]] * ew[sel, None])
    if root is not None:
        root = np.asarray(root, dtype=np.float64)
        out += x @ root
        if grads is not None:
            grads["root"] = x.T @ dout
            grads["x"] += dout @ root.T
    if bias is not None:
        out += np.asarray(bias, dtype=np.float64)[None, :]
    return out, grads


# --------------------------------------------------------------------------------------
# Parameter init as the reference does it
# --------------------------------------------------------------------------------------
def glorot_(t: Tensor, generator: Optional[torch.Generator] = None) -> Tensor:
    """PyG ``glorot``: U(+-sqrt(6 / (size(-2) + size(-1))))."""
    bound = math.sqrt(6.0 / (t.shape[-2] + t.shape[-1]))
    with torch.no_grad():
        return t.uniform_(-bound, bound, generator=generator)


def reference_layer_params(num_relations: int, in_c: int, out_c: int, generator: torch.Generator,
                           dtype=torch.float32):
    """weight/root/bias initialised in the reference's order for ONE conv:
    glorot(weight), glorot(root), zeros(bias) (PyG reset_parameters) then
    kaiming_uniform_(weight, mode='fan_in') (model/layers.py:17-18)."""
    weight = glorot_(torch.empty(num_relations, in_c, out_c, dtype=dtype), generator)
    root = glorot_(torch.empty(in_c, out_c, dtype=dtype), generator)
    bias = torch.zeros(out_c, dtype=dtype)
    # kaiming_uniform_(a=0, fan_in): fan_in of a [R,in,out] tensor = size(1) * prod(size(2:)) = in*out
    bound = math.sqrt(2.0) * math.sqrt(3.0 / (in_c * out_c))
    with torch.no_grad():
        weight.uniform_(-bound, bound, generator=generator)
    return weight, root, bias


# --------------------------------------------------------------------------------------
# Synthetic inputs of SURVEY.md 8d (shared by tests and bench.py's cpu_baseline leg)
# --------------------------------------------------------------------------------------
def synthetic_graph(n: int, e: int, num_rel: int, seed: int = 0, skew: bool = False):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    if skew:
        u = torch.rand(e, generator=g, dtype=torch.float64)
        dst = (torch.floor(u.pow(-1.0 / 0.2)).to(torch.int64) - 1) % n  # Zipf(1.2)-like tail
    else:
        dst = torch.randint(0, n, (e,), generator=g)
    typ = torch.randint(0, num_rel, (e,), generator=g)
    return torch.stack([src, dst]), typ


def synthetic_params(num_rel: int, in_c: int, out_c: int, seed: int = 0):
    g = torch.Generator().manual_seed(seed + 1)
    bw = math.sqrt(6.0 / (in_c * out_c))
    br = math.sqrt(6.0 / (in_c + out_c))
    weight = torch.empty(num_rel, in_c, out_c).uniform_(-bw, bw, generator=g)
    root = torch.empty(in_c, out_c).uniform_(-br, br, generator=g)
    bias = torch.zeros(out_c)
    return weight, root, bias
