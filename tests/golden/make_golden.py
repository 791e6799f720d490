#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Run in the build container only (reads the reference's DATA files under
/root/reference/graphs; never needed on the GPU box):

    python tests/golden/make_golden.py

Topologies come from the N-Triples files the reference ships (SURVEY.md 8c item 3,
Appendix B).  The triple -> (edge_index, edge_type) conversion follows the CONTRACT of
/root/reference/graphs/graph.py:24-69 (lower-casing, ``triple[:-2].split(" ", 2)``,
rdf:type / <type> predicates dropped, nodes = sorted(subjects | objects), forward edge type
2*rel and inverse edge type 2*rel+1, duplicates kept, num_relations = 2R+1 as at
model/modelTrainer.py:78) with ONE deliberate difference: predicate ids are assigned in
SORTED order, because the reference iterates a Python set (graph.py:51) and is therefore
hash-randomised per process.  Expected outputs are produced by oracle/rgcn_oracle.py in
float64 (dense formula where N allows, edge-wise otherwise).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import rgcn_oracle as O  # noqa: E402

REF = "/root/reference/graphs"
TYPE_PREDS = {"<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>", "<type>"}


def triples_to_coo(path: str):
    with open(path, "r") as f:
        lines = f.read().splitlines()
    parsed = []
    for line in lines:
        parts = line[:-2].split(" ", maxsplit=2)
        if parts != [""]:
            parsed.append(tuple(p.lower() for p in parts))
    nodes = sorted({s for s, _, _ in parsed} | {o for _, _, o in parsed})
    preds = sorted({p for _, p, _ in parsed} - TYPE_PREDS)
    nid = {v: i for i, v in enumerate(nodes)}
    rid = {p: i for i, p in enumerate(preds)}
    edges = []
    for s, p, o in parsed:
        if p in rid:
            edges.append((nid[s], nid[o], 2 * rid[p]))
            edges.append((nid[o], nid[s], 2 * rid[p] + 1))
    e = np.asarray(edges, dtype=np.int64).T
    return e[:2].astype(np.int32), e[2].astype(np.int32), len(nodes), 2 * len(preds) + 1


def make_case(name, edge_index, edge_type, n, num_rel, in_c, out_c, seed, mode="full", nb=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, in_c, generator=g)
    dout = torch.randn(n, out_c, generator=g)
    comp = None
    if mode == "full":
        weight, root, bias = O.reference_layer_params(num_rel, in_c, out_c, g)
    elif mode == "basis":
        weight = O.glorot_(torch.empty(nb, in_c, out_c), g)
        comp = O.glorot_(torch.empty(num_rel, nb), g)
        root = O.glorot_(torch.empty(in_c, out_c), g)
        bias = torch.zeros(out_c)
    else:  # block
        weight = O.glorot_(torch.empty(num_rel, nb, in_c // nb, out_c // nb), g)
        root = O.glorot_(torch.empty(in_c, out_c), g)
        bias = torch.zeros(out_c)
    bias = bias + 0.1 * torch.randn(out_c, generator=g)  # non-zero so the bias path is exercised
    w_full = O.effective_weight(weight.double(), None if comp is None else comp.double(), num_rel,
                                nb if mode == "block" else None, in_c, out_c).numpy()
    if n <= 3000:
        out = O.rgcn_conv_dense(x.numpy(), edge_index, edge_type, w_full, root.numpy(), bias.numpy())
        grads = O.rgcn_conv_grads_dense(x.numpy(), edge_index, edge_type, w_full, root.numpy(), dout.numpy())
    else:
        out, grads = O.rgcn_conv_segments(x.numpy(), edge_index, edge_type, w_full, root.numpy(),
                                          bias.numpy(), dout.numpy())
    arrs = dict(edge_index=edge_index, edge_type=edge_type, num_nodes=np.int64(n),
                num_relations=np.int64(num_rel), x=x.numpy(), weight=weight.numpy(), root=root.numpy(),
                bias=bias.numpy(), dout=dout.numpy(), out=out, d_x=grads["x"], d_wfull=grads["weight"],
                d_root=grads["root"], d_bias=grads["bias"], mode=np.array(mode))
    if comp is not None:
        arrs["comp"] = comp.numpy()
    if nb is not None:
        arrs["nb"] = np.int64(nb)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: N={n} E={edge_index.shape[1]} R'={num_rel} {in_c}->{out_c} mode={mode} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    graphs = {
        "test": f"{REF}/TEST/TEST_complete.nt",
        "test_sum_in": f"{REF}/TEST/attr/sum/TEST_sum_in.nt",
        "aifb_sum_in": f"{REF}/AIFB/attr/sum/AIFB_sum_in.nt",
        "aifb_sum_in_out": f"{REF}/AIFB/attr/sum/AIFB_sum_in_out.nt",
        "aifb_bisim_k3": f"{REF}/AIFB/bisim/sum/AIFB_bisim_k3.nt",
        "mutag_bisim_k3": f"{REF}/MUTAG/bisim/sum/MUTAG_bisim_k3.nt",
    }
    coo = {k: triples_to_coo(p) for k, p in graphs.items()}
    for k, (ei, et, n, r) in coo.items():
        deg = np.bincount(ei[1], minlength=n).max()
        print(f"  {k}: N={n} E={ei.shape[1]} R'={r} max in-degree={deg}")
    # layer-1 shapes of the reference (emb 63 -> hidden 16, main.py:79,82) and layer-2 (16 -> C)
    make_case("test_l1", *coo["test"], 63, 16, seed=1)
    make_case("test_l2", *coo["test"], 16, 5, seed=2)
    make_case("test_basis", *coo["test"], 12, 8, seed=3, mode="basis", nb=3)
    make_case("test_block", *coo["test"], 12, 8, seed=4, mode="block", nb=4)
    make_case("test_sum_in_l1", *coo["test_sum_in"], 63, 16, seed=5)
    make_case("aifb_sum_in_l1", *coo["aifb_sum_in"], 63, 16, seed=6)
    make_case("aifb_sum_in_out_l1", *coo["aifb_sum_in_out"], 63, 16, seed=7)
    make_case("aifb_bisim_k3_l1", *coo["aifb_bisim_k3"], 63, 16, seed=8)
    make_case("aifb_bisim_k3_l2", *coo["aifb_bisim_k3"], 16, 7, seed=9)
    make_case("mutag_bisim_k3_l1", *coo["mutag_bisim_k3"], 63, 16, seed=10)
    make_case("mutag_bisim_k3_basis", *coo["mutag_bisim_k3"], 32, 32, seed=11, mode="basis", nb=30)


if __name__ == "__main__":
    main()
