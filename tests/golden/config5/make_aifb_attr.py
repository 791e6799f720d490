"""Generator of tests/golden/config5/aifb_attr_config5.npz (run once in the build container; the reference is not on the GPU box).

BASELINE.json config 5 -- "AIFB attr-summary pre-train -> full-graph transfer" -- needs the AIFB original graph, which the
reference does not ship (graphs/AIFB/AIFB_complete.nt is listed in .MISSING_LARGE_BLOBS).  It DOES ship the three attribute
summaries (graphs/AIFB/attr/sum/AIFB_sum_{in,in_out,out}.nt: one summary triple per original triple, in the original
file's order -- graphs/createAttributeSum.py:47-63 writes them in one pass over the same list) and the three node maps
(graphs/AIFB/attr/map/*.nt: all 8,243 original nodes).  Line i of the three summary files therefore names, for the subject
and the object of original triple i, its class under each of the three partitions; this script picks (seeded) one node of
the intersection of the three classes for each endpoint, which yields an original graph with the real node count (8,243),
triple count (29,043; 24,919 of them message-passing triples -> 49,838 directed edges), 44 predicates and hub structure
whose attribute summaries are EXACTLY the shipped files.  What is synthetic: which member of a class an endpoint is, and
WHICH member of its class an rdf:type object is (the class node): chosen per subject from its in_out class with 15 % noise,
so that the labels are learnable from the structure.

Stored as integer ids (names are rebuilt by tests/aifb_attr.py so that sorted-name order == id order):
  org_s, org_p, org_o   int32 [29043]   original triples in file order; p = 44 means rdf:type (o = the class node)
  sum_s[k], sum_o[k]    int32 [29043]   summary-node ids of the endpoints under summary k (in, in_out, out);
                                        ids = rank of the node's decimal-hash name in sorted(string) order, as Graph.init_graph sorts
  org2sum[k]            int32 [8243]    summary node of every original node
"""
import os
import sys
from collections import defaultdict

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
KINDS = ("in", "in_out", "out")
RDF_TYPE = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"


def split(line):
    return line[:-2].split(" ", 2)


def main():
    base = os.path.join(REF, "graphs", "AIFB", "attr")
    sums = {k: [split(l) for l in open(os.path.join(base, "sum", f"AIFB_sum_{k}.nt")).read().splitlines() if l.strip()] for k in KINDS}
    maps = {}
    for k in KINDS:
        d = {}
        for l in open(os.path.join(base, "map", f"AIFB_map_{k}.nt")).read().splitlines():
            s, _, o = split(l)
            d[o.lower()] = s
        maps[k] = d
    org_nodes = sorted(maps["in"])                               # lower-cased names, sorted: the reference's node order
    assert all(sorted(maps[k]) == org_nodes for k in KINDS) and len(org_nodes) == 8243
    org_id = {n: i for i, n in enumerate(org_nodes)}
    joint = defaultdict(list)
    for n in org_nodes:
        joint[tuple(maps[k][n] for k in KINDS)].append(org_id[n])
    n_lines = len(sums["in"])
    assert all(len(sums[k]) == n_lines for k in KINDS)
    preds = sorted({t[1].lower() for t in sums["in"]} - {RDF_TYPE})
    pid = {p: i for i, p in enumerate(preds)}
    assert len(preds) == 44
    rng = np.random.default_rng(20240605)
    org_s, org_p, org_o = (np.zeros(n_lines, np.int32) for _ in range(3))
    sum_ids = {k: {n: i for i, n in enumerate(sorted({t[0] for t in sums[k]} | {t[2] for t in sums[k]}))} for k in KINDS}
    sum_s = {k: np.zeros(n_lines, np.int32) for k in KINDS}
    sum_o = {k: np.zeros(n_lines, np.int32) for k in KINDS}
    inout_rank = {n: i for i, n in enumerate(sorted(set(maps["in_out"].values())))}
    # every class is named at least as often as it has members (checked below): dealing its members round-robin from a
    # seeded permutation uses every one of the 8,243 nodes
    perm = {c: rng.permutation(np.asarray(m)) for c, m in joint.items()}
    uses = defaultdict(int)

    def deal(c):
        v = int(perm[c][uses[c] % len(perm[c])])
        uses[c] += 1
        return v

    for i in range(n_lines):
        rows = [sums[k][i] for k in KINDS]
        p = rows[0][1].lower()
        assert all(r[1].lower() == p for r in rows)
        s = deal(tuple(r[0] for r in rows))
        org_s[i] = s
        for k, r in zip(KINDS, rows):
            sum_s[k][i], sum_o[k][i] = sum_ids[k][r[0]], sum_ids[k][r[2]]
        oc = tuple(r[2] for r in rows)
        if p == RDF_TYPE:
            # the class node: a member of the object's class (the shipped maps name the swrc ontology classes there), picked
            # from the subject's in_out class so that labels follow structure, with 15 % noise
            members = joint[oc]
            c = inout_rank[maps["in_out"][org_nodes[s]]] % len(members)
            if rng.random() < 0.15:
                c = int(rng.integers(len(members)))
            org_p[i] = len(preds)
            org_o[i] = members[c]
        else:
            org_p[i] = pid[p]
            org_o[i] = deal(oc)
    used = set(org_s.tolist()) | set(org_o.tolist())
    assert len(used) == len(org_nodes), f"{len(org_nodes) - len(used)} nodes unused"
    out = dict(org_s=org_s, org_p=org_p, org_o=org_o, n_org=np.int32(len(org_nodes)), n_pred=np.int32(len(preds)))
    for k in KINDS:
        out[f"sum_s_{k}"], out[f"sum_o_{k}"] = sum_s[k], sum_o[k]
        out[f"n_sum_{k}"] = np.int32(len(sum_ids[k]))
        out[f"org2sum_{k}"] = np.asarray([sum_ids[k][maps[k][n]] for n in org_nodes], np.int32)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aifb_attr_config5.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", {k: int(out[f"n_sum_{k}"]) for k in KINDS})


if __name__ == "__main__":
    main()
