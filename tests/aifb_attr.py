"""TEST INFRASTRUCTURE: materialise tests/golden/config5/aifb_attr_config5.npz (generator + provenance: tests/golden/config5/make_aifb_attr.py)
as the N-Triples files the reference's flow reads -- an original graph, three attribute summaries, three node maps -- with
names whose sorted order is the id order of the fixture."""
import os

import numpy as np

from tests.conftest import GOLDEN_DIR

KINDS = ("in", "in_out", "out")
RDF_TYPE = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"


def load():
    return np.load(os.path.join(GOLDEN_DIR, "config5", "aifb_attr_config5.npz"))


def write_dataset(root: str):
    """-> (org_path, sum_dir, map_dir) under ``root``"""
    z = load()
    n_pred = int(z["n_pred"])
    pred = [f"<p{i:02d}>" for i in range(n_pred)] + [RDF_TYPE]
    org = lambda i: f"<n{int(i):05d}>"
    sm = lambda j: f"<s{int(j):04d}>"
    sum_dir, map_dir = os.path.join(root, "sum"), os.path.join(root, "map")
    os.makedirs(sum_dir, exist_ok=True)
    os.makedirs(map_dir, exist_ok=True)
    org_path = os.path.join(root, "AIFB_like_complete.nt")
    with open(org_path, "w") as f:
        f.write("".join(f"{org(s)} {pred[p]} {org(o)} .\n" for s, p, o in zip(z["org_s"], z["org_p"], z["org_o"])))
    for k in KINDS:
        with open(os.path.join(sum_dir, f"AIFB_sum_{k}.nt"), "w") as f:
            f.write("".join(f"{sm(s)} {pred[p]} {sm(o)} .\n" for s, p, o in zip(z[f"sum_s_{k}"], z["org_p"], z[f"sum_o_{k}"])))
        with open(os.path.join(map_dir, f"AIFB_map_{k}.nt"), "w") as f:
            f.write("".join(f"{sm(j)} <isSummaryOf> {org(i)} .\n" for i, j in enumerate(z[f"org2sum_{k}"])))
    return org_path, sum_dir, map_dir
