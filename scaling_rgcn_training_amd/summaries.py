"""Attribute-summary generation: every node is mapped to the 128-bit MurmurHash3 of its sorted
set of outgoing / incoming / incoming+outgoing predicates; the summary graph replaces every node by its hash and the
map file records ``<hash> <isSummaryOf> node``.  Contract: /root/reference/graphs/createAttributeSum.py:6-67
(``create_sum_map`` / ``write_sum_map_files``), whose hash is ``mmh3.hash128`` -- a C extension that is not in this
image; here it is ``csrc/murmur3_x64_128.c`` (plain C, built into ``librgcn_host.so``).  Integer / byte work: results
are bit-exact and pinned by the reference's shipped ``graphs/TEST/attr`` files (tests/test_summaries.py).

``legacy=True`` reproduces those shipped files byte for byte: they were written by an earlier version of the
reference script that neither lower-cased the terms nor skipped ``rdf:type`` when collecting predicate sets (the
hash-named ids in them are murmur3 of e.g. ``<...#isAbout>`` with its capital A and of the rdf:type predicate alone).
The default follows the script as it is in the reference today (lower-cased, rdf:type excluded from the sets).
"""
from __future__ import annotations

import ctypes as C
import os
from collections import defaultdict
from typing import Dict, Iterable, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "librgcn_host.so")
RDF_TYPE = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"
LITERAL_KEY = "http://example.org/literal"
_host = None


def _lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        _host = C.CDLL(HOST_LIB_PATH)
        _host.rgcn_murmur3_x64_128.restype = None
        _host.rgcn_murmur3_x64_128.argtypes = [C.c_char_p, C.c_int64, C.c_uint32, C.POINTER(C.c_uint64 * 2)]
    return _host


def hash128(key: bytes, seed: int = 0) -> int:
    """``mmh3.hash128(key, seed)``: MurmurHash3 x64 128, unsigned, little-endian (h1 | h2 << 64)."""
    out = (C.c_uint64 * 2)()
    _lib().rgcn_murmur3_x64_128(key, len(key), seed, C.byref(out))
    return int(out[0]) | (int(out[1]) << 64)


def _split(triple: str, lower: bool):
    parts = triple[:-2].split(" ", maxsplit=2)
    if parts == [""]:
        return None
    return tuple(p.lower() for p in parts) if lower else tuple(parts)


def property_hashes(triples: Iterable[str], legacy: bool = False) -> Tuple[Dict[str, int], Dict[str, int], Dict[str, int]]:
    """(outgoing, incoming, incoming + outgoing) hash per entity -- createAttributeSum.py:7-38"""
    outgoing, incoming = defaultdict(set), defaultdict(set)
    for t in triples:
        spo = _split(t, lower=not legacy)
        if spo is None:
            continue
        s, p, o = spo
        if legacy or p != RDF_TYPE:
            outgoing[s].add(p)
            if o.startswith('"'):
                incoming[LITERAL_KEY].add(p)
            else:
                incoming[o].add(p)
    h = lambda ps: hash128(",".join(sorted(ps)).encode("utf8"))
    out_h = {k: h(v) for k, v in outgoing.items()}
    in_h = {k: h(v) for k, v in incoming.items()}
    both = {e: in_h.get(e, 0) + out_h.get(e, 0) for e in set(in_h) | set(out_h)}
    return out_h, in_h, both


def write_sum_map_files(hashes: Dict[str, int], triples: List[str], sum_path: str, map_path: str, legacy: bool = False) -> None:
    """createAttributeSum.py:44-67: the summary graph (every term replaced by its hash, '0' when it has none) and the
    map file (insertion order of first appearance, last assignment wins -- a Python dict, as in the reference)."""
    mapping: Dict[str, object] = {}
    with open(sum_path, "w") as f:
        for t in triples:
            spo = _split(t, lower=not legacy)
            if spo is None:
                continue
            s, p, o = spo
            if o.startswith('"') and LITERAL_KEY in hashes:
                obj = hashes[LITERAL_KEY]
            else:
                obj = hashes[o] if o in hashes else "0"
            sub = hashes[s] if s in hashes else "0"
            mapping[s] = sub
            mapping[o] = obj
            f.write(f"<{sub}> {p} <{obj}> .\n")
    with open(map_path, "w") as m:
        for o_node, s_node in mapping.items():
            m.write(f"<{s_node}> <isSummaryOf> {o_node} .\n")


def create_sum_map(path: str, sum_path: str, map_path: str, dataset: str, legacy: bool = False) -> None:
    """``create_sum_map`` of the reference (main.py:39 ``-create_attr_sum``): writes
    ``{sum_path}{dataset}_sum_{out,in,in_out}.nt`` and ``{map_path}{dataset}_map_{out,in,in_out}.nt``."""
    with open(path, "r") as f:
        triples = f.read().splitlines()
    out_h, in_h, both = property_hashes(triples, legacy)
    for name, h in (("out", out_h), ("in", in_h), ("in_out", both)):
        write_sum_map_files(h, triples, f"{sum_path}{dataset}_sum_{name}.nt", f"{map_path}{dataset}_map_{name}.nt", legacy)
