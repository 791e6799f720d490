"""Graph ingest: N-Triples -> node / relation enumeration -> COO ``edge_index`` / ``edge_type`` tensors, and the
dataset assembly around it (labels, 60/20/20 split, summary <-> original node maps).  SURVEY.md 8f-1: the
producer of the hot path's inputs, with the reference's CONTRACT (/root/reference/graphs/graph.py:24-69,
graphs/dataset.py:14-97, graphs/graphProcessing.py:7-91):

* a line is ``"<s> <p> <o> ."``; it is parsed as ``line[:-2].split(" ", 2)`` and lower-cased;
* nodes = sorted(subjects | objects) (literals, blank nodes and rdf:type endpoints included);
* ``rdf:type`` / ``<type>`` triples carry labels and are excluded from message passing;
* every other triple yields a forward edge of type ``2*rel`` and an inverse edge of type ``2*rel + 1``;
  duplicate triples yield duplicate edges; ``num_relations`` of the models is ``2R + 1``;
* classes = sorted object of every rdf:type triple whose subject is not in the swrc ontology namespace.

One deliberate difference: predicates are enumerated in SORTED order.  The reference iterates a Python ``set``
(graph.py:51), so its relation ids change from process to process and are not guaranteed to agree between a
summary graph and the original graph although weight transfer assumes they do.
"""
from __future__ import annotations

import os
from collections import defaultdict
from copy import deepcopy
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .data import Data

RDF_TYPE = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>"
TYPE_PREDICATES = (RDF_TYPE, "<type>")
ONTOLOGY_PREFIX = "http://swrc.ontoware.org/ontology"


def parse_graph_nt(path: str) -> List[str]:
    with open(path, "r") as f:
        return f.read().splitlines()


def split_triples(lines: List[str]) -> List[Tuple[str, str, str]]:
    out = []
    for line in lines:
        parts = line[:-2].split(" ", maxsplit=2)
        if parts != [""]:
            out.append((parts[0].lower(), parts[1].lower(), parts[2].lower()))
    return out


class Graph:
    """Attribute-compatible with the reference's ``Graph``: ``nodes``, ``node_to_enum``, ``num_nodes``,
    ``num_edges``, ``relations``, ``training_data`` (a ``Data`` with ``edge_index`` / ``edge_type``), ``embedding``,
    and the summary <-> original maps filled by ``Dataset``."""

    def __init__(self, name: str, org2type_dict: Optional[Dict[str, set]] = None) -> None:
        self.name = name
        self.nodes: List[str] = None
        self.node_to_enum: Dict[str, int] = None
        self.num_nodes: int = None
        self.num_edges: int = None
        self.relations: Dict[str, int] = None
        self.orgNode2sumNode_dict: Dict[str, str] = None
        self.sumNode2orgNode_dict: Dict[str, List[str]] = None
        self.org2type_dict = org2type_dict
        self.org2type = None
        self.sum2type = None
        self.training_data: Data = None
        self.embedding = None

    def init_graph(self, graph_triples: List[str]) -> None:
        triples = split_triples(graph_triples)
        self.num_edges = len(set(graph_triples))            # statistic only (includes type triples)
        self.nodes = sorted({s for s, _, _ in triples} | {o for _, _, o in triples})
        self.num_nodes = len(self.nodes)
        self.node_to_enum = {n: i for i, n in enumerate(self.nodes)}
        self.relations = {p: i for i, p in enumerate(sorted({p for _, p, _ in triples} - set(TYPE_PREDICATES)))}
        msg = [(self.node_to_enum[s], self.node_to_enum[o], self.relations[p]) for s, p, o in triples
               if p in self.relations]
        t = np.asarray(msg, dtype=np.int64).reshape(-1, 3)
        # forward and inverse edge interleaved in file order, as the reference appends them
        src = np.stack([t[:, 0], t[:, 1]], 1).reshape(-1)
        dst = np.stack([t[:, 1], t[:, 0]], 1).reshape(-1)
        typ = np.stack([2 * t[:, 2], 2 * t[:, 2] + 1], 1).reshape(-1)
        self.training_data = Data(edge_index=torch.from_numpy(np.stack([src, dst])))
        self.training_data.edge_type = torch.from_numpy(typ)


def get_classes(triples: List[Tuple[str, str, str]]) -> List[str]:
    return sorted({o for s, p, o in triples if p == RDF_TYPE and s.split("#")[0] != ONTOLOGY_PREFIX})


def nodes2type_mapping(triples, classes) -> Dict[str, set]:
    cls = set(classes)
    out = defaultdict(set)
    for s, p, o in triples:
        if p == RDF_TYPE and s.split("#")[0] != ONTOLOGY_PREFIX and o in cls:
            out[s].add(o)
    return out


def node_mappings(map_triples) -> Tuple[Dict[str, str], Dict[str, List[str]]]:
    """``<sumNode> <isSummaryOf> <orgNode> .`` lines -> (org -> sum, sum -> [org])"""
    sum2org, org2sum = defaultdict(list), {}
    for s, _, o in map_triples:
        sum2org[s].append(o)
        org2sum[o] = s
    return dict(sorted(org2sum.items())), dict(sorted(sum2org.items()))


class Dataset:
    """``Dataset(org_path, sum_path, map_path).init_dataset()`` -> ``orgGraph``, ``sumGraphs``, ``num_classes``
    with ``training_data.{x,y}_{train,val,test}`` filled (reference graphs/dataset.py)."""

    def __init__(self, org_path: str, sum_path: str, map_path: str) -> None:
        self.org_path, self.sum_path, self.map_path = org_path, sum_path, map_path
        self.sumGraphs: List[Graph] = []
        self.orgGraph: Graph = None
        self.enum_classes: Dict[str, int] = None
        self.num_classes: int = None

    def get_file_names(self) -> Tuple[List[str], List[str]]:
        ls = lambda d: sorted(f for f in os.listdir(d) if not f.startswith(".") and os.path.isfile(os.path.join(d, f)))
        sums, maps = ls(self.sum_path), ls(self.map_path)
        assert len(sums) == len(maps), f"for every summary file there needs to be a map file: {sums} and {maps}"
        return sums, maps

    def _labels(self, graph: Graph, node2types: Dict[str, list]) -> Tuple[List[int], List[list]]:
        idx, labs = [], []
        for node, lab in node2types.items():
            if sum(lab) != 0.0 and node in graph.node_to_enum:
                idx.append(graph.node_to_enum[node])
                labs.append(list(lab))
        return idx, labs

    def make_training_data(self) -> None:
        from sklearn.model_selection import train_test_split
        org, nc = self.orgGraph, self.num_classes
        org.org2type = {}
        for node, types in org.org2type_dict.items():
            v = [0] * nc
            for t in types:
                v[self.enum_classes[t]] += 1
            org.org2type[node] = v
        g_idx, g_labels = self._labels(org, org.org2type)
        x_train, x_test, y_train, y_test = train_test_split(g_idx, g_labels, test_size=0.2, random_state=1, shuffle=True)
        x_train, x_val, y_train, y_val = train_test_split(x_train, y_train, test_size=0.25, random_state=1, shuffle=True)
        td = org.training_data
        td.x_train, td.x_val, td.x_test = (torch.tensor(v, dtype=torch.long) for v in (x_train, x_val, x_test))
        td.y_train, td.y_val, td.y_test = (torch.tensor(v, dtype=torch.long) for v in (y_train, y_val, y_test))
        # evaluation nodes do not contribute to the summary graphs' (fractional) labels
        held_out = set(x_test) | set(x_val)
        pruned = {n: (set() if org.node_to_enum.get(n) in held_out else set(t)) for n, t in org.org2type_dict.items()}
        for sg in self.sumGraphs:
            sg.sum2type = {}
            for sum_node, org_nodes in sg.sumNode2orgNode_dict.items():
                v = [0.0] * nc
                for n in org_nodes:
                    for t in pruned.get(n, ()):
                        v[self.enum_classes[t]] += 1.0
                div = max(1, len(org_nodes))
                sg.sum2type[sum_node] = [a / div for a in v]
            s_idx, s_lab = self._labels(sg, sg.sum2type)
            sg.training_data.x_train = torch.tensor(s_idx, dtype=torch.long)
            sg.training_data.y_train = torch.tensor(s_lab).reshape(len(s_idx), nc)
            assert len(sg.relations) == len(org.relations), "number of relations in summary graph and original graph differ"

    def init_dataset(self) -> None:
        org_lines = parse_graph_nt(self.org_path)
        org_triples = split_triples(org_lines)
        classes = get_classes(org_triples)
        self.enum_classes, self.num_classes = {c: i for i, c in enumerate(classes)}, len(classes)
        org2type = nodes2type_mapping(org_triples, classes)
        self.orgGraph = Graph(os.path.basename(self.org_path), deepcopy(org2type))
        self.orgGraph.init_graph(org_lines)
        sums, maps = self.get_file_names()
        for sf, mf in zip(sums, maps):
            sg = Graph(sf, deepcopy(org2type))
            sg.init_graph(parse_graph_nt(os.path.join(self.sum_path, sf)))
            sg.orgNode2sumNode_dict, sg.sumNode2orgNode_dict = node_mappings(
                split_triples(parse_graph_nt(os.path.join(self.map_path, mf))))
            self.sumGraphs.append(sg)
        self.make_training_data()


# ---- model/embeddingTricks.py equivalents ---------------------------------------------
def transfer_index(graph: Graph, sum_graph: Graph) -> torch.Tensor:
    """For every original node, the row of its summary node in ``sum_graph.embedding`` (-1: not mapped).
    The reference walks a dict per node and per summary (model/embeddingTricks.py:17-24); this is one index
    tensor and an ``index_select``."""
    idx = torch.full((graph.num_nodes,), -1, dtype=torch.long)
    o2s, s_enum = sum_graph.orgNode2sumNode_dict, sum_graph.node_to_enum
    for node, i in graph.node_to_enum.items():
        s = o2s.get(node)
        if s is not None and s in s_enum:
            idx[i] = s_enum[s]
    return idx


def get_tensor_list(graph: Graph, sum_graphs: List[Graph], emb_dim: int) -> List[torch.Tensor]:
    out = []
    for sg in sum_graphs:
        idx = transfer_index(graph, sg)
        emb = sg.embedding.detach().to("cpu")
        t = torch.rand(graph.num_nodes, emb_dim)           # unmapped nodes keep the reference's U(0,1) fill
        m = idx >= 0
        t[m] = emb.index_select(0, idx[m])
        out.append(t)
    return out


def stack_embeddings(graph, sum_graphs, emb_dim):
    return torch.stack(get_tensor_list(graph, sum_graphs, emb_dim)).detach()


def concat_embeddings(graph, sum_graphs, emb_dim):
    return torch.cat(get_tensor_list(graph, sum_graphs, emb_dim), dim=-1).detach()


def sum_embeddings(graph, sum_graphs, emb_dim):
    return sum(get_tensor_list(graph, sum_graphs, emb_dim)).detach()
