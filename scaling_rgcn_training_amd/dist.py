"""One process per GPU over RCCL (``torch.distributed`` backend ``nccl``) / gloo on CPU for tests.

Edge partition: output nodes are cut into ``world`` equal, tile-aligned ranges.
Rank p owns the forward plan of the edges INTO its range (so every (dst, relation) mean is local)
and the transposed plan of the edges OUT OF its range (so every dX row is complete locally).
Features are replicated (2.56 GB at the 10M-node config, against 288 GB of HBM).  Per layer:

* forward : each rank writes its rows into its slice of the gathered buffer, then ONE all-gather
            (the all-reduce of per-node aggregated features with exactly one contributor per row);
* backward: dX rows likewise (all-gather), weight gradients all-reduced (~0.5 MB).

Because range boundaries are tile multiples, a rank's tiles and chunks are exactly the single-GPU
ones, so P-rank outputs and dX are bit-identical to 1-rank ones; only d_weight sums differ in order.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor

from .conv import DistContext, RGCNConv
from .plan import GraphPlans, build_graph_plans, cached_graph_plans


def rows_per_rank(n_nodes: int, tile: int, world: int) -> int:
    n_tiles = (n_nodes + tile - 1) // tile
    return ((n_tiles + world - 1) // world) * tile


def make_context(n_nodes: int, tile: int, group=None) -> Optional[DistContext]:
    if not dist.is_available() or not dist.is_initialized():
        return None
    world = dist.get_world_size(group)
    if world == 1:
        return None
    return DistContext(group, dist.get_rank(group), world, rows_per_rank(n_nodes, tile, world))


def rank_plans(edge_index: Tensor, edge_type: Tensor, n_nodes: int, num_relations: int, tile: int,
               aggr: str, rank: int, world: int) -> GraphPlans:
    rows = rows_per_rank(n_nodes, tile, world)
    b = min(rank * rows, n_nodes)
    e = min(b + rows, n_nodes)
    return build_graph_plans(edge_index, edge_type, n_nodes, num_relations, tile, aggr,
                             fwd_range=(b, e), bwd_range=(b, e))


def cached_rank_plans(edge_index, edge_type, n_nodes, num_relations, tile, aggr, dctx: DistContext) -> GraphPlans:
    return cached_graph_plans(
        edge_index, edge_type, n_nodes, num_relations, tile, aggr,
        builder=lambda: rank_plans(edge_index, edge_type, n_nodes, num_relations, tile, aggr, dctx.rank, dctx.world),
        extra_key=("rank", dctx.rank, dctx.world))


def attach(module: torch.nn.Module, n_nodes: int, n_edges: int, group=None) -> None:
    """Switch every RGCNConv under ``module`` to the edge-partitioned path for the current process group
    (``n_nodes`` / ``n_edges`` of the graph the module will see: they fix the tile size and with it the
    tile-aligned node ranges)."""
    from .conv import tile_for
    for m in module.modules():
        if isinstance(m, RGCNConv):
            tile = tile_for(m.in_channels, m.out_channels, n_nodes, n_edges, m.num_relations)
            m.dist = make_context(n_nodes, tile, group)
