"""One process per GPU over RCCL (``torch.distributed`` backend ``nccl``) / gloo on CPU for tests.

Edge partition: output nodes are cut into ``pieces * world`` equal, tile-aligned blocks
dealt piece-major (conv.DistContext).  Rank p owns the forward plans of the edges INTO its blocks (so every
(dst, relation) mean is local) and the transposed plans of the edges OUT OF its blocks (so every dX row is
complete locally).  Features are replicated (2.56 GB at the 10M-node config, against 288 GB of HBM).  Per layer:

* forward : each rank writes a block into its place in the gathered buffer and all-gathers the super-block
            asynchronously while the next block's kernels run (the all-reduce of per-node aggregated features
            with exactly one contributor per row, pipelined in ``PIECES`` stages);
* backward: dX blocks likewise, weight gradients all-reduced (~0.5 MB).

Because block boundaries are tile multiples, a rank's tiles and chunks are exactly the single-GPU
ones, so P-rank outputs and dX are bit-identical to 1-rank ones; only d_weight sums differ in order.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor

from .conv import DistContext, RGCNConv
from .plan import GraphPlans, build_plan, cached_graph_plans, edge_weights


PIECES = 4  # all-gather pipeline depth: the collective of piece s runs under the kernels of piece s + 1


def piece_rows(n_nodes: int, tile: int, world: int, pieces: int) -> int:
    n_tiles = (n_nodes + tile - 1) // tile
    return ((n_tiles + world * pieces - 1) // (world * pieces)) * tile


def make_context(n_nodes: int, tile: int, group=None, pieces: int = PIECES) -> Optional[DistContext]:
    if not dist.is_available() or not dist.is_initialized():
        return None
    world = dist.get_world_size(group)
    if world == 1:
        return None
    n_tiles = (n_nodes + tile - 1) // tile
    pieces = max(1, min(pieces, n_tiles // world if n_tiles >= world else 1))
    return DistContext(group, dist.get_rank(group), world, piece_rows(n_nodes, tile, world, pieces), pieces)


class RankPlans:
    """The graph plans of one rank: one forward / transposed pair per owned block (piece)."""

    def __init__(self, pieces):
        self.pieces = pieces
        self.num_edges = sum(p.num_edges for p in pieces)


def rank_plans(edge_index: Tensor, edge_type: Tensor, n_nodes: int, num_relations: int, tile: int,
               aggr: str, dctx: DistContext, chunk: int = 64, split: bool = False) -> RankPlans:
    """Plans of this rank's blocks.  The mean normaliser (a sort of all E keys) is computed ONCE and every piece is
    laid out from the same edge list: on the GPU by the library's plan builder with the piece's node range (it keeps
    the edges that scatter into the range), on the CPU (tests) by the torch form from this rank's share."""
    ranges = [dctx.node_range(s, n_nodes) for s in range(dctx.pieces)]
    if edge_type.device.type == "cuda":
        from .plan import build_graph_plans_device
        return RankPlans(build_graph_plans_device(edge_index, edge_type, n_nodes, num_relations, tile, aggr, chunk=chunk,
                                                  ranges=[(r, r) for r in ranges], split=split))
    src, dst = edge_index[0].to(torch.int64), edge_index[1].to(torch.int64)
    rel = edge_type.to(torch.int64)
    w = edge_weights(src, dst, rel, num_relations, aggr)
    pr, world, rank = dctx.piece_rows, dctx.world, dctx.rank

    def share(scatter, gather):
        blk = scatter // pr
        mine = (blk % world) == rank
        return gather[mine], scatter[mine], rel[mine], w[mine], blk[mine] // world

    fg, fs, fr, fw, fpiece = share(dst, src)       # forward: edges INTO my blocks
    bg, bs, br, bw, bpiece = share(src, dst)       # transposed: edges OUT OF my blocks
    out = []
    for s, (b, e) in enumerate(ranges):
        fm, bm = fpiece == s, bpiece == s
        fwd = build_plan(fg[fm], fs[fm], fr[fm], fw[fm], n_nodes, num_relations, tile, b, e, chunk, split)
        bwd = build_plan(bg[bm], bs[bm], br[bm], bw[bm], n_nodes, num_relations, tile, b, e, chunk, split)
        out.append(GraphPlans(fwd=fwd, bwd=bwd, num_edges=int(fm.sum())))
    return RankPlans(out)


def cached_rank_plans(edge_index, edge_type, n_nodes, num_relations, tile, aggr, dctx: DistContext,
                      chunk: int = 64, split: bool = False) -> RankPlans:
    return cached_graph_plans(
        edge_index, edge_type, n_nodes, num_relations, tile, aggr, chunk=chunk, split=split,
        builder=lambda: rank_plans(edge_index, edge_type, n_nodes, num_relations, tile, aggr, dctx, chunk, split),
        extra_key=("rank", dctx.rank, dctx.world, dctx.pieces, dctx.piece_rows))


def attach(module: torch.nn.Module, n_nodes: int, n_edges: int, group=None) -> None:
    """Switch every RGCNConv under ``module`` to the edge-partitioned path for the current process group
    (``n_nodes`` / ``n_edges`` of the graph the module will see: they fix the tile size and with it the
    tile-aligned node ranges)."""
    from .conv import tile_for
    for m in module.modules():
        if isinstance(m, RGCNConv):
            tile = m.layout(n_nodes, n_edges)[0]
            m.dist = make_context(n_nodes, tile, group)
