"""One process per GPU over RCCL (``torch.distributed`` backend ``nccl``) / gloo on CPU for tests.

Edge partition: output nodes are cut into ``pieces * world`` tile-aligned blocks dealt piece-major
(conv.DistContext) -- equal node blocks while those hold equal edge counts within 5 %, else blocks of about equal edge
count (balanced_bounds).  Rank p owns the forward plans of the edges INTO its blocks (so every
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


PIECES = 4  # gather pipeline depth: the collective of piece s runs under the kernels of piece s + 1
BALANCE_TOLERANCE = 0.05     # equal node blocks are kept while every block's edge count is within 5 % of the mean


def piece_rows(n_nodes: int, tile: int, world: int, pieces: int) -> int:
    n_tiles = (n_nodes + tile - 1) // tile
    return ((n_tiles + world * pieces - 1) // (world * pieces)) * tile


def tile_costs(edge_index: Tensor, n_nodes: int, tile: int) -> Tensor:
    """float64 [n_tiles]: rows a rank that owns the tile walks per layer step -- the edges INTO its nodes (forward plan), the
    edges OUT OF them (transposed plan) and the two root pseudo edges per node."""
    n_tiles = (n_nodes + tile - 1) // tile
    c = (torch.bincount(edge_index[1] // tile, minlength=n_tiles) + torch.bincount(edge_index[0] // tile, minlength=n_tiles)).double()
    rows = torch.full((n_tiles,), float(tile), dtype=torch.float64, device=c.device)
    rows[-1] = n_nodes - (n_tiles - 1) * tile
    return (c + 2.0 * rows).cpu()


def block_costs(costs: Tensor, bounds, tile: int) -> Tensor:
    """cost of every block of a cut (bounds in nodes, tile-aligned)"""
    cum = torch.cat([torch.zeros(1, dtype=torch.float64), torch.cumsum(costs, 0)])
    t = torch.clamp(torch.tensor([b // tile for b in bounds]), max=costs.numel())
    return cum[t[1:]] - cum[t[:-1]]


def balanced_bounds(costs: Tensor, n_nodes: int, tile: int, world: int, pieces: int):
    """SURVEY.md 8e 'ranges balanced by edge count': the tile sequence cut into pieces * world contiguous blocks of about
    equal cost -- block i ends at the first tile boundary where the prefix sum reaches i / (pieces * world) of the total.
    Deterministic integer / float64 arithmetic on the same edge list: every rank computes the same cut.  Boundaries are tile
    multiples (the last one is the graph's end), so a rank's tiles and chunks are the single-rank ones; a tile that holds more
    than a block's share (a hub) makes its block heavier than the mean and leaves empty blocks behind it."""
    nb = pieces * world
    cum = torch.cumsum(costs, 0)
    total = float(cum[-1]) if costs.numel() else 0.0
    targets = torch.arange(1, nb, dtype=torch.float64) * (total / nb)
    cut = torch.searchsorted(cum, targets, right=False) + 1          # tiles in blocks 0 .. i
    cut = torch.clamp(cut, max=costs.numel())
    cut = torch.cummax(cut, 0).values.tolist() if nb > 1 else []
    return [0] + [min(int(c) * tile, n_nodes) for c in cut] + [n_nodes]


def make_context(n_nodes: int, tile: int, group=None, pieces: int = PIECES, edge_index: Optional[Tensor] = None,
                 balance: Optional[bool] = None) -> Optional[DistContext]:
    """edge_index given: keep the uniform cut (one in-place all-gather per piece) while its blocks' edge counts stay within
    BALANCE_TOLERANCE of their mean, else cut by edge count (``balance`` True / False pins the choice)."""
    if not dist.is_available() or not dist.is_initialized():
        return None
    world = dist.get_world_size(group)
    if world == 1:
        return None
    n_tiles = (n_nodes + tile - 1) // tile
    pieces = max(1, min(pieces, n_tiles // world if n_tiles >= world else 1))
    pr = piece_rows(n_nodes, tile, world, pieces)
    ctx = DistContext(group, dist.get_rank(group), world, pr, pieces)
    if edge_index is not None and balance is not False:
        costs = tile_costs(edge_index, n_nodes, tile)
        bc = block_costs(costs, ctx.bounds, tile)
        if balance or float(bc.max()) > (1.0 + BALANCE_TOLERANCE) * float(bc.mean()):
            ctx = DistContext(group, dist.get_rank(group), world, 0, pieces, balanced_bounds(costs, n_nodes, tile, world, pieces))
            bc = block_costs(costs, ctx.bounds, tile)
        ctx.block_costs = bc.view(pieces, world)
    return ctx


class RankPlans:
    """The graph plans of one rank: one forward / transposed pair per owned block (piece)."""

    def __init__(self, pieces):
        self.pieces = pieces
        self.num_edges = sum(p.num_edges for p in pieces)


def rank_plans(edge_index: Tensor, edge_type: Tensor, n_nodes: int, num_relations: int, tile: int,
               aggr: str, dctx: DistContext, chunk: int = 64, split: bool = False, dw_tiles: bool = False,
               paths=("ring", "ring")) -> RankPlans:
    """Plans of this rank's blocks.  The mean normaliser (a sort of all E keys) is computed ONCE and every piece is
    laid out from the same edge list: on the GPU by the library's plan builder with the piece's node range (it keeps
    the edges that scatter into the range), on the CPU (tests) by the torch form from this rank's share."""
    ranges = [dctx.node_range(s, n_nodes) for s in range(dctx.pieces)]
    if edge_type.device.type == "cuda":
        from .plan import build_graph_plans_device
        return RankPlans(build_graph_plans_device(edge_index, edge_type, n_nodes, num_relations, tile, aggr, chunk=chunk,
                                                  ranges=[(r, r) for r in ranges], split=split, dw_tiles=dw_tiles, paths=paths))
    src, dst = edge_index[0].to(torch.int64), edge_index[1].to(torch.int64)
    rel = edge_type.to(torch.int64)
    w = edge_weights(src, dst, rel, num_relations, aggr)
    world, rank = dctx.world, dctx.rank
    cuts = torch.tensor(dctx.bounds[1:], dtype=torch.int64)

    def share(scatter, gather):
        blk = torch.searchsorted(cuts, scatter, right=True)          # the block whose [begin, end) holds the node
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
                      chunk: int = 64, split: bool = False, dw_tiles: bool = False, paths=("ring", "ring"), widths=None) -> RankPlans:
    def build():
        p = paths
        if p == "auto":      # the whole graph's choice per direction (eplan.decide_paths): every rank decides alike
            from .eplan import decide_paths
            p = decide_paths(edge_index, n_nodes, num_relations, widths[0], widths[1], tile, chunk)
        return rank_plans(edge_index, edge_type, n_nodes, num_relations, tile, aggr, dctx, chunk, split, dw_tiles, p)
    return cached_graph_plans(
        edge_index, edge_type, n_nodes, num_relations, tile, aggr, chunk=chunk, split=split, dw_tiles=dw_tiles,
        paths=paths if isinstance(paths, str) else tuple(paths), widths=widths, builder=build,
        extra_key=("rank", dctx.rank, dctx.world, dctx.pieces, tuple(dctx.bounds)))


def attach(module: torch.nn.Module, n_nodes: int, n_edges: int, group=None, edge_index: Optional[Tensor] = None,
           pieces: int = PIECES, balance: Optional[bool] = None) -> None:
    """Switch every RGCNConv under ``module`` to the edge-partitioned path for the current process group
    (``n_nodes`` / ``n_edges`` of the graph the module will see: they fix the tile size and with it the
    tile-aligned node ranges; ``edge_index``: lets the cut follow the edge counts, see make_context)."""
    for m in module.modules():
        if isinstance(m, RGCNConv):
            tile = m.layout(n_nodes, n_edges)[0]
            m.dist = make_context(n_nodes, tile, group, pieces, edge_index, balance)
