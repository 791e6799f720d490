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

import os
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


# workgroups one launch round of a tile kernel holds on an MI355X (one workgroup per CU: their LDS); RGCN_CU_ROUND: rehearsals of
# the pieces-of-whole-rounds cut on graphs too small for it
CU_ROUND = int(os.environ.get("RGCN_CU_ROUND", 256))


def piece_tiles(n_tiles: int, world: int, pieces: int):
    """Tiles per block of every piece of the uniform cut (all blocks of ONE piece are equal: one in-place all-gather per piece).
    Equal pieces, unless a piece is a few launch rounds long: a launch of 1,149 one-tile workgroups takes FIVE rounds of 256 where
    its tiles fill 4.5, so four equal pieces of a rank's 4,596 tiles (headline config, world 8) walk 20 rounds where one piece
    walks 18 (measured on the emulated rank: forward 0.96 -> 1.10 ms).  Then every piece but ONE is a whole number of
    rounds and the short one comes FIRST: where the wire is what binds (2.09 ms per gather at 153 GB/s per link against ~1 ms
    of kernels at world 8) a direction ends when the first piece's kernels and then all the exchanges have run, so the first
    piece is the one to keep short (fewer pieces if nothing is left for it).
    Larger pieces (>= 16 rounds: the kernels walk several tiles per workgroup there) and smaller ones (< 1 round) stay equal."""
    per_rank = (n_tiles + world - 1) // world
    eq = (per_rank + pieces - 1) // pieces
    if pieces == 1 or eq < CU_ROUND or eq >= 16 * CU_ROUND:
        return [eq] * pieces
    b = (eq + CU_ROUND - 1) // CU_ROUND * CU_ROUND
    out, left = [], per_rank
    while left > 0 and len(out) < pieces:
        t = left if len(out) == pieces - 1 else min(b, left)
        out.append(t)
        left -= t
    if len(out) > 1 and out[-1] < CU_ROUND // 8:      # a piece of a few tiles is not worth its launches and its collective
        last = out.pop()
        out[-1] += last
    return out[::-1]


def tile_costs(edge_index: Tensor, n_nodes: int, tile: int) -> Tensor:
    """float64 [n_tiles]: rows a rank that owns the tile walks per layer step -- the edges INTO its nodes (forward plan), the
    edges OUT OF them (transposed plan) and the two root pseudo edges per node."""
    n_tiles = (n_nodes + tile - 1) // tile
    # (an end point in the tile past the last one -- make_context's sentinel for a row that is no block's cost -- falls off the end)
    c = (torch.bincount(edge_index[1] // tile, minlength=n_tiles + 1)[:n_tiles] + torch.bincount(edge_index[0] // tile, minlength=n_tiles + 1)[:n_tiles]).double()
    rows = torch.full((n_tiles,), float(tile), dtype=torch.float64, device=c.device)
    rows[-1] = n_nodes - (n_tiles - 1) * tile
    return (c + 2.0 * rows).cpu()


def block_costs(costs: Tensor, bounds, tile: int) -> Tensor:
    """cost of every block of a cut (bounds in nodes; every bound is a tile multiple except a cut's last one, the graph's
    end, which closes a partial tile: rounded UP, so that the last tile's cost counts)"""
    cum = torch.cat([torch.zeros(1, dtype=torch.float64), torch.cumsum(costs, 0)])
    t = torch.clamp(torch.tensor([(b + tile - 1) // tile for b in bounds]), max=costs.numel())
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


def heavy_edge_masks(edge_index: Tensor, edge_type: Tensor, n_nodes: int):
    """(edges of heavy (dst, relation) segments, edges of heavy (src, relation) segments) as bool [E] or None: the rows that
    eplan.SharedHeavy deals over ALL ranks, whatever block their node lies in"""
    from .eplan import HEAVY, heavy_mask
    return (heavy_mask(edge_index[1], edge_type, n_nodes, HEAVY), heavy_mask(edge_index[0], edge_type, n_nodes, HEAVY))


def make_context(n_nodes: int, tile: int, group=None, pieces: int = PIECES, edge_index: Optional[Tensor] = None,
                 balance: Optional[bool] = None, exchange: str = "full", emulate=None, edge_type: Optional[Tensor] = None,
                 split_hubs: bool = True, paths=("ep", "ep")) -> Optional[DistContext]:
    """edge_index given: keep the uniform cut (one in-place all-gather per piece) while its blocks' edge counts stay within
    BALANCE_TOLERANCE of their mean, else cut by edge count (``balance`` True / False pins the choice).
    ``exchange``: "full" | "needed" (conv.DistContext).  ``emulate = (world, rank)``: no process group -- the context of rank
    ``rank`` of a ``world``-rank job in THIS process, collectives skipped (bench.py --emulate-world)."""
    if emulate is not None:
        world, rank = int(emulate[0]), int(emulate[1])
        assert 0 <= rank < world
    else:
        if not dist.is_available() or not dist.is_initialized():
            return None
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return None
    n_tiles = (n_nodes + tile - 1) // tile
    pieces = max(1, min(pieces, n_tiles // world if n_tiles >= world else 1))
    kw = {"exchange": exchange, "emulate": emulate is not None, "split_hubs": split_hubs}
    pt = piece_tiles(n_tiles, world, pieces)
    if len(set(pt)) == 1 and len(pt) == pieces:
        ctx = DistContext(group, rank, world, piece_rows(n_nodes, tile, world, pieces), pieces, **kw)
    else:       # pieces of whole launch rounds: the blocks of one piece are equal, the pieces are not
        pieces = len(pt)
        bounds, at = [0], 0
        for t in pt:
            for _ in range(world):
                at += t * tile
                bounds.append(at)
        ctx = DistContext(group, rank, world, 0, pieces, bounds, uniform=True, **kw)
    if edge_index is not None and balance is not False:
        ei_cost, shared_rows = edge_index, 0.0
        if split_hubs and edge_type is not None and edge_index.shape[1] > 0:
            # rows of heavy segments are dealt over all ranks (eplan.SharedHeavy) wherever a direction takes the edge-parallel
            # path (``paths``: what the layer will run per direction; a direction on the tile kernels walks its hubs in their
            # block): they are not a block's cost.  What stays in a block: its light rows and one pseudo row per heavy segment.
            hf = hb = None
            if "ep" in (paths[0], paths[1]):       # (a sort of the edge list: only where a direction will use it)
                hf, hb = heavy_edge_masks(edge_index, edge_type, n_nodes)
                hf = hf if paths[0] == "ep" else None
                hb = hb if paths[1] == "ep" else None
            if hf is not None or hb is not None:
                zero = torch.zeros(edge_index.shape[1], dtype=torch.bool, device=edge_index.device)
                hf = zero if hf is None else hf
                hb = zero if hb is None else hb
                shared_rows = float(hf.sum() + hb.sum()) / world
                # an edge counts once for its destination's tile (forward) and once for its source's (transposed): drop each side
                # where it is shared -- a sentinel node past the end keeps the tensor shapes
                past = n_tiles * tile           # (a node id in the tile after the last one: tile_costs drops it)
                ei_cost = torch.stack([torch.where(hb, torch.full_like(edge_index[0], past), edge_index[0]),
                                       torch.where(hf, torch.full_like(edge_index[1], past), edge_index[1])])
        costs = tile_costs(ei_cost, n_nodes, tile)
        bc = block_costs(costs, ctx.bounds, tile)
        # (the uniform cut pads its last blocks past the graph's end: compare the heaviest block with a block's share of the
        # total, not with a mean the empty trailing blocks pull down)
        # (what has to be equal is the ranks' blocks WITHIN a piece: every block against the cost a block of its piece's length
        # holds on average -- not against a mean that the padding past the graph's end pulls down)
        bcv = bc.view(pieces, world)
        nominal = torch.tensor([float(costs.sum()) * t / n_tiles for t in pt], dtype=torch.float64).view(pieces, 1)
        uneven = bool((bcv > (1.0 + BALANCE_TOLERANCE) * nominal).any())
        if balance or uneven:
            ctx = DistContext(group, rank, world, 0, pieces, balanced_bounds(costs, n_nodes, tile, world, pieces), **kw)
            bc = block_costs(costs, ctx.bounds, tile)
        ctx.block_costs = bc.view(pieces, world)
        ctx.shared_rows_per_rank = shared_rows      # rows of heavy segments every rank sums besides its blocks' (eplan.SharedHeavy)
    return ctx


class NeededRows:
    """exchange = "needed", one direction (forward outputs or dX outputs) of one rank.  Per piece s:
    ``send_idx[s]``    int64 row ids of THIS rank's block s that some peer reads, peer after peer (rank order), ascending inside a peer;
    ``send_splits[s]`` rows per peer (0 for this rank itself);
    ``recv_idx[s]``    int64 row ids of the PEERS' blocks s that this rank reads, peer after peer, ascending;
    ``recv_splits[s]`` rows per peer.
    ``rows_needed`` / ``rows_remote``: how many of the rows other ranks own this rank reads at all (bench.py: rows_needed_fraction)."""

    def __init__(self, send_idx, send_splits, recv_idx, recv_splits, rows_needed: int, rows_remote: int):
        self.send_idx, self.send_splits, self.recv_idx, self.recv_splits = send_idx, send_splits, recv_idx, recv_splits
        self.rows_needed, self.rows_remote = rows_needed, rows_remote

    def nbytes(self) -> int:
        return sum(t.numel() * 8 for t in self.send_idx + self.recv_idx)


def needed_rows(edge_index: Tensor, n_nodes: int, dctx: DistContext):
    """(forward, transposed) NeededRows of rank ``dctx.rank``, from the replicated edge list alone -- every rank derives its own
    receive lists AND what every peer will ask of it, so no plan-time communication and every pair of ranks agrees by
    construction.  A plan of rank c gathers, in the forward direction, the rows ``src(e)`` of the edges with ``dst(e)`` in c's
    blocks (the root pseudo edges read c's own rows); the transposed plans gather ``dst(e)`` of the edges with ``src(e)`` in c's
    blocks.  One [world, N] boolean table per direction (80 MB at 10M nodes, world 8), filled by one scatter."""
    dev = edge_index.device
    w, me = dctx.world, dctx.rank
    cuts = torch.tensor(dctx.bounds[1:], dtype=torch.int64, device=dev)
    nodes = torch.arange(n_nodes, device=dev)
    owner = torch.searchsorted(cuts, nodes, right=True) % w       # the rank that owns every node
    src, dst = edge_index[0].long(), edge_index[1].long()
    out = []
    for consumer_of, row in ((dst, src), (src, dst)):
        need = torch.zeros(w, n_nodes, dtype=torch.bool, device=dev)
        need[owner[consumer_of], row] = True
        need[owner, nodes] = False          # a rank's own rows never travel
        send_idx, send_splits, recv_idx, recv_splits = [], [], [], []
        for s in range(dctx.pieces):
            b, e = dctx.node_range(s, n_nodes)
            parts = [torch.nonzero(need[c, b:e]).squeeze(1) + b for c in range(w)]
            send_idx.append(torch.cat(parts))
            send_splits.append([int(p.numel()) for p in parts])
            parts = []
            for q in range(w):
                qb, qe = dctx.node_range(s, n_nodes, q)
                parts.append(torch.nonzero(need[me, qb:qe]).squeeze(1) + qb)
            recv_idx.append(torch.cat(parts))
            recv_splits.append([int(p.numel()) for p in parts])
        out.append(NeededRows(send_idx, send_splits, recv_idx, recv_splits, int(need[me].sum()), int((owner != me).sum())))
    return out[0], out[1]


class RankPlans:
    """The graph plans of one rank: one forward / transposed pair per owned block (piece)."""

    def __init__(self, pieces, needed_fwd: Optional[NeededRows] = None, needed_bwd: Optional[NeededRows] = None, dw_rank=None,
                 shared_fwd=None, shared_bwd=None):
        self.pieces = pieces
        # eplan.SharedHeavy per direction: the heavy (node, relation) segments of the whole graph, their rows dealt over the ranks
        # (hubs split across ranks: conv.py all-reduces the partial H before the pieces' transforms)
        self.shared_fwd, self.shared_bwd = shared_fwd, shared_bwd
        self.num_edges = sum(p.num_edges for p in pieces)
        self.needed_fwd, self.needed_bwd = needed_fwd, needed_bwd      # exchange = "needed" (conv._gather_pieces)
        # full exchange: (tile-major d_weight plan, walk table) over ONE contiguous node range of this rank (dw_range) -- x and
        # the upstream gradient are replicated, so the weight gradients' cut is free: one launch per rank, not one per piece
        self.dw_rank = dw_rank


def dw_range(edge_index: Tensor, n_nodes: int, world: int, rank: int, tile: int = 320):
    """the contiguous node range whose in-edges rank ``rank`` sums into d_weight / d_root / d_bias: ``world`` ranges of about equal
    rows walked (in-edges + one root row per node), cut at multiples of ``tile``; float64 prefix sums of the replicated edge
    list, so every rank computes the same cut"""
    n_tiles = (n_nodes + tile - 1) // tile
    c = torch.bincount(edge_index[1] // tile, minlength=n_tiles).double().cpu() + float(tile)
    cum = torch.cumsum(c, 0)
    total = float(cum[-1]) if n_tiles else 0.0
    cut = [0]
    for i in range(1, world):
        t = int(torch.searchsorted(cum, torch.tensor(total * i / world, dtype=torch.float64))) + 1
        cut.append(max(cut[-1], min(t, n_tiles)))
    cut.append(n_tiles)
    return min(cut[rank] * tile, n_nodes), min(cut[rank + 1] * tile, n_nodes)


def rank_plans(edge_index: Tensor, edge_type: Tensor, n_nodes: int, num_relations: int, tile: int,
               aggr: str, dctx: DistContext, chunk: int = 64, split: bool = False, dw_tiles: bool = False,
               paths=("ring", "ring")) -> RankPlans:
    """Plans of this rank's blocks.  The mean normaliser (a sort of all E keys) is computed ONCE and every piece is
    laid out from the same edge list: on the GPU by the library's plan builder with the piece's node range (it keeps
    the edges that scatter into the range), on the CPU (tests) by the torch form from this rank's share."""
    ranges = [dctx.node_range(s, n_nodes) for s in range(dctx.pieces)]
    nf, nb = needed_rows(edge_index, n_nodes, dctx) if dctx.exchange == "needed" else (None, None)
    if edge_type.device.type == "cuda":
        from .plan import build_graph_plans_device
        from . import _lib
        # exchange = "needed": x holds the rows this rank's FORWARD blocks read, so d_weight stays on the pieces' own plans
        rank_dw = dw_tiles and dctx.exchange == "full" and paths[0] != "ep"
        extras = {}
        pcs = build_graph_plans_device(edge_index, edge_type, n_nodes, num_relations, tile, aggr, chunk=chunk,
                                       ranges=[(r, r) for r in ranges], split=split, dw_tiles=dw_tiles and not rank_dw, paths=paths,
                                       rank_dw_range=dw_range(edge_index, n_nodes, dctx.world, dctx.rank, _lib.dw_tiles_geometry()[0]) if rank_dw else None,
                                       extras=extras, hub_split=(dctx.world, dctx.rank) if dctx.split_hubs else None)
        return RankPlans(pcs, nf, nb, extras.get("dw_rank"), extras.get("shared_fwd"), extras.get("shared_bwd"))
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
    # edge-parallel directions (torch twin of the device path; tests): the pieces' eplan.EdgePlan over the whole edge list, the
    # heavy segments shared across the ranks where dctx.split_hubs
    from .eplan import HEAVY, build_edge_plan, build_shared_heavy
    sh_f = build_shared_heavy(src, dst, rel, w, n_nodes, HEAVY, world, rank) if paths[0] == "ep" and dctx.split_hubs else None
    sh_b = build_shared_heavy(dst, src, rel, w, n_nodes, HEAVY, world, rank) if paths[1] == "ep" and dctx.split_hubs else None
    out = []
    for s, (b, e) in enumerate(ranges):
        fm, bm = fpiece == s, bpiece == s
        gp = GraphPlans(fwd=None, bwd=None, num_edges=int(fm.sum()))
        if e <= b:
            gp.fwd = build_plan(fg[fm], fs[fm], fr[fm], fw[fm], n_nodes, num_relations, tile, b, e, chunk, split)
            gp.bwd = build_plan(bg[bm], bs[bm], br[bm], bw[bm], n_nodes, num_relations, tile, b, e, chunk, split)
            out.append(gp)
            continue
        if paths[0] == "ep":
            gp.ep_fwd = build_edge_plan(src, dst, rel, w, n_nodes, num_relations, b, e, heavy=HEAVY, shared=sh_f)
        else:
            gp.fwd = build_plan(fg[fm], fs[fm], fr[fm], fw[fm], n_nodes, num_relations, tile, b, e, chunk, split)
        if paths[1] == "ep":
            gp.ep_bwd = build_edge_plan(dst, src, rel, w, n_nodes, num_relations, b, e, heavy=HEAVY, shared=sh_b)
        else:
            gp.bwd = build_plan(bg[bm], bs[bm], br[bm], bw[bm], n_nodes, num_relations, tile, b, e, chunk, split)
        out.append(gp)
    return RankPlans(out, nf, nb, None, sh_f, sh_b)


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
        extra_key=("rank", dctx.rank, dctx.world, dctx.pieces, tuple(dctx.bounds), dctx.exchange, dctx.split_hubs))


def attach(module: torch.nn.Module, n_nodes: int, n_edges: int, group=None, edge_index: Optional[Tensor] = None,
           pieces: int = PIECES, balance: Optional[bool] = None, exchange: str = "full", emulate=None,
           edge_type: Optional[Tensor] = None, split_hubs: bool = True) -> None:
    """Switch every RGCNConv under ``module`` to the edge-partitioned path for the current process group
    (``n_nodes`` / ``n_edges`` of the graph the module will see: they fix the tile size and with it the
    tile-aligned node ranges; ``edge_index``: lets the cut follow the edge counts, see make_context).
    ``exchange="needed"`` (opt-in): a rank receives only the rows its plans read -- owned and read rows bit-identical to the
    full exchange, unread rows NOT written (conv.DistContext); for layers whose output feeds another partitioned layer over
    the same graph.  ``emulate=(world, rank)``: one process stands in for one rank, no collectives (bench.py).
    ``edge_type`` + ``split_hubs`` (default on): the cut leaves out the rows of heavy (node, relation) segments, which the
    edge-parallel path deals over all ranks (eplan.SharedHeavy)."""
    for m in module.modules():
        if isinstance(m, RGCNConv):
            tile, chunk = m.layout(n_nodes, n_edges)
            paths = ("ring", "ring")
            if edge_type is not None and split_hubs and edge_index is not None:
                # the path every rank will take per direction (the whole graph's choice: cached_rank_plans decides the same way)
                paths = m.path
                if paths == "auto":
                    from .eplan import decide_paths
                    paths = decide_paths(edge_index, n_nodes, m.num_relations, m.in_channels, m.out_channels, tile, chunk)
                elif isinstance(paths, str):
                    paths = (paths, paths)
            m.dist = make_context(n_nodes, tile, group, pieces, edge_index, balance, exchange, emulate, edge_type, split_hubs, tuple(paths))
