"""Edge plan: the HBM layout of the EDGE-PARALLEL path of the layer (csrc/rgcn_ep.hip).

The tile-major plan (plan.py) gives one workgroup a tile of output nodes and walks the (tile, relation) groups of its
edges chunk by chunk: right where a group fills its chunks (the 10M-node / 100M-edge / 32-relation headline: 70 rows
per group) and there are hundreds of tiles per CU.  It is the wrong shape for the graphs the reference actually trains
on (/root/reference/model/modelTrainer.py:78,92: R' = 2R + 1 = 89 / 45 / ~267 relations; graphs/AIFB/attr/sum/
AIFB_sum_in.nt: 49,838 edges onto 44 nodes, one of them with in-degree 11,825): few tiles (AIFB: 17 of 256 CUs busy),
a chunk per (tile, relation) whatever it holds, and a hub's tile walked by ONE workgroup.

Here the same arithmetic -- ``out[i] = bias + sum_e w_e (x[src_e] @ W_rel_e)`` over the edges e into i, the self loop
being relation R' with one pseudo edge per node (torch_geometric RGCNConv, called at model/layers.py:21,23) -- is cut
the other way:

* rows (merged duplicate triples, as in plan.py) sorted RELATION-MAJOR and packed into dense 64-slot UNITS (only the last
  unit of a relation is padded): ``rgcn_ep_transform`` multiplies every 16-row tile by its relation's weights on the
  matrix cores and writes the weighted products Z, one row per slot -- any number of waves side by side, no ownership;
* a destination-major index over those slots (``seg_ptr`` / ``seg_idx``): ``rgcn_ep_segment_sum`` adds the rows of every
  destination in a fixed order (bit-reproducible, no atomics) and applies bias / activation / ReLU mask.  Destinations
  with more than ``PIECE`` rows are summed in levels (pieces of at most PIECE rows, then the pieces of a destination,
  ...), so that a hub costs as many waves as it has pieces.

The same units are a dense relation-major walk for the weight-gradient kernels (``as_tile_plan``: rgcn_bwd_dw reads
nothing but slots and units), where the tile-major plan's units hold a handful of rows each on such graphs.

Cost: Z is written and read once (2 x rows x 4 x out bytes beside the gathers), which is why the headline graph stays on
the tile kernels (``choose_path``).  On the GPU the arrays come from the library's own plan builder (``build_edge_plan_device``:
rgcn_plan_build_* with layout 2 -- one relation-major "tile" -- and rgcn_eplan_segments); ``build_edge_plan`` is its torch twin
(bit-identical: tests/test_gpu_ep.py) and what the CPU-only tests walk.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch
from torch import Tensor

UNIT = 64          # slots per unit (== plan.UNIT: what the weight-gradient kernels walk)
PIECE = 256        # rows one lane group sums in a row before the sum goes through another level
HEAVY = 16         # rows of one (destination, relation) segment from which the segment is aggregated BEFORE the transform


@dataclass
class HeavyPart:
    """Aggregate-then-transform for the (destination, relation) segments that hold at least HEAVY rows (a hub's): the rows of
    such a segment share the weight matrix, so sum_e w_e (x[src_e] @ W_r) = (sum_e w_e x[src_e]) @ W_r -- their gathered rows are
    summed first (rgcn_ep_segment_sum over x itself, weighted, in levels), and ONE pseudo row per segment goes through the
    transform, the per-destination sums and the weight gradients (d_W_r += H_seg^T g[dst]).  S << E is the regime of the
    reference's summary graphs (graphs/AIFB/attr/sum/AIFB_sum_in.nt: 49,838 edges, at most 44 x 89 segments)."""
    n_seg: int
    levels: list            # [(seg_ptr int32, seg_idx int32 or None, seg_w float32 or None, n_out)]: level 0 gathers rows of x
    n_units: int
    unit_rel: Tensor
    unit_cnt: Tensor
    slot_src: Tensor        # pseudo rows gather row `segment id` of the aggregated matrix H (padding: n_seg)
    slot_w: Tensor          # 1 (the rows' weights went into H)
    slot_row: Tensor
    shared: object = None   # SharedHeavy: the segments are the WHOLE graph's (dist.py, hubs split across ranks): H comes all-reduced,
                            # ``levels`` is empty and slot_src holds global segment ids
    _tile_plan: object = field(default=None, repr=False)


@dataclass
class SharedHeavy:
    """Hubs split ACROSS ranks (round 4, SURVEY.md 8e "ranges balanced by edge count").  Under dist a hub's
    block belongs to one rank, and a heavy (scatter node, relation) segment -- node 0 of the Zipf graph: 12.7M in-edges, 32
    segments of ~400K rows -- is that rank's alone: rows walked per rank max / mean 8.3 at world 8.  Here the heavy segments of
    the WHOLE graph (one direction) form one list, their rows sorted by (segment, gathered row) and dealt over the ranks in equal
    contiguous shares; a rank sums ITS rows of every segment it touches (rgcn_ep_segment_sum over x, weighted, in levels) into a
    zero-initialised H[n_seg, width], ONE all-reduce completes H everywhere (the mean normaliser 1 / c is a plan constant, so
    partial SUMS are all that travels -- the "sum and count" of SURVEY.md 8e with the count folded in at plan time), and the
    owner of a segment's node puts one pseudo row per segment through the transform, the per-destination sums and the weight
    gradients exactly as with a rank-local HeavyPart.  At most world - 1 segments are actually cut (their sums differ from the
    single-rank order by one fp32 re-association each); every other segment is summed by one rank and the all-reduce adds zeros."""
    n_seg: int              # heavy segments of the whole graph
    seg_key: Tensor         # int64 [n_seg], ascending: relation * n_nodes + scatter node
    edge_mask: Tensor       # bool [E]: the edge belongs to a heavy segment (it leaves every rank's light units)
    n_rows: int             # heavy rows of the whole graph
    row_lo: int             # this rank's share of the sorted rows: [row_lo, row_hi)
    row_hi: int
    seg_lo: int             # first segment the share touches
    levels: list            # [(seg_ptr int32, seg_idx int32 or None, seg_w float32 or None, n_out)] over the share; outputs = H[seg_lo : seg_lo + n_out]


def build_shared_heavy(gather: Tensor, scatter: Tensor, rel: Tensor, w: Tensor, n_nodes: int, threshold: int, world: int, rank: int,
                       piece: int = PIECE) -> Optional[SharedHeavy]:
    """gather / scatter / rel / w: the WHOLE edge list of one direction (replicated on every rank: every rank derives the same
    segments and the same shares)."""
    hm = heavy_mask(scatter, rel, n_nodes, threshold)
    if hm is None:
        return None
    g, sc, r_, w_ = gather[hm].to(torch.int64), scatter[hm].to(torch.int64), rel[hm].to(torch.int64), w[hm]
    key = r_ * n_nodes + sc
    ukey, seg = torch.unique(key, return_inverse=True)
    n_seg = int(ukey.shape[0])
    order = torch.sort(seg * n_nodes + g)[1]                    # rows by (segment, gathered row)
    m = int(order.shape[0])
    lo, hi = m * rank // world, m * (rank + 1) // world
    seg_sorted = seg[order]
    levels, seg_lo = [], 0
    if hi > lo:
        seg_lo = int(seg_sorted[lo])
        cnt = torch.bincount(seg_sorted[lo:hi] - seg_lo)
        idx0 = g[order[lo:hi]].to(torch.int32)
        w0 = w_[order[lo:hi]].to(torch.float32)
        lv = segment_levels(cnt, piece)
        levels = [(p.to(torch.int32), idx0 if i == 0 else None, w0 if i == 0 else None, n) for i, (p, n) in enumerate(lv)]
    return SharedHeavy(n_seg=n_seg, seg_key=ukey, edge_mask=hm, n_rows=m, row_lo=lo, row_hi=hi, seg_lo=seg_lo, levels=levels)


def _pseudo_units(prel: Tensor, pdst: Tensor, psrc: Tensor, n_gather_rows: int, n_own: int, num_relations: int):
    """one pseudo row per heavy segment, relation-major in dense units (sequential placement): prel / pdst (row inside the owned
    range) / psrc (row of H) per segment, sorted by (relation, destination)"""
    dev = prel.device
    n_seg = int(prel.shape[0])
    r1 = num_relations + 1
    rcnt = torch.bincount(prel, minlength=r1)
    runits = (rcnt + UNIT - 1) // UNIT
    ubase = torch.cumsum(runits, 0) - runits
    rstart = torch.cumsum(rcnt, 0) - rcnt
    n_units = int(runits.sum())
    slot = ubase[prel] * UNIT + (torch.arange(n_seg, device=dev) - rstart[prel])
    slot_src = torch.full((n_units * UNIT,), n_gather_rows, dtype=torch.int32, device=dev)
    slot_w = torch.zeros(n_units * UNIT, dtype=torch.float32, device=dev)
    slot_row = torch.full((n_units * UNIT,), n_own, dtype=torch.int32, device=dev)
    slot_src[slot] = psrc.to(torch.int32)
    slot_w[slot] = 1.0
    slot_row[slot] = pdst.to(torch.int32)
    unit_rel = torch.repeat_interleave(torch.arange(r1, device=dev), runits).to(torch.int32)
    uidx = torch.arange(n_units, device=dev) - ubase[unit_rel.long()]
    used = torch.clamp(rcnt[unit_rel.long()] - uidx * UNIT, max=UNIT)
    unit_cnt = ((used + 15) // 16 * 16).to(torch.int32)
    return n_units, unit_rel, unit_cnt, slot_src, slot_w, slot_row


def shared_heavy_part(shared: SharedHeavy, n_nodes: int, node_begin: int, node_end: int, num_relations: int) -> Optional[HeavyPart]:
    """the pseudo rows of the shared segments whose node lies in [node_begin, node_end): they gather rows of the all-reduced H"""
    node = shared.seg_key % n_nodes
    mine = torch.nonzero((node >= node_begin) & (node < node_end)).squeeze(1)      # ascending: (relation, node) order
    if mine.numel() == 0:
        return None
    prel = (shared.seg_key[mine] // n_nodes).to(torch.int64)
    n_units, unit_rel, unit_cnt, slot_src, slot_w, slot_row = _pseudo_units(prel, node[mine] - node_begin, mine, shared.n_seg,
                                                                             node_end - node_begin, num_relations)
    return HeavyPart(n_seg=shared.n_seg, levels=[], n_units=n_units, unit_rel=unit_rel, unit_cnt=unit_cnt, slot_src=slot_src,
                     slot_w=slot_w, slot_row=slot_row, shared=shared)


@dataclass
class EdgePlan:
    n_nodes: int            # rows of the gathered matrix
    node_begin: int
    node_end: int
    num_relations: int      # R' (the root pseudo relation is id R')
    n_units: int
    n_rows: int             # real slots (merged edges + root pseudo edges)
    unit_rel: Tensor        # int32 [n_units]
    unit_cnt: Tensor        # int32 [n_units]  used slots of the unit rounded up to 16 (whole MFMA row tiles)
    slot_src: Tensor        # int32 [n_units * 64]  row to gather (padding: n_nodes -> zeros)
    slot_w: Tensor          # float32 [n_units * 64] (padding: 0)
    slot_row: Tensor        # int32 [n_units * 64]  destination row inside the owned range (padding: n_owned)
    levels: List[Tuple[Tensor, Optional[Tensor], int]]   # per level (seg_ptr int32 [n_out + 1], seg_idx int32 or None, n_out);
                            # level 0 indexes the rows of Z = [the units' slots; the heavy part's slots]
    max_rows_per_dst: int
    heavy: Optional[HeavyPart] = None
    _tile_plan: object = field(default=None, repr=False)

    @property
    def n_owned(self) -> int:
        return self.node_end - self.node_begin

    @property
    def device(self):
        return self.slot_src.device

    def nbytes(self) -> int:
        ts = [self.unit_rel, self.unit_cnt, self.slot_src, self.slot_w, self.slot_row]
        for p, i, _ in self.levels:
            ts.append(p)
            if i is not None:
                ts.append(i)
        if self.heavy is not None:
            h = self.heavy
            ts += [h.unit_rel, h.unit_cnt, h.slot_src, h.slot_w, h.slot_row] + [t for lv in h.levels for t in lv[:3] if t is not None]
        return sum(t.numel() * t.element_size() for t in ts)

    def heavy_tile_plan(self):
        """the heavy part's pseudo rows as a layout-2 plan for rgcn_bwd_dw (its gathered matrix is the aggregated H)"""
        h = self.heavy
        if h._tile_plan is None:
            from .plan import TilePlan
            dev = self.device
            n_own = self.n_owned
            tile = min(32768, (n_own + 15) // 16 * 16)
            n_tiles = (n_own + tile - 1) // tile
            z = torch.zeros(h.n_units, dtype=torch.int32, device=dev)
            h._tile_plan = TilePlan(
                n_nodes=h.n_seg, node_begin=self.node_begin, node_end=self.node_end, num_relations=self.num_relations, tile=tile,
                chunk=UNIT, n_tiles=n_tiles, n_chunks=h.n_units, n_edges=h.n_seg,
                tile_ptr=torch.zeros(n_tiles + 1, dtype=torch.int32, device=dev), chunk_rel=h.unit_rel, chunk_cnt=h.unit_cnt,
                chunk_tile=z, chunk_flags=z, rel_order=torch.arange(h.n_units, dtype=torch.int32, device=dev), slot_src=h.slot_src,
                slot_w=h.slot_w, slot_dstl=None, slot_row=h.slot_row, slot_acc=h.slot_row, layout=2)
        return h._tile_plan

    def as_tile_plan(self):
        """The units as a plan.TilePlan the relation-major weight-gradient kernels accept (plan layout 2: they read
        rel_order, chunk_rel, chunk_cnt and the slot arrays only; there are no tiles to walk -- rgcn_fwd / rgcn_bwd_dx
        refuse such a plan)."""
        if self._tile_plan is None:
            from .plan import TilePlan
            dev = self.device
            n_own = self.n_owned
            tile = min(32768, (n_own + 15) // 16 * 16)
            n_tiles = (n_own + tile - 1) // tile
            z = torch.zeros(self.n_units, dtype=torch.int32, device=dev)
            self._tile_plan = TilePlan(
                n_nodes=self.n_nodes, node_begin=self.node_begin, node_end=self.node_end, num_relations=self.num_relations,
                tile=tile, chunk=UNIT, n_tiles=n_tiles, n_chunks=self.n_units, n_edges=self.n_rows - n_own,
                tile_ptr=torch.zeros(n_tiles + 1, dtype=torch.int32, device=dev), chunk_rel=self.unit_rel,
                chunk_cnt=self.unit_cnt, chunk_tile=z, chunk_flags=z,
                rel_order=torch.arange(self.n_units, dtype=torch.int32, device=dev), slot_src=self.slot_src,
                slot_w=self.slot_w, slot_dstl=None, slot_row=self.slot_row, slot_acc=self.slot_row, layout=2)
        return self._tile_plan


def segment_levels(counts: Tensor, piece: int = PIECE):
    """Reduction levels over segments of ``counts[i]`` consecutive input rows.  One level if no segment is longer than
    ``piece``; else level 0 sums pieces of at most ``piece`` consecutive rows and the next levels sum the pieces of a
    segment the same way.  Returns [(seg_ptr int64 [n_out + 1], n_out)], level 0 over the input rows."""
    levels = []
    cnt = counts.to(torch.int64)
    while True:
        if int(cnt.max()) <= piece if cnt.numel() else True:
            ptr = torch.zeros(cnt.numel() + 1, dtype=torch.int64, device=cnt.device)
            ptr[1:] = torch.cumsum(cnt, 0)
            levels.append((ptr, int(cnt.numel())))
            return levels
        npieces = (cnt + piece - 1) // piece                       # pieces of every segment (0 for an empty one)
        seg_of_piece = torch.repeat_interleave(torch.arange(cnt.numel(), device=cnt.device), npieces)
        first_piece = torch.cumsum(npieces, 0) - npieces
        k = torch.arange(seg_of_piece.numel(), device=cnt.device) - first_piece[seg_of_piece]
        start = (torch.cumsum(cnt, 0) - cnt)[seg_of_piece] + k * piece
        size = torch.minimum(cnt[seg_of_piece] - k * piece, torch.full_like(k, piece))
        ptr = torch.cat([start, (start[-1:] + size[-1:])])
        levels.append((ptr, int(seg_of_piece.numel())))
        cnt = npieces


def heavy_mask(scatter: Tensor, rel: Tensor, n_nodes: int, threshold: int) -> Optional[Tensor]:
    """bool [E]: the edge belongs to a (scatter node, relation) segment of at least ``threshold`` edges (duplicates counted);
    None when there is no such segment.  One sort of the E segment keys, at plan time."""
    if threshold <= 0 or scatter.numel() == 0:
        return None
    # A segment of `threshold` edges needs a scatter node of at least that degree: one bincount over the nodes decides for
    # most graphs (none: the uniform 10M / 100M graph's busiest node has ~30 in-edges) and leaves the sort -- torch.unique over
    # E 64-bit keys with inverse and counts, seconds at 100M edges and minutes with four ranks sharing one card -- to the edges
    # at such nodes.  (Negative ids, a caller's "not mine", are never heavy.)
    sc = scatter.to(torch.int64)
    deg = torch.bincount(sc.clamp(min=0), minlength=max(n_nodes, 1))
    if int(deg.max()) < threshold:
        return None
    cand = torch.nonzero((deg[sc.clamp(min=0)] >= threshold) & (sc >= 0)).squeeze(1)
    if cand.numel() == 0:
        return None
    key = rel[cand].to(torch.int64) * n_nodes + sc[cand]
    _, inv, cnt = torch.unique(key, return_inverse=True, return_counts=True)
    if int(cnt.max()) < threshold:
        return None
    mask = torch.zeros(scatter.shape[0], dtype=torch.bool, device=scatter.device)
    mask[cand] = cnt[inv] >= threshold
    return mask


def build_heavy_part(gather: Tensor, loc: Tensor, rel: Tensor, w: Tensor, n_own: int, num_relations: int, piece: int) -> HeavyPart:
    """gather / loc (scatter row inside the owned range) / rel / w of the edges of the heavy segments."""
    dev = gather.device
    key = rel.to(torch.int64) * max(n_own, 1) + loc.to(torch.int64)
    ukey, seg = torch.unique(key, return_inverse=True)                 # segments sorted by (relation, destination)
    n_seg = int(ukey.shape[0])
    order = torch.sort(seg * (int(gather.max()) + 1 if gather.numel() else 1) + gather.to(torch.int64))[1]      # rows by (segment, gathered row)
    idx0 = gather[order].to(torch.int32)
    w0 = w[order].to(torch.float32)
    cnt = torch.bincount(seg, minlength=n_seg)
    lv = segment_levels(cnt, piece)
    levels = [(p.to(torch.int32), idx0 if i == 0 else None, w0 if i == 0 else None, n) for i, (p, n) in enumerate(lv)]
    # one pseudo row per segment, relation-major in dense units (sequential placement)
    prel = ukey // max(n_own, 1)
    pdst = ukey % max(n_own, 1)
    n_units, unit_rel, unit_cnt, slot_src, slot_w, slot_row = _pseudo_units(prel, pdst, torch.arange(n_seg, device=dev), n_seg, n_own,
                                                                             num_relations)
    return HeavyPart(n_seg=n_seg, levels=levels, n_units=n_units, unit_rel=unit_rel, unit_cnt=unit_cnt, slot_src=slot_src,
                     slot_w=slot_w, slot_row=slot_row)


def _finish_levels(slot_row_light: Tensor, heavy: Optional[HeavyPart], n_own: int, piece: int, on_device: bool):
    """destination-major index over Z = [light slots; heavy slots] and its sum levels"""
    rows = slot_row_light if heavy is None else torch.cat([slot_row_light, heavy.slot_row])
    if on_device:
        from . import _lib
        seg_ptr, seg_idx = _lib.eplan_segments(rows, n_own)
        n_rows = int(seg_ptr[-1])
        seg_idx = seg_idx[:n_rows]
        counts = (seg_ptr[1:] - seg_ptr[:-1]).to(torch.int64)
    else:
        n_slots = int(rows.numel())
        real = torch.nonzero(rows < n_own).squeeze(1)
        seg_idx = (torch.sort(rows[real].to(torch.int64) * max(n_slots, 1) + real)[0] % max(n_slots, 1)).to(torch.int32)
        counts = torch.bincount(rows[real].to(torch.int64), minlength=n_own)
        n_rows = int(real.numel())
    lv = segment_levels(counts, piece)
    levels = [(p.to(torch.int32), seg_idx if i == 0 else None, n) for i, (p, n) in enumerate(lv)]
    return levels, n_rows, (int(counts.max()) if n_own else 0)


def build_edge_plan(gather: Tensor, scatter: Tensor, rel: Tensor, w: Tensor, n_nodes: int, num_relations: int,
                    node_begin: int = 0, node_end: Optional[int] = None, piece: int = PIECE, heavy: int = 0,
                    shared: Optional[SharedHeavy] = None) -> EdgePlan:
    """gather / scatter: int64 [E] node ids (forward: src / dst; transposed: dst / src); w: the edge weights of
    plan.edge_weights (1 / max(1, c[dst, rel]) for aggr = 'mean'), kept for both directions.  heavy: segments of at least
    that many rows are aggregated before the transform (HeavyPart); 0: every row goes through the transform."""
    if node_end is None:
        node_end = n_nodes
    dev = gather.device
    n_own = node_end - node_begin
    r1 = num_relations + 1
    gather, scatter, rel = gather.to(torch.int64), scatter.to(torch.int64), rel.to(torch.int64)
    if rel.numel() and (int(rel.min()) < 0 or int(rel.max()) >= num_relations):
        raise ValueError("edge_type out of range [0, num_relations)")
    if gather.numel() and (int(gather.min()) < 0 or int(gather.max()) >= n_nodes
                           or int(scatter.min()) < 0 or int(scatter.max()) >= n_nodes):
        raise ValueError("edge_index out of range [0, num_nodes)")
    if shared is not None:      # the whole graph's heavy segments, their rows dealt over the ranks (SharedHeavy): not this plan's rows
        keep_ = ~shared.edge_mask
        gather, scatter, rel, w = gather[keep_], scatter[keep_], rel[keep_], w[keep_]
    own = (scatter >= node_begin) & (scatter < node_end)
    if not bool(own.all()):
        gather, scatter, rel, w = gather[own], scatter[own], rel[own], w[own]
    hp = shared_heavy_part(shared, n_nodes, node_begin, node_end, num_relations) if shared is not None else None
    hm = heavy_mask(scatter, rel, n_nodes, heavy) if shared is None else None
    if hm is not None:
        hp = build_heavy_part(gather[hm], scatter[hm] - node_begin, rel[hm], w[hm], n_own, num_relations, piece)
        gather, scatter, rel, w = gather[~hm], scatter[~hm], rel[~hm], w[~hm]
    nodes = torch.arange(node_begin, node_end, device=dev, dtype=torch.int64)
    g_all = torch.cat([gather, nodes])
    loc = torch.cat([scatter, nodes]) - node_begin
    r_all = torch.cat([rel, torch.full((n_own,), num_relations, device=dev, dtype=torch.int64)])
    w_all = torch.cat([w.to(torch.float32), torch.ones(n_own, device=dev, dtype=torch.float32)])
    # relation-major, then destination, then gathered row; duplicate (gather, scatter, relation) triples share one slot
    # whose weight is the sum of theirs (plan.build_plan explains why: thousands of identical terms on the summary graphs)
    key, perm = torch.sort((r_all * max(n_own, 1) + loc) * n_nodes + g_all)
    w_all = w_all[perm]
    key, inv = torch.unique_consecutive(key, return_inverse=True)
    if key.shape[0] != w_all.shape[0]:
        w_all = torch.zeros(key.shape[0], dtype=torch.float64, device=dev).index_add_(0, inv, w_all.to(torch.float64)).to(torch.float32)
    n_rows = int(key.shape[0])
    g_all = key % n_nodes
    rd = key // n_nodes
    loc = rd % max(n_own, 1)
    r_all = rd // max(n_own, 1)
    rcnt = torch.bincount(r_all, minlength=r1)
    runits = (rcnt + UNIT - 1) // UNIT
    ubase = torch.cumsum(runits, 0) - runits
    rstart = torch.cumsum(rcnt, 0) - rcnt
    n_units = int(runits.sum())
    n_slots = n_units * UNIT
    # rows of a relation are dealt over its row tiles exactly as the tile-major plan deals a group (plan.build_plan layout 0: row
    # j -> row tile j mod nt, place j div nt, nt = ceil(rows / 16)): the device builder lays a relation-major plan out as ONE tile
    # (csrc/rgcn_plan.hip layout 2) and this is its torch twin, bit for bit
    rnt = (rcnt + 15) // 16
    j = torch.arange(n_rows, device=dev) - rstart[r_all]
    slot = ubase[r_all] * UNIT + (j % rnt[r_all]) * 16 + j // rnt[r_all]
    slot_src = torch.full((n_slots,), n_nodes, dtype=torch.int32, device=dev)
    slot_w = torch.zeros(n_slots, dtype=torch.float32, device=dev)
    slot_row = torch.full((n_slots,), n_own, dtype=torch.int32, device=dev)
    slot_src[slot] = g_all.to(torch.int32)
    slot_w[slot] = w_all
    slot_row[slot] = loc.to(torch.int32)
    unit_rel = torch.repeat_interleave(torch.arange(r1, device=dev), runits).to(torch.int32)
    uidx = torch.arange(n_units, device=dev) - ubase[unit_rel.long()]
    unit_cnt = (torch.clamp(rnt[unit_rel.long()] - uidx * (UNIT // 16), max=UNIT // 16) * 16).to(torch.int32)
    # destination-major index over the slots, a destination's rows in slot order (= relation-major): rgcn_eplan_segments
    levels, n_all, max_rows = _finish_levels(slot_row, hp, n_own, piece, False)
    return EdgePlan(n_nodes=n_nodes, node_begin=node_begin, node_end=node_end, num_relations=num_relations, n_units=n_units,
                    n_rows=n_rows, unit_rel=unit_rel, unit_cnt=unit_cnt, slot_src=slot_src, slot_w=slot_w, slot_row=slot_row,
                    levels=levels, max_rows_per_dst=max_rows, heavy=hp)


def build_edge_plan_device(graph, w: Tensor, transposed: bool, n_nodes: int, num_relations: int, ws: Tensor,
                           node_begin: int = 0, node_end: Optional[int] = None, piece: int = PIECE, heavy: int = 0,
                           edge_index: Optional[Tensor] = None, edge_type: Optional[Tensor] = None,
                           shared: Optional[SharedHeavy] = None, light_graph=None) -> EdgePlan:
    """The same plan by the library's own builder (csrc/rgcn_plan.hip): rgcn_plan_build_begin / _finish with layout 2 lay the
    owned range out as one relation-major "tile", rgcn_eplan_segments sorts the slots by destination; only the sum levels of
    hubs (arithmetic on seg_ptr) and the split-off of the heavy segments (``heavy`` > 0; needs the COO tensors) stay here.
    graph / w / ws: _lib.graph_struct, _lib.edge_weights, _lib.plan_workspace."""
    from . import _lib
    if node_end is None:
        node_end = n_nodes
    n_own = node_end - node_begin
    hp, keep = None, None
    if shared is not None:
        # hubs split across ranks: the light graph (the edge list without the whole graph's heavy segments; built once per
        # direction by the caller: light_graph = (graph struct, its weights)) and the pseudo rows of this range's segments
        graph, w = light_graph
        hp = shared_heavy_part(shared, n_nodes, node_begin, node_end, num_relations)
    elif heavy > 0 and edge_index is not None and int(edge_type.shape[0]) > 0:
        g_, s_ = (edge_index[1], edge_index[0]) if transposed else (edge_index[0], edge_index[1])
        own = (s_ >= node_begin) & (s_ < node_end)
        hm = heavy_mask(torch.where(own, s_, torch.full_like(s_, -1)), edge_type, n_nodes + 1, heavy)      # (-1: not owned, its own keys)
        if hm is not None:
            hm = hm & own
            if bool(hm.any()):
                hp = build_heavy_part(g_[hm], s_[hm] - node_begin, edge_type[hm], w[hm], n_own, num_relations, piece)
                lm = ~hm
                ei_l, et_l, w = edge_index[:, lm].contiguous(), edge_type[lm].contiguous(), w[lm].contiguous()
                graph, keep = _lib.graph_struct(ei_l, et_l, n_nodes, num_relations)
    ps, a, n_edges = _lib.plan_build(graph, w, transposed, node_begin, node_end, 16, UNIT, ws, 2)
    levels, n_all, max_rows = _finish_levels(a["slot_row"], hp, n_own, piece, True)
    n_rows = n_all - (hp.n_seg if hp is not None else 0)
    ep = EdgePlan(n_nodes=n_nodes, node_begin=node_begin, node_end=node_end, num_relations=num_relations, n_units=int(ps.n_chunks),
                  n_rows=n_rows, unit_rel=a["chunk_rel"], unit_cnt=a["chunk_cnt"], slot_src=a["slot_src"], slot_w=a["slot_w"],
                  slot_row=a["slot_row"], levels=levels, max_rows_per_dst=max_rows, heavy=hp)
    # the same arrays are the relation-major walk of the weight-gradient kernels: the struct the builder filled
    from .plan import TilePlan
    tp = TilePlan(n_nodes=n_nodes, node_begin=node_begin, node_end=node_end, num_relations=num_relations, tile=int(ps.tile),
                  chunk=UNIT, n_tiles=int(ps.n_tiles), n_chunks=int(ps.n_chunks), n_edges=n_edges, slot_dstl=None, layout=2, **a)
    tp._cstruct = ps
    ep._tile_plan = tp
    del keep
    return ep


# ---- which path: cost model of one forward / dX launch ---------------------------------------------------------------
def ring_launch_us(n_nodes: int, n_edges: int, num_relations: int, width: int, tile: int, chunk: int,
                   max_tile_rows: int) -> float:
    """Time of a tile-kernel launch in microseconds: a chunk costs ~0.7 us plus ~0.2 us per 16-row tile at 64 columns
    (stamp builds; the AIFB / MUTAG shapes of the bench ladder), a workgroup walks its tiles' chunks one after the other, 256 workgroups at a
    time -- and the tile with the most rows (a hub's) is walked by ONE workgroup."""
    import math
    r1 = max(1, num_relations)
    group = n_edges / max(1.0, float(n_nodes) * r1) * tile          # expected rows of a (tile, relation) group
    wscale = max(16, width) / 64.0

    def group_us(rows):
        return math.ceil(rows / chunk) * 0.7 + 0.2 * wscale * math.ceil(rows / 16.0)

    # the relations present in a tile: all of them once a group expects a few rows, else the expected number of non-empty groups
    present = r1 * (1.0 - math.exp(-group)) if group < 8 else r1
    per_tile = present * group_us(max(group, 1.0)) + group_us(float(tile))
    rounds = math.ceil(math.ceil(n_nodes / tile) / 256)
    return max(rounds * per_tile, group_us(float(max_tile_rows)))


def ep_launch_us(n_nodes: int, n_edges: int, in_width: int, out_width: int) -> float:
    """Edge-parallel path: gathers + Z written and read once at ~4 TB/s, plus two launches."""
    rows = n_edges + n_nodes
    return rows * (4.0 * in_width + 8.0 * out_width + 24.0) / 4.0e6 + 12.0


def choose_path(n_nodes: int, n_edges: int, num_relations: int, in_width: int, out_width: int, tile: int, chunk: int,
                max_tile_rows: int) -> str:
    """'ep' where the edge-parallel path is expected to be faster than the tile kernel for this direction."""
    ring = ring_launch_us(n_nodes, n_edges, num_relations, max(in_width, out_width), tile, chunk, max_tile_rows)
    ep = ep_launch_us(n_nodes, n_edges, in_width, out_width)
    # (calibrated on the bench ladder, round 3: MUTAG shape -- tile kernels 0.169 ms per step replayed, edge-parallel 0.117 -- is
    # the closest call: estimates 35 us against 29 us per launch)
    return "ep" if ep < 0.9 * ring else "ring"


def decide_paths(edge_index: Tensor, n_nodes: int, num_relations: int, in_channels: int, out_channels: int, tile: int,
                 chunk: int) -> Tuple[str, str]:
    """(forward path, dX path) for a layer on this graph.  One pass over the edge list per direction (rows per tile: a hub's
    tile is walked by one workgroup of the tile kernel) and one host read of the two maxima -- at plan time only."""
    e = int(edge_index.shape[1])
    n_tiles = (n_nodes + tile - 1) // tile
    if e:
        mx = torch.stack([torch.bincount(edge_index[1] // tile, minlength=n_tiles).max(),
                          torch.bincount(edge_index[0] // tile, minlength=n_tiles).max()]).tolist()
    else:
        mx = [0, 0]
    fwd = choose_path(n_nodes, e, num_relations, in_channels, out_channels, tile, chunk, int(mx[0]) + tile)
    bwd = choose_path(n_nodes, e, num_relations, out_channels, in_channels, tile, chunk, int(mx[1]) + tile)
    return fwd, bwd
