"""Graph plan: the HBM layout the HIP kernels walk.

Built once per (graph, tile size, direction, node range) from the COO tensors the
reference hands to the layer (``edge_index[2,E]`` int64, ``edge_type[E]`` int64, unsorted,
duplicates kept: /root/reference/graphs/graph.py:55-69) and cached, so the public
``forward(x, edge_index, edge_type)`` signature stays PyG's (SURVEY.md 8b "Ownership").
PyG recomputes masks and counts on every call; here they are computed once.

Layout (all int32 / float32, device resident):

* output nodes of the owned range ``[node_begin, node_end)`` are cut into TILES of ``tile``
  consecutive nodes; one workgroup owns a tile's accumulator in LDS;
* the edges scattering into a tile are grouped by relation; each (tile, relation) group is cut
  into CHUNKS of ``CHUNK`` = 64 edge slots (the last one padded), so a chunk is
  relation-homogeneous (one MFMA B operand) and exactly one LDS-DMA ring slot;
* the self-loop (PyG ``root``) is relation id ``num_relations`` with one pseudo edge per node;
* per slot: ``slot_src`` (row to gather; padding = ``n_nodes``, one past the last row, which a buffer-descriptor
  gather turns into zeros by its range check), ``slot_w`` (edge weight
  ``1 / max(1, c[dst, rel])`` for ``aggr='mean'``, 0 = padding), ``slot_dstl`` (row inside the tile,
  ``tile`` = padding; host-side only) and ``slot_row`` (the same as a row of the owned range, padding = one past the
  end: what the dW kernel gathers the upstream gradient by); inside a chunk the slots are
  sorted by ``slot_dstl``, which the forward kernel's run-sum relies on; ``slot_acc`` packs what that
  run-sum needs per slot (run-end position << 24 | accumulator row, see ``run_metadata``);
  duplicate (src, dst, relation) triples share ONE slot whose weight is the sum of theirs;
* per chunk: ``chunk_rel``, ``chunk_cnt`` (slots of the used 16-slot row tiles), ``chunk_tile``, ``chunk_flags``; ``tile_ptr`` gives the tile-major
  chunk ranges (forward / dX kernels) and ``rel_order`` the relation-major order (dW kernel).

The transposed plan (``direction='bwd'``) swaps the roles of source and destination and keeps
the forward weights, so dX is the same kernel run on it with W_r^T (SURVEY.md 7 "Backward dX
without atomics").  Everything here is torch tensor plumbing and runs on CPU or GPU alike.
"""
from __future__ import annotations

import functools

import numpy as np
from os import environ as _env
_os_environ_get = _env.get
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

CHUNK = 64  # default edge slots per chunk == rows of one LDS ring slot (csrc/rgcn_common.h kChunk)
UNIT = 64   # rows the weight-gradient kernels walk at a time: a chunk is chunk // UNIT units
CHUNKS = (64, 128)   # chunk sizes (slot strides) the kernels are built for
CHUNK_112 = 112      # as the ``chunk`` argument of the builders: 128-slot chunks that hold at most SEVEN row tiles (112 rows) of rows
                     # -- what rgcn_tile3p_kernel's 42 KiB ring slots hold, leaving its accumulator room for tiles up to 272
                     # (TilePlan.chunk stays 128, TilePlan.chunk_rows says 112)


@dataclass
class TilePlan:
    n_nodes: int          # nodes of the whole graph (rows of the gathered matrix)
    node_begin: int       # first owned output node (multiple of tile)
    node_end: int         # one past the last owned output node
    num_relations: int    # R' (the root pseudo relation is id R')
    tile: int
    chunk: int            # edge slots per chunk (64 or 128)
    n_tiles: int
    n_chunks: int
    n_edges: int          # real edges placed (before merging duplicate triples; no root / padding)
    tile_ptr: Tensor      # int32 [n_tiles + 1]
    chunk_rel: Tensor     # int32 [n_chunks]
    chunk_cnt: Tensor     # int32 [n_chunks]
    chunk_tile: Tensor    # int32 [n_chunks]
    chunk_flags: Tensor   # int32 [n_chunks]  bit t: MFMA row tile t of the chunk holds a repeated destination
    rel_order: Tensor     # int32 [n_units]  non-empty 64-row units (unit u = rows [64 u, 64 u + 64) of the slot
                          #   arrays, chunk u // (chunk // 64)), relation-major then tile: the dW kernels' walk
    slot_src: Tensor      # int32 [n_chunks * chunk]
    slot_w: Tensor        # float32 [n_chunks * chunk]
    slot_dstl: Optional[Tensor]   # int32 [n_chunks * chunk]  row inside the tile (padding: tile); host-side checks only:
                          #   None for plans built on the device
    slot_row: Tensor      # int32 [n_chunks * chunk]  row inside the owned range = tile index * tile + slot_dstl (padding: n_owned)
    slot_acc: Tensor      # int32 [n_chunks * chunk]  run-end position << 24 | accumulator row
    layout: int = 0       # 0: rows of a group dealt over all its row tiles; 1: team placement (team_placement);
    #                       3: layout 0 with the runs of equal (destination, relation) on one slot each (compact_runs)
    chunk_rows: int = 0   # rows a chunk may hold: 0 / chunk, or 112 (chunk = 128: seven row tiles of rows per chunk)
    slot_src2: Optional[Tensor] = None    # layout 5 (the tile-major weight-gradient plan): int32 [n_chunks * 8] second rows of the
                                          # slots 0..3 and 32..35 of every unit (padding: n_nodes), see dw_pairs
    _keep: tuple = field(default=(), repr=False)

    @property
    def device(self):
        return self.slot_src.device

    @property
    def n_owned(self) -> int:
        return self.node_end - self.node_begin

    @property
    def n_units(self) -> int:
        return int(self.rel_order.shape[0])

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (
            self.tile_ptr, self.chunk_rel, self.chunk_cnt, self.chunk_tile, self.chunk_flags, self.rel_order,
            self.slot_src, self.slot_w, self.slot_row, self.slot_acc))


def edge_weights(src: Tensor, dst: Tensor, rel: Tensor, num_relations: int, aggr: str = "mean") -> Tensor:
    """w_e = 1 / max(1, c[dst_e, rel_e]) (duplicates counted), float32, in input edge order."""
    if aggr in ("sum", "add"):
        return torch.ones(dst.shape[0], dtype=torch.float32, device=dst.device)
    if aggr != "mean":
        raise ValueError(f"unsupported aggr {aggr!r}")
    key = dst.to(torch.int64) * num_relations + rel.to(torch.int64)
    skey, perm = torch.sort(key)
    _, inv, cnt = torch.unique_consecutive(skey, return_inverse=True, return_counts=True)
    w_sorted = 1.0 / cnt.to(torch.float32)[inv]
    w = torch.empty_like(w_sorted)
    w[perm] = w_sorted
    return w


def build_plan(gather: Tensor, scatter: Tensor, rel: Tensor, w: Tensor, n_nodes: int,
               num_relations: int, tile: int, node_begin: int = 0,
               node_end: Optional[int] = None, chunk: int = CHUNK, split: bool = False) -> TilePlan:
    """Lay out the edges scattering into ``[node_begin, node_end)``.

    gather / scatter: int64 [E] node ids (forward: src / dst; transposed: dst / src).
    chunk: edge slots per chunk (one of CHUNKS).
    split: the plan layout -- False / 0: layout 0; True / 1: the TEAM placement (128-slot chunks only), see ``team_placement``;
           3: layout 0 followed by ``compact_runs`` (128-slot chunks only).
    """
    cap = None            # row tiles a chunk may hold (None: chunk / 16)
    if chunk == CHUNK_112:
        if int(split) not in (0, 3):
            raise ValueError("112-row chunks: layouts 0 and 3 only")
        if int(split) == 3:
            return compact_runs(build_plan(gather, scatter, rel, w, n_nodes, num_relations, tile, node_begin, node_end, chunk, False))
        chunk, cap = 128, 7
    if chunk not in CHUNKS:
        raise ValueError(f"chunk must be one of {CHUNKS} (or {CHUNK_112})")
    if int(split) == 5:
        if chunk != 64:
            raise ValueError("layout 5 (pairs on one slot: the tile-major weight-gradient plan) needs 64-slot chunks")
        return dw_pairs(build_plan(gather, scatter, rel, w, n_nodes, num_relations, tile, node_begin, node_end, chunk, False))
    if split and chunk != 128:
        raise ValueError("the team placement and the run compaction need 128-slot chunks")
    if int(split) == 3:
        return compact_runs(build_plan(gather, scatter, rel, w, n_nodes, num_relations, tile, node_begin, node_end, chunk, False))
    CHUNK = chunk  # noqa: N806  (shadows the module default inside this function)
    if node_end is None:
        node_end = n_nodes
    if node_end <= node_begin:
        return empty_plan(n_nodes, node_begin, num_relations, tile, chunk, gather.device, 1 if split else 0)
    if node_begin % tile != 0:
        raise ValueError("node_begin must be a multiple of the tile size")
    dev = gather.device
    n_own = node_end - node_begin
    r1 = num_relations + 1
    gather = gather.to(torch.int64)
    scatter = scatter.to(torch.int64)
    rel = rel.to(torch.int64)
    if rel.numel() and (int(rel.min()) < 0 or int(rel.max()) >= num_relations):
        raise ValueError("edge_type out of range [0, num_relations)")
    if gather.numel() and (int(gather.min()) < 0 or int(gather.max()) >= n_nodes
                           or int(scatter.min()) < 0 or int(scatter.max()) >= n_nodes):
        raise ValueError("edge_index out of range [0, num_nodes)")
    own = (scatter >= node_begin) & (scatter < node_end)
    if not bool(own.all()):
        gather, scatter, rel, w = gather[own], scatter[own], rel[own], w[own]
    n_edges = int(gather.shape[0])
    # root pseudo edges: node i gathers its own row with weight 1 under relation id R'
    nodes = torch.arange(node_begin, node_end, device=dev, dtype=torch.int64)
    g_all = torch.cat([gather, nodes])
    loc = torch.cat([scatter, nodes]) - node_begin
    r_all = torch.cat([rel, torch.full((n_own,), num_relations, device=dev, dtype=torch.int64)])
    w_all = torch.cat([w.to(torch.float32), torch.ones(n_own, device=dev, dtype=torch.float32)])
    tile_id = loc // tile
    dstl = loc - tile_id * tile
    key = (tile_id * r1 + r_all) * tile + dstl
    # Sort by (tile, relation, row in tile, gathered node) and MERGE duplicate (gather, scatter, relation)
    # triples into one slot whose weight is the sum of theirs (m / c for a triple repeated m times):
    # identical under the multiset semantics of the layer, but a repeated triple is gathered and
    # multiplied once, and -- decisive for the reference's summary graphs, where ~50k edges join 44
    # nodes -- thousands of bit-identical terms are no longer added one by one into an fp32
    # accumulator (that sum has a systematic rounding bias ~n*u; it cost 4e-4 on AIFB_sum_in).
    key2, perm = torch.sort(key * n_nodes + g_all)
    w_all = w_all[perm]
    key2, inv, mult = torch.unique_consecutive(key2, return_inverse=True, return_counts=True)
    if key2.shape[0] != w_all.shape[0]:
        w_all = torch.zeros(key2.shape[0], dtype=torch.float64, device=dev).index_add_(
            0, inv, w_all.to(torch.float64)).to(torch.float32)
    key = key2 // n_nodes
    g_all = key2 - key * n_nodes
    dstl = key % tile
    gk = key // tile
    gvals, gcnt = torch.unique_consecutive(gk, return_counts=True)
    per = cap if cap is not None else CHUNK // ROWS_PER_MFMA_TILE       # row tiles a chunk holds
    gch = ((gcnt + 15) // 16 + (per - 1)) // per if cap is not None else (gcnt + (CHUNK - 1)) // CHUNK
    chunk_base = torch.cumsum(gch, 0) - gch
    n_chunks = int(gch.sum())
    n_groups = gvals.shape[0]
    gstart = torch.cumsum(gcnt, 0) - gcnt
    grp_of_edge = torch.repeat_interleave(torch.arange(n_groups, device=dev), gcnt)
    rank = torch.arange(key.shape[0], device=dev) - gstart[grp_of_edge]
    # Row j of a group (sorted by destination) goes to MFMA row tile (j mod nt), position (j div nt), nt =
    # ceil(n / 16): every tile stays sorted by destination, the tiles of a group are equally full, and a run
    # of c equal destinations is spread over c different tiles (c <= nt), so most tiles hold pairwise distinct
    # destinations and the forward kernel can skip its run-sum product for them (chunk_flags).
    g16 = ROWS_PER_MFMA_TILE
    gnt = (gcnt + (g16 - 1)) // g16
    grp_of_chunk = torch.repeat_interleave(torch.arange(n_groups, device=dev), gch)
    idx_in_grp = torch.arange(n_chunks, device=dev) - chunk_base[grp_of_chunk]
    straddle = None
    if not split:
        nt_e = gnt[grp_of_edge]
        t_e = rank % nt_e                         # the group's row tile: chunk t // per of the group, its row tile t % per
        slot = (chunk_base[grp_of_edge] + t_e // per) * CHUNK + (t_e % per) * g16 + rank // nt_e
        # slots of the chunk's used row tiles (a multiple of 16; padding sits at the end of every tile)
        chunk_cnt = (torch.clamp(gnt[grp_of_chunk] - idx_in_grp * per, max=per) * g16).to(torch.int32)
    else:
        slot, chunk_cnt, straddle = team_placement(dstl, gcnt, grp_of_edge, rank, chunk_base, grp_of_chunk, idx_in_grp)
    n_slots = n_chunks * CHUNK
    slot_src = torch.full((n_slots,), n_nodes, dtype=torch.int32, device=dev)  # padding: one past the last row
    slot_w = torch.zeros(n_slots, dtype=torch.float32, device=dev)
    slot_dstl = torch.full((n_slots,), tile, dtype=torch.int32, device=dev)  # padding -> dummy accumulator row
    slot_src[slot] = g_all.to(torch.int32)
    slot_w[slot] = w_all
    slot_dstl[slot] = dstl.to(torch.int32)
    slot_acc, tile_dup = run_metadata(slot_dstl, tile)
    chunk_flags = (tile_dup.view(-1, CHUNK // g16).to(torch.int32)
                   * (2 ** torch.arange(CHUNK // g16, device=dev, dtype=torch.int32))).sum(1).to(torch.int32)
    if straddle is not None:
        chunk_flags = chunk_flags | (straddle.to(torch.int32) << 8)
    # the dW kernel gathers the upstream-gradient row of every slot: its index in the owned range
    tile_of_slot = torch.repeat_interleave((gvals[grp_of_chunk] // r1).to(torch.int32), CHUNK)
    slot_row = torch.where(slot_dstl < tile, tile_of_slot * tile + slot_dstl,
                           torch.full_like(slot_dstl, n_own))
    chunk_rel = (gvals[grp_of_chunk] % r1).to(torch.int32)
    chunk_tile = (gvals[grp_of_chunk] // r1).to(torch.int32)
    n_tiles = (n_own + tile - 1) // tile
    per_tile = torch.bincount(chunk_tile.to(torch.int64), minlength=n_tiles)
    tile_ptr = torch.zeros(n_tiles + 1, dtype=torch.int32, device=dev)
    tile_ptr[1:] = torch.cumsum(per_tile, 0).to(torch.int32)
    # the dW kernels walk 64-row units, relation-major: units of a chunk that hold no row tile are left out
    upc = CHUNK // UNIT
    order_key = chunk_rel.to(torch.int64) * max(n_tiles, 1) + chunk_tile.to(torch.int64)
    chunk_order = torch.sort(order_key, stable=True)[1]
    units = (chunk_order[:, None] * upc + torch.arange(upc, device=dev)[None, :]).reshape(-1)
    used = (chunk_cnt.to(torch.int64)[chunk_order][:, None] > UNIT * torch.arange(upc, device=dev)[None, :]).reshape(-1)
    rel_order = interleave_walk(units[used], chunk_rel.to(torch.int64)[chunk_order].repeat_interleave(upc)[used],
                                r1).to(torch.int32)
    return TilePlan(n_nodes=n_nodes, node_begin=node_begin, node_end=node_end,
                    num_relations=num_relations, tile=tile, chunk=CHUNK, n_tiles=n_tiles, n_chunks=n_chunks,
                    n_edges=n_edges, tile_ptr=tile_ptr, chunk_rel=chunk_rel, chunk_cnt=chunk_cnt,
                    chunk_tile=chunk_tile, rel_order=rel_order, slot_src=slot_src, slot_w=slot_w,
                    slot_dstl=slot_dstl, slot_row=slot_row, slot_acc=slot_acc, chunk_flags=chunk_flags,
                    layout=1 if split else 0, chunk_rows=per * g16)


def compact_runs(plan: TilePlan) -> TilePlan:
    """Twin of ``compact_runs_kernel`` (csrc/rgcn_plan.hip): a layout-0 plan with 128-slot chunks -> layout 3.  The rows of a
    (destination, relation) run all go through the same W_r into the same output row, so the producers of ``rgcn_tile3p_kernel``
    may add them in fp32 before they cut them (aggregate, then transform -- the reference's own order) and the run takes ONE slot.
    Chunk-local, and only where it is simple: chunks that are a whole (tile, relation) group, runs of at most 3 rows, at most 32
    runs of 2+ rows and 16 of 3; any other chunk keeps its layout-0 slots.  New chunk: heads (first row of every run; runs of 3
    first, then of 2, then single rows, each class in destination order) on slots 0 .. H-1; the second row of head h on row tile
    7 - h // 16, place h % 16; the third on row tile 7 - ns1 (right below the tiles of second rows), place h.  ``chunk_cnt`` = 16 ceil(H / 16); ``chunk_flags`` bits 20-23 = the chunk's
    row tiles (every chunk, compacted or not), bits 16-17 = row tiles with second rows, bit 18 = a row tile with third rows, bit 19 = some run's rows differ in weight (the producers then
    scale a shadow row by its weight / its head's weight -- the float32 in the shadow's ``slot_acc`` -- before they add it).  Every
    slot keeps its own weight and its run's row (a walk over all slots with a weight still sums the layer: tests/plan_emulator.py).
    Plain loops over the chunks: small plans (tests) only."""
    if plan.chunk != 128 or plan.layout != 0:
        raise ValueError("compact_runs wants a layout-0 plan with 128-slot chunks")
    if plan.n_chunks == 0:
        plan.layout = 3
        return plan
    dev = plan.slot_src.device
    tile, n_nodes, n_own = plan.tile, plan.n_nodes, plan.n_owned
    src = plan.slot_src.cpu().numpy().copy()
    wbits = plan.slot_w.cpu().numpy().copy().view("uint32")
    row = plan.slot_row.cpu().numpy().copy()
    acc = plan.slot_acc.cpu().numpy().copy()
    cnt = plan.chunk_cnt.cpu().numpy().copy()
    flags = plan.chunk_flags.cpu().numpy().copy()
    crel = plan.chunk_rel.cpu().numpy()
    ctile = plan.chunk_tile.cpu().numpy()
    dstl = None if plan.slot_dstl is None else plan.slot_dstl.cpu().numpy().copy()
    pad_acc = (15 << 24) | tile

    def ratio(ws_bits, wh_bits):          # float32 bits of (shadow weight / head weight), IEEE division as on the device
        import numpy as np
        q = np.array([ws_bits], dtype=np.uint32).view(np.float32) / np.array([wh_bits], dtype=np.uint32).view(np.float32)
        return int(q.astype(np.float32).view(np.int32)[0])

    for c in range(plan.n_chunks):
        nt = int(cnt[c]) // 16
        flags[c] |= nt << 20           # every chunk: its row tiles beside its flags (one scalar word for the producers)
        if c > 0 and crel[c - 1] == crel[c] and ctile[c - 1] == ctile[c]:
            continue
        if c + 1 < plan.n_chunks and crel[c + 1] == crel[c] and ctile[c + 1] == ctile[c]:
            continue
        if nt == 0:
            continue
        base = c * 128
        tbase = int(ctile[c]) * tile
        rows = []
        for j in range(nt * 16):
            sl = base + (j % nt) * 16 + j // nt
            if int(src[sl]) == n_nodes:
                break
            rows.append((int(src[sl]), int(row[sl]) - tbase, int(wbits[sl])))
        runs, j, ok, uneq = [], 0, True, 0
        while j < len(rows):
            ln = 1
            while j + ln < len(rows) and rows[j + ln][1] == rows[j][1]:
                if rows[j + ln][2] != rows[j][2]:
                    uneq = 1
                ln += 1
            if ln > 3:
                ok = False
            runs.append((j, ln))
            j += ln
        if not ok:
            continue
        n1 = sum(1 for _, ln in runs if ln == 1)
        n2 = sum(1 for _, ln in runs if ln == 2)
        n3 = sum(1 for _, ln in runs if ln == 3)
        if n2 + n3 == 0 or n3 > 16 or n2 + n3 > 32:
            continue
        heads = n1 + n2 + n3
        nh, ns1, ns2 = (heads + 15) // 16, (n2 + n3 + 15) // 16, 1 if n3 else 0
        if nh >= nt or nh + ns1 + ns2 > 8:
            continue
        src[base:base + 128] = n_nodes
        wbits[base:base + 128] = 0
        row[base:base + 128] = n_own
        acc[base:base + 128] = pad_acc
        if dstl is not None:
            dstl[base:base + 128] = tile
        nxt = {3: 0, 2: n3, 1: n3 + n2}
        for j0, ln in runs:
            h = nxt[ln]
            nxt[ln] += 1
            s0, d0, w0 = rows[j0]
            src[base + h], wbits[base + h], row[base + h] = s0, w0, tbase + d0
            acc[base + h] = ((h & 15) << 24) | d0
            if dstl is not None:
                dstl[base + h] = d0
            if ln >= 2:
                s1 = base + (7 - h // 16) * 16 + (h & 15)
                src[s1], wbits[s1], row[s1] = rows[j0 + 1][0], rows[j0 + 1][2], tbase + d0
                acc[s1] = ratio(rows[j0 + 1][2], w0)
            if ln == 3:
                s2 = base + (7 - ns1) * 16 + h
                src[s2], wbits[s2], row[s2] = rows[j0 + 2][0], rows[j0 + 2][2], tbase + d0
                acc[s2] = ratio(rows[j0 + 2][2], w0)
        cnt[c] = nh * 16
        flags[c] = (ns1 << 16) | (ns2 << 18) | (uneq << 19) | (nh << 20)
    plan.slot_src = torch.from_numpy(src).to(dev)
    plan.slot_w = torch.from_numpy(wbits.view("float32")).to(dev)
    plan.slot_row = torch.from_numpy(row).to(dev)
    plan.slot_acc = torch.from_numpy(acc).to(dev)
    plan.chunk_cnt = torch.from_numpy(cnt).to(dev)
    plan.chunk_flags = torch.from_numpy(flags).to(dev)
    if dstl is not None:
        plan.slot_dstl = torch.from_numpy(dstl).to(dev)
    plan.layout = 3
    return plan


def dw_pairs(plan: TilePlan) -> TilePlan:
    """Twin of ``dw_pairs_kernel`` (csrc/rgcn_plan.hip): a layout-0 plan with 64-slot chunks -> layout 5, the plan the tile-major
    weight-gradient kernel walks.  d_W_r is a sum over slots of (w x[src])^T g[row]: two rows with ONE (row, weight) -- a pair --
    take ONE slot if x[src] + x[src2] is formed first.  Group-local: (tile, relation) groups of at most two chunks; runs of equal
    (row, weight) are cut into pairs; heads = pairs + single rows are dealt back densely, 64 per unit; the head of a pair needs one
    of the unit's pair places (slots 0..3, and 32..35 once the unit holds more than 32 heads); pairs beyond the group's places stay
    two single rows.  ``slot_src2[unit * 8 + k]``: the second row of pair place k.  Chunks the group no longer needs get
    ``chunk_cnt`` 0 and leave ``rel_order``.  Plain loops over the groups: small plans (tests) only."""
    if plan.chunk != 64 or plan.layout != 0:
        raise ValueError("dw_pairs wants a layout-0 plan with 64-slot chunks")
    dev = plan.slot_src.device
    n_nodes, n_own = plan.n_nodes, plan.n_owned
    src = plan.slot_src.cpu().numpy().copy()
    wbits = plan.slot_w.cpu().numpy().copy().view("uint32")
    row = plan.slot_row.cpu().numpy().copy()
    cnt = plan.chunk_cnt.cpu().numpy().copy()
    crel = plan.chunk_rel.cpu().numpy()
    ctile = plan.chunk_tile.cpu().numpy()
    nc = plan.n_chunks
    src2 = np.full(max(nc * 8, 1), n_nodes, dtype=np.int32)
    c = 0
    while c < nc:
        m = 1
        while c + m < nc and crel[c + m] == crel[c] and ctile[c + m] == ctile[c]:
            m += 1
        c0, c = c, c + m
        if m > 2:
            continue
        nt = sum(int(cnt[c0 + i]) // 16 for i in range(m))
        if nt == 0:
            continue
        base = c0 * 64
        rows = []
        for j in range(nt * 16):
            tt = j % nt
            sl = base + (tt // 4) * 64 + (tt % 4) * 16 + j // nt
            if int(src[sl]) == n_nodes:
                break
            rows.append((int(src[sl]), int(row[sl]), int(wbits[sl])))
        n = len(rows)
        runs, j = [], 0
        while j < n:
            ln = 1
            while j + ln < n and rows[j + ln][1:] == rows[j][1:]:
                ln += 1
            runs.append((j, ln))
            j += ln
        pairs = sum(ln // 2 for _, ln in runs)
        if pairs == 0:
            continue
        heads = n - pairs

        def unit_heads(u):
            return min(64, heads - 64 * u)

        def places(u):
            nu = unit_heads(u)
            return min(4, nu) + (min(4, nu - 32) if nu > 32 else 0)

        while pairs > sum(places(u) for u in range((heads + 63) // 64)):
            pairs -= 1
            heads += 1
        units = (heads + 63) // 64
        pairs_in, left = [0, 0], pairs
        for u in range(units):
            pairs_in[u] = min(left, places(u))
            left -= pairs_in[u]
        src[base:base + m * 64] = n_nodes
        wbits[base:base + m * 64] = 0
        row[base:base + m * 64] = n_own

        def is_pair_place(u, sl):
            if sl < 4:
                return sl < pairs_in[u]
            if 32 <= sl < 36:
                return sl - 28 < pairs_in[u]
            return False

        pair_left, next_pair, next_single, single_u = pairs, [0, 0], [0, 0], 0
        for j0, ln in runs:
            k = 0
            while k < ln:
                as_pair = k + 1 < ln and pair_left > 0
                if as_pair:
                    u = 0 if next_pair[0] < pairs_in[0] else 1
                    kk = next_pair[u]
                    next_pair[u] += 1
                    sl = kk if kk < 4 else 28 + kk
                    pair_left -= 1
                else:
                    u = single_u
                    while True:
                        while next_single[u] < unit_heads(u) and is_pair_place(u, next_single[u]):
                            next_single[u] += 1
                        if next_single[u] < unit_heads(u):
                            break
                        u += 1
                        single_u = u
                    sl = next_single[u]
                    next_single[u] += 1
                g = base + u * 64 + sl
                src[g], row[g], wbits[g] = rows[j0 + k]
                if as_pair:
                    src2[(c0 + u) * 8 + (sl if sl < 4 else sl - 28)] = rows[j0 + k + 1][0]
                    k += 2
                else:
                    k += 1
        for i in range(m):
            cnt[c0 + i] = (unit_heads(i) + 15) // 16 * 16 if i < units else 0
    plan.slot_src = torch.from_numpy(src).to(dev)
    plan.slot_w = torch.from_numpy(wbits.view("float32")).to(dev)
    plan.slot_row = torch.from_numpy(row).to(dev)
    plan.chunk_cnt = torch.from_numpy(cnt).to(dev)
    plan.slot_src2 = torch.from_numpy(src2).to(dev)
    if plan.slot_dstl is not None:
        plan.slot_dstl = None          # (host-side checks of the forward plans only)
    # the unit list: non-empty 64-slot units, relation-major then tile -- as build_plan emits it, over the new counts
    order_key = plan.chunk_rel.to(torch.int64) * max(plan.n_tiles, 1) + plan.chunk_tile.to(torch.int64)
    chunk_order = torch.sort(order_key, stable=True)[1]
    used = plan.chunk_cnt.to(torch.int64)[chunk_order] > 0
    plan.rel_order = chunk_order[used].to(torch.int32)
    plan.layout = 5
    return plan


def team_placement(dstl: Tensor, gcnt: Tensor, grp_of_edge: Tensor, rank: Tensor, chunk_base: Tensor,
                   grp_of_chunk: Tensor, idx_in_grp: Tensor):
    """Layout 1 (128-slot chunks): chunk c of a group takes the group's sorted rows [128 c, 128 c + 128) -- n_c of them, on
    nt = ceil(n_c / 16) row tiles.  The rows are cut at a RUN BOUNDARY s (the destination changes between rows s - 1 and s)
    into part A = rows [0, s) on the chunk's first nA = ceil(nt / 2) row tiles and part B = rows [s, n_c) on the other
    nB = nt - nA (row j of a part -> the part's tile j mod n_part, place j div n_part).  So the two parts of a chunk hold
    DISJOINT destination sets, and two teams of consumer waves of the forward / dX kernel (csrc/rgcn_tile3p.hip) accumulate
    them at the same time without ever touching the same accumulator row.  s lies in [max(1, n_c - 16 nB), min(n_c - 1,
    16 nA)] -- both parts fit their tiles and need all of them, so a chunk takes exactly the nt row tiles of layout 0 --
    and is the boundary closest to the middle of that window (the lower one on a tie); a chunk without one (a single
    destination's run covers the window) is cut at the middle and flagged (chunk_flags bit 8: the parts share a destination,
    one team takes the whole chunk).  nt = 1: everything in part A.
    Returns (slot of every row, chunk_cnt, straddle flag per chunk)."""
    C, g16 = 128, ROWS_PER_MFMA_TILE
    dev = dstl.device
    n_chunks = int(grp_of_chunk.shape[0])
    n_c = torch.clamp(gcnt[grp_of_chunk] - idx_in_grp * C, max=C)
    nt = (n_c + (g16 - 1)) // g16
    n_a = (nt + 1) // 2
    n_b = nt - n_a
    two = nt > 1
    lo_c = torch.clamp(n_c - g16 * n_b, min=1)
    hi_c = torch.minimum(n_c - 1, g16 * n_a)
    mid_c = (lo_c + hi_c + 1) // 2
    cidx = rank // C
    jc = rank - cidx * C
    chunk_of_row = chunk_base[grp_of_edge] + cidx
    prev_dstl = torch.cat([dstl[:1], dstl[:-1]])
    boundary = (jc >= 1) & (dstl != prev_dstl)             # jc >= 1: row jc - 1 belongs to the same chunk range
    cand = boundary & two[chunk_of_row] & (jc >= lo_c[chunk_of_row]) & (jc <= hi_c[chunk_of_row])
    dist = jc - mid_c[chunk_of_row]
    score = 2 * dist.abs() + (dist > 0).to(torch.int64)    # mid, mid - 1, mid + 1, mid - 2, ...
    big_score = 1 << 20
    best = torch.full((n_chunks,), big_score, dtype=torch.int64, device=dev)
    best = best.scatter_reduce(0, chunk_of_row[cand], score[cand], reduce="amin", include_self=True)
    has = best < big_score
    k = best // 2
    s_found = torch.where(best % 2 == 1, mid_c + k, mid_c - k)
    s_c = torch.where(two, torch.where(has, s_found, mid_c), n_c)
    straddle = two & ~has
    chunk_cnt = (nt * g16).to(torch.int32)
    s_r = s_c[chunk_of_row]
    na_r, nb_r = n_a[chunk_of_row], torch.clamp(n_b[chunk_of_row], min=1)
    j1 = torch.clamp(jc - s_r, min=0)
    in_chunk = torch.where(jc < s_r, (jc % na_r) * g16 + jc // na_r, (na_r + j1 % nb_r) * g16 + j1 // nb_r)
    return chunk_of_row * C + in_chunk, chunk_cnt, straddle


import os as _os
_WALK_MODE = _os.environ.get("RGCN_WALK", "sorted")   # experiment knob (read once at import): sorted | rr | phase
DW_WALKERS = 2048   # waves that walk rel_order side by side in the largest dW launch (512 workgroups x 4 waves)


def interleave_walk(units: Tensor, unit_rel: Tensor, r1: int, walkers: int = DW_WALKERS, mode: Optional[str] = None) -> Tensor:
    """Order of the weight-gradient walk.  ``units`` arrive sorted by (relation, tile); a dW launch hands every wave
    one CONTIGUOUS range of the walk (few relation changes = few accumulator flushes).

    ``sorted`` (the product default) keeps that order.  ``rr`` and ``phase`` are the two interleaves tried in round 2
    to let L2 / the Infinity Cache serve the gathered upstream-gradient rows (E * 4 * out bytes per launch although
    only N rows exist: 55 GB moved for a 29 GB job at the headline config): the units of relation r are dealt over the
    ~``walkers`` concurrent waves so that all of them pass the same tiles at the same point of their walk (``rr``:
    round-robin over J_r = round(U_r * walkers / n) pieces; ``phase``: exactly aligned with the waves' ranges).
    Measured (tools/debug/dw_walk_experiment.py, 10M / 100M / 32, 64 -> 64): sorted 9.44 ms, rr 10.00, phase 9.51, and
    the same with nt loads on the x rows (9.33 / 9.92 / 9.31) -- no gain: ~5.5 TB/s of gathers push 256 MiB through the
    memory-side cache every ~46 us, i.e. a row survives about four walk steps of the 2,048 waves, while free-running
    waves drift apart by far more than that.  Relations stay contiguous and ascending in every mode
    (what the slab reduction relies on); any order inside a relation gives the same sums up to fp32 re-association."""
    n = int(units.shape[0])
    if n == 0:
        return units
    dev = units.device
    mode = mode or _WALK_MODE
    if mode == "sorted":
        return units
    cnt = torch.bincount(unit_rel, minlength=r1)                       # U_r
    start = torch.cumsum(cnt, 0) - cnt                                  # A_r
    if mode == "phase":
        # walker w owns positions [w n / walkers, (w + 1) n / walkers): the PHASE of position p is how far into its
        # walker's range it lies, frac(p * walkers / n) = ((p * walkers) mod n) / n.  Inside a relation the k-th
        # position in (phase, p) order takes the relation's k-th unit in tile order, so every walker -- also one
        # that straddles two relations -- passes tile fraction ~phase at the same point of its own walk.
        p = torch.arange(n, device=dev)
        phase = (p * walkers) % n
        o1 = torch.sort(phase, stable=True)[1]
        o2 = torch.sort(unit_rel[o1], stable=True)[1]
        pos = o1[o2]                                                    # positions, relation-major, by (phase, p)
        out = torch.empty_like(units)
        out[pos] = units
        return out
    pieces = torch.clamp((cnt * walkers + n // 2) // n, min=1)          # J_r
    q = torch.arange(n, device=dev) - start[unit_rel]
    jr, ur = pieces[unit_rel], cnt[unit_rel]
    j = q % jr
    base, rem = ur // jr, ur % jr                                       # the first `rem` pieces hold base + 1 units
    pos = start[unit_rel] + j * base + torch.minimum(j, rem) + q // jr
    out = torch.empty_like(units)
    out[pos] = units
    return out


ROWS_PER_MFMA_TILE = 16
LDS_BYTES = 160 * 1024
ACC_PAD = 4    # floats of padding per accumulator row in the tile kernel's LDS tile (bank spread)


def padded_width(w: int) -> int:
    """the kernels pad feature widths to 16 / 32 / 64 / 128 (csrc/rgcn_common.h padded_width)"""
    return 16 if w <= 16 else 32 if w <= 32 else 64 if w <= 64 else 128


# measured ms per launch at the headline config over the model's cycle count, per kernel: what makes the two kernels' costs comparable
# (exact fp32: 10.15 ms where the model says 8.6; bf16 x 3: 8.4 ms where its model -- 1,800 cycles per chunk, 420 per row tile,
# DESIGN.md 8.0f -- says 9.2)
_KERNEL_MODEL = {"fp32": (800.0, 650.0, 1.18), "bf16x3": (1800.0, 420.0, 0.91)}
DW_PLAN_LAYOUT = int(_os_environ_get("RGCN_DW_PLAN_LAYOUT", "5"))      # 5: pairs on one slot (round 4); 0: one slot per row
P3_MAX_TILE = 224          # rgcn_tile3p_kernel: two 48 KiB ring slots (any 128-slot chunk) + the fp32 accumulator in 160 KiB
P3_MAX_TILE_112 = 272      # ... two 42 KiB slots (chunks of at most 112 rows: CHUNK_112)


def choose_layout(n_nodes: int, n_edges: int, num_relations: int, in_channels: int, out_channels: int, kernel: str = "fp32",
                  with_cost: bool = False):
    """cached front of ``_choose_layout`` (a layer asks on every forward call; the model is a few hundred erfc evaluations)"""
    return _choose_layout(int(n_nodes), int(n_edges), int(num_relations), int(in_channels), int(out_channels), kernel, bool(with_cost))


@functools.lru_cache(maxsize=256)
def _choose_layout(n_nodes: int, n_edges: int, num_relations: int, in_channels: int, out_channels: int, kernel: str = "fp32",
                   with_cost: bool = False):
    """(tile, chunk) for a layer: output nodes per tile and edge slots per chunk.  ``kernel="bf16x3"``: the layout for
    rgcn_tile3p_kernel (64 x 64 layers: 128-slot chunks, tiles up to 224, its own per-chunk / per-row-tile cycles);
    ``with_cost``: also the modelled time of one launch, comparable between the two kernels (conv.RGCNConv.layout picks with it).

    Cost model of the forward / dX kernel, calibrated on the 10M-node / 100M-edge graph (tools/debug/stamps.py): a
    chunk costs ~800 cycles whatever it holds (barrier, metadata, pipeline fill and drain) and every 16-row MFMA tile
    ~650 (512 of them MFMA issue).  A (tile, relation) group of g = tile * E / (N R') edges (roughly normal, sd
    sqrt(g)) takes E[ceil(g / chunk)] chunks and ~g / 16 + 1/2 row tiles, so larger tiles and 128-slot chunks
    amortise the fixed part -- within the LDS: (tile + 1) * (pad(width) + 4) * 4 B of accumulator plus at least two
    ring slots of chunk * (pad(other width) + 2) * 4 B (three cost nothing extra; with two the producers run only one
    chunk ahead: +3 %).  128-slot chunks are built for widths <= 64.  The launch runs in rounds of 256 workgroups that
    walk up to 16 tiles each, so what is minimised is rounds x cycles per tile."""
    import math
    kp, np_ = padded_width(in_channels), padded_width(out_channels)
    density = n_edges / max(1.0, float(n_nodes) * max(1, num_relations))

    def lds(t, chunk, ring):
        fwd = (t + 1) * (np_ + ACC_PAD) * 4 + ring * chunk * (kp + 2) * 4
        bwd = (t + 1) * (kp + ACC_PAD) * 4 + ring * chunk * (np_ + 2) * 4
        return max(fwd, bwd)

    def expect_ceil(g, unit):
        if g <= 0:
            return 0.0
        sd = math.sqrt(g)
        return sum(0.5 * math.erfc((unit * k - g) / (sd * math.sqrt(2.0))) for k in range(0, int(g / unit) + 6))

    c_chunk, c_tile, scale = _KERNEL_MODEL[kernel]
    p3 = kernel == "bf16x3"

    def cost(t, chunk):
        g = density * t
        ring_penalty = 1.0 if (p3 or lds(t, chunk, 3) <= LDS_BYTES) else 1.03
        per_rel = expect_ceil(g, chunk) * c_chunk + (g / 16.0 + 0.5) * c_tile if g > 0 else 0.0
        root = math.ceil(t / chunk) * c_chunk + (t / 16.0) * c_tile
        return scale * ring_penalty * (max(1, num_relations) * per_rel + root) / t      # cycles per output node

    def launch_rounds(n_tiles):
        # tile times one launch takes: a workgroup (one per CU, 256 CUs) walks up to 16 tiles, start-up ~4 % of a tile;
        # the library picks the count the same way (csrc/rgcn_kernels_shared.h tiles_per_workgroup)
        return min(math.ceil(math.ceil(n_tiles / k) / 256) * (k + 0.04) for k in range(1, 17))

    if p3:      # 128-slot chunks up to tile 224, 112-row chunks (seven row tiles; the kernel's smaller ring slots) up to 272
        cands = ([(t, 128) for t in range(64, P3_MAX_TILE + 1, 16)] + [(t, CHUNK_112) for t in range(P3_MAX_TILE + 16, P3_MAX_TILE_112 + 1, 16)]
                 if max(kp, np_) == 64 and min(kp, np_) == 64 else [])
    else:
        cands = [(t, c) for c in CHUNKS for t in range(64, 513, 16)
                 if lds(t, c, 2) <= LDS_BYTES and (c == CHUNK or max(kp, np_) <= 64)]
    if not cands:
        return (64, CHUNK, float("inf")) if with_cost else (64, CHUNK)
    # time of a launch ~ rounds x (cycles per tile): on large graphs this is the cost per node, on small ones the round
    # count decides (100k nodes: 285 tiles of 352 are two rounds with the second one a ninth full)
    key = lambda tc: (round(launch_rounds(math.ceil(n_nodes / tc[0])) * cost(*tc) * tc[0] / 1e3, 1), -tc[0])
    best = min(cands, key=key)
    return (best[0], best[1], key(best)[0]) if with_cost else best


def run_metadata(slot_dstl: Tensor, tile: int):
    """Per slot, for the forward kernel's run-sum (csrc/rgcn_tile_fp32_kernel.h / rgcn_tile3p.hip: the Y-orientation run-sum product), precomputed here so
    the kernel spends no vector instructions on it: inside every 16-slot MFMA row tile, slots with equal
    destination are adjacent (tiles are sorted by destination) and form a RUN; the run's sum is written by
    its LAST slot only.  Returns (slot_acc, tile_dup).  slot_acc = (position 0..15 inside the row tile of the
    slot that ends this slot's run) << 24 | (accumulator row the slot writes).  The accumulator row is the
    slot's destination if it ends a run, else the dummy row ``tile`` (padding slots carry destination ``tile``
    already); it sits in the low 24 bits so that the kernel forms the LDS address with one 24-bit
    multiply-add, which ignores the top byte by itself.
    tile_dup[t]: row tile t contains a run longer than one slot (needs the run-sum product)."""
    g = ROWS_PER_MFMA_TILE
    d = slot_dstl.view(-1, g).to(torch.int64)
    nxt = torch.cat([d[:, 1:], torch.full_like(d[:, :1], -1)], dim=1)
    is_end = d != nxt                                    # last column always ends (next = -1)
    pos = torch.arange(g, device=d.device).expand_as(d)
    endpos = torch.where(is_end, pos, torch.full_like(pos, g))
    runend = torch.flip(torch.cummin(torch.flip(endpos, [1]), dim=1).values, [1])
    acc = torch.where(is_end, d, torch.full_like(d, tile))
    tile_dup = ((~is_end) & (d != tile)).any(dim=1)     # runs of padding slots do not count
    return ((runend << 24) | acc).to(torch.int32).reshape(-1), tile_dup


@dataclass
class GraphPlans:
    """Forward plan (edges grouped by destination) + transposed plan (grouped by source).  A direction that runs the
    edge-parallel path (eplan.choose_path) has an eplan.EdgePlan instead of a tile-major plan."""
    fwd: Optional[TilePlan]
    bwd: Optional[TilePlan]
    num_edges: int
    dw: Optional[TilePlan] = None        # forward-direction plan in the geometry of the tile-major dW kernel (dw_walk_table)
    dw_walk: Optional[Tensor] = None     # int32 [R'][walkers + 1]
    ep_fwd: Optional[object] = None      # eplan.EdgePlan: forward on rgcn_ep_*, weight gradients on its dense units
    ep_bwd: Optional[object] = None      # eplan.EdgePlan of the transposed graph: dX on rgcn_ep_*

    @property
    def fwd_walk(self) -> TilePlan:
        """the plan whose 64-slot units the relation-major weight-gradient kernels walk"""
        return self.fwd if self.fwd is not None else self.ep_fwd.as_tile_plan()


def build_graph_plans_torch(edge_index: Tensor, edge_type: Tensor, n_nodes: int, num_relations: int,
                            tile: int, aggr: str = "mean",
                            fwd_range: Optional[Tuple[int, int]] = None,
                            bwd_range: Optional[Tuple[int, int]] = None, chunk: int = CHUNK,
                            split: bool = False, paths: Tuple[str, str] = ("ring", "ring")) -> GraphPlans:
    """The plans as torch tensor ops (any device): the TEST ORACLE of the device-side builder and what the CPU-only
    tests walk with tests/plan_emulator.py.  paths: 'ring' (tile-major plan) or 'ep' (eplan.EdgePlan) per direction."""
    from .eplan import build_edge_plan
    src, dst = edge_index[0], edge_index[1]
    w = edge_weights(src, dst, edge_type, num_relations, aggr)
    fb, fe = fwd_range if fwd_range is not None else (0, n_nodes)
    bb, be = bwd_range if bwd_range is not None else (0, n_nodes)
    gp = GraphPlans(fwd=None, bwd=None, num_edges=int(edge_type.shape[0]))
    from .eplan import HEAVY
    if paths[0] == "ep":
        gp.ep_fwd = build_edge_plan(src, dst, edge_type, w, n_nodes, num_relations, fb, fe, heavy=HEAVY)
    else:
        gp.fwd = build_plan(src, dst, edge_type, w, n_nodes, num_relations, tile, fb, fe, chunk, split)
    if paths[1] == "ep":
        gp.ep_bwd = build_edge_plan(dst, src, edge_type, w, n_nodes, num_relations, bb, be, heavy=HEAVY)
    else:
        gp.bwd = build_plan(dst, src, edge_type, w, n_nodes, num_relations, tile, bb, be, chunk, split)
    return gp


def empty_plan(n_nodes: int, node_begin: int, num_relations: int, tile: int, chunk: int, device, layout: int = 0) -> TilePlan:
    """the plan of an empty node range (dist.py: a block wholly past the last node, or squeezed out by a hub's block)"""
    z = lambda dt=torch.int32: torch.zeros(0, dtype=dt, device=device)
    rows = chunk
    chunk = 128 if chunk == CHUNK_112 else chunk
    return TilePlan(n_nodes=n_nodes, node_begin=node_begin, node_end=node_begin, num_relations=num_relations, tile=tile,
                    chunk_rows=rows, chunk=chunk, n_tiles=0, n_chunks=0, n_edges=0, tile_ptr=torch.zeros(1, dtype=torch.int32, device=device),
                    chunk_rel=z(), chunk_cnt=z(), chunk_tile=z(), chunk_flags=z(), rel_order=z(), slot_src=z(),
                    slot_w=z(torch.float32), slot_dstl=None, slot_row=z(), slot_acc=z(), layout=layout)


def _device_plan(graph, w, transposed: bool, n_nodes: int, num_relations: int, tile: int, chunk: int,
                 node_begin: int, node_end: int, ws, split: bool = False, aligned: bool = True) -> TilePlan:
    from . import _lib
    if node_end <= node_begin:           # nothing to lay out
        return empty_plan(n_nodes, node_begin, num_relations, tile, chunk, ws.device, int(split))
    if aligned and node_begin % tile != 0:       # forward / dX plans of a rank: its tiles must be the single-rank tiles
        raise ValueError("node_begin must be a multiple of the tile size")
    ps, a, n_edges = _lib.plan_build(graph, w, transposed, node_begin, node_end, tile, chunk, ws, split)
    if int(split) == 5:          # the pairs left fewer units than _begin sized rel_order for
        a["rel_order"] = a["rel_order"][:int(ps.n_units)]
    plan = TilePlan(n_nodes=n_nodes, node_begin=node_begin, node_end=node_end, num_relations=num_relations, tile=tile,
                    chunk=int(ps.chunk), n_tiles=int(ps.n_tiles), n_chunks=int(ps.n_chunks), n_edges=n_edges, slot_dstl=None,
                    layout=int(split), chunk_rows=int(ps.chunk_rows), **a)
    plan._cstruct = ps
    return plan


def build_graph_plans_device(edge_index: Tensor, edge_type: Tensor, n_nodes: int, num_relations: int, tile: int,
                             aggr: str = "mean", fwd_range: Optional[Tuple[int, int]] = None,
                             bwd_range: Optional[Tuple[int, int]] = None, chunk: int = CHUNK,
                             ranges=None, split: bool = False, dw_tiles: bool = False,
                             paths: Tuple[str, str] = ("ring", "ring"), rank_dw_range: Optional[Tuple[int, int]] = None,
                             extras: Optional[dict] = None, hub_split: Optional[Tuple[int, int]] = None):
    """The plans built by the HIP library itself (csrc/rgcn_plan.hip through rgcn_edge_weights / rgcn_plan_build_*):
    what every GPU forward uses.  ``ranges``: a list of (begin, end) owned ranges -> a list of GraphPlans that share one
    edge-weight pass and one workspace (dist.py: one pair of plans per owned block).  ``rank_dw_range`` (dist.py, full exchange):
    ONE tile-major weight-gradient plan over that contiguous node range besides the pieces' forward / transposed plans, returned
    in ``extras["dw_rank"] = (plan, walk table)`` -- both operands of d_weight are replicated, so its cut need not be the
    forward's and one launch per rank replaces one per piece.  ``hub_split = (world, rank)`` (dist.py): the heavy segments of an
    edge-parallel direction are the WHOLE graph's, their rows dealt over the ranks (eplan.SharedHeavy, returned in
    ``extras["shared_fwd"]`` / ``["shared_bwd"]``); the pieces' plans hold the light rows and the pseudo rows of their own segments."""
    from . import _lib
    if chunk not in CHUNKS and chunk != CHUNK_112:
        raise ValueError(f"chunk must be one of {CHUNKS} (or {CHUNK_112})")
    graph, keep = _lib.graph_struct(edge_index, edge_type, n_nodes, num_relations)
    e = int(edge_type.shape[0])
    rs = ranges if ranges is not None else [(fwd_range or (0, n_nodes), bwd_range or (0, n_nodes))]
    own_max = max([1] + [max(f[1] - f[0], b[1] - b[0]) for f, b in rs] + ([rank_dw_range[1] - rank_dw_range[0]] if rank_dw_range else []))
    ws_tile = min(tile, _lib.dw_tiles_geometry()[0]) if dw_tiles else tile      # the smallest tile sizes the group arrays
    ws = _lib.plan_workspace(e, own_max, num_relations, ws_tile, edge_type.device)
    try:
        w = _lib.edge_weights(graph, aggr, ws)
    except _lib.RgcnLibraryError as err:
        if "out of range" in str(err):
            raise ValueError("edge_index / edge_type out of range [0, num_nodes) / [0, num_relations)") from err
        raise
    from .eplan import HEAVY
    heavy = HEAVY
    shared, light, keep_light = [None, None], [None, None], []
    if hub_split is not None and e > 0:
        from .eplan import build_shared_heavy
        for d in (0, 1):
            if paths[d] != "ep":
                continue
            g_, s_ = (edge_index[1], edge_index[0]) if d else (edge_index[0], edge_index[1])
            sh = build_shared_heavy(g_, s_, edge_type, w, n_nodes, heavy, hub_split[0], hub_split[1])
            if sh is None:
                continue
            lm = ~sh.edge_mask
            ei_l, et_l, w_l = edge_index[:, lm].contiguous(), edge_type[lm].contiguous(), w[lm].contiguous()
            g_l, k_l = _lib.graph_struct(ei_l, et_l, n_nodes, num_relations)
            keep_light.append((k_l, ei_l, et_l))
            shared[d], light[d] = sh, (g_l, w_l)
        if extras is not None:
            extras["shared_fwd"], extras["shared_bwd"] = shared
    out = []
    for (fb, fe), (bb, be) in rs:
        gp = GraphPlans(fwd=None, bwd=None, num_edges=e)
        if fe <= fb or be <= bb:       # an empty block of a rank (dist.py)
            out.append(GraphPlans(fwd=empty_plan(n_nodes, fb, num_relations, tile, chunk, ws.device, int(split)),
                                  bwd=empty_plan(n_nodes, bb, num_relations, tile, chunk, ws.device, int(split)), num_edges=0))
            continue
        if paths[0] == "ep":       # edge-parallel direction: relation-major units + destination-major segments (eplan.py)
            from .eplan import build_edge_plan_device
            gp.ep_fwd = build_edge_plan_device(graph, w, False, n_nodes, num_relations, ws, fb, fe, heavy=heavy,
                                               edge_index=edge_index, edge_type=edge_type, shared=shared[0], light_graph=light[0])
        else:
            gp.fwd = _device_plan(graph, w, False, n_nodes, num_relations, tile, chunk, fb, fe, ws, split)
            if ranges is not None:
                gp.num_edges = gp.fwd.n_edges
        if paths[1] == "ep":
            from .eplan import build_edge_plan_device
            gp.ep_bwd = build_edge_plan_device(graph, w, True, n_nodes, num_relations, ws, bb, be, heavy=heavy,
                                               edge_index=edge_index, edge_type=edge_type, shared=shared[1], light_graph=light[1])
        else:
            gp.bwd = _device_plan(graph, w, True, n_nodes, num_relations, tile, chunk, bb, be, ws, split)
        if dw_tiles and paths[0] != "ep" and fe > fb:
            # the tile-major weight-gradient kernel's own layout of the same edges (a rank's piece: its node range need not
            # be a multiple of THAT tile -- the sums of a partitioned d_weight differ in order from the single-rank ones anyway)
            t_dw, walkers, max_rel = _lib.dw_tiles_geometry()
            if num_relations <= max_rel:
                # (layout 5: the two rows of a (destination, relation) pair on one slot -- dw_pairs / dw_pairs_kernel)
                gp.dw = _device_plan(graph, w, False, n_nodes, num_relations, t_dw, 64, fb, fe, ws, DW_PLAN_LAYOUT, aligned=False)
                gp.dw_walk = _lib.dw_tiles_walk(_lib.plan_struct(gp.dw), edge_type.device)
        out.append(gp)
    if rank_dw_range is not None and extras is not None and paths[0] != "ep":
        t_dw, walkers, max_rel = _lib.dw_tiles_geometry()
        if num_relations <= max_rel:
            if rank_dw_range[1] > rank_dw_range[0]:
                pl = _device_plan(graph, w, False, n_nodes, num_relations, t_dw, 64, rank_dw_range[0], rank_dw_range[1], ws, DW_PLAN_LAYOUT, aligned=False)
                extras["dw_rank"] = (pl, _lib.dw_tiles_walk(_lib.plan_struct(pl), edge_type.device))
            else:
                extras["dw_rank"] = (None, None)      # an empty range (fewer tiles than ranks): this rank adds zeros
    del keep, keep_light
    return out if ranges is not None else out[0]


def dw_walk_table(plan: TilePlan, walkers: int) -> Tensor:
    """Torch form (test oracle) of rgcn_dw_tiles_walk -- walk_ptr of rgcn_bwd_dw_tiles: for relation r and walker p the first position in ``plan.rel_order`` (sorted by
    (relation, tile)) of a unit of relation r whose tile is >= p * n_tiles // walkers; column ``walkers`` = end of relation r."""
    dev = plan.rel_order.device
    units = plan.rel_order.long()
    upc = plan.chunk // UNIT
    key = plan.chunk_rel.long()[units // upc] * plan.n_tiles + plan.chunk_tile.long()[units // upc]
    r = torch.arange(plan.num_relations, device=dev)[:, None]
    t = (torch.arange(walkers + 1, device=dev) * plan.n_tiles // walkers)[None, :]
    return torch.searchsorted(key, (r * plan.n_tiles + t).reshape(-1)).view(plan.num_relations, walkers + 1).to(torch.int32).contiguous()


def build_graph_plans(edge_index: Tensor, edge_type: Tensor, n_nodes: int, num_relations: int,
                      tile: int, aggr: str = "mean",
                      fwd_range: Optional[Tuple[int, int]] = None,
                      bwd_range: Optional[Tuple[int, int]] = None, chunk: int = CHUNK,
                      split: bool = False, dw_tiles: bool = False, paths: Tuple[str, str] = ("ring", "ring")) -> GraphPlans:
    """Device tensors: the HIP plan builder behind the C ABI.  CPU tensors (tests without a GPU): the torch form."""
    if edge_type.device.type == "cuda" and _WALK_MODE == "sorted":
        return build_graph_plans_device(edge_index, edge_type, n_nodes, num_relations, tile, aggr, fwd_range, bwd_range, chunk,
                                        split=split, dw_tiles=dw_tiles, paths=paths)
    return build_graph_plans_torch(edge_index, edge_type, n_nodes, num_relations, tile, aggr, fwd_range, bwd_range, chunk, split,
                                   paths=paths)


def balanced_ranges(counts_per_tile: Tensor, world: int, tile: int, n_nodes: int):
    """Cut the tile sequence into ``world`` contiguous node ranges of about equal edge count
    (SURVEY.md 8e "ranges balanced by edge count").  Boundaries are multiples of ``tile`` so a
    rank's tiles -- and therefore every accumulation order -- equal the single-rank ones."""
    n_tiles = counts_per_tile.shape[0]
    cum = torch.cumsum(counts_per_tile.to(torch.float64) + 1e-3, 0)  # +eps: spread empty tiles too
    total = float(cum[-1]) if n_tiles else 0.0
    bounds = [0]
    for p in range(1, world):
        t = int(torch.searchsorted(cum, torch.tensor(total * p / world, dtype=torch.float64,
                                                      device=cum.device)).item())
        t = min(max(t, bounds[-1]), n_tiles)
        bounds.append(t)
    bounds.append(n_tiles)
    return [(min(b * tile, n_nodes), min(e * tile, n_nodes)) for b, e in zip(bounds[:-1], bounds[1:])]


# ------------------------------------------------------------------------------------------
# plan cache, keyed on the identity of the tensors the caller passes every forward
# ------------------------------------------------------------------------------------------
_CACHE: Dict[tuple, tuple] = {}     # insertion-ordered: least recently used first
_CACHE_MAX = 16
_CACHE_MAX_BYTES = int(float(_os.environ.get("RGCN_PLAN_CACHE_GB", "48")) * (1 << 30))   # read once at import


def _plans_nbytes(plans) -> int:
    pieces = getattr(plans, "pieces", None) or [plans]
    extra = sum(q.nbytes() for q in (getattr(plans, "needed_fwd", None), getattr(plans, "needed_bwd", None)) if q is not None)
    if getattr(plans, "dw_rank", None) is not None and plans.dw_rank[0] is not None:
        extra += plans.dw_rank[0].nbytes()
    return extra + sum(sum(q.nbytes() for q in (p.fwd, p.bwd, getattr(p, "dw", None), getattr(p, "ep_fwd", None), getattr(p, "ep_bwd", None))
                           if q is not None) for p in pieces)


def cached_graph_plans(edge_index: Tensor, edge_type: Tensor, n_nodes: int, num_relations: int,
                       tile: int, aggr: str, builder=None, extra_key=(), chunk: int = CHUNK, split: bool = False,
                       dw_tiles: bool = False, paths=("ring", "ring"), widths: Optional[Tuple[int, int]] = None) -> GraphPlans:
    """LRU over (edge tensors' identity, layout): at most ``_CACHE_MAX`` entries and ``RGCN_PLAN_CACHE_GB`` (48) GiB of
    plan arrays (4.3 GB per 100M edges), least recently used evicted first."""
    key = (edge_index.data_ptr(), edge_type.data_ptr(), tuple(edge_index.shape), edge_index._version,
           edge_type._version, str(edge_index.device), n_nodes, num_relations, tile, chunk, aggr, int(split), bool(dw_tiles),
           paths if isinstance(paths, str) else tuple(paths), widths) + tuple(extra_key)
    hit = _CACHE.pop(key, None)
    if hit is not None:
        _CACHE[key] = hit           # most recently used last
        return hit[0]
    if builder is None:
        if paths == "auto":      # per direction: tile kernels or the edge-parallel path (eplan.choose_path), decided once per graph
            from .eplan import decide_paths
            paths = decide_paths(edge_index, n_nodes, num_relations, widths[0], widths[1], tile, chunk)
        plans = build_graph_plans(edge_index, edge_type, n_nodes, num_relations, tile, aggr, chunk=chunk, split=split,
                                  dw_tiles=dw_tiles, paths=paths)
    else:
        plans = builder()
    nbytes = _plans_nbytes(plans)
    while _CACHE and (len(_CACHE) >= _CACHE_MAX or sum(v[3] for v in _CACHE.values()) + nbytes > _CACHE_MAX_BYTES):
        _CACHE.pop(next(iter(_CACHE)))
    # hold the key tensors so their storage (and data_ptr) cannot be recycled while cached
    _CACHE[key] = (plans, edge_index, edge_type, nbytes)
    return plans


def clear_plan_cache() -> None:
    _CACHE.clear()
