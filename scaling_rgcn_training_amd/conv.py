"""``RGCNConv``: constructor-, attribute- and forward-compatible with PyG 2.3.1
``torch_geometric.nn.RGCNConv`` as the reference uses it (/root/reference/model/layers.py:15-16,
21-23, 33-46), with forward and backward running as the HIP kernels of
``csrc/rgcn_tile_fp32*.hip`` / ``rgcn_tile3p.hip`` / ``rgcn_ep.hip`` / ``rgcn_dw_*.hip`` through the C ABI of ``include/rgcn_mi355x.h``.

Mutability contract (model/layers.py:33-46, model/modelTrainer.py:26-39): ``weight`` / ``root`` /
``bias`` are plain ``nn.Parameter`` attributes that callers REPLACE after construction and may
freeze; forward reads them at call time and backward skips the frozen ones.

There is no CPU path: CPU tensors (or a missing HIP library) raise.
"""
from __future__ import annotations

import math
import os
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from . import _lib
from .plan import GraphPlans, TilePlan, cached_graph_plans


def _round4(n: int) -> int:
    return (n + 3) // 4 * 4


def _rows16(t: Tensor, width: int) -> Tensor:
    """float32, contiguous rows whose stride is a multiple of 4 elements, zero padded (C ABI rule)."""
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    if width % 4 != 0:
        return torch.nn.functional.pad(t, (0, _round4(width) - width))
    return t.contiguous()


# experiment overrides of the (tile, chunk) layout: read ONCE at import, never on the forward path
_ENV_TILE = int(os.environ["RGCN_TILE"]) if "RGCN_TILE" in os.environ else None
_ENV_CHUNK = int(os.environ["RGCN_CHUNK"]) if "RGCN_CHUNK" in os.environ else None
# forward / dX on the bf16 x 3 kernel whose PRODUCER waves split the gathered rows (csrc/rgcn_tile3p.hip):
# fp32-equivalent arithmetic; layers padded to 64 x 64 on graphs dense enough for 128-slot chunks.  "1" / "0" / "auto"
_SPLIT_PRODUCERS_DEFAULT = os.environ.get("RGCN_SPLIT_PRODUCERS", "1")
# "1": plans of those layers in the TEAM placement (plan layout 1, plan.team_placement) for experiment builds of the kernel
# with two teams of consumer waves (csrc/rgcn_tile3p.hip RGCN_P3_TEAMS=2: measured no faster).  Default: layout 0
_TEAM_LAYOUT_DEFAULT = os.environ.get("RGCN_TEAM_LAYOUT", "0") == "1"
_MERGE_RUNS_DEFAULT = os.environ.get("RGCN_MERGE_RUNS", "1") == "1"
_PATH_DEFAULT = os.environ.get("RGCN_PATH", "auto")       # auto | ring | ep
SPLIT_PRODUCERS_TILE = 224       # the largest tile whose fp32 accumulator fits beside the kernel's two 48 KiB ring slots
# exact-fp32 forward / dX of 64 x 64 layers on layout-3 plans (round 4): the largest tile it takes.  Measured at the headline
# config, forward / dX launch, A/B on one box (profiles/r04g_exact_merge_timing.txt): layout 0 at the cost model's 352: 10.42 /
# 10.33 ms; layout 3 at 352 (40 % of the chunks compacted, 5.8 % fewer head row tiles) 10.12 / 10.18; at 320: 10.21 / 10.19; at
# 288 (60 %, -10 %): 10.32 / 10.41 -- smaller tiles compact more chunks but pay more of them: the cap stays at 352
EXACT_MERGE_TILE = int(os.environ.get("RGCN_EXACT_MERGE_TILE", 352))
DW_TILES_MIN_EDGES = 4_000_000

_ACT_CODES = {None: _lib.ACT_NONE, "relu": _lib.ACT_RELU, "sigmoid": _lib.ACT_SIGMOID}
# graphs from this many nodes run the root / bias gradient kernel on a side stream beside the dX launch (below it the step is
# launch-bound and a second stream only adds event traffic)
_SIDE_STREAM_MIN_ROWS = int(os.environ.get("RGCN_SIDE_STREAM_MIN_ROWS", 262144))
_side_streams = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = torch.device(device).index
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


def layout_for(in_channels: int, out_channels: int, n_nodes: int = 0, n_edges: int = 0,
               num_relations: int = 1) -> Tuple[int, int]:
    """(output nodes per tile, edge slots per chunk) for a layer (plan.choose_layout): bounded by the LDS budget of
    the wider side, tuned to the graph's density.  ``RGCN_TILE`` / ``RGCN_CHUNK`` in the environment at import time
    override (experiments)."""
    if not (1 <= in_channels <= 128 and 1 <= out_channels <= 128):
        raise ValueError(f"RGCNConv widths must be in 1..128, got {in_channels}->{out_channels}")
    from .plan import choose_layout
    tile, chunk = choose_layout(n_nodes, n_edges, num_relations, in_channels, out_channels)
    return (_ENV_TILE or tile), (_ENV_CHUNK or chunk)


def tile_for(in_channels: int, out_channels: int, n_nodes: int = 0, n_edges: int = 0, num_relations: int = 1) -> int:
    """Output nodes per tile of ``layout_for``."""
    return layout_for(in_channels, out_channels, n_nodes, n_edges, num_relations)[0]


class DistContext:
    """One process per GPU.  Output nodes are cut into ``pieces * world`` tile-aligned blocks, dealt piece-major: block
    (s, r) = rows [bounds[s * world + r], bounds[s * world + r + 1]) belongs to rank r.  A rank computes piece s straight into
    its block of the gathered buffer and the blocks of super-block s travel asynchronously while it computes piece s + 1
    (dist.py).  Two cuts:
      * uniform (``piece_rows`` rows per block, the last blocks padded past the graph's end): one in-place
        ``all_gather_into_tensor`` per piece -- the default wherever equal node blocks hold equal edge counts within 5 %;
      * balanced (``dist.balanced_bounds``: blocks of about equal EDGE count): unequal blocks, gathered by one
        broadcast per rank and piece (what an uneven all-gather is underneath).
    Two exchanges (round 4):
      * ``exchange = "full"`` (default): every rank ends up with every row of the gathered matrix -- the per-layer all-reduce
        of north_star with one contributor per row;
      * ``exchange = "needed"`` (opt-in, dist.attach(..., exchange="needed")): a rank receives only the rows its own plans
        gather (dist.NeededRows, derived from the edge list at plan time: the sources of the edges into its blocks for a
        forward output, the destinations of the edges out of them for a dX output) -- one ``all_to_all_single`` with split sizes
        per piece instead of the all-gather, packed rows scattered into place.  Owned rows and read rows are bit-identical to
        the full exchange; rows no plan of this rank reads are NOT written (they hold whatever the allocator returned), so
        the mode is for layers whose output feeds another partitioned layer over the same graph, not for a model's last layer.
    ``emulate``: no process group at all -- ONE process stands in for rank ``rank`` of a ``world``-rank job: it builds that
    rank's plans, launches that rank's kernels, packs / unpacks that rank's rows, and skips the collectives (bench.py
    --emulate-world: what a rank's share of a step costs, measured on one GPU).
    ``stats`` counts what the collectives moved (bench.py reports it per step)."""

    def __init__(self, group, rank: int, world: int, piece_rows: int, pieces: int, bounds=None, exchange: str = "full",
                 emulate: bool = False, split_hubs: bool = True, uniform: Optional[bool] = None):
        if exchange not in ("full", "needed"):
            raise ValueError("exchange must be 'full' or 'needed'")
        self.group, self.rank, self.world = group, rank, world
        self.pieces = pieces
        self.exchange, self.emulate = exchange, emulate
        # edge-parallel pieces: the heavy (node, relation) segments of the whole graph are summed by ALL ranks, an equal share of
        # their rows each, and one small all-reduce completes the sums (eplan.SharedHeavy) -- a hub no longer belongs to one rank
        self.split_hubs = split_hubs
        # uniform: the blocks of ONE piece are equal (one in-place all-gather per piece); ``bounds`` given with uniform = True:
        # pieces of different lengths (dist.piece_tiles: whole launch rounds), still equal blocks inside each
        self.uniform = (bounds is None) if uniform is None else bool(uniform)
        self.bounds_equal = bounds is None                 # every block of every piece has ``piece_rows`` rows
        self.piece_rows = piece_rows if bounds is None else None
        self.bounds = [i * piece_rows for i in range(pieces * world + 1)] if bounds is None else [int(b) for b in bounds]
        assert len(self.bounds) == pieces * world + 1
        if self.uniform:
            for s_ in range(pieces):
                sizes = {self.bounds[s_ * world + r + 1] - self.bounds[s_ * world + r] for r in range(world)}
                assert len(sizes) == 1, "uniform cut: the blocks of a piece are equal"
        self.stats = {"all_gather": 0, "all_gather_bytes": 0, "all_reduce": 0, "all_reduce_bytes": 0,
                      "wait_events": []}
        self.time_waits = False     # bench.py: HIP events around the waits on the collectives, piece by piece

    @property
    def total_rows(self) -> int:
        return self.bounds[-1]

    def block(self, s: int, r: Optional[int] = None) -> Tuple[int, int]:
        r = self.rank if r is None else r
        return self.bounds[s * self.world + r], self.bounds[s * self.world + r + 1]

    def node_range(self, s: int, n_nodes: int, r: Optional[int] = None) -> Tuple[int, int]:
        """owned node range of block (s, r), clipped to the graph; a block that lies wholly past the last node is
        the empty range at the tile-aligned end (n_nodes rounded up would not be a valid begin otherwise)"""
        b, e = self.block(s, r)
        if b >= n_nodes:
            return b, b
        return b, min(e, n_nodes)

    def src_rank(self, r: int) -> int:
        """global rank of group rank r (broadcast sources are global ranks)"""
        if self.group is None:
            return r
        return torch.distributed.get_global_rank(self.group, r)


def _gather_pieces(dctx: "DistContext", plans_list, launch, ld: int, n: int, device,
                   dtype: torch.dtype = torch.float32, needed=None, defer: bool = False):
    """Run ``launch(plan, out_rows)`` for every piece this rank owns and exchange the pieces, overlapping the collective of
    piece s (RCCL's own stream) with the kernels of piece s + 1 (current stream).
    Full exchange, uniform cut: an IN-PLACE all-gather (this rank's block is already where the collective would put it:
    sendbuf == recvbuf + rank * count, the aliasing NCCL / RCCL document for in-place all-gather).  Balanced cut: every rank
    broadcasts its block of the piece in place.  ``needed`` (dist.NeededRows of this direction, exchange = "needed"): the rows
    the peers read of this rank's block are packed (index_select, peer by peer), travel in one all_to_all_single with split
    sizes, and the rows this rank reads of the peers' blocks are scattered into place (index_copy_) one piece behind the
    launches.  Returns the gathered matrix, or with ``defer`` (the backward: the weight-gradient kernels need none of the
    gathered rows and run under the dX exchange) a pair (matrix, finish) -- ``finish()`` makes the current stream wait for what
    is still in flight."""
    full = torch.empty(max(dctx.total_rows, n), ld, dtype=dtype, device=device)
    if getattr(dctx, "poison_unread", False):      # tests: a row no exchange wrote is a NaN wherever it is read
        full.fill_(float("nan"))
    handles = []
    w = dctx.world
    esz = full.element_size()
    live = not dctx.emulate

    def unpack(s_idx, hs, recv):
        for h in hs:
            h.wait()
        if recv is not None and recv.shape[0] > 0:
            full.index_copy_(0, needed.recv_idx[s_idx], recv)

    for s_idx, plan in enumerate(plans_list):
        b, e = dctx.block(s_idx)
        mine = full[b:e]
        if plan.n_owned > 0:
            launch(plan, mine)
        if plan.n_owned < e - b:
            mine[plan.n_owned:].zero_()     # rows past the graph's end: defined bytes on the wire
        hs, recv = [], None
        if needed is not None:
            send = full.index_select(0, needed.send_idx[s_idx])
            recv = torch.empty(int(needed.recv_idx[s_idx].shape[0]), ld, dtype=dtype, device=device)
            if live:
                hs.append(torch.distributed.all_to_all_single(recv, send, needed.recv_splits[s_idx], needed.send_splits[s_idx],
                                                              group=dctx.group, async_op=True))
            dctx.stats["all_gather_bytes"] += recv.shape[0] * ld * esz
        elif dctx.uniform:
            sup = full[dctx.bounds[s_idx * w]:dctx.bounds[(s_idx + 1) * w]]
            if live:
                hs.append(torch.distributed.all_gather_into_tensor(sup, mine, group=dctx.group, async_op=True))
            dctx.stats["all_gather_bytes"] += (w - 1) * (e - b) * ld * esz      # bytes this rank RECEIVES
        else:
            for r in range(w):
                rb, re = dctx.block(s_idx, r)
                if re > rb:
                    if live:
                        hs.append(torch.distributed.broadcast(full[rb:re], src=dctx.src_rank(r), group=dctx.group, async_op=True))
                    if r != dctx.rank:
                        dctx.stats["all_gather_bytes"] += (re - rb) * ld * esz
        handles.append((s_idx, hs, recv))
        dctx.stats["all_gather"] += 1
        if needed is not None and len(handles) >= 2 and handles[-2] is not None:
            # the packed rows of the piece before: their collective had this piece's kernels to finish under
            unpack(*handles[-2])
            handles[-2] = None

    def finish():
        timed = dctx.time_waits and device.type == "cuda"
        evs = []
        for item in handles:          # piece by piece: which piece's exchange the launch stream had to wait for
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            if item is not None:
                unpack(*item)
            if timed:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                evs.append((e0, e1))
        if timed:
            dctx.stats["wait_events"].append(evs)
        handles.clear()

    if defer:
        return full[:n], finish
    finish()
    return full[:n]


def _shared_heavy_sums(shared, x: Tensor, width: int, dctx: "DistContext") -> Optional[Tensor]:
    """H[segment] = sum of the weighted rows of every heavy (node, relation) segment of the WHOLE graph: this rank's share of the
    rows (rgcn_ep_segment_sum), then one all-reduce over the ranks (eplan.SharedHeavy)"""
    if shared is None:
        return None
    hmat = _lib.ep_aggregate_shared(shared, x, width)
    if not dctx.emulate:
        torch.distributed.all_reduce(hmat, group=dctx.group)
    dctx.stats["all_reduce"] += 1
    dctx.stats["all_reduce_bytes"] += hmat.numel() * 4
    dctx.stats["shared_heavy_rows"] = dctx.stats.get("shared_heavy_rows", 0) + (shared.row_hi - shared.row_lo)
    return hmat


class _RGCNLayerFn(torch.autograd.Function):
    """a = act(sum_r mean-aggregate_r(x) @ W_r + x @ root + bias)   (forward: rgcn_fwd with the activation fused
    into its store; backward: rgcn_bwd_dx on the transposed plan + rgcn_bwd_dw)."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, comp: Optional[Tensor], root: Optional[Tensor], bias: Optional[Tensor],
                plans: GraphPlans, dctx: Optional[DistContext], act: int, input_relu: bool, grad_premasked: bool,
                flags: int, num_rel: int, dout: int):
        # weight / comp: the layer's OWN parameters -- dense [R, in, out], bases [B, in, out] + comp [R, B], or blocks
        # [R, nb, in / nb, out / nb]: a decomposition is composed inside the weight packer and differentiated from the dense
        # d_W scratch of the weight-gradient kernels (rgcn_pack_weights_basis / _block, rgcn_basis_backward / rgcn_block_backward),
        # so autograd never holds an [R, in, out] tensor (PyG materialises it on every call)
        n, din = x.shape
        fp: Optional[TilePlan] = plans.fwd if dctx is None else None
        xp = _rows16(x, din)
        wf = weight.detach().float().contiguous()
        cp = None if comp is None else comp.detach().float().contiguous()
        rt = None if root is None else root.detach().float().contiguous()
        bs = None if bias is None else bias.detach().float().contiguous()
        packed = _lib.pack_weights_decomposed(wf, cp, rt, num_rel, din, dout, transpose=False)
        ldo = _round4(dout)
        if dctx is None and plans.ep_fwd is not None:
            # edge-parallel path (eplan.py): relation-major dense units -> weighted products per slot -> per-destination sums
            out = torch.empty(n, ldo, dtype=torch.float32, device=x.device)
            ctx.ep_heavy = _lib.ep_layer(plans.ep_fwd, xp, din, packed, bs, out, dout, act, None, flags)
        elif dctx is None:
            out = torch.empty(n, ldo, dtype=torch.float32, device=x.device)
            _lib.fwd(_lib.plan_struct(fp), xp, din, packed, bs, out, dout, act, flags)
        else:
            # every rank computes its own blocks straight into the gathered buffer; with destination-range
            # ownership the per-layer all-reduce of SURVEY.md 8e degenerates to an all-gather (each row has
            # exactly one non-zero contributor), issued piece by piece under the next piece's kernels
            # (a piece whose forward runs the edge-parallel path -- a hub's block -- carries an eplan.EdgePlan instead)
            # (hubs split across ranks, eplan.SharedHeavy: every rank sums its share of the heavy segments' rows, one all-reduce
            # of the [segments, in] sums, then the owners' pseudo rows go through the transform)
            hm_f = _shared_heavy_sums(getattr(plans, "shared_fwd", None), xp, din, dctx)
            ctx.ep_heavy = hm_f

            def launch_fwd(pl, rows):
                if isinstance(pl, TilePlan):
                    _lib.fwd(_lib.plan_struct(pl), xp, din, packed, bs, rows, dout, act, flags)
                else:
                    _lib.ep_layer(pl, xp, din, packed, bs, rows, dout, act, None, flags, hmat=hm_f)
            out = _gather_pieces(dctx, [p.fwd if p.fwd is not None else p.ep_fwd for p in plans.pieces], launch_fwd, ldo, n, x.device,
                                 needed=plans.needed_fwd if dctx.exchange == "needed" else None)
        ctx.plans, ctx.dctx = plans, dctx
        ctx.dims = (n, din, dout, num_rel)
        ctx.has_root, ctx.has_bias = root is not None, bias is not None
        ctx.act, ctx.input_relu, ctx.flags = act, input_relu, flags
        # the activated output is only needed to differentiate the activation; a ReLU whose consumer folds the mask
        # into its dX store (grad_premasked) needs nothing
        need_a = act == _lib.ACT_SIGMOID or (act == _lib.ACT_RELU and not grad_premasked)
        ctx.need_a = need_a
        ctx.save_for_backward(xp, wf, cp, rt, out if need_a else None)
        return out if ldo == dout else out[:, :dout]

    @staticmethod
    def backward(ctx, g: Tensor):
        xp, wf, cp, rt, a_out = ctx.saved_tensors
        plans, dctx, flags = ctx.plans, ctx.dctx, ctx.flags
        n, din, dout, num_rel = ctx.dims
        need_x, need_wparam, need_comp, need_root, need_bias = ctx.needs_input_grad[:5]
        decomposed = cp is not None or wf.dim() == 4
        need_comp = need_comp and cp is not None
        need_w = need_wparam or need_comp              # the dense d_W[R, in, out] (for a decomposition: scratch)
        gp = _rows16(g, dout)
        if ctx.need_a:
            gp = _lib.act_backward(a_out, gp, ctx.act)       # dL/dz = dL/da * act'(a)
        dx = dw = droot = dbias = None
        finish_dx = None
        need_root = need_root and ctx.has_root
        need_bias = need_bias and ctx.has_bias
        dev = g.device
        # ONE flat buffer for the three weight gradients (a single all-reduce in the distributed case)
        sizes = [num_rel * din * dout if need_w else 0, din * dout if need_root else 0, dout if need_bias else 0]

        def views(flat):
            o0, o1 = sizes[0], sizes[0] + sizes[1]
            return (flat[:o0].view(num_rel, din, dout) if need_w else None,
                    flat[o0:o1].view(din, dout) if need_root else None,
                    flat[o1:].view(dout) if need_bias else None)

        # Single-GPU layers whose d_weight goes to the tile-major kernel: d_root / d_bias come from the plan-free streaming
        # kernel (rgcn_bwd_dw_root).  On large graphs it is enqueued on a SIDE stream before the dX launch: it is HBM-bound
        # with a tenth of a launch's MFMAs, uses no LDS and few registers, so its workgroups share the CUs with the MFMA-bound
        # dX kernel instead of adding their ~1 ms behind it.  The join is a stream wait, never a host sync.
        dwp = getattr(plans, "dw", None) if dctx is None else None
        merged = plans.fwd is not None and getattr(plans.fwd, "layout", 0) == 3 if dctx is None else False
        tiles_path = (dwp is not None and (need_w or merged) and plans.fwd is not None and plans.fwd.n_owned > 0 and
                      not (flags & (_lib.FLAG_DW_RING | _lib.FLAG_DW_DIRECT | _lib.FLAG_POINTER_GATHER)) and
                      _lib.buffer_addressable(n, xp.shape[1]) and _lib.buffer_addressable(n, gp.shape[1]))
        tiles_part = side = None
        if tiles_path:
            tiles_part = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
            _, pr, pb = views(tiles_part)
            if need_root or need_bias:
                if need_x and n >= _SIDE_STREAM_MIN_ROWS:
                    side = _side_stream(dev)
                    side.wait_stream(torch.cuda.current_stream(dev))      # gp (and xp) are produced on the current stream
                    with torch.cuda.stream(side):
                        _lib.bwd_dw_root(xp, din, gp, dout, pr, pb)
                else:
                    _lib.bwd_dw_root(xp, din, gp, dout, pr, pb)
        if need_x:
            packed_t = _lib.pack_weights_decomposed(wf, cp, rt, num_rel, din, dout, transpose=True)
            ldx = _round4(din)
            mask = xp if ctx.input_relu else None            # x = relu(z_prev): store dL/dz_prev = dx * (x > 0)
            if dctx is None and plans.ep_bwd is not None:
                dxp = torch.empty(n, ldx, dtype=torch.float32, device=dev)
                _lib.ep_layer(plans.ep_bwd, gp, dout, packed_t, None, dxp, din, _lib.ACT_NONE, mask, flags)
            elif dctx is None:
                dxp = torch.empty(n, ldx, dtype=torch.float32, device=dev)
                _lib.bwd_dx(_lib.plan_struct(plans.bwd), gp, dout, packed_t, dxp, din, mask, flags)
            else:
                hm_b = _shared_heavy_sums(getattr(plans, "shared_bwd", None), gp, dout, dctx)

                def launch_dx(pl, rows):
                    m = None if mask is None else mask[pl.node_begin:pl.node_end]
                    if isinstance(pl, TilePlan):
                        _lib.bwd_dx(_lib.plan_struct(pl), gp, dout, packed_t, rows, din, m, flags)
                    else:
                        _lib.ep_layer(pl, gp, dout, packed_t, None, rows, din, _lib.ACT_NONE, m, flags, hmat=hm_b)
                # the exchange of the dX pieces stays in flight under the weight-gradient kernels below (they read x and the
                # rank's own rows of g, none of the gathered rows); finish_dx() is the wait
                dxp, finish_dx = _gather_pieces(dctx, [p.bwd if p.bwd is not None else p.ep_bwd for p in plans.pieces], launch_dx, ldx, n, dev,
                                                needed=plans.needed_bwd if dctx.exchange == "needed" else None, defer=True)
            dx = dxp if ldx == din else dxp[:, :din]
        if tiles_path:
            # relations: tile-major kernel (gradient rows staged in LDS)
            if need_w:
                _lib.bwd_dw_tiles(_lib.plan_struct(dwp), plans.dw_walk, xp, din, gp, dout, views(tiles_part)[0], flags)
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)
            dw, droot, dbias = views(tiles_part)
        elif need_w or need_root or need_bias:
            # (an edge-parallel forward hands its dense relation-major units to the same kernels: GraphPlans.fwd_walk)
            pieces = [plans] if dctx is None else plans.pieces
            pinned = flags & (_lib.FLAG_DW_RING | _lib.FLAG_DW_DIRECT | _lib.FLAG_POINTER_GATHER)
            acc = None
            rank_dw = getattr(plans, "dw_rank", None) if dctx is not None else None
            if rank_dw is not None and not pinned and _lib.buffer_addressable(n, xp.shape[1]):
                # full exchange: x and g are replicated, so this rank's share of the weight gradients is ONE contiguous node
                # range of its own (dist.dw_range) -- one tile-major launch + the streaming root part, whatever the forward's pieces
                dwp_, walk_ = rank_dw
                if dwp_ is None:        # an empty range
                    acc = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
                    pieces = []
                b, e = (dwp_.node_begin, dwp_.node_end) if dwp_ is not None else (0, 0)
                if dwp_ is not None and _lib.buffer_addressable(e - b, gp.shape[1]):
                    acc = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
                    pw, pr, pb = views(acc)
                    if need_w:
                        _lib.bwd_dw_tiles(_lib.plan_struct(dwp_), walk_, xp, din, gp[b:e], dout, pw, flags)
                    if need_root or need_bias:
                        _lib.bwd_dw_root(xp[b:e], din, gp[b:e], dout, pr, pb)
                    dctx.stats["dw_tiles_rank"] = dctx.stats.get("dw_tiles_rank", 0) + 1
                    pieces = []
            for pc in pieces:
                fp = pc.fwd_walk
                if fp.n_owned <= 0:
                    continue
                part = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
                pw, pr, pb = views(part)
                b, e = fp.node_begin, fp.node_end
                if (dctx is not None and (need_w or getattr(fp, "layout", 0) == 3) and getattr(pc, "dw", None) is not None and not pinned
                        and _lib.buffer_addressable(n, xp.shape[1]) and _lib.buffer_addressable(e - b, gp.shape[1])):
                    # a rank's piece on the tile-major kernel, as the single-GPU step (its root part: the piece's own rows)
                    if need_w:
                        _lib.bwd_dw_tiles(_lib.plan_struct(pc.dw), pc.dw_walk, xp, din, gp[b:e], dout, pw, flags)
                    if need_root or need_bias:
                        _lib.bwd_dw_root(xp[b:e], din, gp[b:e], dout, pr, pb)
                    dctx.stats["dw_tiles_pieces"] = dctx.stats.get("dw_tiles_pieces", 0) + 1
                else:
                    _lib.bwd_dw(_lib.plan_struct(fp), xp, din, gp[b:e], dout, pw, pr, pb, flags)
                    epf = getattr(pc, "ep_fwd", None)
                    if need_w and epf is not None and epf.heavy is not None:
                        # the heavy segments' share: d_W_r += H_seg^T g[dst] over their pseudo rows (H from the forward)
                        hmat = getattr(ctx, "ep_heavy", None)
                        if hmat is None:
                            hmat = _lib.ep_aggregate_heavy(epf, xp, din)
                        pw2 = torch.empty_like(pw)
                        _lib.bwd_dw(_lib.plan_struct(epf.heavy_tile_plan()), hmat, din, gp[b:e], dout, pw2, None, None, flags)
                        pw.add_(pw2)
                acc = part if acc is None else acc.add_(part)
            if acc is None:
                acc = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
            if dctx is not None:
                if not dctx.emulate:
                    torch.distributed.all_reduce(acc, group=dctx.group)
                dctx.stats["all_reduce"] += 1
                dctx.stats["all_reduce_bytes"] += acc.numel() * 4
            dw, droot, dbias = views(acc)
        if finish_dx is not None:
            finish_dx()
        dcomp = None
        if decomposed and dw is not None:
            dw, dcomp = _lib.decomposed_weight_grads(dw.contiguous(), wf, cp, need_wparam, need_comp)
        return dx, dw, dcomp, droot, dbias, None, None, None, None, None, None, None, None


def rgcn_conv_function(x: Tensor, weight: Tensor, root: Optional[Tensor], bias: Optional[Tensor],
                       plans: GraphPlans, dctx: Optional[DistContext] = None, activation: Optional[str] = None,
                       input_relu: bool = False, grad_premasked: bool = False, flags: int = 0,
                       comp: Optional[Tensor] = None, num_relations: Optional[int] = None, out_channels: Optional[int] = None) -> Tensor:
    """weight: dense [R, in, out]; or, with ``comp [R, B]``, the bases [B, in, out]; or blocks [R, nb, in / nb, out / nb]
    (then ``out_channels`` = nb * weight.shape[3])."""
    if x.device.type != "cuda":
        raise RuntimeError("RGCNConv runs only on an MI355X (ROCm 'cuda' device); there is no CPU fallback")
    if activation not in _ACT_CODES:
        raise ValueError(f"fused activation must be one of {list(_ACT_CODES)}")
    _lib.load()
    if comp is not None:
        num_rel, dout = int(comp.shape[0]), int(weight.shape[2])
    elif weight.dim() == 4:
        num_rel, dout = int(weight.shape[0]), int(weight.shape[1] * weight.shape[3])
    else:
        num_rel, dout = int(weight.shape[0]), int(weight.shape[2])
    return _RGCNLayerFn.apply(x, weight, comp, root, bias, plans, dctx, _ACT_CODES[activation], bool(input_relu),
                              bool(grad_premasked), int(flags), num_relations or num_rel, out_channels or dout)


def glorot_(t: Tensor) -> Tensor:
    """PyG ``glorot``: U(+-sqrt(6 / (size(-2) + size(-1))))"""
    bound = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        return t.uniform_(-bound, bound)


class RGCNConv(nn.Module):
    r"""Drop-in for ``torch_geometric.nn.RGCNConv`` (PyG 2.3.1):

    ``RGCNConv(in_channels, out_channels, num_relations, num_bases=None, num_blocks=None,
    aggr='mean', root_weight=True, is_sorted=False, bias=True)``;
    ``forward(x[N,in] f32, edge_index[2,E] int64, edge_type[E] int64) -> [N,out] f32``.

    Parameters (registered in PyG's order): ``weight`` ``[R,in,out]`` (``[B,in,out]`` with
    ``num_bases=B``; ``[R,nb,in/nb,out/nb]`` with ``num_blocks=nb``), ``comp`` ``[R,B]`` or ``None``,
    ``root`` ``[in,out]`` or ``None``, ``bias`` ``[out]`` or ``None``.
    """

    def __init__(self, in_channels: int, out_channels: int, num_relations: int,
                 num_bases: Optional[int] = None, num_blocks: Optional[int] = None, aggr: str = "mean",
                 root_weight: bool = True, is_sorted: bool = False, bias: bool = True, **kwargs):
        super().__init__()
        if num_bases is not None and num_blocks is not None:
            raise ValueError("Can not apply both basis-decomposition and block-diagonal-decomposition "
                             "at the same time.")
        if isinstance(in_channels, (tuple, list)):
            if in_channels[0] != in_channels[1]:
                raise NotImplementedError("bipartite RGCNConv is not used by the reference and not built")
            in_channels = in_channels[0]
        if aggr not in ("mean", "sum", "add"):
            raise ValueError(f"unsupported aggr {aggr!r} (mean / sum)")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_relations = num_relations
        self.num_bases = num_bases
        self.num_blocks = num_blocks
        self.aggr = "sum" if aggr == "add" else aggr
        self.is_sorted = is_sorted  # only meaningful for PyG's pyg_lib path; plans are order-independent
        self.dist: Optional[DistContext] = None
        self._dist_plans = None
        self.kernel_flags = 0     # RGCN_FLAG_* passed to every launch of this layer (tests pin kernel paths with it)
        self.dw_tiles = True      # d_weight by the tile-major kernel where it applies (_plans); False: relation-major kernels
        # forward / dX on the producer-split bf16 x 3 kernel where it applies (64 x 64, 128-slot chunks, single GPU): fp32-
        # equivalent arithmetic (24-bit operand significands, exact products, fp32 accumulation), 1 ms per step faster at
        # the headline config.  False (or RGCN_SPLIT_PRODUCERS=0): the exact-fp32 MFMA kernel everywhere
        self.split_producers = _SPLIT_PRODUCERS_DEFAULT == "1"
        # "auto": per direction, the tile kernels or the edge-parallel path (csrc/rgcn_ep.hip), whichever eplan.choose_path
        # expects to be faster on the graph (many relations / few tiles / hubs -> edge-parallel); "ring" / "ep" or a
        # (forward, dX) pair pins it.  RGCN_PATH in the environment at import time sets the default.
        self.path = _PATH_DEFAULT
        self.team_layout = _TEAM_LAYOUT_DEFAULT     # plans of such layers in the team placement (experiment builds: two consumer teams)
        # forward / dX plans in layout 3 where the producer-split kernel runs them and the tile-major kernel takes d_weight: the rows
        # of a (destination, relation) run on ONE slot, summed by the producers before the cut -- aggregate, then transform, as the
        # reference does; 13 % fewer row tiles at the headline config (plan.compact_runs).  RGCN_MERGE_RUNS=0 / False: layout 0
        self.merge_runs = _MERGE_RUNS_DEFAULT
        if num_bases is not None:
            self.weight = nn.Parameter(torch.empty(num_bases, in_channels, out_channels))
            self.comp = nn.Parameter(torch.empty(num_relations, num_bases))
        elif num_blocks is not None:
            assert in_channels % num_blocks == 0 and out_channels % num_blocks == 0, \
                "in_channels and out_channels must be divisible by num_blocks"
            self.weight = nn.Parameter(torch.empty(num_relations, num_blocks, in_channels // num_blocks,
                                                   out_channels // num_blocks))
            self.register_parameter("comp", None)
        else:
            self.weight = nn.Parameter(torch.empty(num_relations, in_channels, out_channels))
            self.register_parameter("comp", None)
        if root_weight:
            self.root = nn.Parameter(torch.empty(in_channels, out_channels))
        else:
            self.register_parameter("root", None)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        tile_for(in_channels, out_channels)  # validates the widths early
        self.reset_parameters()

    def reset_parameters(self) -> None:
        glorot_(self.weight)
        if self.comp is not None:
            glorot_(self.comp)
        if self.root is not None:
            glorot_(self.root)
        if self.bias is not None:
            with torch.no_grad():
                self.bias.zero_()

    def effective_weight(self) -> Tensor:
        """Dense ``[R, in, out]`` relation weights by differentiable torch ops -- NOT on the forward path (the library composes
        a decomposition inside its weight packer: rgcn_pack_weights_basis / _block); kept as the reference the tests compare
        that packer with.  Basis: ``(comp @ weight.view(B,-1)).view(R,in,out)``; block-diagonal: blocks placed on the diagonal
        of a zero matrix."""
        if self.num_bases is not None:
            return (self.comp @ self.weight.view(self.num_bases, -1)).view(
                self.num_relations, self.in_channels, self.out_channels)
        if self.num_blocks is not None:
            nb = self.num_blocks
            bi, bo = self.in_channels // nb, self.out_channels // nb
            eye = torch.eye(nb, device=self.weight.device, dtype=self.weight.dtype)
            # [R, nb, bi, bo] -> [R, nb, bi, nb, bo] with zeros off the block diagonal
            w = torch.einsum("rbio,bc->rbico", self.weight, eye)
            return w.reshape(self.num_relations, self.in_channels, self.out_channels)
        return self.weight

    def _plans(self, x: Tensor, edge_index: Tensor, edge_type: Tensor) -> GraphPlans:
        n = x.shape[0]
        e = int(edge_type.shape[0])
        tile, chunk = self.layout(n, e)
        split = self.team_layout and chunk == 128 and self._use_split_producers(chunk)       # plan layout 1 (team placement)
        # the tile-major weight-gradient kernel: 64 x 64 layers with few relations on graphs large enough to fill it
        # (it gathers through buffer descriptors only: above 2^24 rows / 4 GiB the relation-major kernels run)
        from .plan import padded_width
        dw_tiles = (self.dw_tiles and padded_width(self.in_channels) == 64 and padded_width(self.out_channels) == 64
                    and self.num_relations <= 32 and e >= DW_TILES_MIN_EDGES and x.is_cuda
                    and _lib.buffer_addressable(n, _round4(self.in_channels))
                    and _lib.buffer_addressable(n, _round4(self.out_channels)))
        # layout 3 only where nothing but rgcn_tile3p_kernel walks the forward / transposed plans: the split kernels unpinned
        # (no kernel flags), d_weight on its own tile-major plan, d_root / d_bias on the plan-free streaming kernel
        if (not split and self.merge_runs and dw_tiles and self.kernel_flags == 0 and
                (self._use_split_producers(chunk) or self._exact_merge(chunk))):
            split = 3
        if self.dist is None:
            paths = self.path if self.path == "auto" else ((self.path, self.path) if isinstance(self.path, str) else tuple(self.path))
            if not x.is_cuda:
                paths = ("ring", "ring")
            return cached_graph_plans(edge_index, edge_type, n, self.num_relations, tile, self.aggr, chunk=chunk, split=split,
                                      dw_tiles=dw_tiles, paths=paths, widths=(self.in_channels, self.out_channels))
        # edge-partitioned: every piece on the kernels the whole graph would take per direction (+ its own tile-major
        # weight-gradient plan where the forward runs tile kernels)
        from .dist import cached_rank_plans
        paths = ("ring", "ring")
        if x.is_cuda:
            paths = self.path if self.path == "auto" else ((self.path, self.path) if isinstance(self.path, str) else tuple(self.path))
        return cached_rank_plans(edge_index, edge_type, n, self.num_relations, tile, self.aggr, self.dist, chunk, split, dw_tiles,
                                 paths=paths, widths=(self.in_channels, self.out_channels))

    def _use_split_producers(self, chunk: int) -> bool:
        """whether a plan of ``self.layout`` with that chunk runs on the bf16 x 3 kernel (layout() returns 128-slot chunks for a
        64 x 64 layer with split_producers only where that kernel is the modelled choice, at a tile it has room for)"""
        from .plan import padded_width
        return (self.split_producers and chunk in (112, 128) and not _ENV_TILE
                and padded_width(self.in_channels) == 64 and padded_width(self.out_channels) == 64)

    def _exact_merge(self, chunk: int) -> bool:
        """the exact-fp32 kernel on layout-3 plans: 64 x 64 layers, 128-slot chunks, where the bf16 x 3 kernel is switched off"""
        from .plan import padded_width
        return (not self.split_producers and self.merge_runs and chunk == 128 and not _ENV_TILE
                and padded_width(self.in_channels) == 64 and padded_width(self.out_channels) == 64)

    def layout(self, n_nodes: int, n_edges: int) -> Tuple[int, int]:
        """(tile, chunk) of this layer's plans on a graph of that size: ``layout_for``, capped at the producer-split kernel's
        tile where that kernel will run (dist.attach aligns the ranks' node ranges to the same tile), or at the tile that leaves
        the exact-fp32 kernel's chunks room for their shadow row tiles where it will walk layout-3 plans."""
        tile, chunk = layout_for(self.in_channels, self.out_channels, n_nodes, n_edges, self.num_relations)
        if self._use_split_producers(128) and not _ENV_CHUNK:
            # 64 x 64 with the bf16 x 3 kernel available: its own layout (128-slot chunks, tiles up to 224, its own cycles per chunk
            # and row tile) against the exact-fp32 kernel's, by modelled launch time -- round 4: on a 100k-node / 1M-edge graph the
            # exact model's (400, 64) kept the layer off the faster kernel: 0.416 ms per step replayed against 0.333 at (208, 128)
            from .plan import choose_layout
            args = (n_nodes, n_edges, self.num_relations, self.in_channels, self.out_channels)
            t0, c0, cost0 = choose_layout(*args, with_cost=True)
            t3, c3, cost3 = choose_layout(*args, kernel="bf16x3", with_cost=True)
            if cost3 <= cost0:
                return t3, c3
            if c0 == 128:      # (the exact-fp32 kernel with 128-slot chunks would be taken for the other one: keep it on 64)
                tile, chunk = layout_for(self.in_channels, self.out_channels, n_nodes, n_edges, self.num_relations)
                return min(tile, SPLIT_PRODUCERS_TILE), chunk
            return t0, c0
        if self._use_split_producers(chunk):
            tile = min(tile, SPLIT_PRODUCERS_TILE)
        elif (self._exact_merge(chunk) and self.dw_tiles and self.kernel_flags == 0 and self.num_relations <= 32
              and n_edges >= DW_TILES_MIN_EDGES):
            tile = min(tile, EXACT_MERGE_TILE)
        return tile, chunk

    def forward(self, x: Tensor, edge_index: Tensor, edge_type: Optional[Tensor] = None, *,
                _activation: Optional[str] = None, _input_relu: bool = False,
                _grad_premasked: bool = False) -> Tensor:
        """PyG's ``forward(x, edge_index, edge_type)``.  The keyword-only arguments are the private hook the model
        wrappers use to fuse the activations either side of the layer (layers._RGCNStack._tail):
        ``_activation`` ('relu' | 'sigmoid') is applied in the forward kernel's store; ``_input_relu`` says x is the
        ReLU output of the previous layer, so the dX kernel stores dL/dz_prev = dX * (x > 0); ``_grad_premasked`` says
        every consumer of THIS layer's ReLU output does that, so no ReLU backward runs here."""
        assert edge_type is not None, "edge_type is required (PyG RGCNConv asserts the same)"
        if x is None or not torch.is_floating_point(x):
            raise NotImplementedError("featureless (integer / None x) RGCNConv is never used by the reference "
                                      "(x is always float: model/layers.py:21,62,108) and is not built")
        if x.dim() != 2 or x.shape[1] != self.in_channels:
            raise ValueError(f"x must be [N, {self.in_channels}], got {tuple(x.shape)}")
        plans = self._plans(x, edge_index, edge_type)
        flags = self.kernel_flags
        # (the arithmetic mode follows the layer and the graph's size -- the chunk of self.layout -- not which path the other
        # direction happened to take: an edge-parallel pair on a graph too small for 128-slot chunks stays on exact fp32)
        if self._use_split_producers(self.layout(x.shape[0], int(edge_type.shape[0]))[1]):
            # rgcn_fwd / rgcn_bwd_dx / rgcn_bwd_dw_tiles / rgcn_ep_transform: the bf16 x 3 (fp32-equivalent) forms of 64 x 64 layers;
            # the library falls back where they do not fit
            flags |= _lib.FLAG_SPLIT_PRODUCERS
        return rgcn_conv_function(x, self.weight, self.root, self.bias, plans, self.dist,
                                  _activation, _input_relu, _grad_premasked and _activation == "relu", flags,
                                  comp=self.comp, num_relations=self.num_relations, out_channels=self.out_channels)

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, "
                f"num_relations={self.num_relations})")
