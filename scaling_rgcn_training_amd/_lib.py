"""ctypes binding of librgcn_mi355x.so (C ABI: include/rgcn_mi355x.h).

There is deliberately no fallback: if the library is missing or a call fails, an exception is
raised.  The library is built in-tree by ``__graft_entry__.build()`` / ``tools/build_lib.sh``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RGCN_LIB: an alternative build of the same library (kernel experiments: tools/debug/)
LIB_PATH = os.environ.get("RGCN_LIB") or os.path.join(_HERE, "librgcn_mi355x.so")
ABI_VERSION = 17

EXPORTS = (
    "rgcn_abi_version", "rgcn_status_string", "rgcn_padded_width", "rgcn_packed_weight_floats",
    "rgcn_pack_weights", "rgcn_fwd", "rgcn_bwd_dx", "rgcn_act_backward", "rgcn_bwd_dw_workspace_bytes", "rgcn_bwd_dw",
    "rgcn_plan_workspace_bytes", "rgcn_edge_weights", "rgcn_plan_build_begin", "rgcn_plan_build_finish",
    "rgcn_dw_tiles_geometry", "rgcn_dw_tiles_walk", "rgcn_bwd_dw_tiles_workspace_bytes", "rgcn_bwd_dw_tiles",
    "rgcn_bwd_dw_root_workspace_bytes", "rgcn_bwd_dw_root", "rgcn_ep_transform", "rgcn_ep_segment_sum",
    "rgcn_pack_weights_basis", "rgcn_pack_weights_block", "rgcn_basis_backward", "rgcn_block_backward", "rgcn_eplan_segments",
)

# enum rgcn_act / RGCN_FLAG_* of include/rgcn_mi355x.h
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
ERR_PLAN = -4          # inconsistent plan, or a plan layout the called kernel does not walk
ERR_ADDRESS = -10      # rgcn_bwd_dw_tiles: operands not addressable through a buffer descriptor
FLAG_POINTER_GATHER, FLAG_DW_RING, FLAG_DW_DIRECT, FLAG_EXACT_FP32, FLAG_DW_ROOT_ONLY, FLAG_SPLIT_PRODUCERS = 1, 2, 4, 8, 16, 32


class RgcnPlanStruct(C.Structure):
    """struct rgcn_plan of include/rgcn_mi355x.h"""
    _fields_ = [
        ("n_nodes", C.c_int32), ("n_owned", C.c_int32), ("num_relations", C.c_int32),
        ("tile", C.c_int32), ("n_tiles", C.c_int32), ("n_chunks", C.c_int32), ("chunk", C.c_int32), ("n_units", C.c_int32),
        ("layout", C.c_int32), ("chunk_rows", C.c_int32),
        ("tile_ptr", C.c_void_p), ("chunk_rel", C.c_void_p), ("chunk_cnt", C.c_void_p),
        ("chunk_tile", C.c_void_p), ("chunk_flags", C.c_void_p), ("rel_order", C.c_void_p), ("slot_src", C.c_void_p),
        ("slot_w", C.c_void_p), ("slot_row", C.c_void_p), ("slot_acc", C.c_void_p), ("slot_src2", C.c_void_p),
    ]


class RgcnEdgeUnits(C.Structure):
    """struct rgcn_edge_units of include/rgcn_mi355x.h"""
    _fields_ = [("n_nodes", C.c_int32), ("n_units", C.c_int32), ("num_relations", C.c_int32), ("reserved", C.c_int32),
                ("unit_rel", C.c_void_p), ("unit_cnt", C.c_void_p), ("slot_src", C.c_void_p), ("slot_w", C.c_void_p)]


class RgcnGraphStruct(C.Structure):
    """struct rgcn_graph: the int64 COO exactly as the caller holds it (strided views allowed)"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("type", C.c_void_p),
                ("src_stride", C.c_int64), ("dst_stride", C.c_int64), ("type_stride", C.c_int64),
                ("num_edges", C.c_int64), ("num_nodes", C.c_int32), ("num_relations", C.c_int32)]


class RgcnPlanSizes(C.Structure):
    """struct rgcn_plan_sizes"""
    _fields_ = [("n_tiles", C.c_int32), ("n_chunks", C.c_int32), ("n_units", C.c_int32), ("reserved", C.c_int32),
                ("n_slots", C.c_int64), ("n_edges", C.c_int64), ("opaque", C.c_uint64 * 16)]


class RgcnLibraryError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the HIP library once; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RgcnLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the R-GCN layer.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.rgcn_abi_version.restype = i32
    lib.rgcn_abi_version.argtypes = []
    lib.rgcn_status_string.restype = C.c_char_p
    lib.rgcn_status_string.argtypes = [i32]
    lib.rgcn_padded_width.restype = i32
    lib.rgcn_padded_width.argtypes = [i32]
    lib.rgcn_packed_weight_floats.restype = sz
    lib.rgcn_packed_weight_floats.argtypes = [i32, i32, i32]
    lib.rgcn_pack_weights.restype = i32
    lib.rgcn_pack_weights.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    lib.rgcn_fwd.restype = i32
    u32 = C.c_uint
    lib.rgcn_fwd.argtypes = [C.POINTER(RgcnPlanStruct), vp, i32, i32, vp, vp, vp, i32, i32, i32, u32, vp]
    lib.rgcn_bwd_dx.restype = i32
    lib.rgcn_bwd_dx.argtypes = [C.POINTER(RgcnPlanStruct), vp, i32, i32, vp, vp, i32, i32, vp, i32, u32, vp]
    lib.rgcn_act_backward.restype = i32
    lib.rgcn_act_backward.argtypes = [vp, vp, vp, C.c_long, i32, i32, vp]
    lib.rgcn_bwd_dw_workspace_bytes.restype = sz
    lib.rgcn_bwd_dw_workspace_bytes.argtypes = [C.POINTER(RgcnPlanStruct), i32, i32]
    lib.rgcn_bwd_dw.restype = i32
    lib.rgcn_bwd_dw.argtypes = [C.POINTER(RgcnPlanStruct), vp, i32, i32, vp, i32, i32, vp, sz, vp, vp, vp, u32, vp]
    i64 = C.c_int64
    lib.rgcn_plan_workspace_bytes.restype = sz
    lib.rgcn_plan_workspace_bytes.argtypes = [i64, i32, i32, i32]
    lib.rgcn_edge_weights.restype = i32
    lib.rgcn_edge_weights.argtypes = [C.POINTER(RgcnGraphStruct), i32, vp, vp, sz, vp]
    lib.rgcn_plan_build_begin.restype = i32
    lib.rgcn_plan_build_begin.argtypes = [C.POINTER(RgcnGraphStruct), vp, i32, i32, i32, i32, i32, i32, vp, sz,
                                          C.POINTER(RgcnPlanSizes), vp]
    lib.rgcn_plan_build_finish.restype = i32
    lib.rgcn_plan_build_finish.argtypes = [C.POINTER(RgcnPlanSizes), vp, sz, C.POINTER(RgcnPlanStruct), vp]
    lib.rgcn_dw_tiles_geometry.restype = i32
    lib.rgcn_dw_tiles_geometry.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.rgcn_dw_tiles_walk.restype = i32
    lib.rgcn_dw_tiles_walk.argtypes = [C.POINTER(RgcnPlanStruct), vp, vp]
    lib.rgcn_bwd_dw_tiles_workspace_bytes.restype = sz
    lib.rgcn_bwd_dw_tiles_workspace_bytes.argtypes = [i32]
    lib.rgcn_bwd_dw_tiles.restype = i32
    lib.rgcn_bwd_dw_tiles.argtypes = [C.POINTER(RgcnPlanStruct), vp, vp, i32, i32, vp, i32, i32, vp, sz, vp, u32, vp]
    lib.rgcn_bwd_dw_root_workspace_bytes.restype = sz
    lib.rgcn_bwd_dw_root_workspace_bytes.argtypes = []
    lib.rgcn_bwd_dw_root.restype = i32
    lib.rgcn_bwd_dw_root.argtypes = [vp, i32, i32, vp, i32, i32, C.c_long, vp, sz, vp, vp, vp]
    lib.rgcn_pack_weights_basis.restype = i32
    lib.rgcn_pack_weights_basis.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.rgcn_pack_weights_block.restype = i32
    lib.rgcn_pack_weights_block.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.rgcn_basis_backward.restype = i32
    lib.rgcn_basis_backward.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
    lib.rgcn_block_backward.restype = i32
    lib.rgcn_block_backward.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.rgcn_eplan_segments.restype = i32
    lib.rgcn_eplan_segments.argtypes = [vp, i64, i32, vp, sz, vp, vp, vp]
    lib.rgcn_ep_transform.restype = i32
    lib.rgcn_ep_transform.argtypes = [C.POINTER(RgcnEdgeUnits), vp, i32, i32, vp, vp, i32, i32, u32, vp]
    lib.rgcn_ep_segment_sum.restype = i32
    lib.rgcn_ep_segment_sum.argtypes = [vp, i32, vp, vp, vp, i32, i32, vp, i32, vp, i32, i32, vp, i32, vp]
    if lib.rgcn_abi_version() != ABI_VERSION:
        raise RgcnLibraryError(f"ABI version mismatch: library {lib.rgcn_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().rgcn_status_string(status).decode()
        err = RgcnLibraryError(f"{what} failed with status {status}: {msg}")
        err.status = status
        raise err


def buffer_addressable(rows: int, ld: int) -> bool:
    """Can a [rows, ld] fp32 matrix be gathered through a buffer descriptor (csrc/rgcn_kernels_shared.h buffer_bytes: 24-bit
    row index, 32-bit offsets with the one-past-the-end padding row in range)?  The tile-major weight-gradient kernel
    addresses both operands that way only; the other kernels fall back to 64-bit pointers."""
    return rows < (1 << 24) and (rows + 1) * ld * 4 < 0xFFFFFF00


def plan_struct(plan) -> RgcnPlanStruct:
    """Fill the C struct from a plan.TilePlan whose tensors live on the GPU."""
    cached = getattr(plan, "_cstruct", None)       # the plan's tensors never change: build the struct once
    if cached is not None:
        return cached
    if plan.slot_src.device.type != "cuda":
        raise RgcnLibraryError("the graph plan must live on the GPU (plan tensors are on %s)" % plan.slot_src.device)
    plan._cstruct = RgcnPlanStruct(
        plan.n_nodes, plan.n_owned, plan.num_relations, plan.tile, plan.n_tiles, plan.n_chunks, plan.chunk, plan.n_units,
        int(getattr(plan, "layout", 0)), int(getattr(plan, "chunk_rows", 0) or plan.chunk),
        plan.tile_ptr.data_ptr(), plan.chunk_rel.data_ptr(), plan.chunk_cnt.data_ptr(),
        plan.chunk_tile.data_ptr(), plan.chunk_flags.data_ptr(), plan.rel_order.data_ptr(), plan.slot_src.data_ptr(),
        plan.slot_w.data_ptr(), plan.slot_row.data_ptr(), plan.slot_acc.data_ptr(),
        _ptr(getattr(plan, "slot_src2", None)))
    return plan._cstruct


def _stream(t: torch.Tensor) -> int:
    """the current torch stream OF THE TENSOR'S DEVICE (launches go where the data lives, like every torch op)"""
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def padded_width(w: int) -> int:
    return load().rgcn_padded_width(int(w))


# ---- thin typed wrappers: device tensors in, device tensors out, current torch stream of the tensors' device ----
def pack_weights(weight: torch.Tensor, root: Optional[torch.Tensor], transpose: bool) -> torch.Tensor:
    lib = load()
    r, din, dout = weight.shape
    n = lib.rgcn_packed_weight_floats(r, din, dout)
    if n == 0:
        raise RgcnLibraryError(f"unsupported layer widths {din}->{dout} (1..128 per side)")
    packed = torch.empty(n, dtype=torch.float32, device=weight.device)
    with torch.cuda.device(weight.device):
        check(lib.rgcn_pack_weights(weight.data_ptr(), _ptr(root), r, din, dout, int(transpose),
                                    packed.data_ptr(), _stream(weight)), "rgcn_pack_weights")
    return packed


def pack_weights_decomposed(weight: torch.Tensor, comp: Optional[torch.Tensor], root: Optional[torch.Tensor], num_relations: int,
                            din: int, dout: int, transpose: bool) -> torch.Tensor:
    """The pack of a layer's weights whatever their parametrisation (PyG RGCNConv): dense ``weight [R, in, out]``, basis
    decomposition (``weight [B, in, out]`` + ``comp [R, B]``) or block-diagonal (``weight [R, nb, in/nb, out/nb]``).  The
    decompositions are composed inside the packer: no [R, in, out] tensor exists."""
    if comp is None and weight.dim() == 3:
        return pack_weights(weight, root, transpose)
    lib = load()
    n = lib.rgcn_packed_weight_floats(num_relations, din, dout)
    if n == 0:
        raise RgcnLibraryError(f"unsupported layer widths {din}->{dout} (1..128 per side)")
    packed = torch.empty(n, dtype=torch.float32, device=weight.device)
    with torch.cuda.device(weight.device):
        if comp is not None:
            check(lib.rgcn_pack_weights_basis(weight.data_ptr(), comp.data_ptr(), _ptr(root), num_relations, weight.shape[0], din, dout,
                                              int(transpose), packed.data_ptr(), _stream(weight)), "rgcn_pack_weights_basis")
        else:
            check(lib.rgcn_pack_weights_block(weight.data_ptr(), _ptr(root), num_relations, weight.shape[1], din, dout,
                                              int(transpose), packed.data_ptr(), _stream(weight)), "rgcn_pack_weights_block")
    return packed


def decomposed_weight_grads(d_w: torch.Tensor, weight: torch.Tensor, comp: Optional[torch.Tensor], need_weight: bool, need_comp: bool):
    """(d_weight, d_comp) of the layer's own parameters from the dense ``d_w [R, in, out]`` scratch the weight-gradient kernels
    wrote (rgcn_basis_backward / rgcn_block_backward)."""
    lib = load()
    r, din, dout = d_w.shape
    with torch.cuda.device(d_w.device):
        if comp is not None:
            dv = torch.empty_like(weight) if need_weight else None
            dc = torch.empty_like(comp) if need_comp else None
            if need_weight or need_comp:
                check(lib.rgcn_basis_backward(d_w.data_ptr(), weight.data_ptr(), comp.data_ptr(), r, weight.shape[0], din, dout,
                                              _ptr(dv), _ptr(dc), _stream(d_w)), "rgcn_basis_backward")
            return dv, dc
        db = None
        if need_weight:
            db = torch.empty_like(weight)
            check(lib.rgcn_block_backward(d_w.data_ptr(), r, weight.shape[1], din, dout, db.data_ptr(), _stream(d_w)), "rgcn_block_backward")
        return db, None


def fwd(ps: RgcnPlanStruct, x: torch.Tensor, din: int, packed: torch.Tensor,
        bias: Optional[torch.Tensor], out: torch.Tensor, dout: int, act: int = ACT_NONE, flags: int = 0) -> None:
    with torch.cuda.device(x.device):
        check(load().rgcn_fwd(C.byref(ps), x.data_ptr(), x.stride(0), din, packed.data_ptr(), _ptr(bias),
                              out.data_ptr(), out.stride(0), dout, int(act), int(flags), _stream(x)), "rgcn_fwd")


def bwd_dx(ps_t: RgcnPlanStruct, g: torch.Tensor, dout: int, packed_t: torch.Tensor,
           dx: torch.Tensor, din: int, relu_of: Optional[torch.Tensor] = None, flags: int = 0) -> None:
    with torch.cuda.device(g.device):
        check(load().rgcn_bwd_dx(C.byref(ps_t), g.data_ptr(), g.stride(0), dout, packed_t.data_ptr(),
                                 dx.data_ptr(), dx.stride(0), din, _ptr(relu_of),
                                 0 if relu_of is None else relu_of.stride(0), int(flags), _stream(g)), "rgcn_bwd_dx")


def act_backward(a: torch.Tensor, da: torch.Tensor, act: int) -> torch.Tensor:
    """dz = da * act'(a), a = act(z) (both [rows, ld] with the same 16-byte-aligned stride)"""
    assert a.stride(0) == da.stride(0) and a.stride(1) == 1 and da.stride(1) == 1 and a.stride(0) % 4 == 0
    dz = torch.empty_strided(da.shape, da.stride(), dtype=torch.float32, device=da.device)
    with torch.cuda.device(a.device):
        check(load().rgcn_act_backward(a.data_ptr(), da.data_ptr(), dz.data_ptr(), a.shape[0], a.stride(0), int(act),
                                       _stream(a)), "rgcn_act_backward")
    return dz


def bwd_dw(ps: RgcnPlanStruct, x: torch.Tensor, din: int, g: torch.Tensor, dout: int,
           d_weight: Optional[torch.Tensor], d_root: Optional[torch.Tensor],
           d_bias: Optional[torch.Tensor], flags: int = 0) -> None:
    lib = load()
    nbytes = lib.rgcn_bwd_dw_workspace_bytes(C.byref(ps), din, dout)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.rgcn_bwd_dw(C.byref(ps), x.data_ptr(), x.stride(0), din, g.data_ptr(), g.stride(0), dout,
                              ws.data_ptr(), nbytes, _ptr(d_weight), _ptr(d_root), _ptr(d_bias), int(flags), _stream(x)),
              "rgcn_bwd_dw")


# ---- graph plan, built on the device (rgcn_plan.hip) --------------------------------------------------------------
def graph_struct(edge_index: torch.Tensor, edge_type: torch.Tensor, n_nodes: int, num_relations: int):
    """(struct rgcn_graph, tensors it points into).  int64 device tensors are passed as they are -- strided views such
    as the rows of the reference's transposed [E, 3] edge tensor included; other integer dtypes are converted."""
    src, dst = edge_index[0], edge_index[1]
    if src.dtype != torch.int64:
        src, dst = src.long(), dst.long()
    typ = edge_type if edge_type.dtype == torch.int64 else edge_type.long()
    if src.device.type != "cuda" or typ.device != src.device:
        raise RgcnLibraryError("the graph must live on the GPU")
    e = int(typ.shape[0])
    g = RgcnGraphStruct(src.data_ptr() if e else None, dst.data_ptr() if e else None, typ.data_ptr() if e else None,
                        src.stride(0) if e else 1, dst.stride(0) if e else 1, typ.stride(0) if e else 1,
                        e, int(n_nodes), int(num_relations))
    return g, (src, dst, typ)


def plan_workspace(num_edges: int, n_owned: int, num_relations: int, tile: int, device) -> torch.Tensor:
    n = load().rgcn_plan_workspace_bytes(int(num_edges), int(n_owned), int(num_relations), int(tile))
    if n == 0:
        raise RgcnLibraryError("rgcn_plan_workspace_bytes: bad arguments")
    return torch.empty(n, dtype=torch.uint8, device=device)


def edge_weights(graph: RgcnGraphStruct, aggr: str, ws: torch.Tensor) -> torch.Tensor:
    if aggr not in ("mean", "sum", "add"):
        raise ValueError(f"unsupported aggr {aggr!r}")
    w = torch.empty(max(int(graph.num_edges), 1), dtype=torch.float32, device=ws.device)
    with torch.cuda.device(ws.device):
        check(load().rgcn_edge_weights(C.byref(graph), int(aggr != "mean"), w.data_ptr(), ws.data_ptr(), ws.numel(),
                                       _stream(ws)), "rgcn_edge_weights")
    return w[:int(graph.num_edges)]


def plan_build(graph: RgcnGraphStruct, w: torch.Tensor, transposed: bool, node_begin: int, node_end: int, tile: int,
               chunk: int, ws: torch.Tensor, split=False):
    """-> (RgcnPlanStruct, dict of the ten device arrays, n_edges placed).  split: plan layout -- 0 (False) rows of a (tile, relation) group dealt
    over its row tiles; 1 (True) the team placement of experiment builds; 2 relation-major units of the
    edge-parallel path (one pseudo tile); 3 layout 0 with the rows of a (destination, relation) run compacted onto one head slot
    (compact_runs_kernel: only rgcn_tile3p_kernel walks it, every weight-gradient entry point refuses it)"""
    lib, dev = load(), ws.device
    sizes = RgcnPlanSizes()
    with torch.cuda.device(dev):
        check(lib.rgcn_plan_build_begin(C.byref(graph), w.data_ptr() if graph.num_edges else None, int(transposed),
                                        int(node_begin), int(node_end), int(tile), int(chunk), int(split), ws.data_ptr(),
                                        ws.numel(), C.byref(sizes), _stream(ws)), "rgcn_plan_build_begin")
        i32 = dict(dtype=torch.int32, device=dev)
        arr = {
            "tile_ptr": torch.empty(sizes.n_tiles + 1, **i32), "chunk_rel": torch.empty(sizes.n_chunks, **i32),
            "chunk_cnt": torch.empty(sizes.n_chunks, **i32), "chunk_tile": torch.empty(sizes.n_chunks, **i32),
            "chunk_flags": torch.empty(sizes.n_chunks, **i32), "rel_order": torch.empty(sizes.n_units, **i32),
            "slot_src": torch.empty(sizes.n_slots, **i32), "slot_w": torch.empty(sizes.n_slots, dtype=torch.float32, device=dev),
            "slot_row": torch.empty(sizes.n_slots, **i32), "slot_acc": torch.empty(sizes.n_slots, **i32),
        }
        if int(split) == 5:      # second rows of the pairs (the tile-major weight-gradient plan)
            arr["slot_src2"] = torch.empty(max(sizes.n_chunks * 8, 1), **i32)
        ps = RgcnPlanStruct()
        for k, t in arr.items():
            setattr(ps, k, t.data_ptr())
        check(lib.rgcn_plan_build_finish(C.byref(sizes), ws.data_ptr(), ws.numel(), C.byref(ps), _stream(ws)),
              "rgcn_plan_build_finish")
    return ps, arr, int(sizes.n_edges)


def dw_tiles_geometry():
    """(tile, walkers, max relations) of the tile-major weight-gradient kernel"""
    t, w, r = C.c_int(), C.c_int(), C.c_int()
    check(load().rgcn_dw_tiles_geometry(C.byref(t), C.byref(w), C.byref(r)), "rgcn_dw_tiles_geometry")
    return t.value, w.value, r.value


def dw_tiles_walk(ps: RgcnPlanStruct, device) -> torch.Tensor:
    """walk_ptr of rgcn_bwd_dw_tiles for a plan of the tile-major geometry: int32 [num_relations, walkers + 1]"""
    walkers = dw_tiles_geometry()[1]
    out = torch.empty(int(ps.num_relations), walkers + 1, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        check(load().rgcn_dw_tiles_walk(C.byref(ps), out.data_ptr(), _stream(out)), "rgcn_dw_tiles_walk")
    return out


def bwd_dw_tiles(ps: RgcnPlanStruct, walk_ptr: torch.Tensor, x: torch.Tensor, din: int, g: torch.Tensor, dout: int,
                 d_weight: torch.Tensor, flags: int = 0) -> None:
    lib = load()
    nbytes = lib.rgcn_bwd_dw_tiles_workspace_bytes(int(ps.num_relations))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.rgcn_bwd_dw_tiles(C.byref(ps), walk_ptr.data_ptr(), x.data_ptr(), x.stride(0), din, g.data_ptr(), g.stride(0),
                                    dout, ws.data_ptr(), nbytes, d_weight.data_ptr(), int(flags), _stream(x)), "rgcn_bwd_dw_tiles")


def bwd_dw_root(x: torch.Tensor, din: int, g: torch.Tensor, dout: int, d_root: Optional[torch.Tensor],
                d_bias: Optional[torch.Tensor]) -> None:
    """d_root = x^T g, d_bias = column sums of g (rgcn_bwd_dw_root): rows of x and g pair up one to one."""
    lib = load()
    if x.shape[0] != g.shape[0]:
        raise RgcnLibraryError("rgcn_bwd_dw_root: x and g must have the same number of rows")
    with torch.cuda.device(x.device):
        ws = torch.empty(lib.rgcn_bwd_dw_root_workspace_bytes(), dtype=torch.uint8, device=x.device)
        check(lib.rgcn_bwd_dw_root(x.data_ptr(), x.stride(0), din, g.data_ptr(), g.stride(0), dout, x.shape[0],
                                   ws.data_ptr(), ws.numel(), _ptr(d_root), _ptr(d_bias), _stream(x)), "rgcn_bwd_dw_root")


# ---- edge-parallel path (eplan.EdgePlan) ----------------------------------------------------------------------------
def edge_units_struct(ep) -> RgcnEdgeUnits:
    cached = getattr(ep, "_cunits", None)
    if cached is None:
        if ep.slot_src.device.type != "cuda":
            raise RgcnLibraryError("the edge plan must live on the GPU (plan tensors are on %s)" % ep.slot_src.device)
        cached = ep._cunits = RgcnEdgeUnits(ep.n_nodes, ep.n_units, ep.num_relations, 0, ep.unit_rel.data_ptr(),
                                            ep.unit_cnt.data_ptr(), ep.slot_src.data_ptr(), ep.slot_w.data_ptr())
    return cached


def _edge_units(n_rows_gathered: int, n_units: int, num_relations: int, unit_rel, unit_cnt, slot_src, slot_w) -> RgcnEdgeUnits:
    return RgcnEdgeUnits(n_rows_gathered, n_units, num_relations, 0, unit_rel.data_ptr(), unit_cnt.data_ptr(), slot_src.data_ptr(),
                         slot_w.data_ptr())


def ep_segment_sum(src: torch.Tensor, ptr: torch.Tensor, idx, w, n_out: int, width: int, out: torch.Tensor, bias=None,
                   act: int = ACT_NONE, mask=None, final: bool = False) -> None:
    with torch.cuda.device(src.device):
        check(load().rgcn_ep_segment_sum(src.data_ptr(), src.stride(0), ptr.data_ptr(), _ptr(idx), _ptr(w), n_out, width,
                                         _ptr(bias), int(act), _ptr(mask), mask.stride(0) if mask is not None else 0, int(final),
                                         out.data_ptr(), out.stride(0), _stream(src)), "rgcn_ep_segment_sum")


def ep_aggregate_heavy(ep, x: torch.Tensor, din: int) -> Optional[torch.Tensor]:
    """H[seg] = sum_e w_e x[src_e] over the rows of every heavy (destination, relation) segment of the plan (eplan.HeavyPart):
    rgcn_ep_segment_sum over x itself, weighted, in levels.  None when the plan has no heavy part."""
    h = getattr(ep, "heavy", None)
    if h is None:
        return None
    if getattr(h, "shared", None) is not None:
        raise RgcnLibraryError("this plan's heavy segments are shared across ranks: H comes from ep_aggregate_shared + all-reduce")
    cur = x
    for ptr, idx, w, n_out in h.levels:
        dst = torch.empty(max(n_out, 1), x.stride(0), dtype=torch.float32, device=x.device)
        ep_segment_sum(cur, ptr, idx, w, n_out, din, dst)
        cur = dst
    return cur


def ep_aggregate_shared(shared, x: torch.Tensor, din: int) -> torch.Tensor:
    """this rank's share of H[seg] = sum_e w_e x[src_e] over the heavy segments of the whole graph (eplan.SharedHeavy): zeros but
    for the segments its rows belong to; the all-reduce over the ranks (conv.py) completes it"""
    hmat = torch.zeros(max(shared.n_seg, 1), x.stride(0), dtype=torch.float32, device=x.device)
    cur = x
    for li, (ptr, idx, w, n_out) in enumerate(shared.levels):
        last = li == len(shared.levels) - 1
        dst = hmat[shared.seg_lo:shared.seg_lo + n_out] if last else torch.empty(max(n_out, 1), x.stride(0), dtype=torch.float32, device=x.device)
        ep_segment_sum(cur, ptr, idx, w, n_out, din, dst)
        cur = dst
    return hmat


def ep_layer(ep, x: torch.Tensor, din: int, packed: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, dout: int,
             act: int = ACT_NONE, mask: Optional[torch.Tensor] = None, flags: int = 0, hmat: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """out[:n_owned] = act(bias + sum over the plan's rows of w * (x[src] @ W_rel)) * (mask > 0): the heavy segments' rows
    summed first (ep_aggregate_heavy), rgcn_ep_transform over the units (and over the heavy part's pseudo rows, gathered from
    H), then one rgcn_ep_segment_sum per level of the plan.  ``out``: [n_owned, ld] with ld a multiple of 4.  Returns H (the
    weight gradients of the heavy part need it) or None."""
    lib = load()
    ldz = out.stride(0)
    st = _stream(x)
    h = getattr(ep, "heavy", None)
    if hmat is None:      # (given: the all-reduced H of the heavy segments shared across ranks)
        hmat = ep_aggregate_heavy(ep, x, din)
    with torch.cuda.device(x.device):
        n_light = ep.n_units * 64
        z = torch.empty(max(n_light + (h.n_units * 64 if h is not None else 0), 1), ldz, dtype=torch.float32, device=x.device)
        check(lib.rgcn_ep_transform(C.byref(edge_units_struct(ep)), x.data_ptr(), x.stride(0), din, packed.data_ptr(),
                                    z.data_ptr(), ldz, dout, int(flags), st), "rgcn_ep_transform")
        if h is not None:
            hu = getattr(h, "_cunits", None)
            if hu is None:
                hu = h._cunits = _edge_units(h.n_seg, h.n_units, ep.num_relations, h.unit_rel, h.unit_cnt, h.slot_src, h.slot_w)
            check(lib.rgcn_ep_transform(C.byref(hu), hmat.data_ptr(), hmat.stride(0), din, packed.data_ptr(),
                                        z[n_light:].data_ptr(), ldz, dout, int(flags), st), "rgcn_ep_transform (heavy part)")
    cur = z
    for li, (ptr, idx, n_out) in enumerate(ep.levels):
        final = li == len(ep.levels) - 1
        dst = out if final else torch.empty(max(n_out, 1), ldz, dtype=torch.float32, device=x.device)
        ep_segment_sum(cur, ptr, idx, None, n_out, dout, dst, bias if final else None, act if final else ACT_NONE,
                       mask if final else None, final)
        cur = dst
    return hmat


def eplan_segments(slot_row: torch.Tensor, n_owned: int):
    """(seg_ptr int32 [n_owned + 1], seg_idx int32 [real slots]) of a relation-major plan's slots (rgcn_eplan_segments)"""
    lib = load()
    n_slots = int(slot_row.numel())
    dev = slot_row.device
    seg_ptr = torch.empty(n_owned + 1, dtype=torch.int32, device=dev)
    seg_idx = torch.empty(max(n_slots, 1), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        ws = torch.empty(lib.rgcn_plan_workspace_bytes(n_slots, 0, 1, 16), dtype=torch.uint8, device=dev)
        check(lib.rgcn_eplan_segments(slot_row.data_ptr(), n_slots, int(n_owned), ws.data_ptr(), ws.numel(), seg_ptr.data_ptr(),
                                      seg_idx.data_ptr(), _stream(slot_row)), "rgcn_eplan_segments")
    return seg_ptr, seg_idx
