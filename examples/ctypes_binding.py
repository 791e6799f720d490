"""The whole reference-side binding of librgcn_mi355x.so (ABI v17) in one file: ctypes + torch tensors as device memory,
nothing imported from this repository's Python package.  This is what a maintainer of the reference would drop next to
model/layers.py to replace ``torch_geometric.nn.RGCNConv`` (model/layers.py:7, call sites :21,23) and what autograd
derives from it (model/modelTrainer.py:66) without taking the package; ``scaling_rgcn_training_amd/_lib.py`` + ``conv.py``
are the maintained, cached, multi-GPU version of the same calls.  Checked against the oracle by
tests/test_gpu_binding_example.py.

    layer = RGCNLayer("scaling_rgcn_training_amd/librgcn_mi355x.so", edge_index, edge_type, num_nodes, num_relations)
    out = layer(x, weight, root, bias)        # x [N, in] fp32 on the MI355X; weight [R, in, out]; differentiable
"""
import ctypes as C

import torch

vp, i32, i64, u32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_size_t
PLAN_ARRAYS = ("tile_ptr", "chunk_rel", "chunk_cnt", "chunk_tile", "chunk_flags", "rel_order",
               "slot_src", "slot_w", "slot_row", "slot_acc")


class rgcn_plan(C.Structure):           # struct rgcn_plan, include/rgcn_mi355x.h
    _fields_ = [(k, i32) for k in ("n_nodes", "n_owned", "num_relations", "tile", "n_tiles", "n_chunks", "chunk", "n_units",
                                   "layout", "chunk_rows")] + [(k, vp) for k in PLAN_ARRAYS] + [("slot_src2", vp)]      # (layout 5 only: NULL here)


class rgcn_graph(C.Structure):          # struct rgcn_graph: the COO tensors of graphs/graph.py:55-69, as they are
    _fields_ = [("src", vp), ("dst", vp), ("type", vp), ("src_stride", i64), ("dst_stride", i64), ("type_stride", i64),
                ("num_edges", i64), ("num_nodes", i32), ("num_relations", i32)]


class rgcn_plan_sizes(C.Structure):     # struct rgcn_plan_sizes
    _fields_ = [("n_tiles", i32), ("n_chunks", i32), ("n_units", i32), ("reserved", i32), ("n_slots", i64), ("n_edges", i64),
                ("opaque", C.c_uint64 * 16)]


def load(path):
    lib = C.CDLL(path)
    P, G, S = C.POINTER(rgcn_plan), C.POINTER(rgcn_graph), C.POINTER(rgcn_plan_sizes)
    sig = {
        "rgcn_abi_version": (i32, []),
        "rgcn_status_string": (C.c_char_p, [i32]),
        "rgcn_plan_workspace_bytes": (sz, [i64, i32, i32, i32]),
        "rgcn_edge_weights": (i32, [G, i32, vp, vp, sz, vp]),
        "rgcn_plan_build_begin": (i32, [G, vp, i32, i32, i32, i32, i32, i32, vp, sz, S, vp]),
        "rgcn_plan_build_finish": (i32, [S, vp, sz, P, vp]),
        "rgcn_packed_weight_floats": (sz, [i32, i32, i32]),
        "rgcn_pack_weights": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
        "rgcn_fwd": (i32, [P, vp, i32, i32, vp, vp, vp, i32, i32, i32, u32, vp]),
        "rgcn_bwd_dx": (i32, [P, vp, i32, i32, vp, vp, i32, i32, vp, i32, u32, vp]),
        "rgcn_bwd_dw_workspace_bytes": (sz, [P, i32, i32]),
        "rgcn_bwd_dw": (i32, [P, vp, i32, i32, vp, i32, i32, vp, sz, vp, vp, vp, u32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    assert lib.rgcn_abi_version() == 17
    return lib


def _ok(lib, status, what):
    if status != 0:
        raise RuntimeError(f"{what}: {lib.rgcn_status_string(status).decode() if status < 0 else 'hipError %d' % status}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _pad4(t):
    """fp32 rows with a 16-byte-aligned stride: the one layout rule of the ABI (a [N, 63] tensor gets one zero column)"""
    t = t.float()
    if t.shape[1] % 4 == 0 and t.is_contiguous():
        return t
    out = t.new_zeros(t.shape[0], (t.shape[1] + 3) // 4 * 4)
    out[:, :t.shape[1]] = t
    return out


class Plan:
    """one direction of the graph plan: the struct the kernels take + the tensors that own its arrays"""

    def __init__(self, lib, graph, w, transposed, num_nodes, tile, chunk, ws):
        sizes = rgcn_plan_sizes()
        _ok(lib, lib.rgcn_plan_build_begin(C.byref(graph), w.data_ptr(), int(transposed), 0, num_nodes, tile, chunk, 0,
                                           ws.data_ptr(), ws.numel(), C.byref(sizes), _stream()), "rgcn_plan_build_begin")
        n = {"tile_ptr": sizes.n_tiles + 1, "rel_order": sizes.n_units}
        self.arrays = {k: torch.empty(n.get(k, sizes.n_slots if k.startswith("slot") else sizes.n_chunks),
                                      dtype=torch.float32 if k == "slot_w" else torch.int32, device=ws.device) for k in PLAN_ARRAYS}
        self.struct = rgcn_plan()
        for k, t in self.arrays.items():
            setattr(self.struct, k, t.data_ptr())
        _ok(lib, lib.rgcn_plan_build_finish(C.byref(sizes), ws.data_ptr(), ws.numel(), C.byref(self.struct), _stream()),
            "rgcn_plan_build_finish")


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, root, bias, layer):
        lib, (R, din, dout) = layer.lib, weight.shape
        xp = _pad4(x)
        packed = torch.empty(lib.rgcn_packed_weight_floats(R, din, dout), device=x.device)
        _ok(lib, lib.rgcn_pack_weights(weight.data_ptr(), root.data_ptr(), R, din, dout, 0, packed.data_ptr(), _stream()), "pack")
        out = torch.empty(x.shape[0], (dout + 3) // 4 * 4, device=x.device)
        _ok(lib, lib.rgcn_fwd(C.byref(layer.fwd.struct), xp.data_ptr(), xp.stride(0), din, packed.data_ptr(), bias.data_ptr(),
                              out.data_ptr(), out.stride(0), dout, 0, 0, _stream()), "rgcn_fwd")
        ctx.save_for_backward(xp, weight, root)
        ctx.layer = layer
        return out[:, :dout]

    @staticmethod
    def backward(ctx, g):
        xp, weight, root = ctx.saved_tensors
        layer, lib, (R, din, dout) = ctx.layer, ctx.layer.lib, weight.shape
        gp = _pad4(g)
        packed_t = torch.empty(lib.rgcn_packed_weight_floats(R, dout, din), device=g.device)
        _ok(lib, lib.rgcn_pack_weights(weight.data_ptr(), root.data_ptr(), R, din, dout, 1, packed_t.data_ptr(), _stream()), "pack")
        dx = torch.empty(xp.shape[0], (din + 3) // 4 * 4, device=g.device)
        # dX = the forward kernel on the TRANSPOSED plan with W_r^T: no float atomics, bit-reproducible
        _ok(lib, lib.rgcn_bwd_dx(C.byref(layer.bwd.struct), gp.data_ptr(), gp.stride(0), dout, packed_t.data_ptr(),
                                 dx.data_ptr(), dx.stride(0), din, None, 0, 0, _stream()), "rgcn_bwd_dx")
        dw, dr, db = torch.empty_like(weight), torch.empty_like(root), torch.empty(dout, device=g.device)
        nbytes = lib.rgcn_bwd_dw_workspace_bytes(C.byref(layer.fwd.struct), din, dout)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
        _ok(lib, lib.rgcn_bwd_dw(C.byref(layer.fwd.struct), xp.data_ptr(), xp.stride(0), din, gp.data_ptr(), gp.stride(0), dout,
                                 ws.data_ptr(), nbytes, dw.data_ptr(), dr.data_ptr(), db.data_ptr(), 0, _stream()), "rgcn_bwd_dw")
        return dx[:, :din], dw, dr, db, None


class RGCNLayer:
    """R-GCN layer with mean aggregation over one fixed graph: out = sum_r D_r^-1 A_r x W_r + x root + bias"""

    def __init__(self, lib_path, edge_index, edge_type, num_nodes, num_relations, tile=352, chunk=128):
        self.lib = lib = load(lib_path)
        ei, et = edge_index.long(), edge_type.long()          # int64 device tensors; strided views are fine
        e = int(et.shape[0])
        graph = rgcn_graph(ei[0].data_ptr(), ei[1].data_ptr(), et.data_ptr(), ei[0].stride(0), ei[1].stride(0), et.stride(0),
                           e, num_nodes, num_relations)
        ws = torch.empty(lib.rgcn_plan_workspace_bytes(e, num_nodes, num_relations, tile), dtype=torch.uint8, device=et.device)
        w = torch.empty(max(e, 1), device=et.device)          # 1 / c[dst, rel]: the mean normaliser, once per graph
        _ok(lib, lib.rgcn_edge_weights(C.byref(graph), 0, w.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "rgcn_edge_weights")
        self.fwd = Plan(lib, graph, w, False, num_nodes, tile, chunk, ws)     # edges grouped by destination
        self.bwd = Plan(lib, graph, w, True, num_nodes, tile, chunk, ws)      # ... by source: the dX launch
        torch.cuda.current_stream().synchronize()             # ei / et / w / ws may go once the builder has run

    def __call__(self, x, weight, root, bias):
        return _Fn.apply(x, weight, root, bias, self)
