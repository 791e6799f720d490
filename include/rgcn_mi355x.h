/*
 * rgcn_mi355x.h -- C ABI of librgcn_mi355x.so: the R-GCN layer hot path (forward + backward of
 * the per-relation sparse message passing  out = sum_r D_r^-1 A_r X W_r + X root + b) as
 * hand-written HIP kernels for gfx950 (MI355X).
 *
 * What each entry point replaces in the reference (paths relative to /root/reference):
 *   - the arithmetic of torch_geometric.nn.RGCNConv (torch_geometric==2.3.1, requirements.txt:7),
 *     which the reference reaches from model/layers.py:21,23 (Emb_Layers.forward), :62,64
 *     (Emb_ATT_Layers.forward) and :108,110 (Emb_MLP_Layers.forward), and differentiates through at
 *     model/modelTrainer.py:66 (output.backward()).
 * The reference has no FFI of its own (it is pure Python); INTEGRATION.md shows the ctypes stub a
 * maintainer would add.
 *
 * Conventions (SURVEY.md 8b "C ABI"):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 *   - every function returns an int status (0 = RGCN_OK, negative = argument error, positive =
 *     hipError_t of a failed launch); nothing throws;
 *   - nothing allocates: workspaces are sized by the *_bytes / *_floats queries and owned by the caller;
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream); no host synchronisation, no global mutable state -> re-entrant on distinct streams
 *     and capturable into a hipGraph.
 *   - float32 features, int32 indices.  Feature row strides (ld*) are in ELEMENTS, must be multiples
 *     of 4 (16-byte rows) and >= the feature width; columns between the width and the width rounded
 *     up to a multiple of 4 must hold zeros (the Python host pads when needed).
 *   - feature widths 1..128 per side.
 */
#ifndef RGCN_MI355X_H
#define RGCN_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGCN_ABI_VERSION 17
#define RGCN_UNIT 64 /* edge slots per unit of the weight-gradient walk (rel_order); a chunk is 1 or 2 units */
#define RGCN_CHUNK_MAX 128 /* plan->chunk is 64 or 128 edge slots (rows of one LDS ring slot of the forward / dX kernel) */
#define RGCN_MAX_WIDTH 128
#define RGCN_DW_WALKERS 2048 /* waves that walk rel_order side by side (512 workgroups x 4): the interleave of rel_order */

/* activation fused into rgcn_fwd's store (reference model/layers.py:22 F.relu, :24 activation = torch.sigmoid) */
enum rgcn_act { RGCN_ACT_NONE = 0, RGCN_ACT_RELU = 1, RGCN_ACT_SIGMOID = 2 };

/* per-call options (bit mask); 0 = let the library choose */
#define RGCN_FLAG_POINTER_GATHER 1u /* address gathered rows with 64-bit pointers even where a buffer descriptor fits */
#define RGCN_FLAG_DW_RING 2u        /* rgcn_bwd_dw: LDS-ring kernels whatever the size */
#define RGCN_FLAG_DW_DIRECT 4u      /* rgcn_bwd_dw: direct-gather kernel whenever the widths allow (64 x 64) */
#define RGCN_FLAG_DW_ROOT_ONLY 16u   /* rgcn_bwd_dw: d_root and d_bias only (the relations went to rgcn_bwd_dw_tiles) */
#define RGCN_FLAG_SPLIT_PRODUCERS 32u /* rgcn_fwd / rgcn_bwd_dx: the bf16 x 3 kernel whose PRODUCER waves split the gathered rows
                                       * (fp32-equivalent: 24 significant bits on both operands, six bf16 products): 64 x 64 layers,
                                       * 128-slot chunks, tile <= 224 (layout-1 plans: two teams of consumer waves); other shapes take the exact-fp32 kernel.
                                       * rgcn_bwd_dw_tiles: the same walk with both operands split into three bf16 pieces in registers
                                       * (six bf16 products, fp32 accumulation; same fp32-equivalence) */
#define RGCN_FLAG_EXACT_FP32 8u     /* rgcn_fwd / rgcn_bwd_dx: the exact-fp32 MFMA kernel whatever else the flags ask for */

enum rgcn_status {
    RGCN_OK = 0,
    RGCN_ERR_NULL = -1,      /* a required pointer is NULL */
    RGCN_ERR_WIDTH = -2,     /* feature width outside 1..128 */
    RGCN_ERR_STRIDE = -3,    /* a row stride is not a multiple of 4 or smaller than the width */
    RGCN_ERR_PLAN = -4,      /* inconsistent plan (sizes <= 0, tile not a multiple of 16, ...) */
    RGCN_ERR_LDS = -5,       /* plan tile too large for the 160 KiB LDS at these widths */
    RGCN_ERR_WORKSPACE = -6, /* workspace smaller than the *_workspace_bytes query */
    RGCN_ERR_DEVICE = -7,    /* current device is not gfx950 / no device */
    RGCN_ERR_ACT = -8,       /* unknown activation code */
    RGCN_ERR_GRAPH = -9,     /* rgcn_plan_build: an edge_index / edge_type value is out of range */
    RGCN_ERR_ADDRESS = -10   /* rgcn_bwd_dw_tiles: a gathered matrix cannot be addressed through a buffer descriptor (2^24 rows or
                              * 4 GiB and more): use rgcn_bwd_dw, whose kernels fall back to 64-bit pointers */
};

/* Graph plan in HBM, built once per graph (scaling_rgcn_training_amd/plan.py documents the layout;
 * PyG rebuilds the per-relation masks and counts on every forward call instead).
 * Rows scattered into are the plan's OWNED node range, numbered from 0 (= node_begin). */
typedef struct rgcn_plan {
    int32_t n_nodes;       /* rows of the gathered matrix (whole graph) */
    int32_t n_owned;       /* output rows (node_end - node_begin) */
    int32_t num_relations; /* R'; the self-loop ("root") is relation id R' */
    int32_t tile;          /* output nodes per tile (multiple of 16) */
    int32_t n_tiles;
    int32_t n_chunks;
    int32_t chunk;         /* edge slots per chunk: 64 or 128 (rows of one LDS ring slot of the forward / dX kernel) */
    int32_t n_units;       /* entries of rel_order */
    int32_t layout;        /* 0: the rows of a (tile, relation) group are dealt over all its row tiles; 1 (chunk = 128): TEAM
                            * placement -- a chunk's rows are cut at a change of destination into part A on its first
                            * ceil(nt / 2) row tiles and part B on the others, so the two parts scatter into disjoint rows
                            * (chunk_flags bit 8: they do not) and experiment builds of the forward / dX kernel of 64 x 64 layers give
                            * each part to its own team of consumer waves; same chunks and row-tile counts as layout 0;
                            * 2: no tiles -- dense relation-major units for rgcn_bwd_dw only (rgcn_edge_units);
                            * 3 (chunk = 128): layout 0 with the rows of a (destination, relation) run on ONE slot where a chunk is
                            * a whole (tile, relation) group with runs of at most 3 rows: heads on slots 0 .. H-1, second rows on
                            * row tile 7 - h / 16 (place h % 16), third rows on the row tile below those (6 or 5); chunk_cnt counts the head row tiles,
                            * chunk_flags bits 20-23 repeat every chunk's row-tile count, bits 16-17 / 18 count the row tiles of
                            * second / third rows, bit 19 "the rows of a run differ in
                            * weight" (a shadow slot's slot_acc then holds the float weight / head's weight).  Walked by rgcn_fwd / rgcn_bwd_dx with RGCN_FLAG_SPLIT_PRODUCERS on 64 x 64 layers only (the
                            * producer waves add a run's rows before they cut them: aggregate, then transform); every other
                            * entry point answers RGCN_ERR_PLAN;
                            * 5 (chunk = 64; the plan rgcn_bwd_dw_tiles walks): layout 0 with PAIRS of rows of one (destination, relation,
                            * weight) on one slot in the (tile, relation) groups of at most two chunks, see slot_src2; n_units / rel_order
                            * hold the units that are left (its rgcn_plan_build_finish synchronises the stream to count them);
                            * rgcn_bwd_dw answers RGCN_ERR_PLAN */
    int32_t chunk_rows;    /* rows a chunk may hold: = chunk, or 112 (chunk = 128: seven row tiles of rows, the eighth free for shadow
                            * rows; built by rgcn_plan_build_begin(chunk = 112)): what rgcn_tile3p_kernel's 42 KiB ring slots hold,
                            * which leaves its accumulator room for tiles up to 272.  0 is read as `chunk` */
    const int32_t* tile_ptr;   /* [n_tiles + 1] tile-major chunk ranges */
    const int32_t* chunk_rel;  /* [n_chunks] relation id, R' for root chunks */
    const int32_t* chunk_cnt;  /* [n_chunks] slots of the chunk's used 16-slot MFMA row tiles (16, 32, ... chunk);
                                * padding slots sit at the end of every row tile */
    const int32_t* chunk_tile; /* [n_chunks] */
    const int32_t* chunk_flags; /* [n_chunks] bit t: row tile t holds a repeated destination (needs the run-sum); layout 1: bit 8;
                                 * layout 3: bits 16-19, see `layout` */
    const int32_t* rel_order;  /* [n_units] the weight-gradient walk: non-empty 64-slot units (unit u = slots
                                * [64 u, 64 u + 64), chunk u / (chunk / 64)) sorted by (relation, tile) */
    const int32_t* slot_src;   /* [n_chunks * chunk] row to gather; padding = n_nodes (one past the last row) */
    const float* slot_w;       /* [n_chunks * chunk] edge weight 1/max(1,c[dst,rel]), 0 = padding */
    const int32_t* slot_row;   /* [n_chunks * chunk] row of the owned range the slot scatters into (tile * tile_size + row
                                * in tile), ascending inside a 16-slot row tile; padding = n_owned */
    const int32_t* slot_acc;   /* [n_chunks * chunk] forward run-sum metadata per slot: (position 0..15 in the 16-slot
                                * MFMA row tile of the slot ending this slot's run of equal destinations) << 24 |
                                * (accumulator row written); the row is the slot's row in the tile if the slot
                                * ends its run, else `tile` (dummy row) */
    const int32_t* slot_src2;  /* layout 5 only (else NULL): [n_chunks * 8] the SECOND gathered row of the slots 0..3 and 32..35 of
                                * every 64-slot unit (padding = n_nodes): rgcn_bwd_dw_tiles adds it to the slot's first row before
                                * the contraction -- two rows with one (destination, relation) and one weight on ONE slot */
} rgcn_plan_t;

int rgcn_abi_version(void);
const char* rgcn_status_string(int status);

/* ---- graph plan, built on the device ---------------------------------------------------------------------------
 * The COO the reference's Graph.init_graph produces and hands to every forward call (graphs/graph.py:55-69):
 * int64, unsorted, duplicate triples kept, forward / inverse edges interleaved.  edge_index[0] / [1] / edge_type
 * are rows of a TRANSPOSED [E, 3] tensor there, hence the element strides.  PyG rebuilds per-relation masks and
 * counts from it on every call; here it is laid out once. */
typedef struct rgcn_graph {
    const int64_t* src;  /* edge_index[0]: source j of edge j -> i */
    const int64_t* dst;  /* edge_index[1]: target i */
    const int64_t* type; /* edge_type, 0 .. num_relations - 1 */
    int64_t src_stride, dst_stride, type_stride; /* in elements */
    int64_t num_edges;
    int32_t num_nodes;
    int32_t num_relations;
} rgcn_graph_t;

/* What rgcn_plan_build_begin found: sizes of the arrays the caller allocates for rgcn_plan_build_finish. */
typedef struct rgcn_plan_sizes {
    int32_t n_tiles;  /* tile_ptr: n_tiles + 1 */
    int32_t n_chunks; /* chunk_rel / chunk_cnt / chunk_tile / chunk_flags */
    int32_t n_units;  /* rel_order */
    int32_t reserved;
    int64_t n_slots;  /* slot_src / slot_w / slot_row / slot_acc: n_chunks * chunk */
    int64_t n_edges;  /* edges placed (scatter node inside the owned range), before duplicate triples are merged */
    uint64_t opaque[16]; /* state handed from _begin to _finish */
} rgcn_plan_sizes_t;

/* Bytes of scratch for rgcn_edge_weights / rgcn_plan_build_* on a graph of num_edges edges whose plan owns n_owned
 * output nodes (0 on bad arguments).  The same workspace serves all three; it is dead after _finish. */
size_t rgcn_plan_workspace_bytes(int64_t num_edges, int32_t n_owned, int32_t num_relations, int32_t tile);

/* w[e] = 1 / max(1, c[dst_e, type_e]) for aggr = mean (c counts duplicates: PyG's scatter-mean normaliser), 1 for
 * aggr = sum (aggr_sum != 0); float32, in input edge order.  Shared by the forward and the transposed plan.
 * SYNCHRONISES the stream (reads a data-dependent count back); RGCN_ERR_GRAPH on out-of-range ids. */
int rgcn_edge_weights(const rgcn_graph_t* graph, int aggr_sum, float* w, void* workspace, size_t workspace_bytes, void* stream);

/* Plan of the edges scattering into nodes [node_begin, node_end) (node_begin a multiple of tile).
 * transposed = 0: forward plan (gather source rows, scatter into targets) for rgcn_fwd / rgcn_bwd_dw;
 * transposed = 1: the plan rgcn_bwd_dx runs on (gather target rows, scatter into sources), same weights w.
 * _begin sorts, merges duplicate triples and sizes the plan (SYNCHRONISES the stream: the sizes are data-dependent);
 * the caller then allocates the ten device arrays of `plan` (sizes above; nothing in this library allocates) and
 * _finish fills them and the scalar fields, asynchronously on `stream`.  tile: output nodes per tile (multiple of
 * 16), chunk: 64 or 128 -- or 112 (layouts 0 and 3): a plan of 128-slot chunks that hold at most 112 rows, see rgcn_plan.chunk_rows --,
 * layout: 0, 2 (chunk = 64) or (chunk = 128) 1 / 3, see struct rgcn_plan.  Replaces scaling_rgcn_training_amd/plan.py (torch tensor ops), which stays as the test
 * oracle: all arrays are bit-identical. */
int rgcn_plan_build_begin(const rgcn_graph_t* graph, const float* w, int transposed, int32_t node_begin, int32_t node_end,
                          int32_t tile, int32_t chunk, int32_t layout, void* workspace, size_t workspace_bytes,
                          rgcn_plan_sizes_t* sizes, void* stream);
int rgcn_plan_build_finish(const rgcn_plan_sizes_t* sizes, void* workspace, size_t workspace_bytes, rgcn_plan_t* plan,
                           void* stream);

/* Widths are padded to 16/32/64/128 inside the kernels; returns that padded value (0 if unsupported). */
int rgcn_padded_width(int width);

/* Floats of the MFMA-fragment-ordered weight pack: (R' + 1) * pad(K) * pad(N), plus for 64 x 64 layers the bf16 x 3 split
 * of the same weights ((R' + 1) * 6144 floats) that the split-precision kernel reads. */
size_t rgcn_packed_weight_floats(int num_relations, int din, int dout);

/* Pack weight[R', din, dout] (+ root[din, dout], may be NULL = zeros) into B-fragment order.
 * transpose = 0: B_r = W_r (K = din, N = dout), used by rgcn_fwd;
 * transpose = 1: B_r = W_r^T (K = dout, N = din), used by rgcn_bwd_dx.
 * Replaces nothing in PyG (it indexes weight[i] directly); cost O(R' * din * dout) per call. */
int rgcn_pack_weights(const float* weight, const float* root, int num_relations, int din, int dout,
                      int transpose, float* packed, void* stream);
/* The same pack for PyG's two weight decompositions (BASELINE.json configs[2]: num_bases = 30),
 * composed inside the packer -- [R', din, dout] is never materialised:
 *   basis: W_r = sum_b comp[r, b] * bases[b], bases [B, din, dout], comp [R', B]  (torch: comp @ weight.view(B, -1));
 *   block: W_r = blockdiag(blocks[r, 0 .. nb - 1]), blocks [R', nb, din / nb, dout / nb]. */
int rgcn_pack_weights_basis(const float* bases, const float* comp, const float* root, int num_relations, int num_bases,
                            int din, int dout, int transpose, float* packed, void* stream);
int rgcn_pack_weights_block(const float* blocks, const float* root, int num_relations, int num_blocks, int din, int dout,
                            int transpose, float* packed, void* stream);
/* Gradients of the decomposition's parameters from the dense d_w [R', din, dout] that rgcn_bwd_dw / rgcn_bwd_dw_tiles
 * produce (a scratch buffer, not an autograd tensor): d_bases[b] = sum_r comp[r, b] d_w[r], d_comp[r, b] = <d_w[r], bases[b]>
 * (either may be NULL); d_blocks = the diagonal blocks of d_w.  Fixed summation orders: bit-reproducible. */
int rgcn_basis_backward(const float* d_w, const float* bases, const float* comp, int num_relations, int num_bases, int din,
                        int dout, float* d_bases, float* d_comp, void* stream);
int rgcn_block_backward(const float* d_w, int num_relations, int num_blocks, int din, int dout, float* d_blocks, void* stream);

/* Forward of RGCNConv.forward (aggr mean/sum folded into the plan's edge weights):
 *   out[i, :] = act(bias + sum_{slots scattering into i} w_e * x[src_e, :] @ W_{rel_e})   (root = rel R')
 * x: [plan->n_nodes, ldx]; out: [plan->n_owned, ldo]; packed_w from rgcn_pack_weights(transpose=0);
 * bias: [dout] or NULL.  Columns dout..roundup4(dout) of out are written as zeros.
 * act: RGCN_ACT_* applied in the tile store -- what model/layers.py:22 (F.relu) and :24 (activation) run as
 * separate elementwise kernels over [N, out]. */
int rgcn_fwd(const rgcn_plan_t* plan, const float* x, int ldx, int din, const float* packed_w,
             const float* bias, float* out, int ldo, int dout, int act, unsigned flags, void* stream);

/* dX of the layer (autograd of index_select/scatter-mean/matmul in PyG's loop), atomics-free:
 *   dx[j, :] = sum_{edges j->i, r} w_e * g[i, :] @ W_r^T + g[j, :] @ root^T
 * `plan_t` is the TRANSPOSED plan (edges grouped by source); g: [plan_t->n_nodes, ldg] upstream
 * gradient; packed_wt from rgcn_pack_weights(transpose=1); dx: [plan_t->n_owned, lddx].
 * relu_of (NULL or [plan_t->n_owned, ldr], the rows of the layer INPUT that dx belongs to): when the input was
 * produced by a ReLU (the previous layer's fused RGCN_ACT_RELU), dx is stored as dx * (relu_of > 0), i.e. the
 * gradient w.r.t. the previous layer's pre-activation: autograd's relu backward never runs as a kernel. */
int rgcn_bwd_dx(const rgcn_plan_t* plan_t, const float* g, int ldg, int dout, const float* packed_wt,
                float* dx, int lddx, int din, const float* relu_of, int ldr, unsigned flags, void* stream);

/* dz = da * act'(a) for an output a = act(z) of rgcn_fwd: relu -> da * (a > 0), sigmoid -> da * a * (1 - a).
 * a, da, dz: [rows, ld] (dz may alias da).  For layers whose consumer cannot fold the mask (rgcn_bwd_dx relu_of). */
int rgcn_act_backward(const float* a, const float* da, float* dz, long rows, int ld, int act, void* stream);

/* Weight gradients: d_weight[r] = H_r^T g, d_root = X^T g, d_bias = column sums of g, over the
 * plan's owned rows (g: [plan->n_owned, ldg] is the upstream gradient of those rows).
 * Any of d_weight / d_root / d_bias may be NULL (frozen parameter, model/layers.py:33-46).
 * Deterministic: per-workgroup partial slabs in `workspace` are summed in a fixed order. */
size_t rgcn_bwd_dw_workspace_bytes(const rgcn_plan_t* plan, int din, int dout);
int rgcn_bwd_dw(const rgcn_plan_t* plan, const float* x, int ldx, int din, const float* g, int ldg,
                int dout, void* workspace, size_t workspace_bytes, float* d_weight, float* d_root,
                float* d_bias, unsigned flags, void* stream);

/* d_weight alone, tile-major (64 x 64 layers with at most 32 relations on large graphs): every wave owns one relation and
 * keeps its 64 x 64 accumulator in registers for the whole launch, the upstream-gradient rows of a tile are staged in LDS
 * once per relation quarter instead of being gathered per edge (37 GB instead of 55 GB moved at the headline config).
 * `plan`: a FORWARD-direction plan built with the geometry rgcn_dw_tiles_geometry reports (tile = 320, chunk = 64, layout
 * 0); walk_ptr: int32 [num_relations][walkers + 1], walk_ptr[r][p] = first position in plan->rel_order of relation r
 * whose tile is >= p * n_tiles / walkers (integer division), walk_ptr[r][walkers] = end of relation r; filled by
 * rgcn_dw_tiles_walk (once per plan; walk_ptr: device memory, num_relations * (walkers + 1) int32).
 * d_root / d_bias: rgcn_bwd_dw(..., RGCN_FLAG_DW_ROOT_ONLY) on any forward plan of the same graph. */
int rgcn_dw_tiles_geometry(int* tile, int* walkers, int* max_relations);
int rgcn_dw_tiles_walk(const rgcn_plan_t* plan, int32_t* walk_ptr, void* stream);
size_t rgcn_bwd_dw_tiles_workspace_bytes(int num_relations);
int rgcn_bwd_dw_tiles(const rgcn_plan_t* plan, const int32_t* walk_ptr, const float* x, int ldx, int din, const float* g,
                      int ldg, int dout, void* workspace, size_t workspace_bytes, float* d_weight, unsigned flags,
                      void* stream);

/* d_root = x^T g and d_bias = column sums of g over rows [0, rows) of the two matrices -- the self-loop ("root") part of
 * autograd's backward of RGCNConv (reference model/modelTrainer.py:66), which needs no plan: the rows the root relation
 * "gathers" are the nodes' own.  Widths up to 64 per side (RGCN_ERR_WIDTH beyond: use rgcn_bwd_dw with
 * RGCN_FLAG_DW_ROOT_ONLY).  A streaming kernel without LDS whose workgroups fit a CU next to rgcn_bwd_dx's: enqueue it on a
 * second stream beside the dX launch.  d_root or d_bias may be NULL (not both).  Bit-reproducible. */
size_t rgcn_bwd_dw_root_workspace_bytes(void);
int rgcn_bwd_dw_root(const float* x, int ldx, int din, const float* g, int ldg, int dout, long rows, void* workspace,
                     size_t workspace_bytes, float* d_root, float* d_bias, void* stream);

/* ---- edge-parallel path (scaling_rgcn_training_amd/eplan.py) ---------------------------------------------------
 * The same layer arithmetic for graphs on which the tile-major plan is the wrong shape -- the reference's own datasets
 * (model/modelTrainer.py:78,92: 45 .. ~267 relation ids, hubs of in-degree 10^4 on 8k nodes): rows (merged edges + one
 * root pseudo edge per node) sorted RELATION-MAJOR and packed into dense 64-slot units; rgcn_ep_transform writes
 * Z[slot] = w_slot * (x[src_slot] @ W_rel) for every slot, rgcn_ep_segment_sum adds the rows of every destination in a
 * fixed order and applies bias / activation / ReLU mask.  dX: the same two calls on the transposed units with W^T.
 * The units double as a dense relation-major walk for rgcn_bwd_dw (a rgcn_plan_t with layout = 2, chunk = 64, rel_order
 * = 0 .. n_units - 1, chunk_rel / chunk_cnt = unit_rel / unit_cnt: rgcn_fwd / rgcn_bwd_dx refuse it). */
typedef struct rgcn_edge_units {
    int32_t n_nodes;           /* rows of the gathered matrix (padding slots gather row n_nodes -> zeros) */
    int32_t n_units;           /* 64-slot units */
    int32_t num_relations;     /* R' (the root pseudo relation is id R') */
    int32_t reserved;
    const int32_t* unit_rel;   /* [n_units] relation of the unit, ascending */
    const int32_t* unit_cnt;   /* [n_units] used slots rounded up to 16 (whole MFMA row tiles): 16 .. 64 */
    const int32_t* slot_src;   /* [n_units * 64] */
    const float* slot_w;       /* [n_units * 64] edge weight, 0 = padding */
} rgcn_edge_units_t;

/* The unit arrays come from the plan builder itself: rgcn_plan_build_begin / _finish with layout = 2, chunk = 64 (tile is
 * ignored) lay the owned range out as ONE tile, i.e. relation-major: chunk_rel / chunk_cnt are unit_rel / unit_cnt, slot_src /
 * slot_w the slots, slot_row the destination row of every slot (n_owned = padding), rel_order = 0 .. n_units - 1 -- the same
 * rgcn_plan_t is what rgcn_bwd_dw walks.  rgcn_eplan_segments then sorts the slots by destination: seg_idx [n_slots] (its first
 * seg_ptr[n_owned] entries are the real slots ordered by (destination row, slot)), seg_ptr [n_owned + 1].  Workspace:
 * rgcn_plan_workspace_bytes(n_slots, 0, 1, 16).  Destinations with more than a few hundred rows are summed in levels: cut
 * [seg_ptr[i], seg_ptr[i + 1]) into pieces of at most P rows (scaling_rgcn_training_amd/eplan.py segment_levels, P = 256). */
int rgcn_eplan_segments(const int32_t* slot_row, int64_t n_slots, int32_t n_owned, void* workspace, size_t workspace_bytes,
                        int32_t* seg_ptr, int32_t* seg_idx, void* stream);
/* z: [n_units * 64, ldz] (rows of unused row tiles are left untouched); packed_w: rgcn_pack_weights(..., transpose).
 * flags & RGCN_FLAG_SPLIT_PRODUCERS: 64 x 64 layers multiply on bf16 MFMAs over three-way split operands (fp32-equivalent, as
 * rgcn_fwd under the same flag): the exact-fp32 MFMA rate binds the transform at that width. */
int rgcn_ep_transform(const rgcn_edge_units_t* units, const float* x, int ldx, int din, const float* packed_w, float* z,
                      int ldz, int dout, unsigned flags, void* stream);
/* out[i] = sum of rows seg_idx[q] (q itself when seg_idx is NULL) of `in` -- times seg_w[q] when seg_w is given -- for q in
 * [seg_ptr[i], seg_ptr[i + 1]), i < n_out, added in index order; final_level != 0: + bias (may be NULL), activation (RGCN_ACT_*), then out *= (mask > 0) when mask
 * is given (rows of the layer input when it is a ReLU output, as rgcn_bwd_dx's relu_of).  Long segments are summed in
 * levels: pieces first (final_level = 0, seg_idx of the first level only), the pieces of a segment last. */
int rgcn_ep_segment_sum(const float* in, int ldin, const int32_t* seg_ptr, const int32_t* seg_idx, const float* seg_w, int n_out,
                        int width, const float* bias, int act, const float* mask, int ldm, int final_level, float* out, int ldo,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RGCN_MI355X_H */
