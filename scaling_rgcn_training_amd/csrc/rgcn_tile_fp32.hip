// rgcn_tile_fp32.hip -- forward / dX of the R-GCN layer in exact fp32 (v_mfma_f32_16x16x4_f32) for every width class, and
// the entry points rgcn_fwd / rgcn_bwd_dx of include/rgcn_mi355x.h (which hand 64 x 64 layers to rgcn_tile3p.hip).
//
// Replaces the arithmetic of torch_geometric.nn.RGCNConv (PyG 2.3.1) that the reference calls at
// /root/reference/model/layers.py:21,23,62,64,108,110 and differentiates at model/modelTrainer.py:66.
//   rgcn_tile_kernel      forward AND dX: per output tile, out_tile(LDS) += w_e * (x[src_e] @ B_rel)
//                         (tile-major walk of the plan; dX = same kernel on the transposed plan with W^T)
#include "rgcn_kernels_shared.h"

namespace rgcn {

// the kernel's instantiations live in two translation units (rgcn_tile_fp32_kernel.h)
int dispatch_tile_narrow(int KP, int NP, const TileArgs& a, int n_tiles, int chunk, void* stream);      // gathered width 16 / 32
int dispatch_tile_wide(int KP, int NP, const TileArgs& a, int n_tiles, int chunk, void* stream);        // gathered width 64 / 128

static int dispatch_tile(int KP, int NP, const TileArgs& a, int n_tiles, int chunk, hipStream_t s) {
    return KP <= 32 ? dispatch_tile_narrow(KP, NP, a, n_tiles, chunk, s) : dispatch_tile_wide(KP, NP, a, n_tiles, chunk, s);
}

#ifdef RGCN_DEBUG_KNOBS
static std::atomic<int> g_debug_mode{0};
#endif

// shared by rgcn_fwd and rgcn_bwd_dx: gather rows of `x` (width kin), scatter into `out` (width nout)
static int run_tile(const rgcn_plan_t* plan, const float* x, int ldx, int kin, const float* packed, const float* bias,
                    float* out, int ldo, int nout, int act, const float* mask, int ldm, unsigned flags, void* stream) {
    int st = check_plan(plan);
    if (st != RGCN_OK) return st;
    if (plan->layout == 2) return RGCN_ERR_PLAN;        // relation-major units: nothing tile-major to walk (rgcn_ep_*)
    if (!x || !packed || !out) return RGCN_ERR_NULL;
    if ((st = check_stride(ldx, kin)) != RGCN_OK) return st;
    if ((st = check_stride(ldo, nout)) != RGCN_OK) return st;
    if (mask != nullptr && (st = check_stride(ldm, nout)) != RGCN_OK) return st;
    if (act != RGCN_ACT_NONE && act != RGCN_ACT_RELU && act != RGCN_ACT_SIGMOID) return RGCN_ERR_ACT;
    if ((st = check_device()) != RGCN_OK) return st;
    TileArgs a;
    a.tile_ptr = plan->tile_ptr;
    a.chunk_rel = plan->chunk_rel;
    a.chunk_cnt = plan->chunk_cnt;
    a.chunk_flags = plan->chunk_flags;
    a.slot_src = plan->slot_src;
    a.slot_w = plan->slot_w;
    a.slot_acc = plan->slot_acc;
    a.x = x;
    a.wp = packed;
    a.bias = bias;
    a.out = out;
    a.ldx = ldx;
    a.x_bytes = buffer_bytes(plan->n_nodes, ldx, flags);
    a.n_rows = plan->n_nodes;
    a.din4 = (kin + 3) / 4;
    a.dout = nout;
    a.ldo = ldo;
    a.tile = plan->tile;
    a.n_owned = plan->n_owned;
    a.mask = mask;
    a.ldm = ldm;
    a.act = act;
    a.n_tiles = plan->n_tiles;
    a.n_chunks = plan->n_chunks;
    a.merged = plan->layout == 3 ? 1 : 0;
    a.tiles_per_wg = tiles_per_workgroup(plan->n_tiles);
#ifdef RGCN_TPW_ENV        // experiment build only (tools/debug/tpw_sweep.py)
    if (const char* e = getenv("RGCN_TPW")) a.tiles_per_wg = atoi(e) > 0 ? atoi(e) : a.tiles_per_wg;
#endif
#ifdef RGCN_DEBUG_KNOBS
    a.dbg = g_debug_mode.load();
#else
    a.dbg = 0;
#endif
    const int KP = padded_width(kin), NP = padded_width(nout);
    // Producer-split bf16 x 3 kernel (rgcn_tile3p.hip): 64 x 64 layers, 128-slot chunks, tiles that leave room for its 48 KiB
    // ring slots (layout-1 plans: two consumer teams); anything else falls through to the kernels below
    if ((flags & RGCN_FLAG_SPLIT_PRODUCERS) && !(flags & RGCN_FLAG_EXACT_FP32) && KP == 64 && NP == 64 && plan->chunk == 128 &&
        a.x_bytes != 0) {
        TileArgs b = a;
        b.wp = packed + (size_t)(plan->num_relations + 1) * KP * NP;
        const int st3 = launch_tile3p(b, plan->n_tiles, plan->layout, plan->chunk_rows, stream);
        if (st3 != RGCN_ERR_LDS || plan->layout == 3) return st3;
    }
    // layout 3 (runs of equal (destination, relation) on ONE slot, their other rows in shadow row tiles): besides the kernel above
    // the exact-fp32 kernel of 64 x 64 layers adds the shadows (its producers: rgcn_tile_fp32_kernel.h), through buffer descriptors only
    if (plan->layout == 3 && !(KP == 64 && NP == 64 && plan->chunk == 128 && a.x_bytes != 0)) return RGCN_ERR_PLAN;
    return dispatch_tile(KP, NP, a, plan->n_tiles, plan->chunk, (hipStream_t)stream);
}

}  // namespace rgcn

using namespace rgcn;

#ifdef RGCN_DEBUG_KNOBS
extern "C" void rgcn_debug_set_mode(int mode) { rgcn::g_debug_mode.store(mode); }
#endif

extern "C" int rgcn_fwd(const rgcn_plan_t* plan, const float* x, int ldx, int din, const float* packed_w,
                        const float* bias, float* out, int ldo, int dout, int act, unsigned flags, void* stream) {
    return run_tile(plan, x, ldx, din, packed_w, bias, out, ldo, dout, act, nullptr, 0, flags, stream);
}

extern "C" int rgcn_bwd_dx(const rgcn_plan_t* plan_t, const float* g, int ldg, int dout, const float* packed_wt,
                           float* dx, int lddx, int din, const float* relu_of, int ldr, unsigned flags, void* stream) {
    return run_tile(plan_t, g, ldg, dout, packed_wt, nullptr, dx, lddx, din, RGCN_ACT_NONE, relu_of, ldr, flags, stream);
}
