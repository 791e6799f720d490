// rgcn_tile_common.h -- what the forward / dX kernels of the library share: launch arguments, the accumulator tile's row
// stride and the tile epilogue (store + fused activation / ReLU mask).  Included by rgcn_tile_fp32_kernel.h (exact-fp32 kernel,
// consumer-split bf16 kernel) and rgcn_tile3p.hip (producer-split bf16 kernel).
#pragma once
#include "rgcn_common.h"

namespace rgcn {

struct TileArgs {
    const int* tile_ptr;
    const int* chunk_rel;
    const int* chunk_cnt;
    const int* chunk_flags;
    const int* slot_src;
    const float* slot_w;
    const int* slot_acc;
    const float* x;
    const float* wp;
    const float* bias;
    float* out;
    unsigned x_bytes;  // rows * ldx * 4 when buffer-descriptor gathers are possible, else 0
    int n_rows;        // rows of x (padding slots carry this index)
    int ldx, din4, dout, ldo, tile, n_owned;
    const float* mask;  // dX only: rows of the layer INPUT when that input is a ReLU output (dx *= mask > 0), or NULL
    int ldm;
    int act;            // forward only: RGCN_ACT_* applied in the tile store
    int dbg;            // diagnostic builds only (RGCN_DBG)
    int n_tiles;        // tiles of the plan
    int n_chunks;       // chunks of the plan (bounds of the scalar buffer loads of the per-chunk arrays)
    int merged;         // the plan is layout 3 (runs of equal (destination, relation) compacted: rgcn_plan.hip compact_runs_kernel)
    int tiles_per_wg;   // consecutive tiles one workgroup walks (rgcn_tile_kernel): the next tile's first gathers are in
                        // flight while the finished tile is stored
};

// accumulator row stride of the tile kernel's LDS tile (floats)
template <int NP>
constexpr int kAccStride = NP + 4;

constexpr int kTileConsumerWaves = 4;

// ---- epilogue of the forward / dX kernels: the finished tile, whole 16-byte pieces, coalesced ------------------------
// REINIT: every element is also reset to the bias for the NEXT tile of a workgroup that walks several tiles (each thread
// resets exactly the elements it has just read; columns beyond the width never leave zero, the dummy row is never stored).
// tid / nthreads: the calling threads' rank and count (all 512, or the 256 consumer threads between two tiles).
// The accumulator tile of a workgroup's first tile: every row = the bias (columns beyond the width and the pad: zero), row
// `tile` (the dummy row of padding slots) included.  A thread keeps ONE 16-byte column piece and walks rows: one bias load per
// thread where the per-element form had one per element (a dependent global load in each of its ~36 iterations: ~10 us at
// the start of every workgroup, a fifth of a launch on graphs of a few hundred tiles).
template <int LDO>
__device__ __forceinline__ void tile_init(const TileArgs& a, float* out_lds, int tid, int nthreads) {
    static_assert(LDO % 4 == 0, "16-byte pieces");
    constexpr int P = LDO / 4;                 // pieces per row
    const int per = nthreads / P;              // rows written per pass
    if (tid >= per * P) return;
    const int c4 = tid % P, r0 = tid / P;
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) {
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = c4 * 4 + c < a.dout ? a.bias[c4 * 4 + c] : 0.f;
    }
    float* p = out_lds + r0 * LDO + c4 * 4;
    for (int r = r0; r <= a.tile; r += per, p += per * LDO) *(f32x4*)p = b;
}

struct EpiRows {
    float* lp;            // this thread's piece of the first row in the LDS tile
    float* gp;            // ... in the output
    const float* mp;      // ... in the mask rows, or NULL
    int lstep;            // floats between two rows of this thread, LDS
    size_t gstep, mstep;  // ... output, mask
    int n;                // rows of this thread
    f32x4 bfix;           // what the LDS piece is reset to (REINIT)
};

// one thread's rows of the tile store: four pieces in flight (four LDS reads, then four stores), no index arithmetic
template <bool REINIT, int ACT, bool MASK>
__device__ __forceinline__ void epilogue_rows(EpiRows e) {
    auto finish = [&](f32x4 v, f32x4 m) {
        if constexpr (ACT == RGCN_ACT_RELU) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = v[c] > 0.f ? v[c] : 0.f;
        }
        if constexpr (MASK) {
            // v = m > 0 ? v : 0 through VCC: sixteen compares in flight would want sixteen SGPR pairs, and the kernels that
            // inline this are at their SGPR limit (spills into VGPR lanes inside their main loops otherwise)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float t = v[c];
                asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(t) : "v"(m[c]) : "vcc");
                v[c] = t;
            }
        }
        return v;
    };
    int k = 0;
    for (; k + 4 <= e.n; k += 4) {
        f32x4 v[4], m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = *(const f32x4*)(e.lp + u * e.lstep);
            if constexpr (MASK) m[u] = *(const f32x4*)(e.mp + u * e.mstep);
            else m[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (REINIT) *(f32x4*)(e.lp + u * e.lstep) = e.bfix;
            *(f32x4*)(e.gp + u * e.gstep) = finish(v[u], m[u]);
        }
        e.lp += 4 * e.lstep;
        e.gp += 4 * e.gstep;
        if constexpr (MASK) e.mp += 4 * e.mstep;
    }
    for (; k < e.n; ++k) {
        f32x4 v = *(const f32x4*)e.lp, m = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MASK) m = *(const f32x4*)e.mp;
        if constexpr (REINIT) *(f32x4*)e.lp = e.bfix;
        *(f32x4*)e.gp = finish(v, m);
        e.lp += e.lstep;
        e.gp += e.gstep;
        if constexpr (MASK) e.mp += e.mstep;
    }
}

template <int LDO, bool REINIT>
__device__ __forceinline__ void tile_epilogue(const TileArgs& a, float* out_lds, int tile, int tid, int nthreads) {
    const int row0 = tile * a.tile;
    const int rows = min(a.tile, a.n_owned - row0);
    const int o4 = (a.dout + 3) >> 2;
    // Fused epilogues (reference model/layers.py:22,24: F.relu / activation applied to the layer output): the
    // activation costs nothing here, as a separate kernel it re-reads and re-writes [N, out].  In the dX launch of the
    // NEXT layer the ReLU backward of this layer's output is the mask (input > 0) on the stored gradient rows.
    // Padding columns (dout .. 4 * o4) stay zero: relu(0) = 0, and sigmoid is applied to real columns only.
    const int act = a.act;
    auto bias4 = [&](int c4) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr) {
#pragma unroll
            for (int c = 0; c < 4; ++c) b[c] = c4 * 4 + c < a.dout ? a.bias[c4 * 4 + c] : 0.f;
        }
        return b;
    };
    // A thread meets one column group only when the thread count is a multiple of the groups per row (every width whose
    // o4 is a power of two: 64 -> o4 = 16): its bias is loaded once and its three addresses advance by constants -- one
    // division per tile instead of one per 16-byte piece.  Round 4: the per-piece index arithmetic made the store of a tile
    // VALU-bound (~8,000 cycles of a tile's ~115,000 at the headline config, profiles/r04w_*); this loop is bound by the LDS
    // and the store path.
    if ((nthreads % o4) == 0 && act != RGCN_ACT_SIGMOID) {      // (sigmoid: the output layer's one launch takes the loop below)
        const int c4 = tid % o4, r0 = tid / o4, rstep = nthreads / o4;
        f32x4 bfix = {0.f, 0.f, 0.f, 0.f};
        if constexpr (REINIT) bfix = bias4(c4);
        EpiRows e;
        e.lp = out_lds + r0 * LDO + c4 * 4;
        e.gp = a.out + (size_t)(row0 + r0) * a.ldo + c4 * 4;
        e.mp = a.mask != nullptr ? a.mask + (size_t)(row0 + r0) * a.ldm + c4 * 4 : nullptr;
        e.lstep = rstep * LDO;
        e.gstep = (size_t)rstep * a.ldo;
        e.mstep = (size_t)rstep * a.ldm;
        e.n = rows > r0 ? (rows - r0 + rstep - 1) / rstep : 0;
        e.bfix = bfix;
        // the activation and the mask are launch constants: one straight-line loop per combination that occurs
        if (a.mask != nullptr) {
            if (act == RGCN_ACT_NONE) epilogue_rows<REINIT, RGCN_ACT_NONE, true>(e);
            else epilogue_rows<REINIT, RGCN_ACT_RELU, true>(e);
        } else {
            if (act == RGCN_ACT_NONE) epilogue_rows<REINIT, RGCN_ACT_NONE, false>(e);
            else epilogue_rows<REINIT, RGCN_ACT_RELU, false>(e);
        }
        return;
    }
    // any other width (o4 = 3, 5, ...: the narrow / wide kernels): a thread's pieces change columns from row to row
    for (int i = tid; i < rows * o4; i += nthreads) {
        const int r = i / o4, c4 = i - r * o4;
        f32x4 v = *(const f32x4*)(out_lds + r * LDO + c4 * 4);
        if constexpr (REINIT) *(f32x4*)(out_lds + r * LDO + c4 * 4) = bias4(c4);
        if (act == RGCN_ACT_RELU) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = v[c] > 0.f ? v[c] : 0.f;
        } else if (act == RGCN_ACT_SIGMOID) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (c4 * 4 + c < a.dout) ? 1.f / (1.f + expf(-v[c])) : 0.f;
        }
        if (a.mask != nullptr) {
            const f32x4 m = *(const f32x4*)(a.mask + (size_t)(row0 + r) * a.ldm + c4 * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = m[c] > 0.f ? v[c] : 0.f;
        }
        *(f32x4*)(a.out + (size_t)(row0 + r) * a.ldo + c4 * 4) = v;
    }
}

// rgcn_tile3p.hip: producer-split bf16 kernel (64 -> 64, 128-slot chunks, buffer-addressable x; layout 1: two consumer teams)
int launch_tile3p(const TileArgs& a, int n_tiles, int layout, int chunk_rows, void* stream);

}  // namespace rgcn
