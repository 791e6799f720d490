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
    // a thread meets one column group only when the thread count is a multiple of the groups per row: its bias then
    // is loaded once, not per element
    const bool fixed_c4 = REINIT && (nthreads % o4) == 0;
    f32x4 bfix = {0.f, 0.f, 0.f, 0.f};
    if (fixed_c4) bfix = bias4(tid % o4);
    for (int i = tid; i < rows * o4; i += nthreads) {
        const int r = i / o4, c4 = i - r * o4;
        f32x4 v = *(const f32x4*)(out_lds + r * LDO + c4 * 4);
        if constexpr (REINIT) *(f32x4*)(out_lds + r * LDO + c4 * 4) = fixed_c4 ? bfix : bias4(c4);
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
