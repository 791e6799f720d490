// rgcn_kernels_shared.h -- what the kernel translation units of librgcn_mi355x.so share besides rgcn_common.h: the diagnostic
// build knobs, the LDS-DMA row gather of the ring kernels, and the host-side argument checks of the C ABI.
//   rgcn_tile_fp32.hip   rgcn_fwd / rgcn_bwd_dx; the exact-fp32 forward / dX kernel (every width class) is rgcn_tile_fp32_kernel.h,
//                        instantiated in rgcn_tile_fp32_narrow.hip / _wide.hip
//   rgcn_tile3p.hip      forward / dX of 64 x 64 layers on bf16 x 3 MFMAs (the default there)
//   rgcn_dw_relmajor.hip weight gradients, relation-major walks (every width class) + rgcn_bwd_dw
//   rgcn_dw_tile.hip     weight gradients, tile-major walk (64 x 64, <= 32 relations) + rgcn_bwd_dw_tiles
//   rgcn_dw_root.hip     d_root / d_bias by a plan-free streaming product
//   rgcn_abi.hip         version / status strings, weight packing, activation backward
//   rgcn_plan.hip        device-side graph plan builder
#pragma once
#include <atomic>
#include <cstring>
#include <type_traits>
#include "rgcn_common.h"
#include "rgcn_tile_common.h"

namespace rgcn {

// Diagnostic build only (-DRGCN_STAMPS, tools/debug/stamps.py): per-segment cycle sums of consumer wave 4
// and producer wave 0 of every workgroup, written to a buffer no other code reads.  Never in the product .so.
#ifdef RGCN_STAMPS
static __device__ unsigned long long* g_stamps = nullptr;     // one copy per translation unit (diagnostic builds only)
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP(v) const unsigned long long v = stamp()
#define STAMP_ADD(acc, a, b) acc += (b) - (a)
#else
#define STAMP(v)
#define STAMP_ADD(acc, a, b)
#endif
// compile-time ablations for the diagnostic build: 1 no main MFMA, 2 no run-sum MFMA, 4 no accumulator RMW
#ifndef RGCN_ABL
#define RGCN_ABL 0
#endif
#ifndef RGCN_PRIO
#define RGCN_PRIO 3
#endif
// 1: the Y^T path's 16 MFMAs of a row tile form ONE dependent chain (back-to-back dependent v_mfma_f32_16x16x4_f32 issue at
// full rate on gfx950), so the accumulate is 2 packed FMAs per tile; 0: two chains folded by 6 FMAs (measured 0.9 % slower)
#ifndef RGCN_ONE_CHAIN
#define RGCN_ONE_CHAIN 1
#endif
// tile-major dW: cut a unit's tail at the 4-row k-step instead of the 16-row group
#ifndef RGCN_DW_KSTEP_GATE
#define RGCN_DW_KSTEP_GATE 0
#endif
// cache policy of the direct dW kernel's x-row gathers (aux bits of buffer_load): 0 default, 2 = nt (streamed once)
#ifndef RGCN_DW_X_AUX
#define RGCN_DW_X_AUX 0
#endif
// run-time ablations of the tile kernel (1 skip MFMA + accumulate, 2 skip DMA, 4 skip B loads): only in diagnostic
// builds (-DRGCN_DEBUG_KNOBS, set through rgcn_debug_set_mode); the product library has no such switch
#ifdef RGCN_DEBUG_KNOBS
#define RGCN_DBG(a) ((a).dbg)
#else
#define RGCN_DBG(a) 0
#endif

// ------------------------------------------------------------------------------------------------
// producers: gather the 64 rows of a chunk into a ring slot by LDS-DMA -- ONE wave per chunk
// ------------------------------------------------------------------------------------------------
// Chunk k of a workgroup's sequence belongs to producer wave k % 4, so a wave issues every 4th
// iteration.  That cadence hides the latency of the chunk's row-index load (one coalesced 256-B
// vector load per chunk, issued right after the previous chunk's DMAs and first used 4 iterations
// later); a scalar index load issued in the same iteration as its DMAs was a demand miss to HBM
// per chunk and capped the first version at ~5 us per chunk.
// W = padded row width (floats).  One LDS-DMA instruction moves 64 lanes x 16 B = RPI rows.
// idxv: lane l holds the index of chunk row l; padding slots carry index == n_rows (one past the end).
// BUF = true : the matrix has < 2^24 rows and < 4 GiB and is addressed through a buffer descriptor:
//              offset = idx * row_bytes + column offset is ONE v_mad_u32_u24, and a padding row is out of
//              range by construction, so the hardware range check feeds its zeros with no select at all;
// BUF = false: 64-bit pointers, zeros from a 16-byte zero constant.
// The swizzled column offset of a lane depends on (row & 15) only, i.e. on (DMA index mod V): V lane-constant
// offsets are computed once per kernel, not per DMA.
template <int W, int MODE, bool BUF>
struct RowGather {
    static constexpr int LPR = W / 4;       // 16-byte lanes per row
    static constexpr int RPI = 64 / LPR;    // rows per DMA instruction
    static constexpr int NOPS = 64 / RPI;   // DMA instructions per chunk
    static constexpr int V = RPI >= 16 ? 1 : 16 / RPI;
    unsigned coff[V];   // byte offset of the 16-B column chunk this lane fetches (0xFFFFFFF0: beyond the width)
    unsigned rowb[V];   // BUF: bytes per row, or 0 where coff is the out-of-range marker -- so that
                        // offset = idx * rowb + coff is ONE v_mad_u32_u24 per DMA with no select behind it
    int rsub;
    int perm_addr;      // ds_bpermute address of chunk row `rsub` (further rows: immediate offsets)

    __device__ __forceinline__ void init(int lane, int n4, int ld) {
        rsub = lane / LPR;
        perm_addr = rsub * 4;
        const int p = lane % LPR;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int c = p ^ swizzle<MODE, LPR>(v * RPI + rsub);   // which 16-B column chunk lands at position p
            coff[v] = c < n4 ? (unsigned)c * 16u : 0xFFFFFFF0u;
            rowb[v] = c < n4 ? (unsigned)ld * 4u : 0u;
        }
    }

    // This wave's quarter (rows 16*pw .. 16*pw+15) of a chunk; row indices come from an LDS copy of the
    // chunk's index vector (landed there by LDS-DMA iterations earlier), read with same-address broadcasts.
    __device__ __forceinline__ void issue_quarter(const float* __restrict__ base, unsigned bytes, int n_rows, int ld,
                                                  const int* idx_lds, float* slot_base, int pw) const {
        constexpr int QOPS = NOPS / 4;
        int idx[QOPS];
#pragma unroll
        for (int i = 0; i < QOPS; ++i) idx[i] = idx_lds[16 * pw + i * RPI + rsub];
        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(base, bytes);
#pragma unroll
        for (int i = 0; i < QOPS; ++i) {
            float* dst = slot_base + (16 * pw + i * RPI) * W;
            unsigned co, rb;
            if constexpr (V >= 4) {   // pw-dependent variant (uniform select)
                co = coff[(pw * QOPS + i) % V];
                rb = rowb[(pw * QOPS + i) % V];
            } else {
                co = coff[i % V];
                rb = rowb[i % V];
            }
            if constexpr (BUF) {
                const unsigned off = __umul24((unsigned)idx[i], rb) + co;
                dma16_buf(rsrc, off, dst);
            } else {
                const float* gp = (idx[i] < n_rows && co != 0xFFFFFFF0u)
                                      ? (const float*)((const char*)(base + (size_t)idx[i] * ld) + co) : g_zero16;
                dma16(gp, dst);
            }
        }
    }

    // NOPS_PART consecutive DMA instructions of a 64-row block, starting at instruction `op0` (a multiple of V, so
    // that instruction i uses the lane constants i % V): rows op0 * RPI .. of the block whose 64 row indices are in
    // `idxv`.  part_base = LDS address of the first of those rows.
    template <int NOPS_PART>
    __device__ __forceinline__ void issue_part(const float* __restrict__ base, unsigned bytes, int n_rows, int ld,
                                               int idxv, float* part_base, int op0) const {
        int idx[NOPS_PART];
        const int pa = perm_addr + op0 * RPI * 4;
#pragma unroll
        for (int i = 0; i < NOPS_PART; ++i) idx[i] = __builtin_amdgcn_ds_bpermute(pa + i * RPI * 4, idxv);
        if constexpr (BUF) {
            const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(base, bytes);
#pragma unroll
            for (int i = 0; i < NOPS_PART; ++i) {
                const unsigned off = __umul24((unsigned)idx[i], rowb[i % V]) + coff[i % V];
                dma16_buf(rsrc, off, part_base + i * RPI * W);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NOPS_PART; ++i) {
                const unsigned co = coff[i % V];
                const float* gp = (idx[i] < n_rows && co != 0xFFFFFFF0u)
                                      ? (const float*)((const char*)(base + (size_t)idx[i] * ld) + co) : g_zero16;
                dma16(gp, part_base + i * RPI * W);
            }
        }
    }

    __device__ __forceinline__ void issue(const float* __restrict__ base, unsigned bytes, int n_rows, int ld,
                                          int idxv, float* slot_base) const {
        // all cross-lane index fetches first (one LDS-crossbar round trip for the batch, not one per DMA);
        // constant address + immediate offset per fetch: no address arithmetic in the loop
        int idx[NOPS];
#pragma unroll
        for (int i = 0; i < NOPS; ++i) idx[i] = __builtin_amdgcn_ds_bpermute(perm_addr + i * RPI * 4, idxv);
        if constexpr (BUF) {
            const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(base, bytes);
#pragma unroll
            for (int i = 0; i < NOPS; ++i) {
                const unsigned off = __umul24((unsigned)idx[i], rowb[i % V]) + coff[i % V];
                dma16_buf(rsrc, off, slot_base + i * RPI * W);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NOPS; ++i) {
                const unsigned co = coff[i % V];
                const float* gp = (idx[i] < n_rows && co != 0xFFFFFFF0u)
                                      ? (const float*)((const char*)(base + (size_t)idx[i] * ld) + co) : g_zero16;
                dma16(gp, slot_base + i * RPI * W);
            }
        }
    }
};

// KT x 16 B per lane of one column slice's B fragments (consecutive j are 1 KiB apart)
template <int KT>
__device__ __forceinline__ void prefetch_b(f32x4 (&dst)[KT], const f32x4* p) {
    prefetch16<0>(dst[0], p);
    if constexpr (KT > 1) prefetch16<1024>(dst[1], p);
    if constexpr (KT > 2) {
        prefetch16<2048>(dst[2], p);
        prefetch16<3072>(dst[3], p);
    }
    if constexpr (KT > 4) {
        const f32x4* q = p + 4 * 64;
        prefetch16<0>(dst[4], q);
        prefetch16<1024>(dst[5], q);
        prefetch16<2048>(dst[6], q);
        prefetch16<3072>(dst[7], q);
    }
}

// ------------------------------------------------------------------------------------------------
// host side: argument checks shared by the entry points
// ------------------------------------------------------------------------------------------------
template <int KP>
constexpr int tile_nbuf() { return KP == 128 ? 2 : 4; }
template <int KP, int NP>
constexpr int dw_nbuf() { return (KP == 128 || NP == 128) ? 2 : 4; }

// Tiles one workgroup of rgcn_tile_kernel walks (1..16).  One workgroup fits a CU, so a launch runs in rounds of 256
// workgroups, and the round count is what the time follows (tools/debug/tpw_sweep.py, forward launch, 28,410 tiles: 16
// tiles -> 1,776 workgroups = 7 rounds x 16 = 112 tile times, 10.36 ms; 12 -> 2,368 = 10 rounds x 12 = 120, 11.48 ms;
// 1 -> 111 rounds, 10.67 ms: a workgroup's start-up costs ~3-4 % of a tile.  2,841 tiles: 12 -> 1 round, 1.11 ms; 8 ->
// 2 rounds x 8, 1.46 ms).  Pick the count with the least rounds x (tiles + start-up), the larger one on ties.
static int tiles_per_workgroup(int n_tiles) {
    constexpr int kCUs = 256;
    int best = 1;
    double best_cost = 1e30;
    for (int t = 1; t <= 16; ++t) {
        const int wgs = (n_tiles + t - 1) / t;
        const int rounds = (wgs + kCUs - 1) / kCUs;
        const double cost = rounds * (t + 0.04);
        if (cost <= best_cost * 1.002) {
            best_cost = cost < best_cost ? cost : best_cost;
            best = t;
        }
    }
    return best;
}

// (layout 2 -- the edge-parallel path's dense relation-major units handed to rgcn_bwd_dw -- has no tiles to walk: a heavy part
// of a few units on a graph of several 32768-node pseudo tiles is a valid plan, so the chunks-per-tile bound does not apply to it;
// layout 5 -- pairs of rows on one slot -- leaves chunks EMPTY: fewer units than chunks)
static int check_plan(const rgcn_plan_t* p) {
    if (p == nullptr) return RGCN_ERR_NULL;
    if (!p->tile_ptr || !p->chunk_rel || !p->chunk_cnt || !p->chunk_tile || !p->chunk_flags || !p->rel_order ||
        !p->slot_src ||
        !p->slot_w || !p->slot_row || !p->slot_acc)
        return RGCN_ERR_NULL;
    if (p->n_nodes <= 0 || p->n_owned <= 0 || p->num_relations <= 0 || p->tile <= 0 || (p->tile % 16) != 0 || p->tile > 32768 ||
        p->n_tiles <= 0 || (p->layout != 2 && p->n_chunks < p->n_tiles) || p->n_chunks <= 0 || (long)p->n_tiles * p->tile < p->n_owned ||
        (p->chunk != 64 && p->chunk != 128) || (p->layout != 5 && p->n_units < p->n_chunks) || p->n_units > p->n_chunks * (p->chunk / 64))
        return RGCN_ERR_PLAN;
    return RGCN_OK;
}

// bytes of a [rows, ld] fp32 matrix if it can be gathered through a buffer descriptor: 24-bit row index and
// row size (v_mad_u32_u24), 32-bit offsets with the one-past-the-end padding row and the all-ones "beyond
// the width" offset out of range; else 0 -> the kernels fall back to 64-bit pointers
// (RGCN_FLAG_POINTER_GATHER asks for that fallback on any input: how the tests reach it on small graphs)
static unsigned buffer_bytes(int rows, int ld, unsigned flags) {
    if (flags & RGCN_FLAG_POINTER_GATHER) return 0u;
    const size_t bytes = (size_t)rows * ld * sizeof(float);
    const size_t with_pad_row = bytes + (size_t)ld * sizeof(float);
    return (rows < (1 << 24) && with_pad_row < 0xFFFFFF00ull) ? (unsigned)bytes : 0u;
}

// Opt a kernel instantiation into the full 160 KiB of dynamic LDS: once per (instantiation, device), not per launch.
template <auto KERN>
static hipError_t allow_full_lds() {
    static std::atomic<unsigned long long> done{0};     // one per kernel instantiation (KERN is a template argument)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

}  // namespace rgcn
