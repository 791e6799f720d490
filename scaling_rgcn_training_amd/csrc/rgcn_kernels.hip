// rgcn_kernels.hip -- R-GCN layer forward / backward for gfx950 (MI355X) and the C ABI of
// include/rgcn_mi355x.h.  Written for CDNA4 only: 64-wide waves, LDS-DMA gathers, fp32 MFMA.
//
// Replaces the arithmetic of torch_geometric.nn.RGCNConv (PyG 2.3.1) that the reference calls at
// /root/reference/model/layers.py:21,23,62,64,108,110 and differentiates at model/modelTrainer.py:66.
//
// Kernels
//   rgcn_pack_kernel      weight[R',din,dout] (+root) -> MFMA B-fragment order, zero padded to 16/32/64/128
//   rgcn_tile_kernel      forward AND dX: per output tile, out_tile(LDS) += w_e * (x[src_e] @ B_rel)
//                         (tile-major walk of the plan; dX = same kernel on the transposed plan with W^T)
//   rgcn_dw_kernel        weight gradients: per relation, dB_rel += (w_e x[src_e])^T g[dst_e]
//                         (relation-major walk; register accumulators; one partial slab per (workgroup, rel))
//   rgcn_dw_reduce_kernel fixed-order sum of the slabs -> d_weight / d_root / d_bias
#include <atomic>
#include <cstring>
#include <type_traits>
#include "rgcn_common.h"
#include "rgcn_tile_common.h"

namespace rgcn {

// Diagnostic build only (-DRGCN_STAMPS, tools/debug/stamps.py): per-segment cycle sums of consumer wave 4
// and producer wave 0 of every workgroup, written to a buffer no other code reads.  Never in the product .so.
#ifdef RGCN_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
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
// weight pack
// ------------------------------------------------------------------------------------------------
// packed[((rel*NT + s)*KT + j)*256 + lane*4 + t] = B_rel[k = 16j + 4*(lane>>4) + t][col = 16s + (lane&15)]
// which is exactly the (a, b) pairing the consumers use for v_mfma_f32_16x16x4_f32:
// lane l supplies A[row = l&15][k' = l>>4] and B[k' = l>>4][col = l&15]; MFMA step (j,t) stands for
// k = 16j + 4k' + t on both operands.
__global__ void rgcn_pack_kernel(const float* __restrict__ weight, const float* __restrict__ root, int num_rel,
                                 int din, int dout, int transpose, int KP, int NP, float* __restrict__ packed) {
    const int per_rel = KP * NP;
    const long total = (long)(num_rel + 1) * per_rel;
    const int KT = KP / 16, NT = NP / 16;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int rel = (int)(idx / per_rel);
        int rem = (int)(idx % per_rel);
        const int t = rem & 3;
        const int lane = (rem >> 2) & 63;
        rem >>= 8;
        const int j = rem % KT;
        const int s = rem / KT;
        (void)NT;
        const int k = 16 * j + 4 * (lane >> 4) + t;
        const int col = 16 * s + (lane & 15);
        const float* m = rel < num_rel ? weight + (size_t)rel * din * dout : root;
        float v = 0.f;
        if (m != nullptr) {
            if (!transpose) {
                if (k < din && col < dout) v = m[(size_t)k * dout + col];
            } else {
                if (k < dout && col < din) v = m[(size_t)col * dout + k];
            }
        }
        packed[idx] = v;
    }
}

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
// forward / dX kernel
// ------------------------------------------------------------------------------------------------
// second launch bound = waves per SIMD the register allocation must allow.  2: one 8-wave workgroup per CU at
// full register budget; two workgroups per CU (bound 4 = 128 VGPRs, tiles of 160 nodes, RGCN_LDS_KB=80) measured
// no faster (11.5 vs 11.6 ms): the limiter is SIMD issue, not latency
#ifndef RGCN_TILE_WAVES
#define RGCN_TILE_WAVES 2
#endif
// Producer waves of the tile kernel (experiment knob): 4 = one 8-wave workgroup per CU; 2 = 6-wave workgroups, two per
// CU when their LDS fits twice (3 waves per SIMD at the full register budget)
#ifndef RGCN_TILE_PW
#define RGCN_TILE_PW 4
#endif
constexpr int kTileProducers = RGCN_TILE_PW;
constexpr int kTileThreads = 64 * (kTileProducers + 4);

// ---- producers of the forward / dX kernels: LDS-DMA gather, D chunks ahead of the consumers -----------------------
// c0 / nch: the chunks of ALL the tiles the workgroup walks (one sequence); tile0: its first tile.  Where a chunk closes a
// tile the consumers store and re-initialise the accumulator between two extra barriers' worth of time: the producers
// join that one extra barrier (E) so that the barrier counts of the two roles stay equal.
template <int KP, int NBUF, bool BUF, int CH>
__device__ __forceinline__ void tile_producer_loop(const TileArgs& a, float* ring, float* wring, int* dring, int c0, int nch,
                                                   int lane, int wave, int tile0) {
    constexpr int D = NBUF - 1;
    int tile_cur = tile0;
    int tend = ldc(a.tile_ptr, tile0 + 1) - c0;        // first chunk (relative) of the next tile
    auto tile_boundary = [&](int it) {                  // after the barrier that closes chunk `it`
        if (it + 1 == tend && it + 1 < nch) {
            ++tile_cur;
            tend = ldc(a.tile_ptr, tile_cur + 1) - c0;
            wg_barrier();
        }
    };
        // The producers' few instructions must not queue behind the consumer wave's MFMAs on the shared SIMD
        // (issue is arbitrated by priority, then age; an fp32 MFMA holds the pipe 32 cycles): RGCN_PRIO
        __builtin_amdgcn_s_setprio(RGCN_PRIO);
        // ---- producers: LDS-DMA gather, D chunks ahead of the consumers; wave (k % 4) owns chunk k ----
        const int pw = wave;
        int knext = pw;                                   // this wave's next chunk
        RowGather<KP, kRowRead, BUF> gather;
        gather.init(lane, (RGCN_DBG(a) & 2) ? 0 : a.din4, a.ldx);
#ifdef RGCN_STAMPS
        unsigned long long sp_issue = 0, sp_wait = 0, sp_bar = 0;
#endif
        using Gather = RowGather<KP, kRowRead, BUF>;
        constexpr int RW = CH / kTileProducers;           // rows of a chunk per producer wave in the spread scheme
        if constexpr (D == 1 && RW >= 16 && RW <= 64 && RW % Gather::RPI == 0) {
            // ---- one chunk ahead (two ring slots): EVERY wave issues its RW rows of EVERY chunk -------------------
            // With a single chunk in flight its round trip is on the critical path of every iteration; four waves
            // issuing a quarter each put the whole chunk on the wire in a quarter of the time (and spread the
            // producers' vector instructions over the four SIMDs instead of loading one consumer's).
            constexpr int NOPS_PART = RW / Gather::RPI;
            const int row0 = pw * RW;                     // first row of this wave's part inside the chunk
            const int half = row0 / 64, op0 = (row0 % 64) / Gather::RPI;
            const bool meta = row0 % 64 == 0;             // this wave also moves the half's weights / run metadata
            auto load_idx = [&](int k) {
                const int kk = k < nch ? k : nch - 1;
                return a.slot_src[(size_t)(c0 + kk) * CH + 64 * half + lane];
            };
            auto issue_part = [&](int k, int idxv) {
                const int chunk = c0 + k, buf = k % NBUF;
                gather.template issue_part<NOPS_PART>(a.x, a.x_bytes, a.n_rows, a.ldx, idxv,
                                                      ring + (buf * CH + row0) * KP, op0);
                if (meta) {
                    dma4(a.slot_w + (size_t)chunk * CH + 64 * half + lane, wring + buf * CH + 64 * half);
                    dma4(a.slot_acc + (size_t)chunk * CH + 64 * half + lane, dring + buf * CH + 64 * half);
                }
            };
            int idx_cur = load_idx(0);
            issue_part(0, idx_cur);                       // (its index vector is waited for here, once per tile)
            idx_cur = load_idx(1);
            wait_vmcnt<0>();                              // chunk 0 landed (and the indices of chunk 1)
            wg_barrier();                                 // chunk 0 (and the accumulator init) visible
            for (int it = 0; it < nch; ++it) {
                STAMP(p0);
                int idx_next = idx_cur;
                if (it + 1 < nch) {
                    issue_part(it + 1, idx_cur);
                    idx_next = load_idx(it + 2);          // youngest operation: lands with the rows
                }
                STAMP(p1);
                wait_vmcnt<0>();                          // chunk it + 1 landed
                idx_cur = idx_next;
                STAMP(p2);
                wg_barrier();
                STAMP(p3);
                STAMP_ADD(sp_issue, p0, p1);
                STAMP_ADD(sp_wait, p1, p2);
                STAMP_ADD(sp_bar, p2, p3);
                tile_boundary(it);
            }
        } else {
        // Row indices of this wave's NEXT chunk: one coalesced load, issued right after the current
        // chunk's DMAs and not touched until the wave's next turn 4 iterations later (any use here would
        // make hipcc wait vmcnt(0) on the spot, i.e. for the DMAs just issued).  The address is clamped
        // so the load is always valid; issue() only runs for k < nch.
        constexpr int HALVES = CH / 64;
        auto load_idx = [&](int k, int h) {
            const int kk = k < nch ? k : nch - 1;
            return a.slot_src[(size_t)(c0 + kk) * CH + 64 * h + lane];
        };
        int idxv[HALVES];
#pragma unroll
        for (int h = 0; h < HALVES; ++h) idxv[h] = load_idx(knext, h);
        auto issue = [&](int k) {                         // k == knext
            const int chunk = c0 + k, buf = k % NBUF;
#pragma unroll
            for (int h = 0; h < HALVES; ++h) {
                gather.issue(a.x, a.x_bytes, a.n_rows, a.ldx, idxv[h], ring + (buf * CH + 64 * h) * KP);
                dma4(a.slot_w + (size_t)chunk * CH + 64 * h + lane, wring + buf * CH + 64 * h);
                dma4(a.slot_acc + (size_t)chunk * CH + 64 * h + lane, dring + buf * CH + 64 * h);
            }
            knext += kTileProducers;
#pragma unroll
            for (int h = 0; h < HALVES; ++h) idxv[h] = load_idx(knext, h);   // youngest ops of this wave from here on
        };
#pragma unroll
        for (int k = 0; k < D; ++k)
            if (k % kTileProducers == pw && k < nch) issue(k);
        if (pw == 0) wait_vmcnt<0>();                     // chunk 0 landed
        wg_barrier();                                     // chunk 0 (and the accumulator init) visible
        for (int it = 0; it < nch; ++it) {
            // slot (it+D)%NBUF held chunk it-1, which the consumers finished before the last barrier
            const int ki = it + D, kw = it + 1;
            STAMP(p0);
            if (ki % kTileProducers == pw && ki < nch) issue(ki);
            STAMP(p1);
            // a wave has at most ONE chunk in flight (D <= 4), plus the index load issued with it (which
            // hipcc may schedule among the DMAs): vmcnt(0) is exact
            if (kw % kTileProducers == pw && kw < nch) wait_vmcnt<0>();   // chunk it+1 landed
            STAMP(p2);
            wg_barrier();
            STAMP(p3);
            STAMP_ADD(sp_issue, p0, p1);
            STAMP_ADD(sp_wait, p1, p2);
            STAMP_ADD(sp_bar, p2, p3);
            tile_boundary(it);
        }
        }
        wait_vmcnt<0>();
#ifdef RGCN_STAMPS
        if (g_stamps && lane == 0) {
            unsigned long long* o = g_stamps + (size_t)blockIdx.x * 32;
            if (pw == 0) { o[4] = sp_issue; o[5] = sp_wait; o[6] = sp_bar; }
            if (pw == 1) o[7] = nch;
        }
#endif
}

// CH = edge slots per chunk = rows of one ring slot (64 or 128): a 128-slot chunk is consumed as two 64-row parts
// with no barrier, metadata fetch or B swap between them
template <int KP, int NP, int NBUF, bool BUF, int CH>
__global__ void __launch_bounds__(kTileThreads, RGCN_TILE_WAVES) rgcn_tile_kernel(const TileArgs a) {
    constexpr int KT = KP / 16, NT = NP / 16;
    constexpr int D = NBUF - 1;                  // chunks the producers run ahead
    static_assert(D >= 1 && D <= kTileProducers, "one chunk in flight per producer wave");
    constexpr int CW = NT < 4 ? NT : 4;          // consumer waves that own output column slices
    constexpr int SL = NT < 4 ? 1 : NT / 4;      // column slices per consumer wave
    constexpr int LPR = KP / 4;
    constexpr int LDO = kAccStride<NP>;          // accumulator row stride: NP + 4 floats, so that the 16 rows one
                                                 // ds_read/write_b128 touches start in different banks

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* out_lds = lds;                         // [tile + 1][LDO]  (row `tile`: dummy)
    float* ring = lds + (a.tile + 1) * LDO;       // [NBUF][CH][KP]
    float* wring = ring + NBUF * CH * KP;         // [NBUF][CH]
    int* dring = (int*)(wring + NBUF * CH);       // [NBUF][CH]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // a workgroup walks `tiles_per_wg` consecutive tiles: their chunks form ONE sequence for the ring, and between two
    // tiles only the accumulator is stored and reset -- the next tile's first chunk is already in LDS by then
    const int tile0 = blockIdx.x * a.tiles_per_wg;
    const int tile1 = min(tile0 + a.tiles_per_wg, a.n_tiles);
    const int c0 = ldc(a.tile_ptr, tile0);
    const int nch = ldc(a.tile_ptr, tile1) - c0;

    for (int i = tid; i < (a.tile + 1) * LDO; i += kTileThreads) {
        const int col = i % LDO;
        out_lds[i] = (a.bias != nullptr && col < a.dout) ? a.bias[col] : 0.f;
    }

    // The two roles run DIFFERENT loops that meet only at s_barrier (a hardware arrival counter: it
    // does not care which instruction a wave arrives from; both loops execute 1 + nch barriers).
    // The consumer loop is written FIRST on purpose: hipcc's waitcnt pass is program-order based, so
    // with no LDS-DMA ahead of it the consumer code gets exact counted waits for its own B-fragment
    // loads; in a shared loop body every ring read was preceded by s_waitcnt vmcnt(0) ("a DMA may be
    // pending"), which un-overlapped the B prefetch from the MFMAs.
    if (wave >= kTileProducers) {
        // ---- consumers: ring -> MFMA -> tile accumulator in LDS -----------------------------------
        // Consumer wave cw owns output column slices {cw + CW*s}; with fewer than 4 slices (NP < 64) the
        // surplus consumer waves only keep the barrier count (same time per row: the MFMA work per row
        // shrinks with NP).  Exclusive column ownership + the run-sum below make every accumulator
        // update a plain LDS read-modify-write: no LDS float atomics (ds_add_f32 retires ~1 lane per 3
        // cycles on gfx950, ~190 cycles per wave-instruction: tools/probes/lds_atomic_rate.hip; it
        // was 60 % of the first version's kernel time) and bit-reproducible sums.
        const int cwv = wave - kTileProducers;
        const bool active = cwv < CW;
        const int cw = cwv;
        const int rowl = lane & 15, kq = lane >> 4;
        const unsigned lane_col_bytes = (unsigned)(16 * cw + rowl) * 4u;   // this lane's column in slice 0 (Y layout)
        const unsigned lane_col4_bytes = (unsigned)(16 * cw + 4 * kq) * 4u; // its four columns in the Y^T layout
        const f32x4* wp4 = (const f32x4*)a.wp;
        f32x4 bcur[SL][KT], bnext[SL][KT];
        int rel_cur = ldc(a.chunk_rel, c0);
        if (active) {
#pragma unroll
            for (int s = 0; s < SL; ++s)
#pragma unroll
                for (int j = 0; j < KT; ++j)
                    bcur[s][j] = wp4[((size_t)(rel_cur * NT + cw + CW * s) * KT + j) * 64 + lane];
        }
        // retire these loads in the compiler's scoreboard HERE: otherwise it keeps "maybe pending" waits
        // in front of the loop's MFMAs, and those s_waitcnt vmcnt(N) would also wait for the asm prefetch
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        int cnt_pre = ldc(a.chunk_cnt, c0);
        int flags_pre = ldc(a.chunk_flags, c0);
        // Relation ids one and two chunks ahead.  The NEXT relation's weight fragments are prefetched into `bnext` at the
        // END of an iteration, just before the barrier: by then the producers have issued (and waited for) all their
        // LDS-DMAs, so the CU's vector-memory queue is empty and these few loads issue at once.  Issued at the TOP of
        // an iteration -- right behind the barrier, when the producers flood the queue with the next chunk's 32 gathers --
        // every global_load of a consumer wave took hundreds of cycles to ISSUE (~1,000 cycles per chunk, the "fixed cost
        // that does not scale with the chunk" of the stamp profile; tools/debug/stamps.py).
        constexpr bool kAsmPrefetch = SL * KT <= 4;
        int rel_n1 = nch > 1 ? ldc(a.chunk_rel, c0 + 1) : rel_cur;
        int rel_n2 = nch > 2 ? ldc(a.chunk_rel, c0 + 2) : rel_n1;
        auto prefetch_rel = [&](int rel) {
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                const f32x4* bp = wp4 + ((size_t)(rel * NT + cw + CW * s) * KT) * 64 + lane;
                if constexpr (kAsmPrefetch) {
                    prefetch_b<KT>(bnext[s], bp);
                } else {
#pragma unroll
                    for (int j = 0; j < KT; ++j) bnext[s][j] = bp[j * 64];
                }
            }
        };
        bool pending = false;       // bnext is receiving the fragments of chunk it + 1
        if (kAsmPrefetch && active && rel_n1 != rel_cur && !(RGCN_DBG(a) & 4)) {
            prefetch_rel(rel_n1);
            pending = true;
        }
        int tile_cur = tile0;
        int tend = ldc(a.tile_ptr, tile0 + 1) - c0;     // first chunk (relative) of the next tile
        wg_barrier();
#ifdef RGCN_STAMPS
        unsigned long long st_scal = 0, st_comp = 0, st_bwait = 0, st_bar = 0;
        unsigned long long st_nrt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        for (int it = 0; it < nch; ++it) {
            STAMP(t0);
            const int chunk = c0 + it;
            const int buf = it % NBUF;
            // chunk metadata arrives one iteration ahead (scalar loads issued a whole chunk earlier)
            const int cnt = cnt_pre;
            const int flags_chunk = flags_pre;
            const int rel_next = rel_n1;
            const int rel_next2 = rel_n2;
            if (it + 1 < nch) {
                cnt_pre = ldc(a.chunk_cnt, chunk + 1);
                flags_pre = ldc(a.chunk_flags, chunk + 1);
            }
            rel_n1 = rel_n2;
            if (it + 3 < nch) rel_n2 = ldc(a.chunk_rel, chunk + 3);
#ifdef RGCN_STAMPS
            asm volatile("" ::"s"(cnt), "s"(rel_next));
#endif
            STAMP(t1);
            // (the asm prefetch must never be spilled before its wait -- hipcc believes the value is there --: only used
            // where the fragment sets fit the register file comfortably; the plain-load form keeps the early prefetch)
            const bool swap_b = kAsmPrefetch ? pending : (active && rel_next != rel_cur && !(RGCN_DBG(a) & 4));
            if (!kAsmPrefetch && swap_b) prefetch_rel(rel_next);
            const int nrt_all = (!active || (RGCN_DBG(a) & 1)) ? 0 : (cnt + 15) >> 4;
            const int flags_all = flags_chunk & 0xFF;     // bit 8 (layout 1: the chunk's halves share a destination) is not ours
            // A chunk without repeated destinations runs as ONE straight-line block over all its row tiles (up to
            // CH / 16); otherwise 64-row parts of up to four tiles, each on the path its own flags ask for.
            const bool whole = flags_all == 0;
#pragma unroll
            for (int part = 0; part < CH / 64; ++part) {
            if (whole && part > 0) break;
            const float* hb = ring + (buf * CH + 64 * part) * KP;
            const float* wb = wring + buf * CH + 64 * part;
            const int* db = dring + buf * CH + 64 * part;
            const int nrt = whole ? nrt_all : (nrt_all - 4 * part < 4 ? nrt_all - 4 * part : 4);
            const int flags = (flags_all >> (4 * part)) & 15;
            // Operands of one 16-row tile.  Rows of a chunk are sorted by destination, so equal
            // destinations are adjacent runs; a run ends at a change of destination or at the end of the row
            // tile (the next tile is processed after it).  Which slot ends each run, and which accumulator
            // row each slot writes, comes precomputed with the plan (slot_acc): the consumers' vector
            // instructions compete with the fp32 MFMAs for the same SIMD pipe, so none are spent on it here.
            struct Ops {
                f32x4 av[KT];
                f32x4 w4;   // Y layout: weights of rows 4*kq + i (the rows whose MFMA results this lane holds)
                i32x4 d4;   //           their run metadata from the plan: run-end position << 24 | accumulator row
                float w1;   // Y^T layout: weight and metadata of row `rowl`
                int d1;
            };
            // per-tile state carried between the pipeline stages below
            struct Tile {
                f32x4 y[SL];      // H W_r of the tile (main MFMA result); after stage B: one addend of the update
                f32x4 z[SL];      // after stage B: the other addend (old accumulator contents [+ run sums])
                f32x4 old[SL];    // Y^T path: accumulator contents (y, z = the two MFMA chains)
                float* dst[4];    // accumulator rows this lane updates (Y^T path: dst[0] only)
            };
            // this lane's operand addresses for row tile 0 of the part, formed once; row tile rt is rt * 16 rows
            // further, an immediate offset of the DS instruction (left to itself hipcc re-derives every address
            // per tile: 5 vector adds each)
            const float* arow[KT];
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const int pos = (4 * j + kq) ^ swizzle<kRowRead, LPR>(rowl);
                arow[j] = hb + rowl * KP + pos * 4;
            }
            const float* wrow = wb + rowl;
            const int* drow = db + rowl;
            const float* wrow4 = wb + 4 * kq;
            const int* drow4 = db + 4 * kq;
            auto load_ops = [&](Ops& o, int rt, auto tr_c) {
                // weight / metadata first: the accumulator address of the tile is the first thing computed from it
                if constexpr (decltype(tr_c)::value) {
                    o.w1 = wrow[rt * 16];
                    o.d1 = drow[rt * 16];
                } else {
                    o.w4 = *(const f32x4*)(wrow4 + rt * 16);
                    o.d4 = *(const i32x4*)(drow4 + rt * 16);
                }
#pragma unroll
                for (int j = 0; j < KT; ++j) o.av[j] = *(const f32x4*)(arow[j] + rt * 16 * KP);
            };
            // accumulator row (low 24 bits of the plan's metadata word) -> LDS address of this lane's column(s):
            // one v_mad_u32_u24 (it ignores the run-end byte on top by itself)
            auto acc_ptr = [&](int d, unsigned col_bytes) -> float* {
                return (float*)((char*)out_lds + (__umul24((unsigned)d, (unsigned)(LDO * 4)) + col_bytes));
            };

            // ===== chunks WITHOUT repeated destinations inside any row tile (nearly all of them: the plan spreads
            // a run over different tiles whenever it can): the Y^T path ======================================
            // Y^T = W_r^T . H_tile^T -- the same two registers per MFMA as Y = H W, operands swapped -- leaves a
            // lane with FOUR CONSECUTIVE COLUMNS of ONE row, so its accumulator update is one ds_read_b128, four
            // packed FMAs and one ds_write_b128.  Every vector instruction next to an fp32 MFMA costs its full
            // 4+ issue cycles ON TOP of the MFMA time, plus ~10 cycles per MFMA->VALU->MFMA switch
            // (tools/probes/mfma_f32_overlap.hip: only LDS traffic hides under v_mfma_f32_16x16x4_f32); the Y
            // layout spends 4 b32 reads + 4 b32 writes + 4 addresses + 6 more VALU per slice.
            // `half` 0 / 1: the first / second 2 KT MFMAs of each chain pair (the accumulate of the PREVIOUS tile is
            // issued between the halves, see consume)
            auto stage_a_t = [&](const Ops& o, Tile& t, int half) {
#pragma unroll
                for (int s = 0; s < SL; ++s) {
                    f32x4 acc0 = t.y[s], acc1 = t.z[s];
                    if (half == 0) acc0 = acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < 4 * KT; m += 2) {
                        if ((m < 2 * KT) != (half == 0)) continue;
                        const int j = m >> 2, i = m & 3;
                        if (RGCN_ABL & 1) {
                            acc0 += o.av[j];
                            continue;
                        }
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[s][j][i], o.av[j][i], acc0, 0, 0, 0);
                        if (RGCN_ONE_CHAIN)
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[s][j][i + 1], o.av[j][i + 1], acc0, 0, 0, 0);
                        else
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[s][j][i + 1], o.av[j][i + 1], acc1, 0, 0, 0);
                    }
                    t.y[s] = acc0;      // the two chains are folded in stage C's FMAs
                    t.z[s] = acc1;
                }
            };
            auto stage_b_t = [&](const Ops& o, Tile& t) {     // old accumulator contents (after tile t-1's store)
#if RGCN_ABL & 8
                // timing-only diagnostic (WRONG results): the 16 lanes of a ds_read_b128 phase address rows that differ
                // mod 16 -- what a conflict-free accumulator layout could buy
                t.dst[0] = acc_ptr((o.d1 & 0xFFFFF0) | rowl, lane_col4_bytes);
#else
                t.dst[0] = acc_ptr(o.d1, lane_col4_bytes);
#endif
#pragma unroll
                for (int s = 0; s < SL; ++s) t.old[s] = *(const f32x4*)(t.dst[0] + 16 * CW * s);
            };
            auto stage_c_t = [&](const Ops& o, Tile& t) {     // acc_new = old + w * chain0 + w * chain1
#pragma unroll
                for (int s = 0; s < SL; ++s) {
                    f32x4 v = t.y[s] * o.w1 + t.old[s];
                    if (!RGCN_ONE_CHAIN) v = t.z[s] * o.w1 + v;
                    *(f32x4*)(t.dst[0] + 16 * CW * s) = v;
                }
            };

            // ===== chunks with a repeated destination in some row tile: the Y path ===========================
            // stage A: y = H_tile . W_r (16 MFMAs per column slice, two independent chains)
            auto stage_a = [&](const Ops& o, Tile& t) {
#pragma unroll
                for (int s = 0; s < SL; ++s) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < KT; ++j) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.av[j][0], bcur[s][j][0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.av[j][1], bcur[s][j][1], acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.av[j][2], bcur[s][j][2], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.av[j][3], bcur[s][j][3], acc1, 0, 0, 0);
                    }
                    t.y[s] = acc0 + acc1;
                }
            };
            // stage B: Z = P . Y.  P[m][k] = w_k if row k belongs to the run that ENDS at row m, else 0, so
            // each run's weighted sum lands on its last row and the rows that write in stage C have
            // pairwise distinct destinations inside the tile.  Y's accumulator registers are already in
            // B-operand layout for MFMA step i with k = 4*k' + i: no data movement.  The accumulator reads
            // of stage C are issued here (after the previous tile's stage-C writes in program order).
            auto stage_b = [&](const Ops& o, Tile& t, bool dup) {
#pragma unroll
                for (int i = 0; i < 4; ++i) t.dst[i] = acc_ptr(o.d4[i], lane_col_bytes);
                if (!dup) {
                    // no destination repeats inside THIS row tile: every row ends its own run, P would be
                    // diag(w) -- no product: acc_new = y * w + acc_old
#pragma unroll
                    for (int s = 0; s < SL; ++s) {
                        f32x4 old;
#pragma unroll
                        for (int i = 0; i < 4; ++i) old[i] = t.dst[i][16 * CW * s];
                        t.z[s] = old;
                        t.y[s] = t.y[s] * o.w4;      // finished in stage C as z + y
                    }
                    return;
                }
                float pm[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // row 4*kq + i belongs to the run that ends at tile row (d >> 24); this lane supplies
                    // P[m = rowl][k = 4*kq + i]
                    pm[i] = ((unsigned)o.d4[i] >> 24) == (unsigned)rowl ? o.w4[i] : 0.f;
                }
#pragma unroll
                for (int s = 0; s < SL; ++s) {
                    // the accumulator's old contents are the C operand of the first run-sum MFMA: Z = P.Y + old
                    f32x4 old;
#pragma unroll
                    for (int i = 0; i < 4; ++i) old[i] = t.dst[i][16 * CW * s];
                    f32x4 z1 = {0.f, 0.f, 0.f, 0.f};
                    f32x4 z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pm[0], t.y[s][0], old, 0, 0, 0);
                    z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pm[1], t.y[s][1], z1, 0, 0, 0);
                    t.z[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(pm[2], t.y[s][2], z0, 0, 0, 0);
                    t.y[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(pm[3], t.y[s][3], z1, 0, 0, 0);
                }
            };
            // stage C: write back acc_new = z + y (both branches leave the two addends there); this wave owns
            // these columns and the rows written by one instruction are pairwise distinct: plain stores
            auto stage_c = [&](Tile& t) {
#pragma unroll
                for (int s = 0; s < SL; ++s) {
                    const f32x4 v = t.z[s] + t.y[s];
#pragma unroll
                    for (int i = 0; i < 4; ++i) t.dst[i][16 * CW * s] = v[i];
                }
            };
            // One straight-line block per tile count (1..4) and path, so hipcc can interleave freely.  Software
            // pipeline: A(t+1) is issued before the tail of tile t, and stage C runs a further step behind,
            // so the VALU / LDS work of one tile sits behind the next tiles' MFMAs in program order and its LDS
            // round trips are hidden.
            auto consume = [&](auto nrt_c, auto tr_c) {
                constexpr int NRT = decltype(nrt_c)::value;
                constexpr bool TR = decltype(tr_c)::value;
                Ops ops[NRT];
                Tile tl[NRT];
                // LDS reads of tile t+1 are issued BEFORE the MFMAs of tile t and pinned there with
                // sched_barrier: left alone, hipcc sinks every ds_read_b128 to just in front of the four
                // MFMAs that use it and waits lgkmcnt(0) -- 16 exposed LDS round trips per chunk.
                load_ops(ops[0], 0, tr_c);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int step = 0; step < NRT + 2; ++step) {
                    if constexpr (TR) {
                        // per tile ONE group of vector instructions (store of tile t-1, address + accumulator read
                        // of tile t) in front of tile t's 16 MFMAs: the read has the whole MFMA block to land, and
                        // the MFMA->VALU->MFMA switch is paid once.  Store(t-1) precedes read(t) in program order:
                        // consecutive tiles may hit the same accumulator row.
                        if (step > NRT) continue;
                        // first half of this tile's MFMAs, the next tile's operand reads in between (an LDS
                        // instruction between two MFMAs costs ~2 cycles; in front of the block its full issue slot)
                        if (step + 1 < NRT) {
                            if (RGCN_ABL & 4) {     // diagnostic: no operand reads after tile 0 (opaque copy: no CSE)
                                ops[step + 1] = ops[0];
#pragma unroll
                                for (int j = 0; j < KT; ++j) asm volatile("" : "+v"(ops[step + 1].av[j]));
                            } else {
                                load_ops(ops[step + 1], step + 1, tr_c);
                            }
                        }
                        if (step < NRT) stage_a_t(ops[step], tl[step], 0);
                        if (step + 1 < NRT) {
#pragma unroll
                            for (int i = 0; i < KT; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
                                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        // ONE group of vector instructions per tile, in the MIDDLE of its MFMA block: the store of
                        // tile t-1 (its MFMA results completed during the first half: no pipeline drain) and the
                        // address + accumulator read of tile t (used a whole block later).  Store(t-1) precedes
                        // read(t) in program order: consecutive tiles may hit the same accumulator row.
                        if (RGCN_ABL & 2) {
                            if (step >= 1) asm volatile("" ::"v"(tl[step - 1].y[0]), "v"(tl[step - 1].z[0]), "v"(ops[step - 1].w1), "v"(ops[step - 1].d1));
                        } else {
                            if (step >= 1) stage_c_t(ops[step - 1], tl[step - 1]);
                            if (step < NRT) stage_b_t(ops[step], tl[step]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (step < NRT) stage_a_t(ops[step], tl[step], 1);
                        __builtin_amdgcn_sched_barrier(0);
                        continue;
                    }
                    if (step + 1 < NRT) {
                        load_ops(ops[step + 1], step + 1, tr_c);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (step < NRT) stage_a(ops[step], tl[step]);
                    if (step >= 2) stage_c(tl[step - 2]);
                    if (step >= 1 && step - 1 < NRT) stage_b(ops[step - 1], tl[step - 1], (flags >> (step - 1)) & 1);
                }
            };
            using std::integral_constant;
            if (flags == 0) {
                switch (nrt) {
                    case 1: consume(integral_constant<int, 1>{}, std::true_type{}); break;
                    case 2: consume(integral_constant<int, 2>{}, std::true_type{}); break;
                    case 3: consume(integral_constant<int, 3>{}, std::true_type{}); break;
                    case 4: consume(integral_constant<int, 4>{}, std::true_type{}); break;
                    case 5: if constexpr (CH > 64) consume(integral_constant<int, 5>{}, std::true_type{}); break;
                    case 6: if constexpr (CH > 64) consume(integral_constant<int, 6>{}, std::true_type{}); break;
                    case 7: if constexpr (CH > 64) consume(integral_constant<int, 7>{}, std::true_type{}); break;
                    case 8: if constexpr (CH > 64) consume(integral_constant<int, 8>{}, std::true_type{}); break;
                    default: break;
                }
            } else {
                switch (nrt) {
                    case 1: consume(integral_constant<int, 1>{}, std::false_type{}); break;
                    case 2: consume(integral_constant<int, 2>{}, std::false_type{}); break;
                    case 3: consume(integral_constant<int, 3>{}, std::false_type{}); break;
                    case 4: consume(integral_constant<int, 4>{}, std::false_type{}); break;
                    default: break;
                }
            }
            }   // part
            const int nrt = nrt_all;
            STAMP(t2);
            if (swap_b) {
                if constexpr (kAsmPrefetch) wait_vmcnt<0>();   // the asm prefetch (this wave's only vector-memory traffic)
#pragma unroll
                for (int s = 0; s < SL; ++s)
#pragma unroll
                    for (int j = 0; j < KT; ++j) bcur[s][j] = bnext[s][j];
#ifdef RGCN_STAMPS
                asm volatile("" ::"v"(bcur[0][0][0]), "v"(bcur[SL - 1][KT - 1][3]));
#endif
            }
            rel_cur = rel_next;
            if constexpr (kAsmPrefetch) {      // fragments of chunk it + 2, issued while the memory queue is idle
                pending = active && it + 2 < nch && rel_next2 != rel_next && !(RGCN_DBG(a) & 4);
                if (pending) prefetch_rel(rel_next2);
            }
            STAMP(t3);
            wg_barrier();
            STAMP(t4);
            if (it + 1 == tend && it + 1 < nch) {
                // this chunk closed a tile: store it and reset the accumulator (the 256 consumer threads; the producers
                // wait at the same extra barrier with the next tile's first chunk landed and the second one on its way)
                tile_epilogue<LDO, true>(a, out_lds, tile_cur, tid - 64 * kTileProducers, kTileThreads - 64 * kTileProducers);
                ++tile_cur;
                tend = ldc(a.tile_ptr, tile_cur + 1) - c0;
                // retire the epilogue's loads and stores in the compiler's scoreboard HERE (once per tile): left pending
                // across the back edge they put an s_waitcnt vmcnt(0) at the top of EVERY iteration (first reuse of a
                // register the mask loads had written), which also waits for the B prefetch issued a moment earlier
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                wg_barrier();
            }
            STAMP_ADD(st_scal, t0, t1);
            STAMP_ADD(st_comp, t1, t2);
            STAMP_ADD(st_bwait, t2, t3);
            STAMP_ADD(st_bar, t3, t4);
#ifdef RGCN_STAMPS
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (nrt == i + 1) {
                    st_nrt[i] += t2 - t1;
                    st_cnt[i] += 1;
                }
#endif
        }
#ifdef RGCN_STAMPS
        if (g_stamps && cwv == 0 && lane == 0) {
            unsigned long long* o = g_stamps + (size_t)blockIdx.x * 32;
            o[0] = st_scal; o[1] = st_comp; o[2] = st_bwait; o[3] = st_bar;
            for (int i = 0; i < 8; ++i) { o[8 + i] = st_nrt[i]; o[16 + i] = st_cnt[i]; }
        }
#endif
        // tell the waitcnt pass that no consumer load is pending when the producer code (next in program
        // order) reuses these registers; otherwise it waits vmcnt(0) between the prologue DMAs
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
    if (wave < kTileProducers) tile_producer_loop<KP, NBUF, BUF, CH>(a, ring, wring, dring, c0, nch, lane, wave, tile0);
    tile_epilogue<LDO, false>(a, out_lds, tile1 - 1, tid, kTileThreads);
}

// ------------------------------------------------------------------------------------------------
// forward / dX kernel, split-precision form (64 -> 64, 128-slot chunks, layout-1 plans)
// ------------------------------------------------------------------------------------------------
// Same producers, ring, accumulator tile and epilogue as rgcn_tile_kernel; different consumers.
//
// The exact fp32 MFMA runs at the fp32 vector rate (1/16 of the bf16 rate) and shares the SIMD's FMA pipe with every
// vector instruction, so rgcn_tile_kernel is bound by its own contraction.  Here the contraction is
// x W = (xh + xm + xl)(Wh + Wm + Wl) with bf16 pieces (h = bf16(v), m = bf16(v - h), l = bf16(v - h - m): 24 significant
// bits, i.e. v = h + m + l up to the last fp32 bit) and the SIX products hh, hm, mh, hl, lh, mm on v_mfma_f32_16x16x32_bf16
// (bf16 x bf16 products are exact in the fp32 accumulator; the dropped ml, lm, ll terms are below 2^-24 relative):
// measured error against float64 no larger than the sequential fp32 chain's (tools/debug/bf16_split_error.py:
// 1.35e-6 against 1.86e-6 max on the headline distribution).  W is split once, at pack time (rgcn_pack3_kernel).
// x is split ON THE FLY in registers -- 88 vector instructions for a lane's 16 elements -- which only pays if a row
// is split once: so a consumer wave owns 32 output columns of HALF of a chunk's rows (2 x 2 ownership), which needs
// the two 64-slot halves of a chunk to scatter into disjoint accumulator rows: plan layout 1 (plan.split_placement).
// Waves 4,5 take slots [0, 64) (columns 0..31 / 32..63), waves 6,7 slots [64, 128).  A chunk whose halves share a
// destination (chunk_flags bit 8) is done by waves 4,5 alone.
// Per 16-row tile and wave: 4 ds_read_b128 (+2 b32), the split, 24 MFMAs (2 column tiles x 2 k-steps x 6 products),
// 2 accumulator read-modify-writes of 16 bytes per lane.  Row tiles with a repeated destination take the Y
// orientation (operands swapped) and the same run-sum product P.Y as rgcn_tile_kernel, in exact fp32.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kPack3FragsPerRel = 2 * 3 * 2 * 2;        // [column half c][plane][column tile ct][k-step s]
constexpr size_t kPack3FloatsPerRel = (size_t)kPack3FragsPerRel * 64 * 4;   // 64 lanes x 16 bytes per fragment

// packed3[((((rel * 2 + c) * 3 + pl) * 2 + ct) * 2 + s) * 64 + lane] (16 bytes = 8 bf16): element j =
// plane pl of B_rel[k = 32 s + 8 (lane >> 4) + j][col = 32 c + 16 ct + (lane & 15)] -- the A operand of the Y^T product
// (A[row = column][k]) and, read the other way round, the B operand of the Y product (B[k][col]).
__device__ __forceinline__ unsigned bf16_rne(float v) {
    const unsigned u = __float_as_uint(v);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__global__ void rgcn_pack3_kernel(const float* __restrict__ weight, const float* __restrict__ root, int num_rel, int din,
                                  int dout, int transpose, uint4* __restrict__ packed) {
    const long total = (long)(num_rel + 1) * kPack3FragsPerRel * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long rem = idx;
        const int lane = (int)(rem & 63); rem >>= 6;
        const int s = (int)(rem & 1); rem >>= 1;
        const int ct = (int)(rem & 1); rem >>= 1;
        const int pl = (int)(rem % 3); rem /= 3;
        const int c = (int)(rem & 1); rem >>= 1;
        const int rel = (int)rem;
        const float* m = rel < num_rel ? weight + (size_t)rel * din * dout : root;
        unsigned h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * s + 8 * (lane >> 4) + j, col = 32 * c + 16 * ct + (lane & 15);
            float v = 0.f;
            if (m != nullptr) {
                if (!transpose) {
                    if (k < din && col < dout) v = m[(size_t)k * dout + col];
                } else {
                    if (k < dout && col < din) v = m[(size_t)col * dout + k];
                }
            }
            unsigned b = bf16_rne(v);
            for (int q = 0; q < pl; ++q) {
                v -= __uint_as_float(b << 16);
                b = bf16_rne(v);
            }
            h[j] = b;
        }
        packed[idx] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    }
}

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {      // RNE, lo -> bits 0..15, hi -> bits 16..31
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

template <bool BUF>
__global__ void __launch_bounds__(kTileThreads, RGCN_TILE_WAVES) rgcn_tile3_kernel(const TileArgs a) {
    constexpr int KP = 64, NP = 64, NBUF = 2, CH = 128;
    constexpr int LDO = kAccStride<NP>;
    static_assert(kTileProducers == 4, "four producer waves, four consumer waves");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* out_lds = lds;                         // [tile + 1][LDO]  (row `tile`: dummy)
    float* ring = lds + (a.tile + 1) * LDO;       // [NBUF][CH][KP]
    float* wring = ring + NBUF * CH * KP;         // [NBUF][CH]
    int* dring = (int*)(wring + NBUF * CH);       // [NBUF][CH]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int c0 = ldc(a.tile_ptr, tile);
    const int nch = ldc(a.tile_ptr, tile + 1) - c0;

    for (int i = tid; i < (a.tile + 1) * LDO; i += kTileThreads) {
        const int col = i % LDO;
        out_lds[i] = (a.bias != nullptr && col < a.dout) ? a.bias[col] : 0.f;
    }

    if (wave >= kTileProducers) {
        const int cwv = wave - kTileProducers;
        const int pair = cwv >> 1;                 // which 64-slot half of a chunk
        const int ch = cwv & 1;                    // which 32 output columns
        const int rowl = lane & 15, kq = lane >> 4;
        const unsigned col4_bytes = (unsigned)(32 * ch + 4 * kq) * 4u;     // Y^T layout: four consecutive columns of row rowl
        const unsigned col1_bytes = (unsigned)(32 * ch + rowl) * 4u;       // Y layout: column rowl of rows 4 kq + i
        const uint4* wp4 = (const uint4*)a.wp;
        f32x4 wcur[3][2][2], wnext[3][2][2];       // [plane][column tile][k-step]: 8 bf16 each, kept as 16-byte registers
        int rel_cur = ldc(a.chunk_rel, c0);
        {
            const uint4* bp = wp4 + (size_t)(rel_cur * 2 + ch) * (3 * 2 * 2 * 64) + lane;
#pragma unroll
            for (int f = 0; f < 12; ++f) wcur[f / 4][(f >> 1) & 1][f & 1] = __builtin_bit_cast(f32x4, bp[f * 64]);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): retire these loads in the compiler's scoreboard (see rgcn_tile_kernel)
        int cnt_pre = ldc(a.chunk_cnt, c0);
        int flags_pre = ldc(a.chunk_flags, c0);
        // next relation's fragments: prefetched at the END of an iteration, while the CU's memory queue is idle (rgcn_tile_kernel)
        int rel_n1 = nch > 1 ? ldc(a.chunk_rel, c0 + 1) : rel_cur;
        int rel_n2 = nch > 2 ? ldc(a.chunk_rel, c0 + 2) : rel_n1;
        auto prefetch_rel = [&](int rel) {
            const f32x4* bp = (const f32x4*)(wp4 + (size_t)(rel * 2 + ch) * (3 * 2 * 2 * 64) + lane);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const f32x4* bq = bp + q * 4 * 64;
                prefetch16<0>(wnext[q][0][0], bq);
                prefetch16<1024>(wnext[q][0][1], bq);
                prefetch16<2048>(wnext[q][1][0], bq);
                prefetch16<3072>(wnext[q][1][1], bq);
            }
        };
        bool pending = rel_n1 != rel_cur;
        if (pending) prefetch_rel(rel_n1);
        wg_barrier();
#ifdef RGCN_STAMPS
        unsigned long long st_scal = 0, st_comp = 0, st_bwait = 0, st_bar = 0;
        unsigned long long st_pro = 0, st_top = 0, st_mid = 0, st_acc = 0;
#endif
        for (int it = 0; it < nch; ++it) {
            STAMP(t0);
            const int chunk = c0 + it;
            const int buf = it % NBUF;
            const int cnt = cnt_pre;
            const int flags = flags_pre;
            const int rel_next = rel_n1;
            const int rel_next2 = rel_n2;
            if (it + 1 < nch) {
                cnt_pre = ldc(a.chunk_cnt, chunk + 1);
                flags_pre = ldc(a.chunk_flags, chunk + 1);
            }
            rel_n1 = rel_n2;
            if (it + 3 < nch) rel_n2 = ldc(a.chunk_rel, chunk + 3);
#ifdef RGCN_STAMPS
            asm volatile("" ::"s"(cnt), "s"(rel_next));
#endif
            STAMP(t1);
            const bool swap_b = pending;
            // this wave's row tiles of the chunk: the used tiles are contiguous from tile 0; layout 1 puts a chunk of more
            // than 64 rows into tiles 0..3 (half 0) and 4.. (half 1) with disjoint destinations
            const int nt = cnt >> 4;
            const bool straddle = (flags & 256) != 0;
            int first, n_my;
            if (straddle) {
                first = 0;
                n_my = pair == 0 ? nt : 0;
            } else {
                first = 4 * pair;
                n_my = pair == 0 ? (nt < 4 ? nt : 4) : (nt > 4 ? nt - 4 : 0);
            }
            const float* hb = ring + buf * CH * KP;
            const float* wb = wring + buf * CH;
            const int* db = dring + buf * CH;
            // operand addresses of row tile 0 of the chunk; tile t is t * 16 rows further
            const float* arow[2][2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int pos = (8 * s2 + 2 * kq + h2) ^ swizzle<kRowRead, KP / 4>(rowl);
                    arow[s2][h2] = hb + rowl * KP + pos * 4;
                }
            struct Ops {
                f32x4 x[2][2];   // this lane's 16 elements of row rowl: k = 32 s + 8 kq + 4 h + (0..3)
                float w1; int d1;            // Y^T path: weight / run metadata of row rowl
                f32x4 w4; i32x4 d4;          // Y path: of rows 4 kq + i
            };
            auto load_ops = [&](Ops& o, int t) {
                o.w1 = wb[t * 16 + rowl];
                o.d1 = db[t * 16 + rowl];
                if ((flags >> t) & 1) {      // only the run-sum path needs the per-row-of-this-lane copies
                    o.w4 = *(const f32x4*)(wb + t * 16 + 4 * kq);
                    o.d4 = *(const i32x4*)(db + t * 16 + 4 * kq);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) o.x[s2][h2] = *(const f32x4*)(arow[s2][h2] + t * 16 * KP);
            };
            auto acc_ptr = [&](int d, unsigned col_bytes) -> float* {
                return (float*)((char*)out_lds + (__umul24((unsigned)d, (unsigned)(LDO * 4)) + col_bytes));
            };
            auto process = [&](const Ops& o, int t) {
                // ---- 3-way split of the lane's 16 elements: pl[plane][k-step] ----
                bf16x8 pl[3][2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    unsigned hh[4], mm[4], ll[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        float x0 = o.x[s2][d >> 1][2 * (d & 1)], x1 = o.x[s2][d >> 1][2 * (d & 1) + 1];
                        const unsigned h = cvt_pk_bf16(x0, x1);
                        x0 -= __uint_as_float(h << 16);
                        x1 -= __uint_as_float(h & 0xFFFF0000u);
                        const unsigned m = cvt_pk_bf16(x0, x1);
                        x0 -= __uint_as_float(m << 16);
                        x1 -= __uint_as_float(m & 0xFFFF0000u);
                        hh[d] = h;
                        mm[d] = m;
                        ll[d] = cvt_pk_bf16(x0, x1);
                    }
                    pl[0][s2] = __builtin_bit_cast(bf16x8, make_uint4(hh[0], hh[1], hh[2], hh[3]));
                    pl[1][s2] = __builtin_bit_cast(bf16x8, make_uint4(mm[0], mm[1], mm[2], mm[3]));
                    pl[2][s2] = __builtin_bit_cast(bf16x8, make_uint4(ll[0], ll[1], ll[2], ll[3]));
                }
                constexpr int px[6] = {0, 0, 1, 0, 2, 1}, pw[6] = {0, 1, 0, 2, 0, 1};     // x plane, W plane: hh hm mh hl lh mm
                const bool dup = (flags >> t) & 1;
                if (!dup) {
                    // Y^T = W^T H^T: a lane ends up with four consecutive columns of ONE row
                    float* dst = acc_ptr(o.d1, col4_bytes);
                    f32x4 old[2], y[2];
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        old[ct] = *(const f32x4*)(dst + 16 * ct);
                        y[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct)
                                y[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wcur[pw[q]][ct][s2]), pl[px[q]][s2], y[ct], 0, 0, 0);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) *(f32x4*)(dst + 16 * ct) = y[ct] * o.w1 + old[ct];
                } else {
                    // Y = H W (operands swapped), then the run-sum product Z = P Y + old in exact fp32 (rgcn_tile_kernel)
                    f32x4 y[2];
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) y[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct)
                                y[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl[px[q]][s2], __builtin_bit_cast(bf16x8, wcur[pw[q]][ct][s2]), y[ct], 0, 0, 0);
                    float* dst[4];
                    float pm[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        dst[i] = acc_ptr(o.d4[i], col1_bytes);
                        pm[i] = ((unsigned)o.d4[i] >> 24) == (unsigned)rowl ? o.w4[i] : 0.f;
                    }
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        f32x4 old;
#pragma unroll
                        for (int i = 0; i < 4; ++i) old[i] = dst[i][16 * ct];
                        f32x4 z1 = {0.f, 0.f, 0.f, 0.f};
                        f32x4 z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pm[0], y[ct][0], old, 0, 0, 0);
                        z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pm[1], y[ct][1], z1, 0, 0, 0);
                        z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pm[2], y[ct][2], z0, 0, 0, 0);
                        z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pm[3], y[ct][3], z1, 0, 0, 0);
                        const f32x4 v = z0 + z1;
#pragma unroll
                        for (int i = 0; i < 4; ++i) dst[i][16 * ct] = v[i];
                    }
                }
            };
            // ---- fast path: none of this wave's row tiles repeats a destination (nearly all chunks) -------------------
            // Straight-line over the wave's NT tiles, software-pipelined by hand: the 88-instruction split of tile t + 1
            // is issued BETWEEN the 24 MFMAs of tile t (a bf16 MFMA holds the SIMD's issue port for half of its 16
            // cycles only), the operand reads of tile t + 2 and the accumulator read of tile t go out before them.
            auto split3 = [&](const Ops& o, bf16x8 (&pl)[3][2]) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    unsigned hh[4], mm[4], ll[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        float x0 = o.x[s2][d >> 1][2 * (d & 1)], x1 = o.x[s2][d >> 1][2 * (d & 1) + 1];
                        const unsigned h = cvt_pk_bf16(x0, x1);
                        x0 -= __uint_as_float(h << 16);
                        x1 -= __uint_as_float(h & 0xFFFF0000u);
                        const unsigned m = cvt_pk_bf16(x0, x1);
                        x0 -= __uint_as_float(m << 16);
                        x1 -= __uint_as_float(m & 0xFFFF0000u);
                        hh[d] = h;
                        mm[d] = m;
                        ll[d] = cvt_pk_bf16(x0, x1);
                    }
                    pl[0][s2] = __builtin_bit_cast(bf16x8, make_uint4(hh[0], hh[1], hh[2], hh[3]));
                    pl[1][s2] = __builtin_bit_cast(bf16x8, make_uint4(mm[0], mm[1], mm[2], mm[3]));
                    pl[2][s2] = __builtin_bit_cast(bf16x8, make_uint4(ll[0], ll[1], ll[2], ll[3]));
                }
            };
            auto consume3 = [&](auto nt_c) {
                constexpr int NT = decltype(nt_c)::value;
                constexpr int px[6] = {0, 0, 1, 0, 2, 1}, pw[6] = {0, 1, 0, 2, 0, 1};     // x plane, W plane: hh hm mh hl lh mm
                Ops o[2];
                bf16x8 pl[2][3][2];
                STAMP(q0);
                load_ops(o[0], first);
                if constexpr (NT > 1) load_ops(o[1], first + 1);
                __builtin_amdgcn_sched_barrier(0);
                split3(o[0], pl[0]);
                __builtin_amdgcn_sched_barrier(0);
#ifdef RGCN_STAMPS
                asm volatile("" ::"v"(pl[0][0][0]), "v"(pl[0][2][1]));
#endif
                STAMP(q1);
                STAMP_ADD(st_pro, q0, q1);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int cur = t & 1, nxt = cur ^ 1;
                    STAMP(r0);
                    // accumulator contents (after tile t - 1's store in program order: consecutive tiles may share a row)
                    float* dst = acc_ptr(o[cur].d1, col4_bytes);
                    const float w1 = o[cur].w1;
                    f32x4 old[2], y[2];
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        old[ct] = *(const f32x4*)(dst + 16 * ct);
                        y[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    if (t + 2 < NT) load_ops(o[cur], first + t + 2);       // its fp32 contents were split an iteration ago
                    __builtin_amdgcn_sched_barrier(0);
#ifdef RGCN_STAMPS
                    unsigned long long r1;      // (no lgkmcnt drain here: the reads just issued must stay in flight)
                    asm volatile("s_memtime %0" : "=s"(r1)::"memory");
#endif
                    if (t + 1 < NT) {
                        if (RGCN_ABL & 2) {        // diagnostic: no split (opaque copies keep the operand reads alive)
#pragma unroll
                            for (int a2 = 0; a2 < 3; ++a2)
#pragma unroll
                                for (int b2 = 0; b2 < 2; ++b2) {
                                    f32x4 v = o[nxt].x[b2][a2 & 1];
                                    asm volatile("" : "+v"(v));
                                    pl[nxt][a2][b2] = __builtin_bit_cast(bf16x8, v);
                                }
                        } else {
                            split3(o[nxt], pl[nxt]);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct) {
                                if (RGCN_ABL & 1) {    // diagnostic: no MFMA
                                    y[ct] += __builtin_bit_cast(f32x4, pl[cur][px[q]][s2]);
                                    continue;
                                }
                                y[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wcur[pw[q]][ct][s2]),
                                                                                pl[cur][px[q]][s2], y[ct], 0, 0, 0);
                            }
                    if (t + 1 < NT && RGCN_ABL == 0) {
#pragma unroll
                        for (int i = 0; i < 22; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // 4 VALU of the next tile's split
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#ifdef RGCN_STAMPS
                    asm volatile("" ::"v"(y[0]), "v"(y[1]));
                    unsigned long long r2;
                    asm volatile("s_nop 7\n\ts_memtime %0" : "=s"(r2)::"memory");
#endif
                    if (RGCN_ABL & 4) {            // diagnostic: no accumulator store
                        asm volatile("" ::"v"(y[0]), "v"(y[1]), "v"(old[0]), "v"(old[1]));
                    } else {
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) *(f32x4*)(dst + 16 * ct) = y[ct] * w1 + old[ct];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    STAMP(r3);
#ifdef RGCN_STAMPS
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    st_top += r1 - r0; st_mid += r2 - r1; st_acc += r3 - r2;
#endif
                }
            };
            const int myflags = straddle ? (flags & 0xFF) : ((flags >> first) & 15);
            using std::integral_constant;
            if (n_my > 0 && myflags == 0 && n_my <= 4) {
                switch (n_my) {
                    case 1: consume3(integral_constant<int, 1>{}); break;
                    case 2: consume3(integral_constant<int, 2>{}); break;
                    case 3: consume3(integral_constant<int, 3>{}); break;
                    default: consume3(integral_constant<int, 4>{}); break;
                }
            } else if (n_my > 0) {
                Ops o0, o1;
                load_ops(o0, first);
                for (int t = 0; t < n_my; t += 2) {
                    if (t + 1 < n_my) load_ops(o1, first + t + 1);
                    process(o0, first + t);
                    if (t + 1 < n_my) {
                        if (t + 2 < n_my) load_ops(o0, first + t + 2);
                        process(o1, first + t + 1);
                    }
                }
            }
            STAMP(t2);
            if (swap_b) {
                wait_vmcnt<0>();      // the asm prefetch (this wave's only vector-memory traffic)
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) wcur[q][ct][s2] = wnext[q][ct][s2];
            }
            rel_cur = rel_next;
            pending = it + 2 < nch && rel_next2 != rel_next;
            if (pending) prefetch_rel(rel_next2);
#ifdef RGCN_STAMPS
            asm volatile("" ::"v"(wcur[0][0][0]), "v"(wcur[2][1][1]));
#endif
            STAMP(t3);
            wg_barrier();
            STAMP(t4);
            STAMP_ADD(st_scal, t0, t1);
            STAMP_ADD(st_comp, t1, t2);
            STAMP_ADD(st_bwait, t2, t3);
            STAMP_ADD(st_bar, t3, t4);
        }
#ifdef RGCN_STAMPS
        if (g_stamps && (cwv == 0 || cwv == 2) && lane == 0) {
            unsigned long long* o = g_stamps + (size_t)blockIdx.x * 32 + (cwv == 2 ? 8 : 0);
            o[0] = st_scal; o[1] = st_comp; o[2] = st_bwait; o[3] = st_bar;
            if (cwv == 0) { o[16] = st_pro; o[17] = st_top; o[18] = st_mid; o[19] = st_acc; }
        }
#endif
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see rgcn_tile_kernel
    }
    if (wave < kTileProducers) tile_producer_loop<KP, NBUF, BUF, CH>(a, ring, wring, dring, c0, nch, lane, wave, tile);
    tile_epilogue<LDO, false>(a, out_lds, tile, tid, kTileThreads);
}

// dz = da * act'(a) for an activation fused into rgcn_fwd's store (a = act(z)): relu -> (a > 0), sigmoid -> a (1 - a).
// 16 bytes per lane, grid-stride.  Used where no consumer kernel can fold the mask (rgcn_bwd_dx's `relu_of`).
__global__ void rgcn_act_backward_kernel(const float* __restrict__ av, const float* __restrict__ da, float* __restrict__ dz,
                                         long rows, int ld4, int act) {
    const long total = rows * ld4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const f32x4 y = ((const f32x4*)av)[i];
        f32x4 g = ((const f32x4*)da)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) g[c] = act == RGCN_ACT_RELU ? (y[c] > 0.f ? g[c] : 0.f) : g[c] * y[c] * (1.f - y[c]);
        ((f32x4*)dz)[i] = g;
    }
}

// ------------------------------------------------------------------------------------------------
// weight-gradient kernel
// ------------------------------------------------------------------------------------------------
constexpr int kDwSlabsPer = 4;  // partial slabs per (workgroup, relation): one per consumer wave in the wide kernel
constexpr int kWideConsumers = 4;  // wide dW kernel: one consumer wave per SIMD

struct DwArgs {
    const int* rel_order;
    const int* chunk_rel;
    const int* chunk_cnt;
    const int* chunk_tile;
    const int* slot_src;
    const float* slot_w;
    const int* slot_row;
    const float* x;
    const float* g;
    unsigned x_bytes, g_bytes;
    int n_rows, n_owned;  // rows of x / of g (padding slots gather the row one past the end)
    float* slabs;      // [(nblocks + R' + 1) * 4][KP*NP]
    float* bias_slabs; // [nblocks * 4][NP]
    int ldx, din4, ldg, dout4, tile, n_units, num_rel;
    int ushift;        // log2(units per chunk): unit u belongs to chunk u >> ushift, rows 64 * (u & mask) .. + 63 of it
};

template <int KP, int NP, int NBUF, bool BUF>
__global__ void __launch_bounds__(kThreads, 2) rgcn_dw_kernel(const DwArgs a) {
    constexpr int MT = KP / 16, NT = NP / 16;
    constexpr int D = NBUF - 1;
    constexpr int NSL = NT < 4 ? 1 : NT / 4;              // n-slices per consumer wave
    constexpr int RWM = NT < 4 ? 4 / NT : 1;              // consumer waves across m-tiles
    constexpr int MTW = (MT + RWM - 1) / RWM;             // m-tiles per consumer wave
    constexpr int LPRH = KP / 4, LPRG = NP / 4;

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ringh = lds;                                   // [NBUF][64][KP]
    float* ringg = ringh + NBUF * kChunk * KP;            // [NBUF][64][NP]
    float* wring = ringg + NBUF * kChunk * NP;            // [NBUF][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = gridDim.x, b = blockIdx.x;
    const int i0 = (int)((long)b * a.n_units / nb);
    const int i1 = (int)((long)(b + 1) * a.n_units / nb);
    const int nch = i1 - i0;
    if (nch <= 0) return;

    const int cwv = wave - kProducerWaves;
    const int rowl = lane & 15, kq = lane >> 4;
    const int ntb = NT < 4 ? cwv % NT : cwv;              // first n-slice of this wave (then +4 per s)
    const int mtb = NT < 4 ? cwv / NT : 0;                // first m-tile (then +RWM per i)
    f32x4 acc[NSL][MTW];
    float bsum[NSL];
    int rel_cur = -1;

    auto zero_acc = [&]() {
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            bsum[s] = 0.f;
#pragma unroll
            for (int i = 0; i < MTW; ++i) acc[s][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto flush = [&]() {
        float* slab = a.slabs + (size_t)(b + rel_cur) * kDwSlabsPer * (KP * NP);   // sub-slab 0
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            const int nt = ntb + 4 * s;
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                const int mt = mtb + RWM * i;
                if (mt < MT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[(16 * mt + 4 * kq + r) * NP + 16 * nt + rowl] = acc[s][i][r];
                }
            }
            if (rel_cur == a.num_rel && mtb == 0) {
                float v = bsum[s];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (kq == 0) a.bias_slabs[(size_t)b * kDwSlabsPer * NP + 16 * nt + rowl] = v;
            }
        }
    };

    // separate role loops, consumers first in program order (see rgcn_tile_kernel)
    if (wave >= kProducerWaves) {
        zero_acc();
        // chunk metadata one iteration ahead (three dependent scalar loads per chunk otherwise)
        // the walk is over 64-row UNITS (rel_order); a unit's metadata is its chunk's
        auto unit_cnt = [&](int unit) {
            const int c = ldc(a.chunk_cnt, unit >> a.ushift) - kChunk * (unit & ((1 << a.ushift) - 1));
            return c < kChunk ? c : kChunk;
        };
        // two-deep: the unit id is fetched TWO iterations ahead and its metadata one ahead, so no scalar load waits
        // for another one issued in the same iteration (that dependent round trip was ~450 cycles per chunk)
        int chunk_pre = ldc(a.rel_order, i0);
        int cnt_pre = unit_cnt(chunk_pre);
        int relv_pre = ldc(a.chunk_rel, chunk_pre >> a.ushift);
        int unit_next = ldc(a.rel_order, i0 + (nch > 1 ? 1 : 0));
        wg_barrier();
#ifdef RGCN_STAMPS
        unsigned long long st_scal = 0, st_comp = 0, st_bar = 0;
#endif
        for (int it = 0; it < nch; ++it) {
            STAMP(t0);
            const int buf = it % NBUF;
            const int cnt = cnt_pre;
            const int rel = relv_pre;
            if (it + 1 < nch) {
                cnt_pre = unit_cnt(unit_next);
                relv_pre = ldc(a.chunk_rel, unit_next >> a.ushift);
            }
            unit_next = ldc(a.rel_order, i0 + (it + 2 < nch ? it + 2 : nch - 1));
            STAMP(t1);
            if (rel != rel_cur) {
                if (rel_cur >= 0) flush();
                zero_acc();
                rel_cur = rel;
            }
            const bool is_root = rel == a.num_rel;
            const float* hb = ringh + buf * kChunk * KP;
            const float* gb = ringg + buf * kChunk * NP;
            const float* wb = wring + buf * kChunk;
            // 16 rows (4 MFMA k-steps) per group; operands of the NEXT group are read from LDS before the
            // current group's MFMAs.  Rows beyond cnt were DMA'd as zeros (w = 0 too): no masking.
            struct Grp {
                float av[4][MTW];
                float gv[4][NSL];
                float wv[4];
            };
            auto load_grp = [&](Grp& o, int grp) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = 16 * grp + 4 * t + kq;
                    const int swh = swizzle<kColRead, LPRH>(row), swg = swizzle<kColRead, LPRG>(row);
                    o.wv[t] = wb[row];
#pragma unroll
                    for (int s = 0; s < NSL; ++s) {
                        const int col = 16 * (ntb + 4 * s) + rowl;
                        o.gv[t][s] = gb[row * NP + (((col >> 2) ^ swg) << 2) + (col & 3)];
                    }
#pragma unroll
                    for (int i = 0; i < MTW; ++i) {
                        const int mt = mtb + RWM * i;
                        const int col = 16 * (mt < MT ? mt : 0) + rowl;
                        o.av[t][i] = hb[row * KP + (((col >> 2) ^ swh) << 2) + (col & 3)];
                    }
                }
            };
            auto compute_grp = [&](const Grp& o) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float bv[NSL];
#pragma unroll
                    for (int s = 0; s < NSL; ++s) {
                        if (is_root) bsum[s] += o.gv[t][s];
                        bv[s] = o.gv[t][s] * o.wv[t];
                    }
#pragma unroll
                    for (int i = 0; i < MTW; ++i) {
                        if (mtb + RWM * i < MT) {
#pragma unroll
                            for (int s = 0; s < NSL; ++s)
                                acc[s][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.av[t][i], bv[s], acc[s][i], 0, 0, 0);
                        }
                    }
                }
            };
            const int ngrp = (cnt + 15) >> 4;
            constexpr bool kPrefetch = MTW * NSL < 16;   // 128x128: the second operand set would spill
            Grp grp[kPrefetch ? 2 : 1];
            load_grp(grp[0], 0);
#pragma unroll
            for (int gi = 0; gi < kChunk / 16; ++gi) {
                if (gi < ngrp) {
                    if (kPrefetch) {
                        if (gi + 1 < ngrp) load_grp(grp[(gi + 1) & 1], gi + 1);
                        compute_grp(grp[gi & 1]);
                    } else {
                        if (gi > 0) load_grp(grp[0], gi);
                        compute_grp(grp[0]);
                    }
                }
            }
            STAMP(t2);
            wg_barrier();
            STAMP(t3);
            STAMP_ADD(st_scal, t0, t1);
            STAMP_ADD(st_comp, t1, t2);
            STAMP_ADD(st_bar, t2, t3);
        }
#ifdef RGCN_STAMPS
        if (g_stamps && cwv == 0 && lane == 0) {
            unsigned long long* o = g_stamps + (size_t)blockIdx.x * 32;
            o[0] = st_scal; o[1] = st_comp; o[2] = 0; o[3] = st_bar;
        }
#endif
        if (rel_cur >= 0) flush();
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see rgcn_tile_kernel
    }
    if (wave < kProducerWaves) {
        // The producers' few instructions must not queue behind the consumer wave's MFMAs on the shared SIMD
        // (issue is arbitrated by priority, then age; an fp32 MFMA holds the pipe 32 cycles): RGCN_PRIO
        __builtin_amdgcn_s_setprio(RGCN_PRIO);
        // producers: wave (k % 4) owns chunk k of this workgroup's range (see rgcn_tile_kernel)
        const int pw = wave;
        int knext = pw;
        int idx_h = 0, idx_g = 0;
        RowGather<KP, kColRead, BUF> gather_h;
        RowGather<NP, kColRead, BUF> gather_g;
        gather_h.init(lane, a.din4, a.ldx);
        gather_g.init(lane, a.dout4, a.ldg);
        // raw index loads for the wave's next chunk; combined into row ids only at its next turn
        auto load_idx = [&](int k) {
            const int kk = k < nch ? k : nch - 1;
            const int chunk = ldc(a.rel_order, i0 + kk);
            idx_h = a.slot_src[(size_t)chunk * kChunk + lane];
            idx_g = a.slot_row[(size_t)chunk * kChunk + lane];
        };
        load_idx(knext);
        auto issue = [&](int k) {
            const int chunk = ldc(a.rel_order, i0 + k), buf = k % NBUF;
            gather_h.issue(a.x, a.x_bytes, a.n_rows, a.ldx, idx_h, ringh + buf * kChunk * KP);
            gather_g.issue(a.g, a.g_bytes, a.n_owned, a.ldg, idx_g, ringg + buf * kChunk * NP);
            dma4(a.slot_w + (size_t)chunk * kChunk + lane, wring + buf * kChunk);
            knext += kProducerWaves;
            load_idx(knext);
        };
#pragma unroll
        for (int k = 0; k < D; ++k)
            if (k % kProducerWaves == pw && k < nch) issue(k);
        if (pw == 0) wait_vmcnt<0>();
        wg_barrier();
#ifdef RGCN_STAMPS
        unsigned long long sp_issue = 0, sp_wait = 0, sp_bar = 0;
#endif
        for (int it = 0; it < nch; ++it) {
            const int ki = it + D, kw = it + 1;
            STAMP(p0);
            if (ki % kProducerWaves == pw && ki < nch) issue(ki);
            STAMP(p1);
            if (kw % kProducerWaves == pw && kw < nch) wait_vmcnt<0>();
            STAMP(p2);
            wg_barrier();
            STAMP(p3);
            STAMP_ADD(sp_issue, p0, p1);
            STAMP_ADD(sp_wait, p1, p2);
            STAMP_ADD(sp_bar, p2, p3);
        }
        wait_vmcnt<0>();
#ifdef RGCN_STAMPS
        if (g_stamps && lane == 0) {
            unsigned long long* o = g_stamps + (size_t)blockIdx.x * 32;
            if (pw == 0) { o[4] = sp_issue; o[5] = sp_wait; o[6] = sp_bar; }
            if (pw == 1) o[7] = nch;
        }
#endif
    }
}

// ------------------------------------------------------------------------------------------------
// weight-gradient kernel, wide form (KP, NP multiples of 64)
// ------------------------------------------------------------------------------------------------
// Same producers / ring / walk as rgcn_dw_kernel, different consumer decomposition.  There each consumer
// wave owns 16 output columns and reads its MFMA operands element-wise (ds_read_b32 with a swizzled address
// per element: 6 LDS reads + their address arithmetic per 4 MFMAs, and fp32 MFMAs share the SIMD pipe with
// that arithmetic).  Here a wave owns the WHOLE [KP x NP] accumulator and a quarter of the rows:
//   lane (ml = l & 15, kq = l >> 4) reads H[row][64u + 4 ml .. +3] and G[row][64u + 4 ml .. +3] with ONE
//   ds_read_b128 each; component j of the first is the A operand and component j' of the second the B operand
//   of the MFMA whose 16 x 16 output tile is { dW[64u + 4 m' + j][64u' + 4 n' + j'] } -- a strided set of rows
//   and columns, which an outer-product accumulation does not care about.
// 3 LDS reads per 16 (KP = NP = 64) MFMAs.  The four waves' partial sums go to four sub-slabs.
template <int KP, int NP, int NBUF, bool BUF, int CONS>
__global__ void __launch_bounds__(64 * (kProducerWaves + CONS), (kProducerWaves + CONS) / 4) rgcn_dw_wide_kernel(const DwArgs a) {
    constexpr int UA = KP / 64, UB = NP / 64;
    constexpr int TEAMS = CONS / 4;   // consumer teams of 4 waves; team t takes the row groups g with (g + it) % TEAMS == t
    constexpr int NA = 4 * UA, NB = 4 * UB;
    constexpr int D = NBUF - 1;
    static_assert(D >= 1, "ring of at least two slots");
    static_assert(CONS <= kDwSlabsPer, "one partial slab per consumer wave");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    // small arrays first: their LDS addresses stay below 64 KiB, i.e. inside the immediate-offset field of the
    // DS instructions (an address beyond it costs a vector add per access)
    float* wring = lds;                                   // [NBUF][64]; then the index rings [2][2D+1][64]
    float* ringh = wring + (NBUF + 2 * (2 * D + 1)) * kChunk;   // [NBUF][64][KP]
    float* ringg = ringh + NBUF * kChunk * KP;            // [NBUF][64][NP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = gridDim.x, b = blockIdx.x;
    const int i0 = (int)((long)b * a.n_units / nb);
    const int i1 = (int)((long)(b + 1) * a.n_units / nb);
    const int nch = i1 - i0;
    if (nch <= 0) return;

    if (wave >= kProducerWaves) {
        const int cwv = wave - kProducerWaves;   // slab index of this wave
        const int cw = cwv & 3;                  // rows 4*cw + kq of a 16-row group
        const int team = cwv >> 2;
        const int ml = lane & 15, kq = lane >> 4;
        f32x4 acc[NA][NB];
        float bsum[NB];
        int rel_cur = -1;
        auto zero_acc = [&]() {
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
                bsum[jb] = 0.f;
#pragma unroll
                for (int ia = 0; ia < NA; ++ia) acc[ia][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        };
        auto flush = [&]() {
            float* slab = a.slabs + ((size_t)(b + rel_cur) * kDwSlabsPer + cwv) * (KP * NP);
#pragma unroll
            for (int ia = 0; ia < NA; ++ia)
#pragma unroll
                for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int colh = 64 * (ia >> 2) + 4 * (4 * kq + r) + (ia & 3);
                        const int colg = 64 * (jb >> 2) + 4 * ml + (jb & 3);
                        slab[colh * NP + colg] = acc[ia][jb][r];
                    }
            if (rel_cur == a.num_rel) {
#pragma unroll
                for (int jb = 0; jb < NB; ++jb) {
                    float v = bsum[jb];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    if (kq == 0)
                        a.bias_slabs[((size_t)b * kDwSlabsPer + cwv) * NP + 64 * (jb >> 2) + 4 * ml + (jb & 3)] = v;
                }
            }
        };
        zero_acc();
        // the walk is over 64-row UNITS (rel_order); a unit's metadata is its chunk's
        auto unit_cnt = [&](int unit) {
            const int c = ldc(a.chunk_cnt, unit >> a.ushift) - kChunk * (unit & ((1 << a.ushift) - 1));
            return c < kChunk ? c : kChunk;
        };
        // two-deep: the unit id is fetched TWO iterations ahead and its metadata one ahead, so no scalar load waits
        // for another one issued in the same iteration (that dependent round trip was ~450 cycles per chunk)
        int chunk_pre = ldc(a.rel_order, i0);
        int cnt_pre = unit_cnt(chunk_pre);
        int relv_pre = ldc(a.chunk_rel, chunk_pre >> a.ushift);
        int unit_next = ldc(a.rel_order, i0 + (nch > 1 ? 1 : 0));
        wg_barrier();   // producers: index vectors landed
        wg_barrier();   // producers: chunk 0 landed
#ifdef RGCN_STAMPS
        unsigned long long st_scal = 0, st_comp = 0, st_bar = 0;
#endif
        for (int it = 0; it < nch; ++it) {
            STAMP(t0);
            const int buf = it % NBUF;
            const int cnt = cnt_pre;
            const int rel = relv_pre;
            if (it + 1 < nch) {
                cnt_pre = unit_cnt(unit_next);
                relv_pre = ldc(a.chunk_rel, unit_next >> a.ushift);
            }
            unit_next = ldc(a.rel_order, i0 + (it + 2 < nch ? it + 2 : nch - 1));
            STAMP(t1);
            if (rel != rel_cur) {
                if (rel_cur >= 0) flush();
                zero_acc();
                rel_cur = rel;
            }
            const bool is_root = rel == a.num_rel;
            const float* hb = ringh + buf * kChunk * KP + 4 * ml;
            const float* gb = ringg + buf * kChunk * NP + 4 * ml;
            const float* wb = wring + buf * kChunk;
            // group g = rows 16g .. 16g+15; this wave takes rows 16g + 4cw + kq (one MFMA k-step per group).
            // Rows beyond cnt were DMA'd as zeros (w = 0 too): no masking.
            struct Grp {
                f32x4 a4[UA];
                f32x4 g4[UB];
                float wv;
            };
            // this lane's row of group 0; group g is 16 rows further (immediate offsets)
            const float* hrow = hb + (4 * cw + kq) * KP;
            const float* grow = gb + (4 * cw + kq) * NP;
            const float* wrow = wb + 4 * cw + kq;
            auto load_grp = [&](Grp& o, int g) {
                o.wv = wrow[16 * g];
#pragma unroll
                for (int u = 0; u < UA; ++u) o.a4[u] = *(const f32x4*)(hrow + 16 * g * KP + 64 * u);
#pragma unroll
                for (int u = 0; u < UB; ++u) o.g4[u] = *(const f32x4*)(grow + 16 * g * NP + 64 * u);
            };
            // One group = 16 MFMAs accumulating IN PLACE.  The MFMA is issued through inline asm with the accumulator
            // as a tied "+v" operand: with the builtin (destination free to differ from the C operand) hipcc gives the
            // guarded group blocks different accumulator registers and moves all 64 of them at every merge (60+
            // v_mov per chunk, each costing MFMA issue time).  What the compiler therefore does not see is the MFMA
            // result hazard: the accumulators are only read by flush(), a workgroup barrier and a scalar-load round
            // trip after the last MFMA that wrote them.  `next` (when given) is read in between the MFMAs: an LDS
            // instruction there costs ~2 cycles and has the rest of the block to land.
            auto compute_grp = [&](const Grp& o, Grp* next, int gnext) {
                f32x4 bv4[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    bv4[u] = o.g4[u] * o.wv;
                    asm volatile("" : "+v"(bv4[u]));      // all multiplies in ONE group in front of the MFMAs
                }
                // the MFMAs below are inline asm: the compiler does not see a VALU-write -> MFMA-read hazard
                asm volatile("s_nop 4" ::: "memory");
                int n = 0;
#pragma unroll
                for (int ia = 0; ia < NA; ++ia)
#pragma unroll
                    for (int jb = 0; jb < NB; ++jb) {
                        if (RGCN_ABL & 1) {   // diagnostic build: no MFMA
                            acc[ia][jb][0] += o.a4[ia >> 2][ia & 3] * bv4[jb >> 2][jb & 3];
                            continue;
                        }
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0"
                                     : "+v"(acc[ia][jb])
                                     : "v"(o.a4[ia >> 2][ia & 3]), "v"(bv4[jb >> 2][jb & 3]));
                        ++n;
                        if (next != nullptr && n == 2) {
                            __builtin_amdgcn_sched_barrier(0);
                            load_grp(*next, gnext);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            };
            const int ngrp = (cnt + 15) >> 4;
            static_assert(TEAMS == 1, "one team of four consumer waves (two teams were tried: no gain)");
            {
                // Ping-pong operand sets: group g + 1's operands are read INSIDE group g's MFMA block (LDS instructions
                // between MFMAs are nearly free and their round trip is covered).  Four guarded blocks in a row on
                // purpose: with one straight-line variant per group count (a switch), or with nested guards, the
                // register allocator moves the 64 accumulator registers at the merges.  The read one group past the
                // chunk's last is unconditional (no select / copy of the operand set) and harmless: still inside the
                // rings, never used.
                Grp grp[2];
                load_grp(grp[0], 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < kChunk / 16; ++g) {          // unrolled: every LDS offset is an immediate
                    if (g < ngrp) compute_grp(grp[g & 1], g + 1 < kChunk / 16 ? &grp[(g + 1) & 1] : nullptr, g + 1);
                }
                // Root chunks also feed the bias gradient: a separate pass over the chunk's dOut rows.  Folding it into
                // the MFMA loop costs either 8 selects per group on every chunk or a second copy of the loop, and at
                // the merge of two loop copies the register allocator moves all 64 accumulator registers.
                if (is_root) {
#pragma unroll
                    for (int g = 0; g < kChunk / 16; ++g) {
                        if (g < ngrp) {
#pragma unroll
                            for (int u = 0; u < UB; ++u) {
                                const f32x4 gv = *(const f32x4*)(grow + 16 * g * NP + 64 * u);
#pragma unroll
                                for (int c = 0; c < 4; ++c) bsum[4 * u + c] += gv[c];
                            }
                        }
                    }
                }
            }
            STAMP(t2);
            wg_barrier();
            STAMP(t3);
            STAMP_ADD(st_scal, t0, t1);
            STAMP_ADD(st_comp, t1, t2);
            STAMP_ADD(st_bar, t2, t3);
        }
#ifdef RGCN_STAMPS
        if (g_stamps && cwv == 0 && lane == 0) {
            unsigned long long* o = g_stamps + (size_t)blockIdx.x * 32;
            o[0] = st_scal; o[1] = st_comp; o[2] = 0; o[3] = st_bar;
        }
#endif
        if (rel_cur >= 0) flush();
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see rgcn_tile_kernel
    }
    if (wave < kProducerWaves) {
        // The producers' few instructions must not queue behind the consumer wave's MFMAs on the shared SIMD
        // (issue is arbitrated by priority, then age; an fp32 MFMA holds the pipe 32 cycles): RGCN_PRIO
        __builtin_amdgcn_s_setprio(RGCN_PRIO);
        // producers, wide kernel: EVERY producer wave issues a quarter of every chunk (rows 16*pw..+15 of the H
        // and of the G slot), so the DMA-issue instructions are spread over the four SIMDs instead of landing
        // on one of them per chunk (fp32 MFMAs and these vector instructions share a SIMD's pipe: with one
        // issuing wave per chunk that SIMD's consumer fell ~860 cycles behind and the other three waited at
        // the barrier).  Row indices: wave 0 copies the chunk's slot_src / slot_dstl vectors into an LDS index
        // ring by LDS-DMA 2*D chunks ahead; all waves read them from LDS D chunks ahead.  A wave issues the
        // same number of vector-memory operations every iteration (beyond the end it re-issues the last
        // chunk into a free slot), so "chunk it+1 has landed" is the counted wait vmcnt((D-1) * OPS).
        const int pw = wave;
        constexpr int IR = 2 * D + 1;                           // index ring slots (a chunk's indices live 2D steps)
        int* idxh = (int*)(wring + NBUF * kChunk);              // [IR][64] slot_src
        int* idxg = idxh + IR * kChunk;                         // [IR][64] slot_dstl
        RowGather<KP, kLinear, BUF> gather_h;
        RowGather<NP, kLinear, BUF> gather_g;
        gather_h.init(lane, a.din4, a.ldx);
        gather_g.init(lane, a.dout4, a.ldg);
        constexpr int OPS_ROWS = KP / 16 + NP / 16;             // row DMAs of one wave per chunk
        auto chunk_of = [&](int k) { return ldc(a.rel_order, i0 + (k < nch ? k : nch - 1)); };
        // chunk ids for the NEXT step are fetched (scalar loads) during the current one
        int c_rows = chunk_of(0), c_idx = chunk_of(2 * D);
        auto issue_idx = [&](int k, int chunk) {                // wave 0 only: 2 ops
            dma4(a.slot_src + (size_t)chunk * kChunk + lane, idxh + (k % IR) * kChunk);
            dma4(a.slot_row + (size_t)chunk * kChunk + lane, idxg + (k % IR) * kChunk);
        };
        auto issue_rows = [&](int k, int chunk) {               // OPS_ROWS ops (+1 on wave 0)
            const int buf = k % NBUF;
            gather_h.issue_quarter(a.x, a.x_bytes, a.n_rows, a.ldx, idxh + (k % IR) * kChunk, ringh + buf * kChunk * KP, pw);
            gather_g.issue_quarter(a.g, a.g_bytes, a.n_owned, a.ldg, idxg + (k % IR) * kChunk, ringg + buf * kChunk * NP, pw);
            if (pw == 0) dma4(a.slot_w + (size_t)chunk * kChunk + lane, wring + buf * kChunk);
        };
        auto wait_ahead = [&]() {       // everything but the (D-1) youngest iterations' operations has landed
            if (pw == 0) wait_vmcnt<(D - 1) * (OPS_ROWS + 3)>();
            else wait_vmcnt<(D - 1) * OPS_ROWS>();
        };
        // step s = { wave 0: index vectors of chunk s + 2D ; every wave: its quarter of chunk s } -- the same
        // operation count for every s, which is what makes wait_ahead() exact from the first iteration on
        auto step = [&](int sidx) {
            const int cr = c_rows, ci = c_idx;
            c_rows = chunk_of(sidx + 1);
            c_idx = chunk_of(sidx + 1 + 2 * D);
            if (pw == 0) issue_idx(sidx + 2 * D, ci);
            issue_rows(sidx, cr);
        };
        // prologue: index vectors of chunks 0 .. 2D-1 up front, then steps 0 .. D-1
        if (pw == 0) {
#pragma unroll
            for (int k = 0; k < 2 * D; ++k) issue_idx(k, chunk_of(k));
            wait_vmcnt<0>();
        }
        wg_barrier();
#pragma unroll
        for (int k = 0; k < D; ++k) step(k);
        wait_ahead();                                           // chunk 0 landed
        wg_barrier();
#ifdef RGCN_STAMPS
        unsigned long long sp_issue = 0, sp_wait = 0, sp_bar = 0;
#endif
        for (int it = 0; it < nch; ++it) {
            STAMP(p0);
            step(it + D);
            STAMP(p1);
            wait_ahead();                                       // chunk it+1 landed
            STAMP(p2);
            wg_barrier();
            STAMP(p3);
            STAMP_ADD(sp_issue, p0, p1);
            STAMP_ADD(sp_wait, p1, p2);
            STAMP_ADD(sp_bar, p2, p3);
        }
        wait_vmcnt<0>();
#ifdef RGCN_STAMPS
        if (g_stamps && lane == 0) {
            unsigned long long* o = g_stamps + (size_t)blockIdx.x * 32;
            if (pw == 0) { o[4] = sp_issue; o[5] = sp_wait; o[6] = sp_bar; }
            if (pw == 1) o[7] = nch;
        }
#endif
    }
}

// ------------------------------------------------------------------------------------------------
// weight-gradient kernel, direct form (64 x 64, buffer-addressable operands)
// ------------------------------------------------------------------------------------------------
// In dW every gathered row is used by exactly ONE wave (a wave owns whole rows, see the wide kernel), so staging the
// rows in LDS buys no reuse -- and LDS-DMA gathers top out at ~25 GB/s per CU (~6.4 TB/s per chip), which is where the
// wide kernel sits with its two gathered rows per slot.  Here there are no producers, no LDS and no barriers: each
// wave walks its own range of 64-row units and loads its MFMA operands straight from global memory into registers
// (buffer_load_dwordx4 by slot index, padding rows out of range -> zeros), half a unit (8 k-steps = 16 loads of 16 B
// per lane) ahead of the half it is multiplying, with two waves per SIMD to cover the rest of the latency.  Row
// indices and weights of a unit are one coalesced load each, a whole unit ahead, and reach the lanes that need them
// through ds_bpermute.  Same arithmetic, same slab layout and the same reduce kernel as the wide form.
__global__ void __launch_bounds__(256, 2) rgcn_dw_direct_kernel(const DwArgs a) {
    constexpr int KP = 64, NP = 64;
    constexpr int HS = 8;                        // k-steps (4 rows each) per half unit
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = gridDim.x, b = blockIdx.x;
    const int bi0 = (int)((long)b * a.n_units / nb), bi1 = (int)((long)(b + 1) * a.n_units / nb);
    const int i0 = bi0 + (int)((long)wave * (bi1 - bi0) / 4), i1 = bi0 + (int)((long)(wave + 1) * (bi1 - bi0) / 4);
    const int nun = i1 - i0;
    if (nun <= 0) return;
    const int ml = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes), rg = make_rsrc(a.g, a.g_bytes);
    const unsigned colb = 16u * (unsigned)ml;
    const unsigned rbx = (unsigned)a.ldx * 4u, rbg = (unsigned)a.ldg * 4u;
    const int perm = kq * 4;                     // ds_bpermute address of row kq of a k-step (further steps: +16 each)

    f32x4 acc[4][4];
    float bsum[4];
    int rel_cur = -1;
    auto zero_acc = [&]() {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            bsum[jb] = 0.f;
#pragma unroll
            for (int ia = 0; ia < 4; ++ia) acc[ia][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto flush = [&]() {
        float* slab = a.slabs + ((size_t)(b + rel_cur) * kDwSlabsPer + wave) * (KP * NP);
#pragma unroll
        for (int ia = 0; ia < 4; ++ia)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(4 * (4 * kq + r) + ia) * NP + 4 * ml + jb] = acc[ia][jb][r];
        if (rel_cur == a.num_rel) {
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                float v = bsum[jb];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (kq == 0) a.bias_slabs[((size_t)b * kDwSlabsPer + wave) * NP + 4 * ml + jb] = v;
            }
        }
    };
    zero_acc();

    struct Idx {      // lane l: slot l of the unit
        int h, g;
        float w;
    };
    struct Half {
        f32x4 a4[HS], g4[HS];
    };
    auto unit_of = [&](int k) { return ldc(a.rel_order, i0 + (k < nun ? k : nun - 1)); };
    auto unit_cnt = [&](int unit) {
        const int cc = ldc(a.chunk_cnt, unit >> a.ushift);
        const int c = cc - kChunk * (unit & ((1 << a.ushift) - 1));
        return c < kChunk ? c : kChunk;
    };
    auto load_idx = [&](int unit) {
        const size_t base = (size_t)unit * kChunk + lane;
        return Idx{a.slot_src[base], a.slot_row[base], a.slot_w[base]};
    };
    auto issue_half = [&](Half& o, const Idx& ix, int h) {
        int ih[HS], ig[HS];
#pragma unroll
        for (int s = 0; s < HS; ++s) {
            ih[s] = __builtin_amdgcn_ds_bpermute(perm + 16 * (HS * h + s), ix.h);
            ig[s] = __builtin_amdgcn_ds_bpermute(perm + 16 * (HS * h + s), ix.g);
        }
#pragma unroll
        for (int s = 0; s < HS; ++s) {
            o.a4[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(__umul24((unsigned)ih[s], rbx) + colb), 0, RGCN_DW_X_AUX));
            o.g4[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)(__umul24((unsigned)ig[s], rbg) + colb), 0, 0));
        }
    };
    // groups g0, g0 + 1 of the unit (two k-step quadruples of this half), guarded by the unit's group count
    auto compute_half = [&](const Half& o, const Idx& ix, int h, int ngrp, bool is_root) {
        float wv[HS];
#pragma unroll
        for (int s = 0; s < HS; ++s)
            wv[s] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm + 16 * (HS * h + s), __builtin_bit_cast(int, ix.w)));
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            if (2 * h + gi < ngrp) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int s = 4 * gi + t;
                    f32x4 bv = o.g4[s] * wv[s];
                    // the MFMAs below are inline asm: the compiler does not see a VALU-write -> MFMA-read hazard and
                    // would schedule the last multiply right in front of the first MFMA (wrong acc[0][0] without this)
                    asm volatile("s_nop 4" : "+v"(bv));
#pragma unroll
                    for (int ia = 0; ia < 4; ++ia)
#pragma unroll
                        for (int jb = 0; jb < 4; ++jb) {
                            if (RGCN_ABL & 1) {   // diagnostic build: no MFMA (memory rate of the walk)
                                if (jb == 0) acc[ia][0][0] += o.a4[s][ia] * bv[ia];
                                continue;
                            }
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0"
                                         : "+v"(acc[ia][jb])
                                         : "v"(o.a4[s][ia]), "v"(bv[jb]));
                        }
                }
                if (is_root) {      // bias gradient: plain column sums of the root rows
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int c = 0; c < 4; ++c) bsum[c] += o.g4[4 * gi + t][c];
                }
            }
        }
    };

    // prologue: ids of units 0..2, indices of units 0 and 1, rows of the first half of unit 0
    int uid_cur = unit_of(0), uid_nxt = unit_of(1), uid_nn = unit_of(2);
    int cnt_pre = unit_cnt(uid_cur), rel_pre = ldc(a.chunk_rel, uid_cur >> a.ushift);
    Idx ix_cur = load_idx(uid_cur), ix_nxt = load_idx(uid_nxt);
    Half s0, s1;
    issue_half(s0, ix_cur, 0);
    for (int k = 0; k < nun; ++k) {
        const int cnt = cnt_pre, rel = rel_pre;
        cnt_pre = unit_cnt(uid_nxt);
        rel_pre = ldc(a.chunk_rel, uid_nxt >> a.ushift);
        if (rel != rel_cur) {
            if (rel_cur >= 0) {
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // asm MFMA results -> compiler-scheduled stores
                flush();
            }
            zero_acc();
            rel_cur = rel;
        }
        const bool is_root = rel == a.num_rel;
        const int ngrp = (cnt + 15) >> 4;
        // second half of this unit on its way while the first is multiplied
        issue_half(s1, ix_cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        compute_half(s0, ix_cur, 0, ngrp, is_root);
        __builtin_amdgcn_sched_barrier(0);
        // indices of the unit after next, first half of the next unit
        const Idx ix_nn = load_idx(uid_nn);
        issue_half(s0, ix_nxt, 0);
        __builtin_amdgcn_sched_barrier(0);
        compute_half(s1, ix_cur, 1, ngrp, is_root);
        __builtin_amdgcn_sched_barrier(0);
        ix_cur = ix_nxt;
        ix_nxt = ix_nn;
        uid_cur = uid_nxt;
        uid_nxt = uid_nn;
        uid_nn = unit_of(k + 3);
    }
    // the accumulators are read by plain stores the compiler schedules: keep them clear of the last asm MFMA
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (rel_cur >= 0) flush();
}

// ------------------------------------------------------------------------------------------------
// weight-gradient kernel, tile-major form (64 x 64, at most 32 relations, buffer-addressable operands)
// ------------------------------------------------------------------------------------------------
// The relation-major kernels above gather TWO rows per slot -- x[src] and the upstream gradient g[dst] -- although only
// N gradient rows exist: E * 4 * out bytes of gathers (25.6 GB at the headline config) that no walk ORDER gets the caches
// to serve.  Making the reuse structural means staging a tile's gradient rows in LDS and walking
// tile-major -- and then the relation changes with every (tile, relation) group, which is why the accumulators have to be
// somewhere that survives the whole walk.  Here they are: a wave owns ONE relation for the whole launch and keeps its
// 64 x 64 accumulator in registers (64 VGPRs, as in the direct kernel); a workgroup = 8 waves = 8 relations, FOUR
// workgroups (relation quarters) share a tile range, `walkers` ranges cover the graph.  Per tile: the workgroup's waves
// DMA the tile's T = 304 gradient rows into one of two LDS buffers (2 x 76 KiB) a tile ahead, each wave walks the
// 64-slot units of (tile, its relation) -- a contiguous stretch of rel_order -- loading x rows straight from global
// memory into registers half a unit ahead (rgcn_dw_direct_kernel's pipeline) and reading the gradient rows from LDS.
// Traffic: x gathers E * 4 * in + four sweeps of g (4 N * 4 * out) + indices = 37 GB instead of 55; one barrier per tile.
// The root relation and the bias gradient stay with rgcn_dw_direct_kernel (RGCN_FLAG_DW_ROOT_ONLY): their x rows are
// the tile's own.
#ifndef RGCN_DW_ABL_NOBARRIER
#define RGCN_DW_ABL_NOBARRIER 0
#endif
#ifndef RGCN_DW_TRUNC
#define RGCN_DW_TRUNC 0
#endif
#ifndef RGCN_DW_ABL
#define RGCN_DW_ABL 0      // timing-only ablations of rgcn_dw_tile_kernel<true>: 1 cached gathers, 2 no MFMAs, 4 no split arithmetic
#endif
#ifndef RGCN_DW_VECTOR_WALK
#define RGCN_DW_VECTOR_WALK 0
#endif
#ifndef RGCN_DW_XCD_MAP
#define RGCN_DW_XCD_MAP 1
#endif
constexpr int kDwTileT = 304;                    // gradient rows per LDS buffer = tile size of the plan this kernel walks
constexpr int kDwTileWalkers = 64;               // tile ranges; x 4 relation quarters = 256 workgroups, one per CU
constexpr int kDwTileMaxRel = 32;

struct DwTileArgs {
    const int* rel_order;   // of a plan with tile = kDwTileT, 64-slot chunks (unit == chunk), layout 0
    const int* chunk_cnt;
    const int* chunk_tile;
    const int* slot_src;
    const float* slot_w;
    const int* slot_row;
    const int* walk_ptr;    // [num_rel][walkers + 1]: rel_order positions where walker p's tiles of relation r begin
    const float* x;
    const float* g;
    unsigned x_bytes, g_bytes;
    float* slabs;           // [walkers][num_rel][64 * 64]
    int ldx, ldg, dout4, n_tiles, n_owned, num_rel, walkers;
};

// SPLIT: the same walk with the contraction as a bf16 x 3 split of BOTH fp32 operands (x row values and weight * gradient
// values -> three bf16 pieces each, split in registers by the wave that uses them; six v_mfma_f32_16x16x32_bf16 products
// hh, hm, mh, hl, lh, mm, fp32 accumulation: 24 significant bits on both sides, as rgcn_tile3p_kernel).  A 64-slot unit is two
// 32-row k-steps of that MFMA = the two halves of the register pipeline, with slot 32 h + 4 s + kq as k index 8 kq + s on both
// operands (a sum over k does not care which slot sits where, only that A and B agree).  96 MFMAs of 16 cycles per half
// against 256 of 32: the exact-fp32 form of this kernel is bound by the fp32 MFMA rate.
template <bool SPLIT>
__global__ void __launch_bounds__(512, 2) rgcn_dw_tile_kernel(const DwTileArgs a) {
    constexpr int T = kDwTileT, NP = 64, HS = 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [2][T][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if RGCN_DW_XCD_MAP
    // the four relation quarters of a tile range on ONE XCD (workgroup b runs on XCD b % 8): they stage the same gradient rows
    const int quarter = (blockIdx.x >> 3) & 3, p = (blockIdx.x & 7) | ((blockIdx.x >> 5) << 3);
#else
    const int quarter = blockIdx.x & 3, p = blockIdx.x >> 2;
#endif
    const int rel = 8 * quarter + wave;
    const bool have = rel < a.num_rel;
    const int t0 = (int)((long)p * a.n_tiles / a.walkers), t1 = (int)((long)(p + 1) * a.n_tiles / a.walkers);
    if (t1 <= t0) return;
    const int i0 = have ? ldc(a.walk_ptr, (long)rel * (a.walkers + 1) + p) : 0;
    const int nun = have ? ldc(a.walk_ptr, (long)rel * (a.walkers + 1) + p + 1) - i0 : 0;
    const int ml = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes), rg = make_rsrc(a.g, a.g_bytes);
    const unsigned colb = 16u * (unsigned)ml;
    const unsigned rbx = (unsigned)a.ldx * 4u, rbg = (unsigned)a.ldg * 4u;
    const unsigned gcol = ml < a.dout4 ? colb : 0xFFFFFFF0u;     // columns beyond the width: out of range -> zeros
    const unsigned grow = ml < a.dout4 ? rbg : 0u;
    const int perm = kq * 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int ia = 0; ia < 4; ++ia)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ia][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // tile t -> LDS buffer b: DMA instruction i moves rows 4 i .. 4 i + 3 (64 lanes x 16 bytes); rows past the end read zeros
    auto dma_tile = [&](int t, int b) {
        float* base = lds + b * T * NP;
        for (int i = wave; i < T / 4; i += 8) {
            const unsigned row = (unsigned)(t * T + 4 * i + kq);
            dma16_buf(rg, __umul24(row, grow) + gcol, base + i * 4 * NP);
        }
    };
    struct Idx {      // lane l: slot l of the unit
        int h, g;
        float w;
    };
    auto unit_of = [&](int k) { return ldc(a.rel_order, i0 + (k < nun ? k : (nun > 0 ? nun - 1 : 0))); };
    auto load_idx = [&](int unit) {
        const size_t base = (size_t)unit * kChunk + lane;
        return Idx{a.slot_src[base], a.slot_row[base], a.slot_w[base]};
    };
    auto issue_half = [&](f32x4 (&a4)[HS], const Idx& ix, int h) {
        int ih[HS];
#pragma unroll
        for (int s = 0; s < HS; ++s) ih[s] = __builtin_amdgcn_ds_bpermute(perm + 16 * (HS * h + s), ix.h);
        if (RGCN_DW_ABL & 1)       // (timing only: every gather hits rows 0..63 -- no HBM traffic for x)
#pragma unroll
            for (int s = 0; s < HS; ++s) ih[s] &= 63;
#pragma unroll
        for (int s = 0; s < HS; ++s)
            a4[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(__umul24((unsigned)ih[s], rbx) + colb), 0, 0));
    };
    // half a unit: 8 k-steps of 4 rows; gradient rows from the LDS tile (row ids local to the tile, padding clamped: its
    // weight is 0 and every LDS word is a finite number)
    auto compute_half = [&](const f32x4 (&a4)[HS], const Idx& ix, int h, int ngrp, int nks, const float* gbuf, int tile_row0) {
        const unsigned loc = (unsigned)(ix.g - tile_row0);
        const int goff = (int)((loc < (unsigned)T ? loc : (unsigned)(T - 1)) * (unsigned)(NP * 4));    // byte offset of this lane's slot row
        float wv[HS];
        f32x4 g4[HS];
#pragma unroll
        for (int s = 0; s < HS; ++s) {
            wv[s] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm + 16 * (HS * h + s), __builtin_bit_cast(int, ix.w)));
            const int o = __builtin_amdgcn_ds_bpermute(perm + 16 * (HS * h + s), goff);
            g4[s] = *(const f32x4*)((const char*)gbuf + o + colb);
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            if (2 * h + gi < ngrp) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int s = 4 * gi + t;
                    // (cutting a unit's tail at the 4-row k-step instead of the 16-row group -- ~6 % fewer MFMAs -- measured
                    // 1 % SLOWER: 8.25 against 8.16 ms; the walk is not bound by its MFMA count.  Knob: RGCN_DW_KSTEP_GATE)
                    if (RGCN_DW_KSTEP_GATE && HS * h + s >= nks) break;
                    f32x4 bv = g4[s] * wv[s];
                    asm volatile("s_nop 4" : "+v"(bv));       // VALU write -> asm MFMA operand (see rgcn_dw_direct_kernel)
#pragma unroll
                    for (int ia = 0; ia < 4; ++ia)
#pragma unroll
                        for (int jb = 0; jb < 4; ++jb)
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[ia][jb]) : "v"(a4[s][ia]), "v"(bv[jb]));
                }
            }
        }
    };

    // v0, v1 -> three packed bf16 pairs (low half = v0), round-to-nearest pieces: v = h + m + l to 24 bits
    auto split_pair = [](float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
        if (RGCN_DW_ABL & 4) {      // (timing only: no split arithmetic)
            h = __float_as_uint(v0);
            m = __float_as_uint(v1);
            l = h ^ m;
            return;
        }
#if RGCN_DW_TRUNC      // pieces by truncation (v_perm_b32 packs two upper halves; exact as well): measured, see DESIGN.md 4.3
        auto pk = [](float lo, float hi) { return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u); };
        auto top = [](float v) { return __uint_as_float(__float_as_uint(v) & 0xFFFF0000u); };
        h = pk(v0, v1);
        v0 -= top(v0); v1 -= top(v1);
        m = pk(v0, v1);
        v0 -= top(v0); v1 -= top(v1);
        l = pk(v0, v1);
        return;
#endif
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(v0), "v"(v1));
        v0 -= __uint_as_float(h << 16);
        v1 -= __uint_as_float(h & 0xFFFF0000u);
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m) : "v"(v0), "v"(v1));
        v0 -= __uint_as_float(m << 16);
        v1 -= __uint_as_float(m & 0xFFFF0000u);
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l) : "v"(v0), "v"(v1));
    };
    // half a unit as ONE 32-row k-step (a half with no valid slot is skipped; padding slots inside one have weight 0)
    auto compute_half3 = [&](const f32x4 (&a4)[HS], const Idx& ix, int h, int ngrp, const float* gbuf, int tile_row0) {
        if (2 * h >= ngrp) return;
        const unsigned loc = (unsigned)(ix.g - tile_row0);
        const int goff = (int)((loc < (unsigned)T ? loc : (unsigned)(T - 1)) * (unsigned)(NP * 4));
        float wv[HS];
        f32x4 g4[HS];
#pragma unroll
        for (int s = 0; s < HS; ++s) {
            wv[s] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm + 16 * (HS * h + s), __builtin_bit_cast(int, ix.w)));
            const int o = __builtin_amdgcn_ds_bpermute(perm + 16 * (HS * h + s), goff);
            g4[s] = *(const f32x4*)((const char*)gbuf + o + colb);
        }
        u32x4 ap[3][4];      // [piece][ia]: 8 bf16 = k index 8 kq + 0..7 of input channel 4 ml + ia
#pragma unroll
        for (int ia = 0; ia < 4; ++ia)
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                unsigned h_, m_, l_;
                split_pair(a4[2 * jp][ia], a4[2 * jp + 1][ia], h_, m_, l_);
                ap[0][ia][jp] = h_; ap[1][ia][jp] = m_; ap[2][ia][jp] = l_;
            }
        constexpr int pa[6] = {2, 1, 1, 0, 0, 0}, pb[6] = {0, 1, 0, 2, 1, 0};      // small products first
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            u32x4 bp[3];
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                unsigned h_, m_, l_;
                split_pair(g4[2 * jp][jb] * wv[2 * jp], g4[2 * jp + 1][jb] * wv[2 * jp + 1], h_, m_, l_);
                bp[0][jp] = h_; bp[1][jp] = m_; bp[2][jp] = l_;
            }
            if (RGCN_DW_ABL & 2) {      // (timing only: no MFMAs; the pieces stay alive)
#pragma unroll
                for (int ia = 0; ia < 4; ++ia)
                    asm volatile("" ::"v"(ap[0][ia]), "v"(ap[1][ia]), "v"(ap[2][ia]), "v"(bp[0]), "v"(bp[1]), "v"(bp[2]));
                continue;
            }
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int ia = 0; ia < 4; ++ia)
                    acc[ia][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[pa[q]][ia]),
                                                                          __builtin_bit_cast(bf16x8, bp[pb[q]]), acc[ia][jb], 0, 0, 0);
        }
    };

    // The halves are computed under conditions (a half without valid slots is skipped), and loads whose uses all sit in later
    // blocks get SUNK there by the optimiser -- issued right in front of their first use, their whole latency exposed (that is
    // where the second half's eight row loads of every unit were until this was found: the ISA showed them behind half 0's
    // MFMAs, not at the top of the iteration; sched_barrier only binds the scheduler inside a block).  A compiler-level memory
    // clobber after each batch keeps the loads where issue_half puts them: a read cannot be moved across it.
    auto pin_loads = [] { asm volatile("" ::: "memory"); };
    dma_tile(t0, 0);
    int k = 0;
    int uid_cur = unit_of(0), uid_nxt = unit_of(1), uid_nn = unit_of(2);
    int cnt_cur = ldc(a.chunk_cnt, uid_cur), tile_cur = nun > 0 ? ldc(a.chunk_tile, uid_cur) : t1;
    int cnt_nxt = ldc(a.chunk_cnt, uid_nxt), tile_nxt = nun > 1 ? ldc(a.chunk_tile, uid_nxt) : t1;
#if RGCN_DW_VECTOR_WALK
    // Inside the walk the three per-unit words (unit id, slot count, tile) come by VECTOR loads of a uniform address: scalar
    // loads return out of order, so the first LDS operation of the next unit -- its wait is lgkmcnt(0) -- would wait for the
    // scalar loads issued a few instructions earlier, a full L2 round trip per unit; vector loads retire in order and are
    // waited for by count, a whole unit after they were issued.
    const __amdgpu_buffer_rsrc_t r_ord = make_rsrc(a.rel_order, 0xFFFFFFFCu), r_cnt = make_rsrc(a.chunk_cnt, 0xFFFFFFFCu),
                                 r_til = make_rsrc(a.chunk_tile, 0xFFFFFFFCu);
    auto ldv = [](__amdgpu_buffer_rsrc_t r, int idx) { return __builtin_amdgcn_raw_buffer_load_b32(r, idx * 4, 0, 0); };
#endif
    Idx ix_cur = load_idx(uid_cur), ix_nxt = load_idx(uid_nxt);
    f32x4 s0[HS], s1[HS];
    if (nun > 0) issue_half(s0, ix_cur, 0);
    constexpr int kInFlight = RGCN_DW_VECTOR_WALK ? 14 : 11;      // 8 row loads + 3 index loads (+ 3 walk words)
    bool walked = nun > 0;      // at least kInFlight vector-memory operations were issued after the pending tile's DMAs
    for (int t = t0; t < t1; ++t) {
        // The DMAs of tile t were issued a tile ago (or in the prologue).  If the wave has walked a unit since (or issued the
        // prologue's loads), more than kInFlight younger operations exist and at most kInFlight are in flight at a unit boundary (8 row
        // loads + 3 index loads of the unit after next + the walk words): a counted wait retires the DMAs and leaves the prefetches alone.  A wave
        // without units in between (an empty relation) has nothing younger to count: it waits for everything.
        if (walked) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kInFlight) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !RGCN_DW_ABL_NOBARRIER      // (timing only: the waves of a workgroup run free -- what the tile lockstep costs)
        wg_barrier();          // tile t landed for every wave; every wave is done with the buffer tile t + 1 goes to
#endif
        const int b = (t - t0) & 1;
        if (t + 1 < t1) dma_tile(t + 1, b ^ 1);
        walked = false;
        const float* gbuf = lds + b * T * NP;
        while (k < nun && tile_cur == t) {
            const int ngrp = (cnt_cur + 15) >> 4, nks = (cnt_cur + 3) >> 2;
            walked = true;
            issue_half(s1, ix_cur, 1);
            pin_loads();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SPLIT) compute_half3(s0, ix_cur, 0, ngrp, gbuf, t * T);
            else compute_half(s0, ix_cur, 0, ngrp, nks, gbuf, t * T);
            __builtin_amdgcn_sched_barrier(0);
            const Idx ix_nn = load_idx(uid_nn);
            issue_half(s0, ix_nxt, 0);
            pin_loads();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SPLIT) compute_half3(s1, ix_cur, 1, ngrp, gbuf, t * T);
            else compute_half(s1, ix_cur, 1, ngrp, nks, gbuf, t * T);
            __builtin_amdgcn_sched_barrier(0);
            ++k;
            ix_cur = ix_nxt;
            ix_nxt = ix_nn;
            uid_cur = uid_nxt;
            uid_nxt = uid_nn;
#if RGCN_DW_VECTOR_WALK
            uid_nn = ldv(r_ord, i0 + (k + 2 < nun ? k + 2 : nun - 1));
            cnt_cur = __builtin_amdgcn_readfirstlane(cnt_nxt);
            tile_cur = k < nun ? __builtin_amdgcn_readfirstlane(tile_nxt) : t1;
            cnt_nxt = ldv(r_cnt, uid_nxt);
            tile_nxt = ldv(r_til, uid_nxt);       // (a clamped unit's tile is never looked at: tile_cur = t1 past the end)
#else
            uid_nn = unit_of(k + 2);
            cnt_cur = cnt_nxt;
            tile_cur = k < nun ? tile_nxt : t1;
            cnt_nxt = ldc(a.chunk_cnt, uid_nxt);
            tile_nxt = k + 1 < nun ? ldc(a.chunk_tile, uid_nxt) : t1;
#endif
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the accumulators are read by plain stores the compiler schedules: keep them clear of the last asm MFMA
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (have) {
        float* slab = a.slabs + ((size_t)p * a.num_rel + rel) * (64 * 64);
#pragma unroll
        for (int ia = 0; ia < 4; ++ia)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(4 * (4 * kq + r) + ia) * NP + 4 * ml + jb] = acc[ia][jb][r];
    }
}

// walk_ptr[r][p] = first position in rel_order (sorted by (relation, tile)) of a unit of relation r whose tile is
// >= p * n_tiles / walkers; one thread per entry, binary search (integer work, once per plan)
__global__ void rgcn_dw_walk_table_kernel(const int* __restrict__ rel_order, const int* __restrict__ chunk_rel,
                                          const int* __restrict__ chunk_tile, int n_units, int n_tiles, int num_rel, int walkers,
                                          int* __restrict__ walk_ptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_rel * (walkers + 1)) return;
    const int r = i / (walkers + 1), p = i - r * (walkers + 1);
    const long target = (long)r * n_tiles + (long)p * n_tiles / walkers;
    int lo = 0, hi = n_units;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int c = rel_order[mid];                        // 64-slot chunks: unit == chunk
        if ((long)chunk_rel[c] * n_tiles + chunk_tile[c] < target) lo = mid + 1; else hi = mid;
    }
    walk_ptr[i] = lo;
}

// d_weight[r] = sum over the walkers' slabs, in walker order (bitwise reproducible)
__global__ void rgcn_dw_tile_reduce_kernel(const float* __restrict__ slabs, int walkers, int num_rel, int din, int dout,
                                           float* __restrict__ d_weight) {
    const int r = blockIdx.x;
    for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < din * dout; e += gridDim.y * blockDim.x) {
        const int kk = e / dout, n = e - kk * dout;
        float sum = 0.f;
        for (int w = 0; w < walkers; ++w) sum += slabs[((size_t)w * num_rel + r) * (64 * 64) + kk * 64 + n];
        d_weight[(size_t)r * din * dout + e] = sum;
    }
}

// slabs -> gradients, fixed summation order (block index ascending) => bitwise reproducible.
// grid = (R' + 2, parts): blockIdx.x = relation (R' = root, R'+1 = bias), blockIdx.y = slice of the elements.
// The workgroups whose chunk range touches relation r are a contiguous run [b_lo, b_hi].
__global__ void rgcn_dw_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ bias_slabs,
                                      const int* __restrict__ rel_order, const int* __restrict__ chunk_rel,
                                      int n_chunks, int ushift, int nblocks, int num_rel, int KP, int NP, int din, int dout,
                                      float* __restrict__ d_weight, float* __restrict__ d_root,
                                      float* __restrict__ d_bias) {
    const bool is_bias = blockIdx.x == num_rel + 1;
    const int r = is_bias ? num_rel : blockIdx.x;            // the bias gradient comes from the root relation's rows
    if (is_bias && (d_bias == nullptr || blockIdx.y != 0)) return;
    float* dst = is_bias ? d_bias : (r < num_rel ? (d_weight ? d_weight + (size_t)r * din * dout : nullptr) : d_root);
    if (dst == nullptr) return;
    // which workgroups' unit ranges touch relation r: all threads look (two dependent loads per workgroup -- as a serial
    // scan by one thread this was 0.7 ms with 512 workgroups)
    __shared__ int s_lo, s_hi;
    if (threadIdx.x == 0) {
        s_lo = nblocks;
        s_hi = -1;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
        const int i0 = (int)((long)b * n_chunks / nblocks);
        const int i1 = (int)((long)(b + 1) * n_chunks / nblocks);
        if (i1 <= i0) continue;
        const int first = chunk_rel[rel_order[i0] >> ushift], last = chunk_rel[rel_order[i1 - 1] >> ushift];
        if (r >= first && r <= last) {
            atomicMin(&s_lo, b);
            atomicMax(&s_hi, b);
        }
    }
    __syncthreads();
    const int lo = s_lo, hi = s_hi;
    if (is_bias) {      // only the workgroups that walked root units wrote bias slabs (summing all 2048 was 0.7 ms)
        for (int n = threadIdx.x; n < dout; n += blockDim.x) {
            float s = 0.f;
            for (int b = lo; b <= hi; ++b)
                for (int c = 0; c < kDwSlabsPer; ++c) s += bias_slabs[((size_t)b * kDwSlabsPer + c) * NP + n];
            d_bias[n] = s;
        }
        return;
    }
    for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < din * dout; e += gridDim.y * blockDim.x) {
        const int k = e / dout, n = e - k * dout;
        float s = 0.f;
        for (int b = lo; b <= hi; ++b)
            for (int c = 0; c < kDwSlabsPer; ++c)
                s += slabs[((size_t)(b + r) * kDwSlabsPer + c) * KP * NP + (size_t)k * NP + n];
        dst[e] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// host side: argument checks, LDS sizing, dispatch over the padded widths
// ------------------------------------------------------------------------------------------------
constexpr int kDwBlocks = 512;  // most workgroups a dW launch uses (sizes the slab workspace): two per CU for the direct
                                // kernel, one per CU (LDS-bound) for the ring kernels
constexpr int kDwRingBlocks = 256;
constexpr int kDwDirectMinUnits = 16 * 1024;   // >= 8 units per wave of 512 four-wave workgroups

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

static int check_plan(const rgcn_plan_t* p) {
    if (p == nullptr) return RGCN_ERR_NULL;
    if (!p->tile_ptr || !p->chunk_rel || !p->chunk_cnt || !p->chunk_tile || !p->chunk_flags || !p->rel_order ||
        !p->slot_src ||
        !p->slot_w || !p->slot_row || !p->slot_acc)
        return RGCN_ERR_NULL;
    if (p->n_nodes <= 0 || p->n_owned <= 0 || p->num_relations <= 0 || p->tile <= 0 || (p->tile % 16) != 0 || p->tile > 32768 ||
        p->n_tiles <= 0 || p->n_chunks < p->n_tiles || (long)p->n_tiles * p->tile < p->n_owned ||
        (p->chunk != 64 && p->chunk != 128) || p->n_units < p->n_chunks || p->n_units > p->n_chunks * (p->chunk / 64))
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

template <int KP, int NP, int NBUF, int CH>
static int launch_tile_nbuf(const TileArgs& a, int n_tiles, size_t lds, hipStream_t stream) {
    hipError_t e = a.x_bytes ? allow_full_lds<rgcn_tile_kernel<KP, NP, NBUF, true, CH>>()
                             : allow_full_lds<rgcn_tile_kernel<KP, NP, NBUF, false, CH>>();
    if (e != hipSuccess) return (int)e;
    auto kern = a.x_bytes ? rgcn_tile_kernel<KP, NP, NBUF, true, CH> : rgcn_tile_kernel<KP, NP, NBUF, false, CH>;
    const int nwg = (n_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(kTileThreads), lds, stream, a);
    return (int)hipGetLastError();
}
// deepest DMA ring (4, 3 or 2 slots) that fits beside the tile accumulator in the 160 KiB LDS
template <int KP, int NP>
static int launch_tile(const TileArgs& a, int n_tiles, int chunk, hipStream_t stream) {
    auto bytes = [&](int nbuf) {
        return sizeof(float) * ((size_t)(a.tile + 1) * kAccStride<NP> + (size_t)nbuf * chunk * (KP + 2));
    };
    constexpr size_t cap = (size_t)kLdsBytes;
    if (chunk == 128) {
        // 128-slot chunks: built for the widths whose ring slots leave room for a useful tile (KP <= 64)
        if constexpr (KP <= 64) {
            if (bytes(3) <= cap) return launch_tile_nbuf<KP, NP, 3, 128>(a, n_tiles, bytes(3), stream);
            if (bytes(2) <= cap) return launch_tile_nbuf<KP, NP, 2, 128>(a, n_tiles, bytes(2), stream);
        }
        return RGCN_ERR_LDS;
    }
    if constexpr (kTileProducers >= 3)
        if (KP < 128 && bytes(4) <= cap) return launch_tile_nbuf<KP, NP, 4, 64>(a, n_tiles, bytes(4), stream);
    if (KP < 128 && bytes(3) <= cap) return launch_tile_nbuf<KP, NP, 3, 64>(a, n_tiles, bytes(3), stream);
    if (bytes(2) <= cap) return launch_tile_nbuf<KP, NP, 2, 64>(a, n_tiles, bytes(2), stream);
    return RGCN_ERR_LDS;
}
template <int KP>
static int dispatch_tile_np(int NP, const TileArgs& a, int n_tiles, int chunk, hipStream_t s) {
    switch (NP) {
        case 16: return launch_tile<KP, 16>(a, n_tiles, chunk, s);
        case 32: return launch_tile<KP, 32>(a, n_tiles, chunk, s);
        case 64: return launch_tile<KP, 64>(a, n_tiles, chunk, s);
        case 128: return launch_tile<KP, 128>(a, n_tiles, chunk, s);
    }
    return RGCN_ERR_WIDTH;
}

static int dispatch_tile(int KP, int NP, const TileArgs& a, int n_tiles, int chunk, hipStream_t s) {
    switch (KP) {
        case 16: return dispatch_tile_np<16>(NP, a, n_tiles, chunk, s);
        case 32: return dispatch_tile_np<32>(NP, a, n_tiles, chunk, s);
        case 64: return dispatch_tile_np<64>(NP, a, n_tiles, chunk, s);
        case 128: return dispatch_tile_np<128>(NP, a, n_tiles, chunk, s);
    }
    return RGCN_ERR_WIDTH;
}

#ifdef RGCN_DEBUG_KNOBS
static std::atomic<int> g_debug_mode{0};
#endif

// shared by rgcn_fwd and rgcn_bwd_dx: gather rows of `x` (width kin), scatter into `out` (width nout)
static int run_tile(const rgcn_plan_t* plan, const float* x, int ldx, int kin, const float* packed, const float* bias,
                    float* out, int ldo, int nout, int act, const float* mask, int ldm, unsigned flags, void* stream) {
    int st = check_plan(plan);
    if (st != RGCN_OK) return st;
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
    // Producer-split bf16 x 3 kernel (rgcn_tile3p.hip): 64 x 64 layers, 128-slot chunks, layout 0, tiles that leave room for
    // its 48 KiB ring slots; anything else falls through to the kernels below
    if ((flags & RGCN_FLAG_SPLIT_PRODUCERS) && KP == 64 && NP == 64 && plan->layout == 0 && plan->chunk == 128 && a.x_bytes != 0) {
        TileArgs b = a;
        b.wp = packed + (size_t)(plan->num_relations + 1) * KP * NP;
        const int st3 = launch_tile3p(b, plan->n_tiles, stream);
        if (st3 != RGCN_ERR_LDS) return st3;
    }
    // Split-precision kernel: 64 x 64 layers on layout-1 plans (the halves of a chunk scatter into disjoint rows)
    if (KP == 64 && NP == 64 && plan->layout == 1 && plan->chunk == 128 && !(flags & RGCN_FLAG_EXACT_FP32)) {
        const size_t lds = sizeof(float) * ((size_t)(a.tile + 1) * kAccStride<64> + (size_t)2 * 128 * (64 + 2));
        if (lds > (size_t)kLdsBytes) return RGCN_ERR_LDS;
        a.wp = packed + (size_t)(plan->num_relations + 1) * KP * NP;
        a.tiles_per_wg = 1;
        hipError_t e = a.x_bytes ? allow_full_lds<rgcn_tile3_kernel<true>>() : allow_full_lds<rgcn_tile3_kernel<false>>();
        if (e != hipSuccess) return (int)e;
        auto kern = a.x_bytes ? rgcn_tile3_kernel<true> : rgcn_tile3_kernel<false>;
        hipLaunchKernelGGL(kern, dim3(plan->n_tiles), dim3(kTileThreads), lds, (hipStream_t)stream, a);
        return (int)hipGetLastError();
    }
    return dispatch_tile(KP, NP, a, plan->n_tiles, plan->chunk, (hipStream_t)stream);
}

template <int KP, int NP>
static int launch_dw(const DwArgs& a, int nblocks, hipStream_t stream) {
    constexpr int NBUF = dw_nbuf<KP, NP>();
    constexpr bool kWide = KP % 64 == 0 && NP % 64 == 0 && KP * NP <= 64 * 128;
    const size_t lds = sizeof(float) * ((size_t)NBUF * kChunk * (KP + NP + 1) + (kWide ? 2 * (2 * NBUF - 1) * kChunk : 0));
    if (lds > (size_t)kLdsBytes) return RGCN_ERR_LDS;
    int threads = kThreads;
    hipError_t e;
    void (*kern)(const DwArgs);
    const bool buf = a.x_bytes && a.g_bytes;
    if constexpr (kWide) {
        // 64x64: accumulators take 64 registers, two consumer teams fit; wider: one team
        constexpr int CONS = KP * NP <= 64 * 64 ? kWideConsumers : 4;
        threads = 64 * (kProducerWaves + CONS);
        e = buf ? allow_full_lds<rgcn_dw_wide_kernel<KP, NP, NBUF, true, CONS>>()
                : allow_full_lds<rgcn_dw_wide_kernel<KP, NP, NBUF, false, CONS>>();
        kern = buf ? rgcn_dw_wide_kernel<KP, NP, NBUF, true, CONS> : rgcn_dw_wide_kernel<KP, NP, NBUF, false, CONS>;
    } else {
        e = buf ? allow_full_lds<rgcn_dw_kernel<KP, NP, NBUF, true>>() : allow_full_lds<rgcn_dw_kernel<KP, NP, NBUF, false>>();
        kern = buf ? rgcn_dw_kernel<KP, NP, NBUF, true> : rgcn_dw_kernel<KP, NP, NBUF, false>;
    }
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(threads), lds, stream, a);
    return (int)hipGetLastError();
}

template <int KP>
static int dispatch_dw_np(int NP, const DwArgs& a, int nb, hipStream_t s) {
    switch (NP) {
        case 16: return launch_dw<KP, 16>(a, nb, s);
        case 32: return launch_dw<KP, 32>(a, nb, s);
        case 64: return launch_dw<KP, 64>(a, nb, s);
        case 128: return launch_dw<KP, 128>(a, nb, s);
    }
    return RGCN_ERR_WIDTH;
}

static int dispatch_dw(int KP, int NP, const DwArgs& a, int nb, hipStream_t s) {
    switch (KP) {
        case 16: return dispatch_dw_np<16>(NP, a, nb, s);
        case 32: return dispatch_dw_np<32>(NP, a, nb, s);
        case 64: return dispatch_dw_np<64>(NP, a, nb, s);
        case 128: return dispatch_dw_np<128>(NP, a, nb, s);
    }
    return RGCN_ERR_WIDTH;
}

static size_t dw_slab_floats(int num_rel, int KP, int NP) {
    return (size_t)(kDwBlocks + num_rel + 1) * kDwSlabsPer * KP * NP;
}

}  // namespace rgcn

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
using namespace rgcn;

extern "C" int rgcn_abi_version(void) { return RGCN_ABI_VERSION; }

#ifdef RGCN_STAMPS
extern "C" int rgcn_debug_set_stamps(unsigned long long* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(rgcn::g_stamps), &p, sizeof(p));
}
#endif
#ifdef RGCN_DEBUG_KNOBS
extern "C" void rgcn_debug_set_mode(int mode) { rgcn::g_debug_mode.store(mode); }
#endif

extern "C" const char* rgcn_status_string(int status) {
    switch (status) {
        case RGCN_OK: return "ok";
        case RGCN_ERR_NULL: return "required pointer is NULL";
        case RGCN_ERR_WIDTH: return "feature width outside 1..128";
        case RGCN_ERR_STRIDE: return "row stride must be a multiple of 4 elements and >= the width rounded up to 4";
        case RGCN_ERR_PLAN: return "inconsistent graph plan";
        case RGCN_ERR_LDS: return "plan tile too large for the 160 KiB LDS at these widths";
        case RGCN_ERR_WORKSPACE: return "workspace too small";
        case RGCN_ERR_DEVICE: return "current device is not gfx950 (MI355X)";
        case RGCN_ERR_ACT: return "unknown activation code";
        case RGCN_ERR_GRAPH: return "edge_index / edge_type value out of range";
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown status";
}

extern "C" int rgcn_padded_width(int width) { return padded_width(width); }

// 64 x 64 layers also carry the bf16 x 3 split of the weights (rgcn_tile3_kernel), behind the fp32 fragments
static size_t pack3_floats(int num_relations, int KP, int NP) {
    return (KP == 64 && NP == 64) ? (size_t)(num_relations + 1) * kPack3FloatsPerRel : 0;
}

extern "C" size_t rgcn_packed_weight_floats(int num_relations, int din, int dout) {
    const int a = padded_width(din), b = padded_width(dout);
    if (a == 0 || b == 0 || num_relations <= 0) return 0;
    return (size_t)(num_relations + 1) * a * b + pack3_floats(num_relations, a, b);
}

extern "C" int rgcn_pack_weights(const float* weight, const float* root, int num_relations, int din, int dout,
                                 int transpose, float* packed, void* stream) {
    if (!weight || !packed) return RGCN_ERR_NULL;
    if (num_relations <= 0) return RGCN_ERR_PLAN;
    const int kin = transpose ? dout : din, nout = transpose ? din : dout;
    const int KP = padded_width(kin), NP = padded_width(nout);
    if (KP == 0 || NP == 0) return RGCN_ERR_WIDTH;
    const long total = (long)(num_relations + 1) * KP * NP;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(rgcn_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, root, num_relations,
                       din, dout, transpose, KP, NP, packed);
    if (pack3_floats(num_relations, KP, NP) != 0) {
        const long lanes = (long)(num_relations + 1) * kPack3FragsPerRel * 64;
        hipLaunchKernelGGL(rgcn_pack3_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, root,
                           num_relations, din, dout, transpose, (uint4*)(packed + total));
    }
    return (int)hipGetLastError();
}

extern "C" int rgcn_fwd(const rgcn_plan_t* plan, const float* x, int ldx, int din, const float* packed_w,
                        const float* bias, float* out, int ldo, int dout, int act, unsigned flags, void* stream) {
    return run_tile(plan, x, ldx, din, packed_w, bias, out, ldo, dout, act, nullptr, 0, flags, stream);
}

extern "C" int rgcn_bwd_dx(const rgcn_plan_t* plan_t, const float* g, int ldg, int dout, const float* packed_wt,
                           float* dx, int lddx, int din, const float* relu_of, int ldr, unsigned flags, void* stream) {
    return run_tile(plan_t, g, ldg, dout, packed_wt, nullptr, dx, lddx, din, RGCN_ACT_NONE, relu_of, ldr, flags, stream);
}

extern "C" int rgcn_act_backward(const float* a, const float* da, float* dz, long rows, int ld, int act, void* stream) {
    if (!a || !da || !dz) return RGCN_ERR_NULL;
    if (ld <= 0 || (ld % 4) != 0) return RGCN_ERR_STRIDE;
    if (act != RGCN_ACT_RELU && act != RGCN_ACT_SIGMOID) return RGCN_ERR_ACT;
    if (rows <= 0) return RGCN_OK;
    const long total = rows * (ld / 4);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(rgcn_act_backward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, da, dz, rows, ld / 4, act);
    return (int)hipGetLastError();
}

extern "C" size_t rgcn_bwd_dw_workspace_bytes(const rgcn_plan_t* plan, int din, int dout) {
    if (plan == nullptr) return 0;
    const int KP = padded_width(din), NP = padded_width(dout);
    if (KP == 0 || NP == 0) return 0;
    return sizeof(float) * (dw_slab_floats(plan->num_relations, KP, NP) + (size_t)kDwBlocks * kDwSlabsPer * NP);
}

extern "C" int rgcn_bwd_dw(const rgcn_plan_t* plan, const float* x, int ldx, int din, const float* g, int ldg,
                           int dout, void* workspace, size_t workspace_bytes, float* d_weight, float* d_root,
                           float* d_bias, unsigned flags, void* stream) {
    int st = check_plan(plan);
    if (st != RGCN_OK) return st;
    if (!x || !g || !workspace) return RGCN_ERR_NULL;
    if ((st = check_stride(ldx, din)) != RGCN_OK) return st;
    if ((st = check_stride(ldg, dout)) != RGCN_OK) return st;
    const size_t need = rgcn_bwd_dw_workspace_bytes(plan, din, dout);
    if (workspace_bytes < need) return RGCN_ERR_WORKSPACE;
    if ((st = check_device()) != RGCN_OK) return st;
    const int KP = padded_width(din), NP = padded_width(dout);
    hipStream_t s = (hipStream_t)stream;
    // RGCN_FLAG_DW_ROOT_ONLY: d_root / d_bias alone (the relations went to rgcn_bwd_dw_tiles): walk the root relation's
    // units, which close rel_order -- their count follows from the tile geometry (every node has one root pseudo edge)
    int unit_begin = 0, n_units = plan->n_units;
    if (flags & RGCN_FLAG_DW_ROOT_ONLY) {
        auto units_of = [&](long rows) -> long {
            if (plan->layout == 0 || plan->chunk == 64) return ((rows + 15) / 16 + 3) / 4;
            const long rem = rows % 128;
            return 2 * (rows / 128) + (rem == 0 ? 0 : (rem > 64 ? 2 : 1));
        };
        const long last_rows = (long)plan->n_owned - (long)(plan->n_tiles - 1) * plan->tile;
        const long root_units = (long)(plan->n_tiles - 1) * units_of(plan->tile) + units_of(last_rows);
        if (root_units <= 0 || root_units > n_units) return RGCN_ERR_PLAN;
        unit_begin = n_units - (int)root_units;
        n_units = (int)root_units;
        d_weight = nullptr;
    }
    // The direct-gather kernel (64 x 64, buffer-addressable operands) pays on large walks; small graphs take fewer
    // persistent workgroups (>= 16 units each) of the ring kernels, and only their slabs are cleared / summed.
    // RGCN_FLAG_DW_RING / RGCN_FLAG_DW_DIRECT pin the choice (tests exercise both on small graphs).
    const unsigned xb = buffer_bytes(plan->n_nodes, ldx, flags), gb = buffer_bytes(plan->n_owned, ldg, flags);
    const bool can_direct = KP == 64 && NP == 64 && xb != 0 && gb != 0;
    const bool want_direct = can_direct && !(flags & RGCN_FLAG_DW_RING) &&
                             ((flags & RGCN_FLAG_DW_DIRECT) || n_units >= kDwDirectMinUnits);
    const int max_blocks = want_direct ? kDwBlocks : kDwRingBlocks;
    const int nblocks = n_units / 16 < 1 ? 1 : (n_units / 16 > max_blocks ? max_blocks : n_units / 16);
    const size_t slab_bytes = sizeof(float) * (size_t)(nblocks + plan->num_relations + 1) * kDwSlabsPer * KP * NP;
    float* bias_slabs = (float*)workspace + dw_slab_floats(plan->num_relations, KP, NP);
    hipError_t e = hipMemsetAsync(workspace, 0, slab_bytes, s);
    if (e == hipSuccess) e = hipMemsetAsync(bias_slabs, 0, sizeof(float) * (size_t)nblocks * kDwSlabsPer * NP, s);
    if (e != hipSuccess) return (int)e;
    DwArgs a;
    a.rel_order = plan->rel_order + unit_begin;
    a.chunk_rel = plan->chunk_rel;
    a.chunk_cnt = plan->chunk_cnt;
    a.chunk_tile = plan->chunk_tile;
    a.slot_src = plan->slot_src;
    a.slot_w = plan->slot_w;
    a.slot_row = plan->slot_row;
    a.x = x;
    a.g = g;
    a.slabs = (float*)workspace;
    a.bias_slabs = bias_slabs;
    a.ldx = ldx;
    a.x_bytes = xb;
    a.g_bytes = gb;
    a.n_rows = plan->n_nodes;
    a.n_owned = plan->n_owned;
    a.din4 = (din + 3) / 4;
    a.ldg = ldg;
    a.dout4 = (dout + 3) / 4;
    a.tile = plan->tile;
    a.n_units = n_units;
    a.ushift = plan->chunk == 128 ? 1 : 0;
    a.num_rel = plan->num_relations;
    if (want_direct) {
        hipLaunchKernelGGL(rgcn_dw_direct_kernel, dim3(nblocks), dim3(256), 0, s, a);
        st = (int)hipGetLastError();
    } else {
        st = dispatch_dw(KP, NP, a, nblocks, s);
    }
    if (st != RGCN_OK) return st;
    hipLaunchKernelGGL(rgcn_dw_reduce_kernel, dim3(plan->num_relations + 2, (din * dout + 255) / 256), dim3(256), 0, s, a.slabs, a.bias_slabs,
                       a.rel_order, plan->chunk_rel, n_units, a.ushift, nblocks, plan->num_relations, KP, NP, din,
                       dout, d_weight, d_root, d_bias);
    return (int)hipGetLastError();
}

extern "C" int rgcn_dw_tiles_geometry(int* tile, int* walkers, int* max_relations) {
    if (tile) *tile = kDwTileT;
    if (walkers) *walkers = kDwTileWalkers;
    if (max_relations) *max_relations = kDwTileMaxRel;
    return RGCN_OK;
}

extern "C" int rgcn_dw_tiles_walk(const rgcn_plan_t* plan, int32_t* walk_ptr, void* stream) {
    int st;
    if ((st = check_device()) != RGCN_OK) return st;
    if ((st = check_plan(plan)) != RGCN_OK) return st;
    if (walk_ptr == nullptr) return RGCN_ERR_NULL;
    if (plan->tile != kDwTileT || plan->chunk != 64 || plan->layout != 0 || plan->num_relations > kDwTileMaxRel) return RGCN_ERR_PLAN;
    const int n = plan->num_relations * (kDwTileWalkers + 1);
    hipLaunchKernelGGL(rgcn_dw_walk_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, plan->rel_order,
                       plan->chunk_rel, plan->chunk_tile, plan->n_units, plan->n_tiles, plan->num_relations, kDwTileWalkers, walk_ptr);
    return (int)hipGetLastError();
}

extern "C" size_t rgcn_bwd_dw_tiles_workspace_bytes(int num_relations) {
    return num_relations > 0 ? sizeof(float) * (size_t)kDwTileWalkers * num_relations * 64 * 64 : 0;
}

extern "C" int rgcn_bwd_dw_tiles(const rgcn_plan_t* plan, const int32_t* walk_ptr, const float* x, int ldx, int din, const float* g,
                                 int ldg, int dout, void* workspace, size_t workspace_bytes, float* d_weight, unsigned flags,
                                 void* stream) {
    int st = check_plan(plan);
    if (st != RGCN_OK) return st;
    if (!walk_ptr || !x || !g || !workspace || !d_weight) return RGCN_ERR_NULL;
    if ((st = check_stride(ldx, din)) != RGCN_OK) return st;
    if ((st = check_stride(ldg, dout)) != RGCN_OK) return st;
    if (padded_width(din) != 64 || padded_width(dout) != 64) return RGCN_ERR_WIDTH;
    if (plan->tile != kDwTileT || plan->chunk != 64 || plan->layout != 0 || plan->num_relations > kDwTileMaxRel) return RGCN_ERR_PLAN;
    if (workspace_bytes < rgcn_bwd_dw_tiles_workspace_bytes(plan->num_relations)) return RGCN_ERR_WORKSPACE;
    if ((st = check_device()) != RGCN_OK) return st;
    DwTileArgs a;
    a.rel_order = plan->rel_order;
    a.chunk_cnt = plan->chunk_cnt;
    a.chunk_tile = plan->chunk_tile;
    a.slot_src = plan->slot_src;
    a.slot_w = plan->slot_w;
    a.slot_row = plan->slot_row;
    a.walk_ptr = walk_ptr;
    a.x = x;
    a.g = g;
    a.x_bytes = buffer_bytes(plan->n_nodes, ldx, flags);
    a.g_bytes = buffer_bytes(plan->n_owned, ldg, flags);
    if (a.x_bytes == 0 || a.g_bytes == 0) return RGCN_ERR_PLAN;       // this kernel addresses through buffer descriptors only
    a.slabs = (float*)workspace;
    a.ldx = ldx;
    a.ldg = ldg;
    a.dout4 = (dout + 3) / 4;
    a.n_tiles = plan->n_tiles;
    a.n_owned = plan->n_owned;
    a.num_rel = plan->num_relations;
    a.walkers = kDwTileWalkers;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = sizeof(float) * 2 * kDwTileT * 64;
    const bool split = (flags & RGCN_FLAG_SPLIT_PRODUCERS) != 0;
    hipError_t e = split ? allow_full_lds<rgcn_dw_tile_kernel<true>>() : allow_full_lds<rgcn_dw_tile_kernel<false>>();
    if (e != hipSuccess) return (int)e;
    // walkers without tiles leave their slabs untouched: clear what the reduction reads
    e = hipMemsetAsync(workspace, 0, rgcn_bwd_dw_tiles_workspace_bytes(plan->num_relations), s);
    if (e != hipSuccess) return (int)e;
    if (split) hipLaunchKernelGGL(rgcn_dw_tile_kernel<true>, dim3(4 * kDwTileWalkers), dim3(512), lds, s, a);
    else hipLaunchKernelGGL(rgcn_dw_tile_kernel<false>, dim3(4 * kDwTileWalkers), dim3(512), lds, s, a);
    if ((st = (int)hipGetLastError()) != 0) return st;
    hipLaunchKernelGGL(rgcn_dw_tile_reduce_kernel, dim3(plan->num_relations, (din * dout + 255) / 256), dim3(256), 0, s, a.slabs,
                       kDwTileWalkers, plan->num_relations, din, dout, d_weight);
    return (int)hipGetLastError();
}
