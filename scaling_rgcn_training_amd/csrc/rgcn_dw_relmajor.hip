// rgcn_dw_relmajor.hip -- weight gradients of the R-GCN layer by RELATION-MAJOR walks of the forward plan (every width class),
// and the entry point rgcn_bwd_dw of include/rgcn_mi355x.h (autograd of the reference: model/modelTrainer.py:66).
//   rgcn_dw_kernel        any padded width: per relation, dB_rel += (w_e x[src_e])^T g[dst_e]; ring + register accumulators
//   rgcn_dw_wide_kernel   widths that are multiples of 64 (64 x 64 walks too short for the direct kernel, 64 x 128, 128 x 64)
//   rgcn_dw_direct_kernel 64 x 64, large walks: no ring, every wave gathers its own rows into registers
//   rgcn_dw_reduce_kernel fixed-order sum of the slabs -> d_weight / d_root / d_bias
#include "rgcn_kernels_shared.h"

namespace rgcn {

constexpr int kDwBlocks = 512;  // most workgroups a dW launch uses (sizes the slab workspace): two per CU for the direct
                                // kernel, one per CU (LDS-bound) for the ring kernels
constexpr int kDwRingBlocks = 256;
constexpr int kDwDirectMinUnits = 16 * 1024;   // >= 8 units per wave of 512 four-wave workgroups

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

using namespace rgcn;

#ifdef RGCN_STAMPS
extern "C" int rgcn_debug_set_stamps_dw(unsigned long long* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(rgcn::g_stamps), &p, sizeof(p));
}
#endif

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
    if (plan->layout == 3) return RGCN_ERR_PLAN;       // (shadow slots: only the forward / dX kernel knows them)
    if (plan->layout == 5) return RGCN_ERR_PLAN;       // (second rows of pairs: only the tile-major kernel adds them)
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
        if (plan->layout == 2) return RGCN_ERR_PLAN;       // (the root units' position follows from the tile geometry)
        // (both plan layouts put ceil(rows / 16) row tiles of a group on contiguous tiles of its chunks)
        auto units_of = [&](long rows) -> long { return ((rows + 15) / 16 + 3) / 4; };
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
    // dense relation-major units (layout 2, the edge-parallel path's graphs): 4 units per workgroup instead of 16 -- four times
    // the workgroups on graphs of a few hundred units, and accumulation chains of at most 256 rows before the fixed-order slab
    // sum takes over (one chain over all rows of a relation measured 2.8 x the error of the CPU loop's blocked GEMM)
    const int upb = plan->layout == 2 ? 4 : 16;
    const int nblocks = n_units / upb < 1 ? 1 : (n_units / upb > max_blocks ? max_blocks : n_units / upb);
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

