// rgcn_dw_tile.hip -- weight gradients of 64 x 64 layers with at most 32 relations by a TILE-MAJOR walk (the gradient rows of
// a tile staged in LDS once per relation quarter; a wave owns one relation for the whole launch), and the entry points
// rgcn_dw_tiles_geometry / rgcn_dw_tiles_walk / rgcn_bwd_dw_tiles of include/rgcn_mi355x.h.
#include "rgcn_kernels_shared.h"

namespace rgcn {

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
// DMA the tile's T = 320 gradient rows into one of two LDS buffers (2 x 80 KiB = all 160 KiB of LDS; 304 until the end of round
// 3: 6.67 against 6.74 ms, fewer half-empty 32-slot halves) a tile ahead, each wave walks the
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
// wave priority raised around the MFMA block of every output-column group (1) or around its vector work (2); 0: none.  Two waves
// share a SIMD and alternate between cutting operands and multiplying: 7.00 / 7.01 ms (1), 7.01 / 6.99 (2) against 7.08 / 7.06 (0),
// A/B on one box -- a per cent, kept at 1.  (The two waves of a SIMD at DIFFERENT priorities for the whole launch, so that they
// fall out of step: 7.0-7.1 ms against 6.76, worse in all three forms tried.)
#ifndef RGCN_DW_PRIO
#define RGCN_DW_PRIO 1
#endif
#ifndef RGCN_DW_PIPE
#define RGCN_DW_PIPE 3     // vector instructions issued behind every MFMA of the split form (0: phases, the round-2 form)
#endif
#ifndef RGCN_DW_STAGED
#define RGCN_DW_STAGED 1   // the A pieces cut four pairs at a time, stage by stage
#endif
#ifndef RGCN_DW_XCD_MAP
#define RGCN_DW_XCD_MAP 1
#endif
// Diagnostic build only (-DRGCN_DW_STAMPS, tools/debug/dw_stamps.py): per-phase cycle sums of every wave of the split form,
// written to a buffer no other code reads.  A stamp is s_memtime + lgkmcnt(0): it also waits for the LDS operations in flight.
#ifdef RGCN_DW_STAMPS
__device__ unsigned long long* g_dw_stamps = nullptr;
__device__ __forceinline__ unsigned dw_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return (unsigned)t;
}
#define DWS(i) { const unsigned now_ = dw_stamp(); ph[i] += now_ - last_; last_ = now_; }
#else
#define DWS(i)
#endif
constexpr int kDwTileT = 320;                    // gradient rows per LDS buffer = tile size of the plan this kernel walks
constexpr int kDwTileWalkers = 64;               // tile ranges; x 4 relation quarters = 256 workgroups, one per CU
constexpr int kDwTileMaxRel = 32;
// Split form only: a wave folds its 64 x 64 accumulator into its own fp32 slab every kDwFlushUnits units (vector adds: round to
// nearest) and starts the next period from zero, with the SIGN of the period's products flipped (the weight of every slot times
// -1 on odd periods; the fold subtracts those).  Why: v_mfma_f32_16x16x32_bf16 aligns every product to the accumulator's exponent
// and TRUNCATES it there -- an error of one sign, proportional to ulp(accumulator), that a long-lived accumulator turns into a
// bias growing with the rows per slab (-7.5e-2 on d_weight at 12M edges per relation against the exact-fp32 form's +2e-4,
// profiles/r03b_*).  A period keeps ulp(accumulator) small; alternating signs make consecutive periods' biases cancel instead
// of add.  The slab is private to the wave: no atomics, the same sum order every run.  Measured at 12M edges per relation
// (profiles/r04a_dw_fold_error.txt: worst error / mean signed error against float64; exact-fp32 form 5.0e-2 / +2.2e-4) and at the
// headline config (launch, A/B on one box, profiles/r04a_dw_fold_timing.txt: no fold 6.65-6.70 ms):
//     period 16 units 3.3e-3 / -1.6e-5, 7.05 ms | 64: 4.8e-3 / -1.8e-4, 6.81 | 128: 6.71 | 256: 1.1e-2 / -2.9e-4, 6.72
//     64 without the sign flip: 2.2e-2 / -1.7e-2 (the flip is worth two orders of magnitude of bias)
// 128 units (~7K rows): within 1 % of the launch without folds, worst error 0.15 x and bias ~1 x the exact-fp32 form's.
// The cost is the fold's read-modify-write draining the wave's prefetch queue (16 x {load 16 B, add, store 16 B} per lane);
// the same fold as 64 no-return global_atomic_add_f32 per lane (RGCN_DW_FOLD_ATOMIC=1, equally deterministic on a private slab)
// was SLOWER: 6.95 ms at 64 units, 7.31 at 32, 8.33 at 16.
#ifndef RGCN_DW_FLUSH_UNITS
#define RGCN_DW_FLUSH_UNITS 128
#endif
#ifndef RGCN_DW_FLUSH_SIGNS
#define RGCN_DW_FLUSH_SIGNS 1
#endif
#ifndef RGCN_DW_FOLD_ATOMIC
#define RGCN_DW_FOLD_ATOMIC 0
#endif
constexpr int kDwFlushUnits = RGCN_DW_FLUSH_UNITS;      // 0: one accumulator for the whole launch (round 2 / 3)

struct DwTileArgs {
    const int* rel_order;   // of a plan with tile = kDwTileT, 64-slot chunks (unit == chunk), layout 0
    const int* chunk_cnt;
    const int* chunk_tile;
    const int* slot_src;
    const float* slot_w;
    const int* slot_row;
    const int* slot_src2;   // PAIRS: [units][8] second rows of the unit's pair heads (slots 32 h + kq, h = 0 / 1: the first four slots
                            // of either half), padding = n_nodes; NULL otherwise
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
// PAIRS (round 4, plan layout 5: rgcn_plan.hip dw_pairs_kernel): the two rows of a (destination, relation) pair take ONE k-slot --
// x[src] + x[src2] is formed in registers before the cut (mean aggregation gives both rows the same weight, and both meet the same
// gradient row): the head of a pair sits on one of the first four slots of a half (register 0 of the half's pipeline), its second
// row arrives by ONE more row load per half with the same lane geometry (padding where a slot has no second row: zeros, no traffic).
template <bool SPLIT, bool PAIRS = false>
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
#ifdef RGCN_DW_STAMPS
    unsigned ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned last_ = dw_stamp();
#endif

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
        int h2;       // PAIRS: lanes 0..7: second rows of slots 0..3 (lanes 0..3) and 32..35 (lanes 4..7)
    };
    auto unit_of = [&](int k) { return ldc(a.rel_order, i0 + (k < nun ? k : (nun > 0 ? nun - 1 : 0))); };
    auto load_idx = [&](int unit) {
        const size_t base = (size_t)unit * kChunk + lane;
        int h2 = 0;
        if constexpr (PAIRS) h2 = a.slot_src2[(size_t)unit * 8 + (lane & 7)];
        return Idx{a.slot_src[base], a.slot_row[base], a.slot_w[base], h2};
    };
    auto issue_half = [&](f32x4 (&a4)[HS], f32x4& pair, const Idx& ix, int h) {
        if constexpr (PAIRS) {      // the second rows of slots 32 h + kq: lane 4 h + kq of the index vector
            const int i2 = __builtin_amdgcn_ds_bpermute(perm + 16 * h, ix.h2);
            pair = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(__umul24((unsigned)i2, rbx) + colb), 0, 0));
        }
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
    auto compute_half = [&](f32x4 (&a4)[HS], const f32x4& pair, const Idx& ix, int h, int ngrp, int nks, const float* gbuf, int tile_row0) {
        if constexpr (PAIRS) a4[0] += pair;
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
        split3_pair(v0, v1, h, m, l);
    };
    // half a unit as ONE 32-row k-step (a half with no valid slot is skipped; padding slots inside one have weight 0)
    auto compute_half3 = [&](f32x4 (&a4)[HS], const f32x4& pair, const Idx& ix, int h, int ngrp, const float* gbuf, int tile_row0) {
        if (2 * h >= ngrp) return;      // (ix.w carries the period's sign, see fold_into_slab)
        if constexpr (PAIRS) a4[0] += pair;
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
        DWS(2 + 5 * h)       // weights and gradient rows of the half out of LDS
        u32x4 ap[3][4];      // [piece][ia]: 8 bf16 = k index 8 kq + 0..7 of input channel 4 ml + ia
#if RGCN_DW_STAGED
        // the four pairs of an input channel cut side by side, stage by stage (split3_pair's arithmetic, same pieces): left
        // alone the scheduler emits one dependent chain after the other with a wait state behind every conversion
#pragma unroll
        for (int ia = 0; ia < 4; ++ia) {
            float v0[4], v1[4];
            unsigned pc[4];
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) { v0[jp] = a4[2 * jp][ia]; v1[jp] = a4[2 * jp + 1][ia]; }
#pragma unroll
            for (int piece = 0; piece < 3; ++piece) {
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) pc[jp] = cvt_pk_bf16(v0[jp], v1[jp]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) ap[piece][ia][jp] = pc[jp];
                if (piece < 2) {
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) {
                        v0[jp] -= __uint_as_float(pc[jp] << 16);
                        v1[jp] -= __uint_as_float(pc[jp] & 0xFFFF0000u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#else
#pragma unroll
        for (int ia = 0; ia < 4; ++ia)
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                unsigned h_, m_, l_;
                split_pair(a4[2 * jp][ia], a4[2 * jp + 1][ia], h_, m_, l_);
                ap[0][ia][jp] = h_; ap[1][ia][jp] = m_; ap[2][ia][jp] = l_;
            }
#endif
        DWS(3 + 5 * h)       // the x rows have arrived and are cut
        constexpr int pa[6] = {2, 1, 1, 0, 0, 0}, pb[6] = {0, 1, 0, 2, 1, 0};      // small products first
#if RGCN_DW_PIPE
        // Software pipeline over the four output-column groups: the B pieces of group jb + 1 are cut in the shadow of group jb's
        // 24 MFMAs.  A 16x16x32 bf16 MFMA holds the matrix pipe for 16 cycles = four issue slots, of which it takes one: the
        // other three go to the SAME wave's vector instructions (the second wave of the SIMD does not fill them: its vector
        // phase and this wave's MFMA phase ran one after the other, MFMA time came on top of everything else -- DESIGN.md 4.3).
        // The order is forced with sched_group_barrier (one MFMA, then up to RGCN_DW_PIPE vector instructions, 24 times);
        // left alone the scheduler keeps runs of 24 MFMAs.
        auto cut_b = [&](int jb, u32x4 (&bp)[3]) {
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                unsigned h_, m_, l_;
                if (RGCN_DW_ABL & 8) {      // (timing only: the B side without its cut -- what pieces staged per tile could save at most)
                    h_ = __float_as_uint(g4[2 * jp][jb]); m_ = __float_as_uint(g4[2 * jp + 1][jb]); l_ = __float_as_uint(wv[2 * jp]) ^ __float_as_uint(wv[2 * jp + 1]);
                } else
                split_pair(g4[2 * jp][jb] * wv[2 * jp], g4[2 * jp + 1][jb] * wv[2 * jp + 1], h_, m_, l_);
                bp[0][jp] = h_; bp[1][jp] = m_; bp[2][jp] = l_;
            }
        };
        u32x4 bpp[2][3];
#if RGCN_DW_STAGED
        {      // group 0 is cut in the open: staged like the A pieces
            float v0[4], v1[4];
            unsigned pc[4];
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) { v0[jp] = g4[2 * jp][0] * wv[2 * jp]; v1[jp] = g4[2 * jp + 1][0] * wv[2 * jp + 1]; }
#pragma unroll
            for (int piece = 0; piece < 3; ++piece) {
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) pc[jp] = cvt_pk_bf16(v0[jp], v1[jp]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) bpp[0][piece][jp] = pc[jp];
                if (piece < 2) {
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) {
                        v0[jp] -= __uint_as_float(pc[jp] << 16);
                        v1[jp] -= __uint_as_float(pc[jp] & 0xFFFF0000u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#else
        cut_b(0, bpp[0]);
#endif
        __builtin_amdgcn_sched_barrier(0);
        DWS(4 + 5 * h)       // column group 0 of B cut
        if (RGCN_DW_PRIO == 1) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            if (jb < 3) cut_b(jb + 1, bpp[(jb + 1) & 1]);
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int ia = 0; ia < 4; ++ia)
                    acc[ia][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[pa[q]][ia]),
                                                                          __builtin_bit_cast(bf16x8, bpp[jb & 1][pb[q]]), acc[ia][jb], 0, 0, 0);
            if (jb < 3) {
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, RGCN_DW_PIPE, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (RGCN_DW_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        DWS(5 + 5 * h)       // 96 MFMAs issued, the cuts of groups 1-3 between them
        return;
#endif
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
            if (RGCN_DW_PRIO == 1) __builtin_amdgcn_s_setprio(3);
            if (RGCN_DW_PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int ia = 0; ia < 4; ++ia)
                    acc[ia][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[pa[q]][ia]),
                                                                          __builtin_bit_cast(bf16x8, bp[pb[q]]), acc[ia][jb], 0, 0, 0);
            if (RGCN_DW_PRIO == 1) __builtin_amdgcn_s_setprio(0);
            if (RGCN_DW_PRIO == 2) __builtin_amdgcn_s_setprio(3);
        }
    };

    // The halves are computed under conditions (a half without valid slots is skipped), and loads whose uses all sit in later
    // blocks get SUNK there by the optimiser -- issued right in front of their first use, their whole latency exposed (that is
    // where the second half's eight row loads of every unit were until this was found: the ISA showed them behind half 0's
    // MFMAs, not at the top of the iteration; sched_barrier only binds the scheduler inside a block).  A compiler-level memory
    // clobber after each batch keeps the loads where issue_half puts them: a read cannot be moved across it.
    auto pin_loads = [] { asm volatile("" ::: "memory"); };
    // slab += sgn * acc, acc = 0 (the slab was cleared by the host-side memset).  Both forms fold: the exact-fp32 form rounds every
    // product to nearest, so it has no bias to cancel (sgn stays +1), but a chain of ~7K rows instead of a launch's worth keeps its
    // rounding errors those of a blocked sum.  Its MFMAs are inline asm the compiler's hazard recognizer does not see: explicit wait
    // states before the accumulators are read (as in front of the final store)
    float* const slab = a.slabs + ((size_t)p * a.num_rel + (have ? rel : 0)) * (64 * 64);
    auto fold_into_slab = [&](float sgn) {
        if (!have) return;
        if constexpr (!SPLIT) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#if RGCN_DW_FOLD_ATOMIC
#pragma unroll
        for (int ia = 0; ia < 4; ++ia)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    unsafeAtomicAdd(slab + (4 * (4 * kq + r) + ia) * NP + 4 * ml + jb, sgn * acc[ia][jb][r]);
                    acc[ia][jb][r] = 0.f;
                }
        return;
#endif
#pragma unroll
        for (int ia = 0; ia < 4; ++ia)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4* q = (f32x4*)(slab + (4 * (4 * kq + r) + ia) * NP + 4 * ml);
                f32x4 v = *q;
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    v[jb] += sgn * acc[ia][jb][r];
                    acc[ia][jb][r] = 0.f;
                }
                *q = v;
            }
    };
    float sgn = 1.f;
    // the fold period of THIS wave: kDwFlushUnits where the wave has a long walk (the headline: ~1,000 units per wave), an eighth
    // of its walk (at least 16 units) where it is shorter -- there the folds cost nothing that matters and a chain of ~1K rows
    // instead of ~7K keeps the sums as accurate as ATen's blocked ones (tests/test_gpu_parity.py holds 2 x the CPU loop's error)
    // (the exact-fp32 form has no bias to cancel, only chains to keep short: four times the period)
    const int fold_period = kDwFlushUnits > 0 ? min(SPLIT ? kDwFlushUnits : 4 * kDwFlushUnits, max(16, nun >> 3)) : 0;
    int fold_left = fold_period;
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
    f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};
    if (nun > 0) issue_half(s0, p0, ix_cur, 0);
    constexpr int kInFlight = (RGCN_DW_VECTOR_WALK ? 14 : 11) + (PAIRS ? 2 : 0);      // 8 (+ 1) row loads + 3 (+ 1) index loads (+ 3 walk words)
    bool walked = nun > 0;      // at least kInFlight vector-memory operations were issued after the pending tile's DMAs
    for (int t = t0; t < t1; ++t) {
        // The DMAs of tile t were issued a tile ago (or in the prologue).  If the wave has walked a unit since (or issued the
        // prologue's loads), more than kInFlight younger operations exist and at most kInFlight are in flight at a unit boundary (8 row
        // loads + 3 index loads of the unit after next + the walk words): a counted wait retires the DMAs and leaves the prefetches alone.  A wave
        // without units in between (an empty relation) has nothing younger to count: it waits for everything.
        DWS(12)
        if (walked) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kInFlight) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DWS(13)              // the tile's DMAs
#if !RGCN_DW_ABL_NOBARRIER      // (timing only: the waves of a workgroup run free -- what the tile lockstep costs)
        wg_barrier();          // tile t landed for every wave; every wave is done with the buffer tile t + 1 goes to
#endif
        DWS(14)              // the barrier
        const int b = (t - t0) & 1;
        if (t + 1 < t1) dma_tile(t + 1, b ^ 1);
        walked = false;
        DWS(15)              // next tile's DMAs issued
        const float* gbuf = lds + b * T * NP;
        while (k < nun && tile_cur == t) {
            const int ngrp = (cnt_cur + 15) >> 4, nks = (cnt_cur + 3) >> 2;
            walked = true;
            DWS(0)               // (loop overhead, walk words)
            issue_half(s1, p1, ix_cur, 1);
            pin_loads();
            __builtin_amdgcn_sched_barrier(0);
            DWS(1)               // second half's rows issued
            if constexpr (SPLIT && kDwFlushUnits > 0 && RGCN_DW_FLUSH_SIGNS) ix_cur.w *= sgn;
            if constexpr (SPLIT) compute_half3(s0, p0, ix_cur, 0, ngrp, gbuf, t * T);
            else compute_half(s0, p0, ix_cur, 0, ngrp, nks, gbuf, t * T);
            __builtin_amdgcn_sched_barrier(0);
            const Idx ix_nn = load_idx(uid_nn);
            issue_half(s0, p0, ix_nxt, 0);
            pin_loads();
            __builtin_amdgcn_sched_barrier(0);
            DWS(6)               // next unit's indices and first-half rows issued
            if constexpr (SPLIT) compute_half3(s1, p1, ix_cur, 1, ngrp, gbuf, t * T);
            else compute_half(s1, p1, ix_cur, 1, ngrp, nks, gbuf, t * T);
            __builtin_amdgcn_sched_barrier(0);
            ++k;
            if constexpr (kDwFlushUnits > 0) {
                if (--fold_left == 0) {      // wave-uniform
                    fold_left = fold_period;
                    fold_into_slab(sgn);
                    if (SPLIT && RGCN_DW_FLUSH_SIGNS) sgn = -sgn;
                }
            }
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
#ifdef RGCN_DW_STAMPS
    ph[11] = (unsigned)k;
    if (g_dw_stamps && lane == 0)
        for (int i = 0; i < 16; ++i) g_dw_stamps[((size_t)blockIdx.x * 8 + wave) * 16 + i] = ph[i];
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the accumulators are read by plain stores the compiler schedules: keep them clear of the last asm MFMA
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if constexpr (kDwFlushUnits > 0) {
        fold_into_slab(sgn);
        return;
    }
    if (have) {
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

}  // namespace rgcn

using namespace rgcn;

#ifdef RGCN_DW_STAMPS
extern "C" int rgcn_debug_set_dw_stamps(unsigned long long* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dw_stamps), &p, sizeof(p));
}
#endif

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
    if (plan->tile != kDwTileT || plan->chunk != 64 || (plan->layout != 0 && plan->layout != 5) || plan->num_relations > kDwTileMaxRel) return RGCN_ERR_PLAN;
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
    if (plan->tile != kDwTileT || plan->chunk != 64 || (plan->layout != 0 && plan->layout != 5) || plan->num_relations > kDwTileMaxRel) return RGCN_ERR_PLAN;
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
    if (a.x_bytes == 0 || a.g_bytes == 0) return RGCN_ERR_ADDRESS;    // this kernel addresses through buffer descriptors only
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
    a.slot_src2 = plan->layout == 5 ? plan->slot_src2 : nullptr;
    if (plan->layout == 5 && a.slot_src2 == nullptr) return RGCN_ERR_NULL;
    hipError_t e = split ? allow_full_lds<rgcn_dw_tile_kernel<true>>() : allow_full_lds<rgcn_dw_tile_kernel<false>>();
    if (e != hipSuccess) return (int)e;
    // walkers without tiles leave their slabs untouched: clear what the reduction reads
    e = hipMemsetAsync(workspace, 0, rgcn_bwd_dw_tiles_workspace_bytes(plan->num_relations), s);
    if (e != hipSuccess) return (int)e;
    if (a.slot_src2 != nullptr) {      // plan layout 5: pairs of rows on one k-slot
        e = split ? allow_full_lds<rgcn_dw_tile_kernel<true, true>>() : allow_full_lds<rgcn_dw_tile_kernel<false, true>>();
        if (e != hipSuccess) return (int)e;
        if (split) hipLaunchKernelGGL((rgcn_dw_tile_kernel<true, true>), dim3(4 * kDwTileWalkers), dim3(512), lds, s, a);
        else hipLaunchKernelGGL((rgcn_dw_tile_kernel<false, true>), dim3(4 * kDwTileWalkers), dim3(512), lds, s, a);
    } else if (split) hipLaunchKernelGGL(rgcn_dw_tile_kernel<true>, dim3(4 * kDwTileWalkers), dim3(512), lds, s, a);
    else hipLaunchKernelGGL(rgcn_dw_tile_kernel<false>, dim3(4 * kDwTileWalkers), dim3(512), lds, s, a);
    if ((st = (int)hipGetLastError()) != 0) return st;
    hipLaunchKernelGGL(rgcn_dw_tile_reduce_kernel, dim3(plan->num_relations, (din * dout + 255) / 256), dim3(256), 0, s, a.slabs,
                       kDwTileWalkers, plan->num_relations, din, dout, d_weight);
    return (int)hipGetLastError();
}