// rgcn_tile3p.hip -- forward / dX of the R-GCN layer for gfx950 with the contraction on bf16 MFMAs and the operand split
// done ONCE per gathered row, by the PRODUCER waves (64 -> 64 layers, 128-slot chunks, layout-0 plans).
//
// Replaces the same arithmetic as rgcn_tile_kernel (torch_geometric.nn.RGCNConv.forward and its dX: reference
// model/layers.py:21,23, model/modelTrainer.py:66) and keeps its structure -- 4 producer + 4 consumer waves per workgroup,
// a two-slot ring in LDS, the fp32 accumulator tile in LDS, Y^T accumulate, run-sum path for repeated destinations -- with
// two changes that DESIGN.md 4.5 / 4.6 derive from measurements:
//   * x W = (xh + xm + xl)(Wh + Wm + Wl) with bf16 pieces and the six products hh, hm, mh, hl, lh, mm on
//     v_mfma_f32_16x16x32_bf16: 24 significant bits on both operands, i.e. fp32-equivalent (error against float64 no larger
//     than the sequential fp32 chain's), for 2.7x fewer cycles of the SIMD's arithmetic pipe than v_mfma_f32_16x16x4_f32;
//   * the 3-way split of x (5.5 vector instructions per element) is what killed that gain when the CONSUMERS did it per
//     row tile and per wave (the consumer-split kernel of round 2, deleted in round 3: DESIGN.md 4.6).  Here the producers gather rows into REGISTERS (buffer loads, two chunks
//     ahead), split them once and write three bf16 planes into the ring; their vector work overlaps the consumers' bf16
//     MFMAs on the shared SIMD (tools/probes/mfma_cross_wave_overlap.hip: a VALU wave slows 1.3x beside bf16 MFMAs, the
//     MFMA wave not at all).  W is split at pack time (rgcn_pack3_kernel's planes, already part of the packed weights).
// Ring slot: [plane 0..2][128 rows][64 bf16] = 48 KiB (16-byte pieces of a row XOR-swizzled by (row >> 1) & 7 so that
// the consumers' ds_read_b128 of 16 rows x 4 k-groups are conflict free); two slots + the [tile + 1][68] fp32 accumulator
// fit 160 KiB up to tile = 224.
#include <type_traits>
#include "rgcn_common.h"
#include "rgcn_tile_common.h"

namespace rgcn {

// timing-only ablations of diagnostic builds (wrong results): 1 producers load nothing, 2 consumers skip their MFMAs,
// 4 consumers skip the accumulator read-modify-write, 8 producers skip the split (planes = raw halves)
#ifndef RGCN_P3_ABL
#define RGCN_P3_ABL 0
#endif
// Diagnostic build only (-DRGCN_P3_STAMPS, tools/debug/p3_stamps.py): per-phase cycle sums of producer wave 0 and consumer
// wave 4 of every workgroup, written to a buffer no other code reads.
#ifdef RGCN_P3_STAMPS
__device__ unsigned long long* g_p3_stamps = nullptr;
__device__ __forceinline__ unsigned long long p3_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define P3S(v) const unsigned long long v = p3_stamp()
#define P3A(acc, a, b) acc += (b) - (a)
#else
#define P3S(v)
#define P3A(acc, a, b)
#endif
// row tiles whose operands the consumers read ahead of the one they multiply (LDS round trip under load: 300-450 cycles,
// 12 bf16 MFMAs: ~200).  Round 2 settled on 2; with the consumer waves at priority 3 (their reads go ahead of the producers')
// and chunks of 4.3 row tiles (layout 3) one tile ahead is as good or better: forward / dX 8.31 / 8.39 ms against 8.40 / 8.42
// (and 8.59 / 8.56 at 3), 8.70 / 8.64 against 8.76 / 8.72 on layout-0 plans, A/B on one box, bit-identical results
#ifndef RGCN_P3_LA
#define RGCN_P3_LA 1
#endif
// 0: round-to-nearest pieces (v_cvt_pk_bf16_f32); 1: the producers split by truncation (v_and / v_sub / v_perm_b32: exact too,
// but measured 0.25 ms per launch SLOWER: 9.94 against 9.67 ms)
#ifndef RGCN_P3_TRUNC
#define RGCN_P3_TRUNC 0
#endif
// wave priorities (s_setprio 0..3) of the producer / consumer waves: issue is arbitrated by priority, then age.  The consumers are the
// critical path of a chunk; with priority 3 their LDS and MFMA issue goes ahead of the producer wave on the same SIMD: forward launch
// 8.87 / 8.87 ms against 9.11 / 8.96 at equal priorities, A/B on one box (producers 3: 9.05 / 9.04; producers 1 + consumers 2: 8.85)
#ifndef RGCN_P3_PRIO_PROD
#define RGCN_P3_PRIO_PROD 0
#endif
#ifndef RGCN_P3_PRIO_CONS
#define RGCN_P3_PRIO_CONS 3
#endif
// which waves are consumers: 0 = waves 4 .. (one producer and one consumer per SIMD with the one-team kernel: wave i runs on SIMD i % 4);
// 1 (one-team kernel only) = waves 0, 1, 4, 5 -- the consumers two per SIMD on SIMDs 0 / 1, the producers on SIMDs 2 / 3
#ifndef RGCN_P3_ROLEMAP
#define RGCN_P3_ROLEMAP 0
#endif
// 1: the consumers drain their LDS queue (s_waitcnt lgkmcnt(0)) in front of EVERY chunk barrier; 0: only where a tile closes.
// Per-wave stamps (round 3, profiles/r03a_*) show every consumer wave waiting ~400 cycles per chunk at that barrier with the
// producers long there -- the drain of its last accumulator stores; without it the same wait moves to the next chunk's first
// lgkmcnt(0) (scalar metadata): 9.39 / 9.33 ms with, 9.42 / 9.32 without, A/B on one box.  Kept at 1.
#ifndef RGCN_P3_DRAIN
#define RGCN_P3_DRAIN 1
#endif

constexpr int kP3Threads = 512;                          // 4 producer + 4 consumer waves
constexpr int kP3CH = 128;                               // slots of a plan chunk (stride of the plan's slot arrays)
// Row tiles a ring slot has room for (template parameter ST of the kernel).  8: any 128-slot chunk (48 KiB slots: tiles up to
// 224).  7 (round 4): plans whose chunks hold at most 112 rows (rgcn_plan.chunk_rows; the shadow row tiles of a layout-3 chunk
// never reach LDS) -- 42 KiB slots leave the accumulator room for tiles up to 272: 18 % fewer chunks at the headline config.
// Timing-only builds with clamped chunks (profiles/r04k_*, r04l_*: -DRGCN_P3_SLOT_TILES_EXPERIMENT=6 / 7) priced it first:
// 7 tiles at T = 272: 7.80 / 7.89 ms against 8.27 / 8.30; 6 tiles at T = 304 / 320: 7.80 / 7.75.
template <int ST>
struct P3Geo {
    static constexpr int kPlaneBytes = 16 * ST * 128;       // one bf16 plane of a slot
    static constexpr int kSlotBytes = 3 * kPlaneBytes;
};
constexpr int kP3LDO = kAccStride<64>;
constexpr int kP3FragsPerRel = 2 * 3 * 2 * 2;            // rgcn_pack3_kernel: [column half c][plane][column tile ct][k-step s]

__device__ __forceinline__ unsigned p3_cvt_pk_bf16(float lo, float hi) {      // RNE, lo -> bits 0..15, hi -> bits 16..31
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// ---- producers ----------------------------------------------------------------------------------------------------
// Wave pw moves rows 4 pw .. 4 pw + 3 of EVERY 16-row tile of a chunk (load i <-> row tile i): 8 buffer loads of 4 rows x
// 256 B (lane = 16-byte piece c of row rq), two chunks ahead in registers; split and 3 ds_write_b64 per USED row tile -- a
// chunk at tile 224 holds 4.5 of its 8 row tiles on average, and with this dealing every wave skips the same unused ones
// (their loads are still issued -- padding rows are out of range and cost no memory traffic -- so that the number of
// operations in flight stays what the waits count).  The chunk's weights / run metadata go by LDS-DMA (one
// 256-byte dma4 per wave and chunk).
struct P3Rows {     // what one producer batch loads: set j % 3 holds chunk j
    f32x4 v[8];    // the wave's 32 rows of chunk j (lane: 16-byte piece c of row 4 i + rq)
    int meta;      // chunk j's metadata of the wave's OWN 32 rows: lanes 0..31 their weights, lanes 32..63 their slot_acc words
                   // (run metadata; in a layout-3 chunk with flag bit 19 a shadow row's weight / its head's weight)
    int idx;       // row indices of chunk j + 2 (lanes 0..31)
};

// Every vector-memory operation of a producer wave is an inline-asm load into registers, waited for by ONE s_waitcnt
// vmcnt(0) at the top of the next iteration, BEFORE anything new is issued -- a wait that names the loaded registers as
// operands, so that no use of them can be scheduled above it.  Left to hipcc the waits were wrong for this pipeline twice
// over: (1) with the chunk's metadata on LDS-DMA it put vmcnt(0) in front of every LDS store ("a DMA into LDS may be
// pending"), i.e. behind the gathers just issued; (2) without any DMA its counted waits for the loop-carried row registers
// still came out as vmcnt(8..0) behind the new gathers.  Either way the wave waited for loads it had just issued.
// Batches are issued TWO iterations before they are used (three register sets): with the MFMAs at bf16 rate an iteration
// is shorter than a trip to memory, and a one-iteration lead left the whole workgroup waiting for the next chunk's INDEX
// load -- 5.8 of 9.7 ms with every other cost ablated away.
// Iteration it, wave pw: [wait until only the youngest batch is outstanding: set (it + 1) % 3 = rows + metadata of chunk
// it + 1 and the indices of chunk it + 3] [issue into set it % 3: rows + metadata of chunk it + 3, indices of chunk it + 5]
// [split + store chunk it + 1] [barrier].
__device__ __forceinline__ void p3_load_rows(f32x4& dst, i32x4 rsrc, unsigned voff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void p3_load_int(int& dst, const int* p) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// wait until at most N vector-memory operations of this wave are outstanding; the batch names its registers as operands so
// that no use of them can be scheduled above the wait
template <int N>
__device__ __forceinline__ void p3_wait_batch(P3Rows& r) {
    asm volatile("s_waitcnt vmcnt(%10)"
                 : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]), "+v"(r.v[7]),
                   "+v"(r.meta), "+v"(r.idx)
                 : "n"(N)
                 : "memory");
}

template <int ST>
__device__ __forceinline__ void p3_producer_loop(const TileArgs& a, char* ring, float* wring, int* dring, int c0, int nch,
                                                 int lane, int pw, int tile0) {
    constexpr int kP3PlaneBytes = P3Geo<ST>::kPlaneBytes, kP3SlotBytes = P3Geo<ST>::kSlotBytes, kP3SlotTiles = ST;
    // c0 / nch: the chunks of ALL the tiles the workgroup walks, one sequence for the ring; where a chunk closes a tile the
    // consumers store and reset the accumulator: the producers join one extra barrier there (equal barrier counts)
    int tile_cur = tile0;
    int tend = ldc(a.tile_ptr, tile0 + 1) - c0;
    auto tile_boundary = [&](int it) {
        if (it + 1 == tend && it + 1 < nch) {
            ++tile_cur;
            tend = ldc(a.tile_ptr, tile_cur + 1) - c0;
            wg_barrier();
        }
    };
    const int c = lane & 15, rq = lane >> 4;
    // raw buffer descriptor of x (stride 0, offen addressing, range check on num_records: rgcn_common.h make_rsrc)
    const unsigned long long xb = (unsigned long long)a.x;
    i32x4 rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb);
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)((xb >> 32) & 0xFFFFull));
    rsrc[2] = __builtin_amdgcn_readfirstlane((int)a.x_bytes);
    rsrc[3] = 0x00020000;
    const unsigned coff = c < a.din4 ? (unsigned)c * 16u : 0xFFFFFFF0u;      // beyond the width: out of range -> zeros
    const unsigned rowb = c < a.din4 ? (unsigned)a.ldx * 4u : 0u;
    auto idx_ptr = [&](int k) {
        const int kk = k < nch ? k : nch - 1;
        return a.slot_src + (size_t)(c0 + kk) * kP3CH + 16 * ((lane & 31) >> 2) + 4 * pw + (lane & 3);      // lane 4 i + rq: row 16 i + 4 pw + rq
    };
    auto issue_loads = [&](P3Rows& r, int idxv) {
        int idx[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) idx[i] = __builtin_amdgcn_ds_bpermute((4 * i + rq) * 4, idxv);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (RGCN_P3_ABL & 1) {
                r.v[i] = f32x4{(float)idx[i], 1.f, 2.f, 3.f};
                continue;
            }
            p3_load_rows(r.v[i], rsrc, __umul24((unsigned)idx[i], rowb) + coff);
        }
    };
    // the ring's metadata: [2][128] pairs {weight, run metadata} -- one 8-byte read per row tile for the consumers.  A wave moves
    // the two words of its OWN 32 rows (lane & 31 <-> row 16 i + 4 pw + rq as for the indices; lanes 0..31 the weight, 32..63 the
    // slot_acc word), so that the weight ratio of a layout-3 shadow row is in the wave that adds that row
    // (plans without such rows keep the round-2 mapping: waves 0, 1 the weights of slots 0..63 / 64..127, waves 2, 3 the slot_acc
    // words -- two coalesced 256-byte loads)
    const int mrow = 16 * ((lane & 31) >> 2) + 4 * pw + (lane & 3);
    const int half = pw & 1;
    const int* meta_src = a.merged ? (lane < 32 ? (const int*)a.slot_w : a.slot_acc) + mrow
                                   : (pw < 2 ? (const int*)a.slot_w : a.slot_acc) + 64 * half + lane;
    int* meta_dst = a.merged ? (int*)wring + 2 * mrow + (lane >> 5) : (int*)wring + 2 * (64 * half + lane) + (pw < 2 ? 0 : 1);
    auto issue_meta = [&](P3Rows& r, int k) { p3_load_int(r.meta, meta_src + (size_t)(c0 + k) * kP3CH); };
    auto store_meta = [&](const P3Rows& r, int k) { meta_dst[(k & 1) * 2 * kP3CH] = r.meta; };
    // fl: the chunk's flags -- bits 16-17 / 18 of a layout-3 chunk (rgcn_plan.hip compact_runs_kernel): row tiles 7, 6 hold the
    // second rows of the (destination, relation) runs whose first rows sit in row tiles 0, 1, the row tile below them (6 with one
    // tile of second rows, 5 with two) the third rows of row tile 0's: same lane, added in fp32 before the cut (aggregate, then
    // transform); bit 19: a shadow row times its weight ratio first.  One uniform branch where a chunk has no shadows; the sums replace the head rows in place (the set is reloaded anyway).
    auto split_store = [&](P3Rows& r, int k, int nrt, int fl) {
        char* slot = ring + (k & 1) * kP3SlotBytes;
        if (fl & (7 << 16)) {
            const int ns1 = (fl >> 16) & 3;
            const bool ns2 = ((fl >> 18) & 1) != 0, pre = ((fl >> 19) & 1) != 0;
            if (pre) {      // the rows of a run differ in weight -- a shadow row times (its weight / its head's)
                auto wt = [&](int t) { return __int_as_float(__builtin_amdgcn_ds_bpermute((32 + 4 * t + rq) * 4, r.meta)); };
                r.v[0] += r.v[7] * wt(7);
                if (ns2) r.v[0] += ns1 >= 2 ? r.v[5] * wt(5) : r.v[6] * wt(6);
                if (ns1 >= 2) r.v[1] += r.v[6] * wt(6);
            } else {
                r.v[0] += r.v[7];
                if (ns2) r.v[0] += ns1 >= 2 ? r.v[5] : r.v[6];
                if (ns1 >= 2) r.v[1] += r.v[6];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i >= nrt) break;                   // row tiles the chunk does not use
            const int row = 16 * i + 4 * pw + rq;
            const f32x4 vi = r.v[i];
            float x0 = vi[0], x1 = vi[1], x2 = vi[2], x3 = vi[3];
            if (RGCN_P3_ABL & 8) {
                char* p8 = slot + row * 128 + (((c >> 1) ^ ((row >> 1) & 7)) << 4) + ((c & 1) << 3);
                *(uint2*)p8 = make_uint2(__float_as_uint(x0), __float_as_uint(x1));
                *(uint2*)(p8 + kP3PlaneBytes) = make_uint2(__float_as_uint(x2), __float_as_uint(x3));
                *(uint2*)(p8 + 2 * kP3PlaneBytes) = make_uint2(__float_as_uint(x0), __float_as_uint(x3));
                continue;
            }
            unsigned h0, h1, m0, m1, l0, l1;
            if (RGCN_P3_TRUNC) {
                // split by TRUNCATION: h = the top 8 significant bits of x, m = those of x - h, l = x - h - m (8 bits at most):
                // x = h + m + l exactly, every step a full-rate instruction (v_and / v_sub / v_perm to pack two upper halves);
                // v_cvt_pk_bf16_f32 issues at about a third of that rate
                auto pk = [](float lo, float hi) { return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u); };
                auto top = [](float v) { return __uint_as_float(__float_as_uint(v) & 0xFFFF0000u); };
                h0 = pk(x0, x1); h1 = pk(x2, x3);
                x0 -= top(x0); x1 -= top(x1); x2 -= top(x2); x3 -= top(x3);
                m0 = pk(x0, x1); m1 = pk(x2, x3);
                x0 -= top(x0); x1 -= top(x1); x2 -= top(x2); x3 -= top(x3);
                l0 = pk(x0, x1); l1 = pk(x2, x3);
            } else {
                split3_pair(x0, x1, h0, m0, l0);
                split3_pair(x2, x3, h1, m1, l1);
            }
            char* p = slot + row * 128 + (((c >> 1) ^ ((row >> 1) & 7)) << 4) + ((c & 1) << 3);
            if (RGCN_P3_ABL & 32) {      // timing only: no plane stores
                asm volatile("" ::"v"(h0), "v"(h1), "v"(m0), "v"(m1), "v"(l0), "v"(l1), "v"(p));
                continue;
            }
            *(uint2*)p = make_uint2(h0, h1);
            *(uint2*)(p + kP3PlaneBytes) = make_uint2(m0, m1);
            *(uint2*)(p + 2 * kP3PlaneBytes) = make_uint2(l0, l1);
        }
    };
    auto issue_batch = [&](P3Rows& r, int j, int idx_rows) {      // rows + metadata of chunk j, indices of chunk j + 2: 10 loads
        issue_loads(r, idx_rows);
        issue_meta(r, j < nch ? j : nch - 1);
        p3_load_int(r.idx, idx_ptr(j + 2));
    };
    P3Rows r0, r1, r2;
    auto clear = [&](P3Rows& r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        r.meta = r.idx = 0;
    };
    clear(r0); clear(r1); clear(r2);
    // prologue: indices of chunks 0 and 1, then the batches of chunks 0, 1, 2
    p3_load_int(r1.idx, idx_ptr(0));
    p3_load_int(r2.idx, idx_ptr(1));
    p3_wait_batch<0>(r1);
    p3_wait_batch<0>(r2);
    const int i0 = r1.idx, i1 = r2.idx;
    issue_batch(r0, 0, i0);                        // + indices of chunk 2
    issue_batch(r1, 1, i1);                        // + indices of chunk 3
    p3_wait_batch<10>(r0);                         // all but the youngest batch: rows of chunk 0, indices of chunk 2
    issue_batch(r2, 2, r0.idx);                    // + indices of chunk 4
    // one scalar word per chunk, fetched an iteration ahead: the chunk's slot count, or on a layout-3 plan its flags, whose bits
    // 20-23 repeat the row-tile count (compact_runs_kernel) beside the shadow counts
    auto word_of = [&](int k) { return ldc(a.merged ? a.chunk_flags : a.chunk_cnt, c0 + (k < nch ? k : nch - 1)); };
    auto tiles_in = [&](int wd) { return min(kP3SlotTiles, a.merged ? (wd >> 20) & 15 : (wd + 15) >> 4); };
    auto flags_in = [&](int wd) { return a.merged ? wd : 0; };
    // (the consumers fetch their per-chunk words through scalar BUFFER loads with one running offset; the same change here does
    // not compile -- the backend reports an illegal VGPR-to-SGPR copy on the loop-carried word -- and the producers are not the
    // critical path of a chunk: left on ldc)
    int wd_next = word_of(1);
    {
        const int wd0 = word_of(0);
        split_store(r0, 0, tiles_in(wd0), flags_in(wd0));
    }
    store_meta(r0, 0);
    wg_barrier();                                  // chunk 0 (and the accumulator init) visible
    // (batches are issued for chunks past the end too, from clamped addresses: the count of operations in flight stays what
    // the waits assume; two or three redundant batches per tile)
#ifdef RGCN_P3_STAMPS
    unsigned long long sp_wait = 0, sp_issue = 0, sp_split = 0, sp_bar = 0;
#endif
    auto step = [&](int it, P3Rows& tgt, P3Rows& src) {      // tgt = set it % 3, src = set (it + 1) % 3
        P3S(q0);
        p3_wait_batch<10>(src);
        __builtin_amdgcn_sched_barrier(0);
        P3S(q1);
        if (it + 1 < nch) issue_batch(tgt, it + 3, src.idx);
        __builtin_amdgcn_sched_barrier(0);
        P3S(q2);
        const int wd_now = wd_next;
        wd_next = word_of(it + 2);
        if (it + 1 < nch) {
            split_store(src, it + 1, tiles_in(wd_now), flags_in(wd_now));
            store_meta(src, it + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        P3S(q3);
        wg_barrier();
        tile_boundary(it);
        P3S(q4);
        P3A(sp_wait, q0, q1); P3A(sp_issue, q1, q2); P3A(sp_split, q2, q3); P3A(sp_bar, q3, q4);
    };
    for (int it = 0; it < nch; it += 3) {
        step(it, r0, r1);
        if (it + 1 < nch) step(it + 1, r1, r2);
        if (it + 2 < nch) step(it + 2, r2, r0);
    }
    wait_vmcnt<0>();
#ifdef RGCN_P3_STAMPS
    if (g_p3_stamps && lane == 0) {      // every wave: [workgroup][wave 0..11][wait, issue / compute, split / swap, barrier, chunks]
        unsigned long long* o = g_p3_stamps + ((size_t)blockIdx.x * 12 + pw) * 8;
        o[0] = sp_wait; o[1] = sp_issue; o[2] = sp_split; o[3] = sp_bar; o[4] = nch;
    }
#endif
}

// ---- the kernel ---------------------------------------------------------------------------------------------------
// TEAMS: 1 = every consumer wave sees every row tile of a chunk (any plan layout); 2 = two teams of consumer waves, team A on
//        the first ceil(nt / 2) row tiles of a chunk and team B on the others -- LAYOUT-1 plans only (plan.team_placement: the
//        two parts of a chunk hold disjoint destinations, so the teams never touch the same accumulator row).
// NCT:   16-column tiles a consumer wave owns (1: four column owners per team, 2: two).
// Consumer waves: TEAMS * 4 / NCT (4 or 8) beside the 4 producer waves.
template <int TEAMS, int NCT>
struct P3Cfg {
    static constexpr int kConsumers = TEAMS * 4 / NCT;
    static constexpr int kThreads = 64 * (4 + kConsumers);
    static constexpr int kWavesPerSimd = (4 + kConsumers) / 4;
};

template <int TEAMS, int NCT, int ST>
__global__ void __launch_bounds__((P3Cfg<TEAMS, NCT>::kThreads), (P3Cfg<TEAMS, NCT>::kWavesPerSimd)) rgcn_tile3p_kernel(const TileArgs a) {
    constexpr int kP3PlaneBytes = P3Geo<ST>::kPlaneBytes, kP3SlotBytes = P3Geo<ST>::kSlotBytes, kP3SlotTiles = ST;
    constexpr int LDO = kP3LDO;
    constexpr int kThreadsAll = P3Cfg<TEAMS, NCT>::kThreads;
    constexpr int kConsumers = P3Cfg<TEAMS, NCT>::kConsumers;
    constexpr int CG = 4 / NCT;                                    // column groups (consumer waves) per team
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* out_lds = lds;                                          // [tile + 1][LDO]  (row `tile`: dummy)
    char* ring = (char*)(lds + (a.tile + 1) * LDO);                // [2][3][128][128 B]
    float* wring = (float*)(ring + 2 * kP3SlotBytes);              // [2][128] pairs {weight, run metadata}
    int* dring = (int*)(wring + 2 * kP3CH);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // a workgroup walks `tiles_per_wg` consecutive tiles: one chunk sequence for the ring; between two tiles only the
    // accumulator is stored and reset (the producers' two-iteration lead runs across the boundary)
    const int tile0 = blockIdx.x * a.tiles_per_wg;
    const int tile1 = min(tile0 + a.tiles_per_wg, a.n_tiles);
    const int c0 = ldc(a.tile_ptr, tile0);
    const int nch = ldc(a.tile_ptr, tile1) - c0;

    tile_init<LDO>(a, out_lds, tid, kThreadsAll);

    constexpr bool kRoleMap = RGCN_P3_ROLEMAP && kConsumers == 4;
    const bool is_consumer = kRoleMap ? (wave & 2) == 0 : wave >= 4;
    const int role_idx = kRoleMap ? ((wave >> 2) * 2 + (wave & 1)) : (wave >= 4 ? wave - 4 : wave);       // cw or pw
    if (RGCN_P3_PRIO_PROD != 0 || RGCN_P3_PRIO_CONS != 0) {
        if (is_consumer) __builtin_amdgcn_s_setprio(RGCN_P3_PRIO_CONS);
        else __builtin_amdgcn_s_setprio(RGCN_P3_PRIO_PROD);
    }
    if (is_consumer) {
        // ---- consumers: wave cw = (team, column group cg) owns output columns 16 NCT cg .. + 16 NCT - 1 of its team's rows ----
        const int cw = role_idx;
        const int team = cw / CG, cg = cw % CG;
        const int rowl = lane & 15, kq = lane >> 4;
        const unsigned col4_bytes = (unsigned)(16 * NCT * cg + 4 * kq) * 4u;     // Y^T layout: four consecutive columns of row rowl
        const unsigned col1_bytes = (unsigned)(16 * NCT * cg + rowl) * 4u;       // Y layout: column rowl of rows 4 kq + i
        // W planes of this wave's column tiles: packed3[((((rel * 2 + c) * 3 + pl) * 2 + ct) * 2 + s) * 64 + lane], column
        // 32 c + 16 ct + (lane & 15): fragment (ct, s) of a plane sits 2048 ct + 1024 s bytes behind the plane's first one
        const int wc = NCT == 1 ? (cg >> 1) : cg, wct0 = NCT == 1 ? (cg & 1) : 0;
        const uint4* wp4 = (const uint4*)a.wp + (size_t)(wc * 3 * 2 * 2 + wct0 * 2) * 64 + lane;
        auto wptr = [&](int rel, int pl) { return (const f32x4*)(wp4 + ((size_t)rel * kP3FragsPerRel + pl * 4) * 64); };
        f32x4 wcur[NCT][3][2], wnext[NCT][3][2];
        int rel_cur = ldc(a.chunk_rel, c0);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                wcur[ct][pl][0] = wptr(rel_cur, pl)[128 * ct];
                wcur[ct][pl][1] = wptr(rel_cur, pl)[128 * ct + 64];
            }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): retire these loads in the compiler's scoreboard
        // Per-chunk metadata, one iteration ahead, through scalar BUFFER loads that share one byte offset (rgcn_common.h
        // sbuf_load): iteration `it` fetches chunk it + 1's slot count and flags and chunk it + 3's relation -- three loads and one
        // s_add instead of three 64-bit address computations and two guards (a read past the plan's last chunk returns 0, one past
        // the workgroup's last chunk a neighbour's word: neither is used).  Round 4, VERDICT r3 item 2: the scalar instruction
        // stream of a chunk iteration is serial code between the barrier and the first operand read.
        const i32x4 rs_cnt = make_srsrc(a.chunk_cnt + c0 + 1, 4L * (a.n_chunks - c0 - 1));
        const i32x4 rs_flg = make_srsrc(a.chunk_flags + c0 + 1, 4L * (a.n_chunks - c0 - 1));
        const i32x4 rs_rel = make_srsrc(a.chunk_rel + c0 + 3, 4L * (a.n_chunks - c0 - 3));
        unsigned moff = 0;
        int ld_cnt = ldc(a.chunk_cnt, c0);
        int ld_flg = ldc(a.chunk_flags, c0);
        int rel_n1 = nch > 1 ? ldc(a.chunk_rel, c0 + 1) : rel_cur;
        int ld_rel = nch > 2 ? ldc(a.chunk_rel, c0 + 2) : rel_n1;
        auto prefetch_rel = [&](int rel) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                prefetch16<0>(wnext[0][pl][0], wptr(rel, pl));
                prefetch16<1024>(wnext[0][pl][1], wptr(rel, pl));
                if constexpr (NCT == 2) {
                    prefetch16<2048>(wnext[1][pl][0], wptr(rel, pl));
                    prefetch16<3072>(wnext[1][pl][1], wptr(rel, pl));
                }
            }
        };
        bool pending = rel_n1 != rel_cur;
        if (pending) prefetch_rel(rel_n1);
        int tile_cur = tile0;
        int tend = ldc(a.tile_ptr, tile0 + 1) - c0;     // first chunk (relative) of the next tile
        wg_barrier();
#ifdef RGCN_P3_STAMPS
        unsigned long long sc_meta = 0, sc_comp = 0, sc_swap = 0, sc_bar = 0;
#endif
        for (int it = 0; it < nch; ++it) {
            P3S(t0s);
            const int chunk = c0 + it;
            const int buf = it & 1;
            const int cnt = ld_cnt;                   // (fetched an iteration ago, waited for behind that iteration's barrier)
            const int flags_all = ld_flg;
            const int rel_next = rel_n1;
            const int rel_next2 = ld_rel;
            rel_n1 = ld_rel;
            sbuf_load(ld_cnt, rs_cnt, moff);
            sbuf_load(ld_flg, rs_flg, moff);
            sbuf_load(ld_rel, rs_rel, moff);
            moff += 4;
            (void)chunk;
            const bool swap_b = pending;
            const int nrt = min(kP3SlotTiles, (cnt + 15) >> 4);
            // this wave's row tiles of the chunk: [t0, t0 + n)
            int t0 = 0, n = nrt;
            bool serial = false;                // a chunk whose parts share a destination: team A takes all of it, tile by tile
            if constexpr (TEAMS == 2) {
                const int na = (nrt + 1) >> 1;
                serial = (flags_all & 256) != 0;
                if (serial) {
                    n = team == 0 ? nrt : 0;
                } else {
                    t0 = team == 0 ? 0 : na;
                    n = team == 0 ? na : nrt - na;
                }
            }
            const int flags = (flags_all >> t0) & ((1 << n) - 1);
#ifdef RGCN_P3_STAMPS
            asm volatile("" ::"s"(cnt), "s"(rel_next), "s"(flags));
#endif
            P3S(t1s);
            // this lane's operand addresses of its first row tile: plane pl at + pl * 16 KiB, row tile t at + t * 2 KiB (immediates)
            const char* xrow[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
                xrow[s] = ring + buf * kP3SlotBytes + t0 * 2048 + rowl * 128 + (((4 * s + kq) ^ ((rowl >> 1) & 7)) << 4);
            const int2* mb = (const int2*)wring + buf * kP3CH + t0 * 16;          // {weight bits, run metadata} per slot
            struct Ops {
                bf16x8 pl[3][2];    // [plane][k-step]: 8 bf16 of row rowl, k = 32 s + 8 kq + (0..7)
                float w1;
                int d1;
            };
            auto load_ops = [&](Ops& o, int t) {
                const int2 wd = mb[t * 16 + rowl];
                o.w1 = __int_as_float(wd.x);
                o.d1 = wd.y;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        if (RGCN_P3_ABL & 16) {      // timing only: no operand reads
                            o.pl[pl][s] = __builtin_bit_cast(bf16x8, f32x4{o.w1, 1.f, 2.f, 3.f});
                            continue;
                        }
                        o.pl[pl][s] = *(const bf16x8*)(xrow[s] + pl * kP3PlaneBytes + t * 2048);
                    }
            };
            auto acc_ptr = [&](int d, unsigned col_bytes) -> float* {
                return (float*)((char*)out_lds + (__umul24((unsigned)d, (unsigned)(LDO * 4)) + col_bytes));
            };
            constexpr int px[6] = {0, 0, 1, 0, 2, 1}, pwl[6] = {0, 1, 0, 2, 0, 1};      // x plane, W plane: hh hm mh hl lh mm
            // six products of one half (k-step s) of a row tile and column tile ct, Y^T orientation: a lane ends with 4
            // consecutive columns of a row
            auto mfma_half = [&](const Ops& o, f32x4 y, int s, int ct) {
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    if (RGCN_P3_ABL & 2) {
                        y += __builtin_bit_cast(f32x4, o.pl[px[q]][s]);
                        continue;
                    }
                    y = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wcur[ct][pwl[q]][s]), o.pl[px[q]][s], y, 0, 0, 0);
                }
                return y;
            };
            // ---- row tiles without repeated destinations: straight-line, operands LA tiles ahead -----------------------------
            // (Reading tile t + 1's accumulator row BEFORE tile t is stored -- with the lane's own previous value forwarded
            // where the two addresses match -- was tried to break the store -> read -> FMA -> store chain: 10.3 ms against
            // 9.7, and not sufficient as it stood: a run of equal destinations as long as the group's tile count wraps and puts
            // the same destination at place p + 1 of tile t and place p of tile t + 1.)
            auto consume = [&](auto nrt_c) {
                constexpr int NRT = decltype(nrt_c)::value;
                constexpr int LA = RGCN_P3_LA;
                Ops o[NRT];
                f32x4 y[NRT][NCT], old[NRT][NCT];
                float* dst[NRT];
#pragma unroll
                for (int t = 0; t < LA; ++t)
                    if (t < NRT) load_ops(o[t], t);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int step = 0; step <= NRT; ++step) {
                    if (step < NRT) {
                        if (step + LA < NRT) load_ops(o[step + LA], step + LA);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) y[step][ct] = mfma_half(o[step], f32x4{0.f, 0.f, 0.f, 0.f}, 0, ct);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // one group of vector instructions per tile: store of tile step - 1, then address + accumulator read of
                    // tile step (in this order: consecutive tiles may scatter into the same accumulator row)
                    if (RGCN_P3_ABL & 4) {
                        if (step >= 1) asm volatile("" ::"v"(y[step - 1][0]), "v"(y[step - 1][NCT - 1]), "v"(o[step - 1].w1), "v"(o[step - 1].d1));
                        if (step < NRT)
#pragma unroll
                            for (int ct = 0; ct < NCT; ++ct) old[step][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                    } else if (step >= 1) {
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
                            *(f32x4*)(dst[step - 1] + 16 * ct) = y[step - 1][ct] * o[step - 1].w1 + old[step - 1][ct];
                    }
                    if (step == NRT - 1 && (RGCN_P3_ABL & 128)) {      // timing only: the LAST row tile's accumulator row is not read
                        dst[step] = acc_ptr(o[step].d1, col4_bytes);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) old[step][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                    } else if (step < NRT && !(RGCN_P3_ABL & 4)) {
                        // (RGCN_P3_ABL & 64, timing only, wrong results: the 16 lanes of a phase address rows that differ mod
                        // 16 -- what a conflict-free accumulator order could buy)
                        dst[step] = acc_ptr((RGCN_P3_ABL & 64) ? ((o[step].d1 & 0xFFFFF0) | rowl) : o[step].d1, col4_bytes);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) old[step][ct] = *(const f32x4*)(dst[step] + 16 * ct);
                    }
                    if (step < NRT) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) y[step][ct] = mfma_half(o[step], y[step][ct], 1, ct);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            // ---- tile by tile: chunks with a repeated destination in some row tile (a flagged tile takes the Y orientation
            // and the run-sum product Z = P Y + old in exact fp32, rgcn_tile_kernel stage B), and serial chunks
            auto process_slow = [&](int t, bool dup) {
                Ops o;
                load_ops(o, t);
                if (!dup) {
                    float* d = acc_ptr(o.d1, col4_bytes);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        const f32x4 oldv = *(const f32x4*)(d + 16 * ct);
                        f32x4 yv = mfma_half(o, f32x4{0.f, 0.f, 0.f, 0.f}, 0, ct);
                        yv = mfma_half(o, yv, 1, ct);
                        *(f32x4*)(d + 16 * ct) = yv * o.w1 + oldv;
                    }
                    return;
                }
                f32x4 w4;
                i32x4 d4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int2 wd = mb[t * 16 + 4 * kq + i];
                    w4[i] = __int_as_float(wd.x);
                    d4[i] = wd.y;
                }
                float pm[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) pm[i] = ((unsigned)d4[i] >> 24) == (unsigned)rowl ? w4[i] : 0.f;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    f32x4 yv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int q = 0; q < 6; ++q)
                            yv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o.pl[px[q]][s], __builtin_bit_cast(bf16x8, wcur[ct][pwl[q]][s]), yv, 0, 0, 0);
                    float* d[4];
                    f32x4 oldv;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        d[i] = acc_ptr(d4[i], col1_bytes) + 16 * ct;
                        oldv[i] = *d[i];
                    }
                    f32x4 z1 = {0.f, 0.f, 0.f, 0.f};
                    f32x4 z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pm[0], yv[0], oldv, 0, 0, 0);
                    z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pm[1], yv[1], z1, 0, 0, 0);
                    z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pm[2], yv[2], z0, 0, 0, 0);
                    z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pm[3], yv[3], z1, 0, 0, 0);
                    const f32x4 v = z0 + z1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) *d[i] = v[i];
                }
            };
            using std::integral_constant;
            if (flags == 0 && !serial) {
                switch (n) {
                    case 1: consume(integral_constant<int, 1>{}); break;
                    case 2: consume(integral_constant<int, 2>{}); break;
                    case 3: consume(integral_constant<int, 3>{}); break;
                    case 4: consume(integral_constant<int, 4>{}); break;
                    case 5: if constexpr (TEAMS == 1) consume(integral_constant<int, 5>{}); break;
                    case 6: if constexpr (TEAMS == 1) consume(integral_constant<int, 6>{}); break;
                    case 7: if constexpr (TEAMS == 1) consume(integral_constant<int, 7>{}); break;
                    case 8: if constexpr (TEAMS == 1 && ST >= 8) consume(integral_constant<int, 8>{}); break;
                    default: break;
                }
            } else {
                for (int t = 0; t < n; ++t) process_slow(t, (flags >> t) & 1);
            }
            P3S(t2s);
            // (Round 4, tried: the chunk loop unrolled by two with two register sets that trade places, so that these 24 v_mov
            // go.  With the prefetch where it is, the register allocator copies the set right behind the loads -- before they
            // have landed -- wherever a live range is split at the back edge: wrong results.  With the prefetch at the top of
            // the iteration and the wait down here, correct and copy-free, but 7.61 -> 7.75 ms: six loads and their addresses in
            // front of the first operand read cost more than the copies; behind the first MFMA block: 8.37 ms.
            // profiles/r04z_weight_sets_timing.txt)
            if (swap_b) {
                wait_vmcnt<0>();                     // the asm prefetch (this wave's only vector-memory traffic)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        wcur[ct][pl][0] = wnext[ct][pl][0];
                        wcur[ct][pl][1] = wnext[ct][pl][1];
                    }
            }
            rel_cur = rel_next;
            pending = it + 2 < nch && rel_next2 != rel_next;
            if (pending) prefetch_rel(rel_next2);
            P3S(t3s);
            // The barrier that hands the ring slot back.  What it must order is this wave's READS of the slot -- all consumed by
            // MFMAs by now -- not its accumulator stores: those rows are read again only by this wave (LDS operations of a wave
            // execute in order) until the tile closes.  So no lgkmcnt(0) in front of it except where a tile closes and the other
            // consumer waves read these columns (RGCN_P3_DRAIN=1: always drain, the round-2 behaviour)
            static_assert(RGCN_P3_DRAIN == 1, "the scalar buffer loads of the next chunk's words share lgkmcnt with the LDS queue");
            if (RGCN_P3_DRAIN || (it + 1 == tend) || it + 1 == nch) wg_barrier();
            else asm volatile("s_barrier" ::: "memory");
            // the words of the next chunk, fetched at the top of this iteration: wg_barrier's lgkmcnt(0) has retired them; the
            // tie keeps every use (and every copy the register allocator makes) behind this point
            sbuf_wait(ld_cnt, ld_flg, ld_rel);
            if (it + 1 == tend && it + 1 < nch) {
                // this chunk closed a tile: the consumer threads store it and reset the accumulator; the producers wait at
                // the same extra barrier with the next tile's first chunks already in LDS / in flight
#ifndef RGCN_P3_ABL_NOEPI     // timing-only build: what a tile boundary costs
                // (round 4, profiles/r04w_*: the producer waves storing too, or no vmcnt(0) behind the stores: no change -- what is
                // left of a tile boundary, ~0.26 ms per launch, is the LDS pass, the 2.56 GB of output and two barriers)
                tile_epilogue<LDO, true>(a, out_lds, tile_cur, cw * 64 + lane, 64 * kConsumers);
#endif
                ++tile_cur;
                tend = ldc(a.tile_ptr, tile_cur + 1) - c0;
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): retire the epilogue's memory operations here, once per tile
                wg_barrier();
            }
            P3S(t4s);
            P3A(sc_meta, t0s, t1s); P3A(sc_comp, t1s, t2s); P3A(sc_swap, t2s, t3s); P3A(sc_bar, t3s, t4s);
        }
#ifdef RGCN_P3_STAMPS
        if (g_p3_stamps && lane == 0) {
            unsigned long long* o = g_p3_stamps + ((size_t)blockIdx.x * 12 + 4 + cw) * 8;
            o[0] = sc_meta; o[1] = sc_comp; o[2] = sc_swap; o[3] = sc_bar; o[4] = nch;
        }
#endif
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing of the consumers is pending when the producer code follows
    }
    if (!is_consumer) p3_producer_loop<ST>(a, ring, wring, dring, c0, nch, lane, role_idx, tile0);
    tile_epilogue<LDO, false>(a, out_lds, tile1 - 1, tid, kThreadsAll);
}

// bytes of dynamic LDS at tile size `tile` with ring slots of `st` row tiles
static size_t p3_lds_bytes(int tile, int st) {
    return sizeof(float) * (size_t)(tile + 1) * kP3LDO + 2 * (size_t)(3 * 16 * st * 128) + 2 * kP3CH * 8;
}

// Launch (called by run_tile in rgcn_tile_fp32.hip when RGCN_FLAG_SPLIT_PRODUCERS is set and the shapes fit): `a.wp` points at
// the bf16 planes of the packed weights.  Layout-1 plans run the two-team form, layout-0 plans the one-team form.
// Returns RGCN_ERR_LDS / RGCN_ERR_PLAN when it does not apply.
// The two-team forms are EXPERIMENT builds (tools/debug/build_variant_p3.sh -DRGCN_P3_TEAMS=2 [-DRGCN_P3_NCT=2]): parity-green
// (tests/test_gpu_parity.py::test_split_producers_kernel_matches_oracle runs whatever form the library was built with on
// layout-1 plans), not faster -- round 3, forward launch at the headline config on a graph without repeated (dst, relation)
// pairs, A/B on one box: one team 9.09 ms, two teams x four 16-column owners (8 consumer waves) 9.16, two teams x two
// 32-column owners (half the operand reads) 9.47; stamps: a consumer's time per row tile grows from 468 to 650-820 cycles
// when eight waves share the LDS (profiles/r03a_*).  The product library instantiates the one-team kernel only; any
// placement is valid for it.
#ifndef RGCN_P3_NCT          // 16-column tiles per consumer wave of the two-team form (1: 8 consumer waves, 2: 4)
#define RGCN_P3_NCT 1
#endif
#ifndef RGCN_P3_TEAMS        // 2: layout-1 plans run the two-team kernel
#define RGCN_P3_TEAMS 1
#endif
template <int TEAMS, int NCT, int ST>
static int launch_tile3p_as(const TileArgs& b, int nwg, size_t lds, hipStream_t stream) {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute((const void*)rgcn_tile3p_kernel<TEAMS, NCT, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return (int)e;
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((rgcn_tile3p_kernel<TEAMS, NCT, ST>), dim3(nwg), dim3(P3Cfg<TEAMS, NCT>::kThreads), lds, stream, b);
    return (int)hipGetLastError();
}

int launch_tile3p(const TileArgs& a, int n_tiles, int layout, int chunk_rows, void* stream) {
    if (a.x_bytes == 0) return RGCN_ERR_PLAN;          // buffer-descriptor addressing only
    // plans whose chunks hold at most 112 rows run on 42 KiB ring slots (tiles up to 272), any other 128-slot plan on 48 KiB ones
    const int st = chunk_rows > 0 && chunk_rows <= 112 ? 7 : 8;
    const size_t lds = p3_lds_bytes(a.tile, st);
    if (lds > (size_t)kLdsBytes) return RGCN_ERR_LDS;
    // walking several tiles pays once every CU gets many workgroups either way (start-up: two trips to memory per workgroup);
    // on graphs of a few launch rounds one tile per workgroup balances better (300k nodes / 1,340 tiles: 4.28 against 4.44 ms)
    TileArgs b = a;
    if (n_tiles < 16 * 256) b.tiles_per_wg = 1;
    const int nwg = (n_tiles + b.tiles_per_wg - 1) / b.tiles_per_wg;
    if constexpr (RGCN_P3_TEAMS == 2)
        if (layout == 1) return launch_tile3p_as<2, RGCN_P3_NCT, 8>(b, nwg, lds, (hipStream_t)stream);
    if (st == 7) return launch_tile3p_as<1, 1, 7>(b, nwg, lds, (hipStream_t)stream);
    return launch_tile3p_as<1, 1, 8>(b, nwg, lds, (hipStream_t)stream);
}

#ifdef RGCN_P3_STAMPS
extern "C" int rgcn_debug_set_p3_stamps(unsigned long long* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_p3_stamps), &p, sizeof(p));
}
#endif

}  // namespace rgcn
