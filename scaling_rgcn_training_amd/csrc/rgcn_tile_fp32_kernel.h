// rgcn_tile_fp32_kernel.h -- rgcn_tile_kernel (forward / dX of the R-GCN layer in exact fp32, v_mfma_f32_16x16x4_f32) and its launchers
// as templates over the padded widths.  Instantiated by rgcn_tile_fp32_narrow.hip (gathered width 16 / 32) and
// rgcn_tile_fp32_wide.hip (64 / 128): two translation units so that the library builds in under a minute.
#pragma once
#include "rgcn_kernels_shared.h"

namespace rgcn {

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
            // ---- layout-3 plans (rgcn_plan.hip compact_runs_kernel; 64-wide rows, 128-slot chunks): aggregate, then transform.
            // The rows of a (destination, relation) run sit on ONE head slot (row tiles 0 / 1 of the chunk) and their second / third
            // rows on the SHADOW row tiles 7, 6 / 5 at the head's place.  The head rows travel by LDS-DMA like every row; the wave
            // (third rows: the row tile right below the second rows', 6 or 5.)  The wave that owns row tiles 0 and 1 (wave 0) loads the shadow rows into REGISTERS with the same lane geometry (lane = row
            // 4 i + rsub of its tile, 16-byte position p: exactly where the DMA puts the head's piece), waits for both, and adds
            // them to the head rows in LDS before the barrier hands the slot to the consumers -- who never see a shadow tile
            // (chunk_cnt counts the head row tiles).  The other waves' shadow slots become padding (no gather traffic).
            // chunk_flags: bits 16-17 row tiles of second rows, 18 a tile of third rows, 19 a shadow row times (its weight / its
            // head's) first -- the float in the shadow slot's slot_acc --, 20-23 the head row tiles.
            constexpr bool kCanMerge = KP == 64 && CH == 128 && BUF;
            const bool merged = kCanMerge && a.merged != 0;
            const bool shadow_wave = merged && row0 == 0;
            auto load_word = [&](int k) { return merged ? ldc(a.chunk_flags, c0 + (k < nch ? k : nch - 1)) : 0; };
            auto load_shadow = [&](const int* arr, int k) {      // lanes 16 .. 63 <-> slots 80 .. 127 (row tiles 5, 6, 7)
                const int kk = k < nch ? k : nch - 1;
                return shadow_wave ? arr[(size_t)(c0 + kk) * CH + 64 + lane] : 0;
            };
            f32x4 sh[12];      // shadow rows in flight: [0..3] row tile 7, [4..7] row tile 6 (second rows), [8..11] the third rows' tile
            auto issue_part = [&](int k, int idxv, int wd, int sidx) {
                const int chunk = c0 + k, buf = k % NBUF;
                if (merged) {      // slots of the shadow row tiles (and beyond the heads): padding for the DMA
                    const int heads = ((wd >> 20) & 15) * 16;
                    idxv = 64 * half + lane < heads ? idxv : a.n_rows;
                }
                gather.template issue_part<NOPS_PART>(a.x, a.x_bytes, a.n_rows, a.ldx, idxv,
                                                      ring + (buf * CH + row0) * KP, op0);
                if (meta) {
                    dma4(a.slot_w + (size_t)chunk * CH + 64 * half + lane, wring + buf * CH + 64 * half);
                    dma4(a.slot_acc + (size_t)chunk * CH + 64 * half + lane, dring + buf * CH + 64 * half);
                }
                if constexpr (kCanMerge) {
                    if (shadow_wave && (wd & (7 << 16))) {
                        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.x, a.x_bytes);
                        const int ns1 = (wd >> 16) & 3;
                        auto rows4 = [&](f32x4* dst, int lane0) {      // four DMA-shaped loads: rows lane0 + 4 i + rsub of the half
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int ix = __builtin_amdgcn_ds_bpermute((lane0 + 4 * i) * 4 + gather.perm_addr, sidx);
                                dst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    rsrc, (int)(__umul24((unsigned)ix, gather.rowb[i % 4]) + gather.coff[i % 4]), 0, 0));
                            }
                        };
                        rows4(sh, 48);
                        if (ns1 >= 2) rows4(sh + 4, 32);
                        if ((wd >> 18) & 1) rows4(sh + 8, ns1 >= 2 ? 16 : 32);      // third rows: row tile 5 behind two tiles of second rows, else 6
                    }
                }
            };
            // after the wave's DMAs and shadow loads have landed: head row += shadow row (x ratio), in LDS, this wave's own rows
            auto merge_part = [&](int k, int wd, int srat) {
                if constexpr (kCanMerge) {
                    if (!(shadow_wave && (wd & (7 << 16)))) return;
                    f32x4* slot = (f32x4*)(ring + ((k % NBUF) * CH) * KP);
                    const int ns1 = (wd >> 16) & 3;
                    const bool third = ((wd >> 18) & 1) != 0, pre = ((wd >> 19) & 1) != 0;
                    auto ratio = [&](int lane0, int i) {
                        return pre ? __int_as_float(__builtin_amdgcn_ds_bpermute((lane0 + 4 * i) * 4 + gather.perm_addr, srat)) : 1.f;
                    };
#pragma unroll
                    for (int i = 0; i < 4; ++i) {      // heads of row tile 0: rows 4 i + rsub = f32x4 index (4 i) * 16 + lane
                        f32x4 h = slot[i * 64 + lane];
                        h += sh[i] * ratio(48, i);
                        if (third) h += sh[8 + i] * ratio(ns1 >= 2 ? 16 : 32, i);
                        slot[i * 64 + lane] = h;
                    }
                    if (ns1 >= 2) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {  // heads of row tile 1
                            f32x4 h = slot[(4 + i) * 64 + lane];
                            h += sh[4 + i] * ratio(32, i);
                            slot[(4 + i) * 64 + lane] = h;
                        }
                    }
                }
            };
            int idx_cur = load_idx(0);
            int wd_cur = load_word(0), sidx_cur = load_shadow(a.slot_src, 0), srat_cur = load_shadow(a.slot_acc, 0);
            issue_part(0, idx_cur, wd_cur, sidx_cur);     // (its index vector is waited for here, once per tile)
            idx_cur = load_idx(1);
            int wd_nxt = load_word(1), srat_nxt = load_shadow(a.slot_acc, 1);
            sidx_cur = load_shadow(a.slot_src, 1);
            wait_vmcnt<0>();                              // chunk 0 landed (and the indices of chunk 1)
            merge_part(0, wd_cur, srat_cur);
            wd_cur = wd_nxt;
            srat_cur = srat_nxt;
            wg_barrier();                                 // chunk 0 (and the accumulator init) visible
            for (int it = 0; it < nch; ++it) {
                STAMP(p0);
                int idx_next = idx_cur, sidx_next = sidx_cur;
                if (it + 1 < nch) {
                    issue_part(it + 1, idx_cur, wd_cur, sidx_cur);
                    idx_next = load_idx(it + 2);          // youngest operation: lands with the rows
                    sidx_next = load_shadow(a.slot_src, it + 2);
                    srat_nxt = load_shadow(a.slot_acc, it + 2);
                    wd_nxt = load_word(it + 2);
                }
                STAMP(p1);
                wait_vmcnt<0>();                          // chunk it + 1 landed
                if (it + 1 < nch) merge_part(it + 1, wd_cur, srat_cur);
                idx_cur = idx_next;
                sidx_cur = sidx_next;
                wd_cur = wd_nxt;
                srat_cur = srat_nxt;
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

    tile_init<LDO>(a, out_lds, tid, kTileThreads);

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
        // per-chunk words one iteration ahead through scalar BUFFER loads with one running offset (rgcn_common.h sbuf_load;
        // round 4, as in rgcn_tile3p_kernel): chunk it + 1's slot count and flags, chunk it + 3's relation
        const i32x4 rs_cnt = make_srsrc(a.chunk_cnt + c0 + 1, 4L * (a.n_chunks - c0 - 1));
        const i32x4 rs_flg = make_srsrc(a.chunk_flags + c0 + 1, 4L * (a.n_chunks - c0 - 1));
        const i32x4 rs_rel = make_srsrc(a.chunk_rel + c0 + 3, 4L * (a.n_chunks - c0 - 3));
        unsigned moff = 0;
        int ld_cnt = ldc(a.chunk_cnt, c0);
        int ld_flg = ldc(a.chunk_flags, c0);
        // Relation ids one and two chunks ahead.  The NEXT relation's weight fragments are prefetched into `bnext` at the
        // END of an iteration, just before the barrier: by then the producers have issued (and waited for) all their
        // LDS-DMAs, so the CU's vector-memory queue is empty and these few loads issue at once.  Issued at the TOP of
        // an iteration -- right behind the barrier, when the producers flood the queue with the next chunk's 32 gathers --
        // every global_load of a consumer wave took hundreds of cycles to ISSUE (~1,000 cycles per chunk, the "fixed cost
        // that does not scale with the chunk" of the stamp profile; tools/debug/stamps.py).
        constexpr bool kAsmPrefetch = SL * KT <= 4;
        int rel_n1 = nch > 1 ? ldc(a.chunk_rel, c0 + 1) : rel_cur;
        int ld_rel = nch > 2 ? ldc(a.chunk_rel, c0 + 2) : rel_n1;
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
            const int cnt = ld_cnt;                   // (fetched an iteration ago, waited for behind that iteration's barrier)
            const int flags_chunk = ld_flg;
            const int rel_next = rel_n1;
            const int rel_next2 = ld_rel;
            rel_n1 = ld_rel;
            sbuf_load(ld_cnt, rs_cnt, moff);
            sbuf_load(ld_flg, rs_flg, moff);
            sbuf_load(ld_rel, rs_rel, moff);
            moff += 4;
            (void)chunk;
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
            sbuf_wait(ld_cnt, ld_flg, ld_rel);      // the next chunk's words: retired by wg_barrier's lgkmcnt(0); uses stay behind here
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
            // (layout-3 plans: the two-slot ring, whose producers all issue every chunk and wave 0 adds the shadow rows)
            if (bytes(3) <= cap && !a.merged) return launch_tile_nbuf<KP, NP, 3, 128>(a, n_tiles, bytes(3), stream);
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


}  // namespace rgcn
