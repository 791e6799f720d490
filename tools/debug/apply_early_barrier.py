"""Experiment patch (NOT applied in the product): early-barrier consumers in rgcn_tile_kernel.  Usage: python tools/debug/apply_early_barrier.py
Rewrites csrc/rgcn_kernels.hip in place (git checkout to undo).  Measured 3 % slower than the end-of-chunk barrier."""
import os
p = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scaling_rgcn_training_amd/csrc/rgcn_kernels.hip")
s = open(p).read()


def rep(old, new):
    global s
    assert old in s, old[:80]
    s = s.replace(old, new)


rep('''// tile-major dW: cut a unit's tail at the 4-row k-step instead of the 16-row group''', '''#ifndef RGCN_PREREAD
#define RGCN_PREREAD 1
#endif
// tile-major dW: cut a unit's tail at the 4-row k-step instead of the 16-row group''')
rep("                                                   int lane, int wave, int tile0) {", "                                                   int lane, int wave, int tile0, bool early) {")
rep('''            tend = ldc(a.tile_ptr, tile_cur + 1) - c0;
            wg_barrier();
        }
    };''', '''            tend = ldc(a.tile_ptr, tile_cur + 1) - c0;
            if (early) wg_barrier();
            wg_barrier();
        }
    };''')
rep('''        }
        }
        wait_vmcnt<0>();
#ifdef RGCN_STAMPS
        if (g_stamps && lane == 0) {
            unsigned long long* o = g_stamps + (size_t)blockIdx.x * 32;
            if (pw == 0) { o[4] = sp_issue; o[5] = sp_wait; o[6] = sp_bar; }''', '''        }
        }
        wait_vmcnt<0>();
        if (early) wg_barrier();
#ifdef RGCN_STAMPS
        if (g_stamps && lane == 0) {
            unsigned long long* o = g_stamps + (size_t)blockIdx.x * 32;
            if (pw == 0) { o[4] = sp_issue; o[5] = sp_wait; o[6] = sp_bar; }''')
rep('''    int* dring = (int*)(wring + NBUF * CH);       // [NBUF][CH]

    const int tid = threadIdx.x;''', '''    int* dring = (int*)(wring + NBUF * CH);       // [NBUF][CH]
    constexpr bool PRE = RGCN_PREREAD && NBUF == 2 && CH == 128 && NP >= 64 && kTileProducers == 4;

    const int tid = threadIdx.x;''')
rep("tile_producer_loop<KP, NBUF, BUF, CH>(a, ring, wring, dring, c0, nch, lane, wave, tile0);", "tile_producer_loop<KP, NBUF, BUF, CH>(a, ring, wring, dring, c0, nch, lane, wave, tile0, PRE);")
rep("tile_producer_loop<KP, NBUF, BUF, CH>(a, ring, wring, dring, c0, nch, lane, wave, tile);", "tile_producer_loop<KP, NBUF, BUF, CH>(a, ring, wring, dring, c0, nch, lane, wave, tile, false);")
rep('''        bool pending = false;       // bnext is receiving the fragments of chunk it + 1
        if (kAsmPrefetch && active && rel_n1 != rel_cur && !(RGCN_DBG(a) & 4)) {''', '''        bool pending = false;       // bnext is receiving the fragments of chunk it + 1
        constexpr bool kPrefetchAtTop = PRE && RGCN_PREREAD != 2;
        if (kAsmPrefetch && !kPrefetchAtTop && active && rel_n1 != rel_cur && !(RGCN_DBG(a) & 4)) {''')
rep('''        int tend = ldc(a.tile_ptr, tile0 + 1) - c0;     // first chunk (relative) of the next tile
        wg_barrier();
#ifdef RGCN_STAMPS
        unsigned long long st_scal = 0, st_comp = 0, st_bwait = 0, st_bar = 0;''', '''        int tend = ldc(a.tile_ptr, tile0 + 1) - c0;     // first chunk (relative) of the next tile
        wg_barrier();
        const float* arow_base[KT];
        f32x4 av_pre[KT];
        float w1_pre = 0.f;
        int d1_pre = 0;
        if constexpr (PRE) {
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const int pos = (4 * j + kq) ^ swizzle<kRowRead, LPR>(rowl);
                arow_base[j] = ring + rowl * KP + pos * 4;
                av_pre[j] = *(const f32x4*)arow_base[j];
            }
            w1_pre = wring[rowl];
            d1_pre = dring[rowl];
        }
#ifdef RGCN_STAMPS
        unsigned long long st_scal = 0, st_comp = 0, st_bwait = 0, st_bar = 0;''')
rep('''            const bool swap_b = kAsmPrefetch ? pending : (active && rel_next != rel_cur && !(RGCN_DBG(a) & 4));''', '''            if constexpr (kAsmPrefetch && kPrefetchAtTop) {
                pending = active && it + 1 < nch && rel_next != rel_cur && !(RGCN_DBG(a) & 4);
                if (pending) prefetch_rel(rel_next);
            }
            const bool swap_b = kAsmPrefetch ? pending : (active && rel_next != rel_cur && !(RGCN_DBG(a) & 4));''')
rep('''            const float* arow[KT];
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const int pos = (4 * j + kq) ^ swizzle<kRowRead, LPR>(rowl);
                arow[j] = hb + rowl * KP + pos * 4;
            }''', '''            const float* arow[KT];
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                if constexpr (PRE) {
                    arow[j] = arow_base[j] + (buf * CH + 64 * part) * KP;
                } else {
                    const int pos = (4 * j + kq) ^ swizzle<kRowRead, LPR>(rowl);
                    arow[j] = hb + rowl * KP + pos * 4;
                }
            }
            auto barrier_and_preread = [&]() {
                if constexpr (PRE) {
                    wg_barrier();
#pragma unroll
                    for (int j = 0; j < KT; ++j) av_pre[j] = *(const f32x4*)(arow_base[j] + ((buf ^ 1) * CH) * KP);
                    w1_pre = wring[(buf ^ 1) * CH + rowl];
                    d1_pre = dring[(buf ^ 1) * CH + rowl];
                }
            };''')
rep('''                if constexpr (decltype(tr_c)::value) {
                    o.w1 = wrow[rt * 16];
                    o.d1 = drow[rt * 16];
                } else {''', '''                if constexpr (decltype(tr_c)::value) {
                    if (PRE && part == 0 && rt == 0) {
                        o.w1 = w1_pre;
                        o.d1 = d1_pre;
                    } else {
                        o.w1 = wrow[rt * 16];
                        o.d1 = drow[rt * 16];
                    }
                } else {''')
rep('''#pragma unroll
                for (int j = 0; j < KT; ++j) o.av[j] = *(const f32x4*)(arow[j] + rt * 16 * KP);
            };''', '''                if (PRE && part == 0 && rt == 0) {
#pragma unroll
                    for (int j = 0; j < KT; ++j) o.av[j] = av_pre[j];
                    return;
                }
#pragma unroll
                for (int j = 0; j < KT; ++j) o.av[j] = *(const f32x4*)(arow[j] + rt * 16 * KP);
            };''')
rep('''                        if (step > NRT) continue;
                        // first half of this tile's MFMAs''', '''                        if (step > NRT) continue;
                        if constexpr (PRE) {
                            if (step == NRT - 1 && whole) barrier_and_preread();
                        }
                        // first half of this tile's MFMAs''')
rep('''                    default: break;
                }
            }
            }   // part''', '''                    default: break;
                }
            }
            if constexpr (PRE) {
                const bool in_block = whole && nrt >= 1;
                const bool last_part = whole || part == CH / 64 - 1;
                if (last_part && !in_block) barrier_and_preread();
            }
            }   // part''')
rep('''            if constexpr (kAsmPrefetch) {      // fragments of chunk it + 2, issued while the memory queue is idle''', '''            if constexpr (kAsmPrefetch && !kPrefetchAtTop) {      // fragments of chunk it + 2, issued while the memory queue is idle''')
rep('''            STAMP(t3);
            wg_barrier();
            STAMP(t4);
            if (it + 1 == tend && it + 1 < nch) {''', '''            STAMP(t3);
            if constexpr (!PRE) wg_barrier();
            STAMP(t4);
            if (it + 1 == tend && it + 1 < nch) {
                if constexpr (PRE) wg_barrier();''')
rep('''        // tell the waitcnt pass that no consumer load is pending when the producer code (next in program
        // order) reuses these registers; otherwise it waits vmcnt(0) between the prologue DMAs
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
    if (wave < kTileProducers) tile_producer_loop<KP, NBUF, BUF, CH>(a, ring, wring, dring, c0, nch, lane, wave, tile0, PRE);''', '''        if constexpr (PRE) wg_barrier();
        // tell the waitcnt pass that no consumer load is pending when the producer code (next in program
        // order) reuses these registers; otherwise it waits vmcnt(0) between the prologue DMAs
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
    if (wave < kTileProducers) tile_producer_loop<KP, NBUF, BUF, CH>(a, ring, wring, dring, c0, nch, lane, wave, tile0, PRE);''')
open(p, "w").write(s)
print("patched", p)
