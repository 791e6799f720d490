// Probe: does v_mfma_f32_16x16x32_bf16 (4 passes, 16 cycles of the matrix pipe) leave the SIMD's vector issue free on gfx950?
// One iteration = 16 MFMAs on four independent accumulators + K vector instructions behind each MFMA (independent chains, so
// neither stream waits for itself); ticks (s_memtime) per iteration for one and for two waves per SIMD, and for the
// vector instructions alone.  If MFMA and VALU overlapped, MFMA + K VALU would cost max(16, 4 + 4 K) ticks per MFMA; if the
// MFMA holds the issue port for its passes, 16 + 4 K.
//   hipcc --offload-arch=gfx950 -O2 mfma_bf16_valu_shadow.hip -o mfma_bf16_valu_shadow && ./mfma_bf16_valu_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VFMA(x, a, b) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b))
#define VCVT(d, a, b) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define VAND(d, a) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(d) : "v"(a))
#define VLSH(d, a) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(d) : "v"(a))
#define VPKADD(d, a, b) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define VPERM(d, a, b, s) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(s))

// KIND 0: v_fma_f32, 1: the cut mix (cvt_pk, shift, and, sub as fma), 2: v_perm_b32.  MF: with MFMAs or the vector instructions alone
template <int K, int KIND, bool MF>
__global__ void __launch_bounds__(512) probe(long long* cyc, int iters, float* sink) {
    const int lane = threadIdx.x & 63;
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + lane); b[i] = (short)0x3f00; }
    float f[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { f[i] = (float)(i + lane); u[i] = 0x3f800000u + i; }
    float fa = 1.0f + lane, fb = 0.5f;
    unsigned sel = 0x07060302u;
    f32x2 p2[4] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MF) MFMA(c[j & 3], a, b);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const int r = (j * K + q) & 7;
                if (KIND == 0) VFMA(f[r], fa, fb);
                if (KIND == 1) {
                    const int ph = (j * K + q) & 3;
                    if (ph == 0) VCVT(u[r], f[r], f[(r + 1) & 7]);
                    if (ph == 1) VLSH(u[r], u[(r + 3) & 7]);
                    if (ph == 2) VAND(u[r], u[(r + 5) & 7]);
                    if (ph == 3) VFMA(f[r], fa, fb);
                }
                if (KIND == 2) VPERM(u[r], u[(r + 3) & 7], u[(r + 5) & 7], sel);
                if (KIND == 3) VCVT(u[r], f[r], f[(r + 1) & 7]);
                if (KIND == 4) VLSH(u[r], u[(r + 3) & 7]);
                if (KIND == 5) VAND(u[r], u[(r + 3) & 7]);
                if (KIND == 7) asm volatile("v_mul_u32_u24 %0, 0x10000, %1" : "=v"(u[r]) : "v"(u[(r + 3) & 7]));
                if (KIND == 8) asm volatile("v_lshl_add_u32 %0, %1, 16, %2" : "=v"(u[r]) : "v"(u[(r + 3) & 7]), "v"(u[(r + 5) & 7]));
                if (KIND == 9) asm volatile("v_alignbit_b32 %0, %1, 0, 16" : "=v"(u[r]) : "v"(u[(r + 3) & 7]));
                if (KIND == 10) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f[r]) : "v"(f[(r + 3) & 7]), "v"(f[(r + 5) & 7]));
                if (KIND == 11) asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(u[r]) : "v"(u[(r + 3) & 7]));
                if (KIND == 12) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(u[r]) : "v"(u[(r + 3) & 7]), "v"(sel), "v"(u[(r + 5) & 7]));
                if (KIND == 13) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(u[r]) : "v"(sel), "v"(u[(r + 3) & 7]), "v"(u[(r + 5) & 7]));
                if (KIND == 14) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(u[r]) : "v"(u[(r + 3) & 7]));
                if (KIND == 15) asm volatile("v_lshlrev_b32_e64 %0, 16, %1" : "=v"(u[r]) : "v"(u[(r + 3) & 7]));
                if (KIND == 16) asm volatile("v_lshlrev_b32 %0, %1, %2" : "=v"(u[r]) : "v"(sel), "v"(u[(r + 3) & 7]));
                if (KIND == 17) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(f[r]) : "v"(f[(r + 3) & 7]), "v"(f[(r + 5) & 7]));
                if (KIND == 6) VPKADD(p2[r & 3], p2[(r + 1) & 3], p2[(r + 2) & 3]);
            }
        }
    }
    const long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    float s = c[0][0] + c[1][0] + c[2][0] + c[3][0];
    for (int i = 0; i < 8; ++i) s += f[i] + (float)u[i] + p2[i & 3][0];
    if (s == 12345.f) sink[0] = s;
}

template <int K, int KIND, bool MF>
static void run(long long* d, float* sink) {
    const int iters = 4000;
    static const char* kinds[] = {"v_fma_f32", "cvt_pk / shift / and / fma", "v_perm_b32", "v_cvt_pk_bf16_f32", "v_lshlrev_b32", "v_and_b32", "v_pk_add_f32", "v_mul_u32_u24", "v_lshl_add_u32", "v_alignbit_b32", "v_sub_f32", "v_lshrrev_b32", "v_and_or_b32", "v_bfi_b32", "v_mov_b32_sdwa WORD_0 -> WORD_1", "v_lshlrev_b32_e64", "v_lshlrev_b32 (register shift)", "v_mul_f32"};
    for (int threads = 256; threads <= 512; threads += 256) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        probe<K, KIND, MF><<<256, threads>>>(d, iters, sink);      // (clock ramp)
        hipEventRecord(e0);
        probe<K, KIND, MF><<<256, threads>>>(d, iters, sink);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        long long h[8];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        const double mfmas = MF ? 256.0 * (threads / 64) * iters * 16 : 0.0;
        printf("%-28s %s K = %d   %d wave/SIMD: %8.1f ticks per iteration of a wave (%5.2f per MFMA slot); launch %.3f ms = %.2f ticks/ns, %.2f PFLOP/s\n",
               kinds[KIND], MF ? "16 MFMA +" : "no MFMA, ", K, threads / 256, (double)h[0] / iters, (double)h[0] / iters / 16, ms,
               (double)h[0] / (ms * 1e6), mfmas * 16384.0 / (ms * 1e-3) * 1e-15);
    }
}

int main() {
    long long* d;
    float* sink;
    hipMalloc(&d, 256 * 8 * sizeof(long long));
    hipMalloc(&sink, 4);
    run<0, 0, true>(d, sink);
    run<1, 0, true>(d, sink); run<2, 0, true>(d, sink); run<3, 0, true>(d, sink); run<4, 0, true>(d, sink); run<6, 0, true>(d, sink);
    run<1, 0, false>(d, sink); run<3, 0, false>(d, sink); run<6, 0, false>(d, sink);
    run<3, 1, true>(d, sink); run<3, 1, false>(d, sink); run<6, 1, true>(d, sink); run<6, 1, false>(d, sink);
    run<3, 2, true>(d, sink); run<3, 2, false>(d, sink);
    run<6, 3, false>(d, sink); run<6, 4, false>(d, sink); run<6, 5, false>(d, sink); run<6, 6, false>(d, sink);
    run<3, 3, true>(d, sink); run<3, 6, true>(d, sink);
    run<6, 7, false>(d, sink); run<6, 8, false>(d, sink); run<6, 9, false>(d, sink); run<6, 10, false>(d, sink); run<6, 11, false>(d, sink);
    run<6, 12, false>(d, sink); run<6, 13, false>(d, sink);
    run<6, 14, false>(d, sink); run<6, 15, false>(d, sink); run<6, 16, false>(d, sink); run<6, 17, false>(d, sink);
    return 0;
}
