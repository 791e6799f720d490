// Probe: what one wave can issue in the shadow of v_mfma_f32_16x16x4_f32 on gfx950 (the fp32 MFMA runs at the
// fp32 vector rate: does it leave the SIMD's issue port free like the bf16 forms do?).  One iteration = 16 MFMAs on
// two independent accumulator chains (the tile kernel's stage A) plus a filler mix; cycles per iteration, with one
// and with two such waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VADD(x) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x))
#define DSR128(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define DSR32(dst, addr) asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(addr))
#define DSW32(addr, v) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v))
#define MFMA32(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VFMA(x, a, b) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b))
#define VPKFMA(x, a, b) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b))
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)")

template <int MODE>
__global__ void __launch_bounds__(768) probe(long long* cyc, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    f32x16 d0 = {};
    f32x2 p0 = {0, 0}, p1 = {1, 1}, p2 = {2, 2}, p3 = {0, 0};
    float a = 1.0f + lane, b = 0.5f, f0 = 0, f1 = 0, f2 = 0, f3 = 0;
    f32x4 r[6];
    float q[8];
    for (int i = 0; i < 6; ++i) r[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) q[i] = 0;
    const unsigned addr128 = (threadIdx.x * 16u) & 0xFFFFu;
    const unsigned addr32 = ((lane >> 4) * 256u * 7u + (threadIdx.x >> 6) * 64u + (lane & 15) * 4u) & 0xFFFFu;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 7) { MFMA(c0, a, b); MFMA(c1, a, b); MFMA(c2, a, b); MFMA(c3, a, b); continue; }
            if (MODE == 8) { MFMA32(d0, a, b); continue; }
            if (MODE == 9) { MFMA(c0, a, b); VFMA(f0, a, b); VFMA(f1, a, b); MFMA(c1, a, b); VFMA(f2, a, b); VFMA(f3, a, b); continue; }
            if (MODE == 10) { MFMA(c0, a, b); VPKFMA(p0, p1, p2); MFMA(c1, a, b); VPKFMA(p3, p1, p2); continue; }
            if (MODE == 11) { MFMA(c0, a, b); MFMA(c0, a, b); continue; }       // ONE dependent chain
            MFMA(c0, a, b);
            if (MODE == 1) { VADD(f0); }
            if (MODE == 2) { VADD(f0); VADD(f1); VADD(f2); }
            if (MODE == 3 && j < 6) DSR128(r[j], addr128);
            if (MODE == 4) DSR32(q[j], addr32);
            if (MODE == 6) { VADD(f0); VADD(f1); if (j < 6) DSR128(r[j], addr128); }
            MFMA(c1, a, b);
            if (MODE == 1) { VADD(f1); }
            if (MODE == 2) { VADD(f3); VADD(f1); VADD(f2); }
            if (MODE == 4) DSW32(addr32, f0);
            if (MODE == 6) { VADD(f2); }
        }
        if (MODE == 5) {
#pragma unroll
            for (int j = 0; j < 5; ++j) { VADD(f0); VADD(f1); VADD(f2); VADD(f3); }
        }
        if (MODE == 3 || MODE == 4 || MODE == 6) LGKM0();
    }
    const long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    float s = c0[0] + c1[0] + c2[0] + c3[0] + d0[0] + p0[0] + p3[1] + f0 + f1 + f2 + f3;
    for (int i = 0; i < 6; ++i) s += r[i][0];
    for (int i = 0; i < 8; ++i) s += q[i];
    if (s == 12345.f) sink[0] = s;
}

template <int MODE>
static void run(const char* what, long long* d, float* sink) {
    const int iters = 4000;
    for (int threads = 256; threads <= 768; threads += 256) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        probe<MODE><<<256, threads>>>(d, iters, sink);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        static long long c[4096];
        hipMemcpy(c, d, sizeof(c), hipMemcpyDeviceToHost);
        double avg = 0;
        const int waves = threads / 64;
        for (int i = 0; i < 256; ++i)
            for (int w = 0; w < waves; ++w) avg += c[i * 16 + w];
        avg /= 256.0 * waves;
        printf("%-44s %d wave/SIMD: %7.1f ticks per iteration (per wave); kernel %.3f ms -> %.2f ticks/ns\n", what, threads / 256, avg / iters, ms, avg / (ms * 1e6));
    }
}

int main() {
    long long* d;
    float* sink;
    hipMalloc(&d, 4096 * 8);
    hipMalloc(&sink, 64);
    run<0>("16 MFMA only", d, sink);
    run<1>("+ 1 v_add after each MFMA (16)", d, sink);
    run<2>("+ 3 v_add after each MFMA (48)", d, sink);
    run<3>("+ 6 ds_read_b128, lgkmcnt(0) at end", d, sink);
    run<4>("+ 8 ds_read_b32 + 8 ds_write_b32 (4-way)", d, sink);
    run<5>("+ 20 v_add after the 16 MFMAs", d, sink);
    run<6>("+ 24 v_add + 6 ds_read_b128 interleaved", d, sink);
    run<7>("32 MFMA 16x16x4 on FOUR chains", d, sink);
    run<8>("8 MFMA 32x32x2 (same flops as 16 16x16x4)", d, sink);
    run<9>("16 MFMA + 2 v_fma after each (32)", d, sink);
    run<10>("16 MFMA + 1 v_pk_fma after each (16)", d, sink);
    run<11>("16 MFMA on ONE dependent chain", d, sink);
    return 0;
}
