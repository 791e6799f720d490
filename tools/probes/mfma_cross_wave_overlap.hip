// Probe: can the vector work of ONE wave run in the shadow of ANOTHER wave's MFMAs on the same SIMD of gfx950?
// The tile kernel's consumer and producer waves share a SIMD.  rgcn_tile_kernel's contraction is v_mfma_f32_16x16x4_f32,
// which (tools/probes/mfma_f32_overlap.hip) shares the fp32 FMA pipe with every vector instruction of its own wave; a
// split-precision design would move the operand split (pure VALU work) into the producer waves and keep bf16 MFMAs in the
// consumers.  Whether that pays is decided by this number: time of an MFMA wave and of a VALU wave on one SIMD, each
// alone and both together, for the fp32 and the bf16 MFMA.
// Workgroup = 8 waves: waves 0-3 (one per SIMD) issue MFMAs, waves 4-7 (their SIMD mates) issue v_fma / v_cvt_pk / v_and.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA_F32(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_BF16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VFMA(x, a, b) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b))
#define VCVT(d, a, b) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define VAND(d, a) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(d) : "v"(a))
#define VSUB(d, a, b) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))

// MF: 0 fp32 MFMA, 1 bf16 MFMA.  who: 1 MFMA waves only, 2 VALU waves only, 3 both.  VK: 0 v_fma chain, 1 the split mix
template <int MF, int VK>
__global__ void __launch_bounds__(512) probe(long long* cyc, int iters, int who, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool mfma_wave = wave < 4;
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    f32x4 a4 = {1.f + lane, 2.f, 3.f, 4.f}, b4 = {0.5f, 0.25f, 1.f, 2.f};
    float a = 1.0f + lane, b = 0.5f;
    float f[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    unsigned u[4] = {0, 0, 0, 0};
    long long t0 = 0, t1 = 0;
    if (mfma_wave && (who & 1)) {
        t0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MF == 0) { MFMA_F32(c0, a, b); MFMA_F32(c1, a, b); MFMA_F32(c2, a, b); MFMA_F32(c3, a, b); }
                else { MFMA_BF16(c0, a4, b4); MFMA_BF16(c1, a4, b4); MFMA_BF16(c2, a4, b4); MFMA_BF16(c3, a4, b4); }
            }
        }
        t1 = clock64();
    } else if (!mfma_wave && (who & 2)) {
        t0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (VK == 0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) VFMA(f[k], a, b);
                } else {      // what splitting two floats into three bf16 planes costs: cvt, 2 x and, 2 x sub, cvt, ...
                    VCVT(u[0], f[0], f[1]); VAND(u[1], u[0]); VSUB(f[2], f[0], a); VSUB(f[3], f[1], b);
                    VCVT(u[2], f[2], f[3]); VAND(u[3], u[2]); VSUB(f[4], f[2], a); VSUB(f[5], f[3], b);
                }
            }
        }
        t1 = clock64();
    }
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float s = c0[0] + c1[0] + c2[0] + c3[0];
    for (int k = 0; k < 8; ++k) s += f[k];
    s += (float)(u[0] + u[1] + u[2] + u[3]);
    if (s == 12345.f) sink[0] = s;
}

template <int MF, int VK>
static void run(const char* what, long long* d, float* sink) {
    const int iters = 4000;
    static long long c[2048];
    double res[4][2];
    for (int who = 1; who <= 3; ++who) {
        probe<MF, VK><<<256, 512>>>(d, iters, who, sink);
        hipDeviceSynchronize();
        hipMemcpy(c, d, sizeof(c), hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int i = 0; i < 256; ++i)
            for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += c[i * 8 + w];
        res[who][0] = m / (256.0 * 4 * iters);
        res[who][1] = v / (256.0 * 4 * iters);
    }
    printf("%-52s MFMA wave alone %7.1f | VALU wave alone %7.1f | together: MFMA wave %7.1f, VALU wave %7.1f  (ticks per iteration; 16 MFMAs / 64 VALU)\n",
           what, res[1][0], res[2][1], res[3][0], res[3][1]);
}

int main() {
    long long* d;
    float* sink;
    hipMalloc(&d, 2048 * 8);
    hipMalloc(&sink, 64);
    run<0, 0>("fp32 MFMA 16x16x4   next to a v_fma_f32 wave", d, sink);
    run<0, 1>("fp32 MFMA 16x16x4   next to a cvt/and/sub wave", d, sink);
    run<1, 0>("bf16 MFMA 16x16x32  next to a v_fma_f32 wave", d, sink);
    run<1, 1>("bf16 MFMA 16x16x32  next to a cvt/and/sub wave", d, sink);
    return 0;
}
