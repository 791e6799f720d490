// Probe: at what rate do DEPENDENT v_mfma_f32_16x16x32_bf16 issue on gfx950 -- a chain on ONE accumulator (vDst == SrcC of the
// next one), as the consumers of rgcn_tile3p_kernel run 12 of per row tile -- against the same count dealt over 2, 3, 4, 6
// accumulators?  (tools/probes/mfma_bf16_valu_shadow.hip dealt its MFMAs over four accumulators: 16.5 cycles each; the exact-fp32
// 16x16x4 chains were measured at full rate in round 2, tools/probes/mfma_f32_overlap.hip.)
//   hipcc --offload-arch=gfx950 -O2 mfma_bf16_dependent_chain.hip -o mfma_bf16_dependent_chain && ./mfma_bf16_dependent_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA32(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// CH accumulators, 12 MFMAs per "row tile" dealt round-robin; F32: the exact-fp32 MFMA instead
template <int CH, bool F32>
__global__ void __launch_bounds__(512) probe(long long* cyc, int iters, float* sink) {
    const int lane = threadIdx.x & 63;
    f32x4 c[6] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + lane); b[i] = (short)0x3f00; }
    float fa = 1.0f + lane, fb = 0.5f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            if (F32) MFMA32(c[j % CH], fa, fb);
            else MFMA(c[j % CH], a, b);
        }
    }
    const long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += c[i][0];
    if (s == 12345.f) sink[0] = s;
}

template <int CH, bool F32>
static void run(long long* d, float* sink) {
    const int iters = 4000;
    for (int threads = 256; threads <= 512; threads += 256) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        probe<CH, F32><<<256, threads>>>(d, iters, sink);      // (clock ramp)
        hipEventRecord(e0);
        probe<CH, F32><<<256, threads>>>(d, iters, sink);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        long long h[8];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        const double mfmas = 256.0 * (threads / 64) * iters * 12;
        printf("%s  %d accumulator(s), %d wave/SIMD: %7.1f ticks per 12 MFMAs of a wave; launch %.3f ms -> %.1f ns per MFMA and SIMD\n",
               F32 ? "v_mfma_f32_16x16x4_f32  " : "v_mfma_f32_16x16x32_bf16", CH, threads / 256, (double)h[0] / iters, ms,
               ms * 1e6 / (mfmas / (256.0 * 4)));
    }
}

int main() {
    long long* d;
    float* sink;
    hipMalloc(&d, 256 * 8 * sizeof(long long));
    hipMalloc(&sink, 4);
    run<1, false>(d, sink); run<2, false>(d, sink); run<3, false>(d, sink); run<4, false>(d, sink); run<6, false>(d, sink);
    run<1, true>(d, sink); run<2, true>(d, sink); run<4, true>(d, sink);
    return 0;
}
