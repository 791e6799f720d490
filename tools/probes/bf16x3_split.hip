// Probe: what a split-precision contraction would cost next to the exact fp32 MFMA.
//
// Numerics first (tools/debug/bf16_split_error.py, CPU): on the headline layer's distribution a 2-way bf16 split
// (3 or 4 products) breaks the 1e-5 criterion (max error 2.1e-5 .. 2.8e-5); a 3-WAY split x = h + m + l with the SIX
// products hh, hm, mh, hl, lh, mm meets it (max error 1.4e-6: no worse than the sequential fp32 chain).
//
// This probe prices the compute side of that form on gfx950 for ONE wave that owns 16 rows x all 64 output columns
// (the only ownership in which a row is split once):
//   mode 0  exact fp32: 64 x v_mfma_f32_16x16x4_f32 per row tile (what the tile kernel runs: 32 cycles each)
//   mode 1  bf16x3, W pre-split at pack time, X split ON THE FLY from fp32 registers: per lane 16 elements ->
//           3 x v_cvt_pk_bf16_f32 + 2 x (expand + subtract) per pair, then 6 products x 4 column tiles x 2 k-steps =
//           48 x v_mfma_f32_16x16x32_bf16
//   mode 2  bf16x3 with X pre-split planes (no VALU; 1.5x the gathered bytes): the 48 MFMAs alone
// Prints cycles per row tile with one wave per SIMD.  hipcc --offload-arch=gfx950 -O3 bf16x3_split.hip -o bf16x3_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {      // two floats -> packed bf16 (RNE)
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int MODE>
__global__ void __launch_bounds__(256) probe(long long* cyc, int iters, float* sink, const float* src) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = src[lane * 16 + i];
    // weight fragments: fp32: 4 column tiles x 16 k-steps of one float; bf16: 3 planes x 4 column tiles x 2 k-steps x 8 bf16
    float wf[4];
    bf16x8 wb[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = src[1024 + lane + 64 * i];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 8; ++i) wb[p][i] = (short)(lane * 7 + p * 3 + i);
    bf16x8 pre[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) pre[p][k][i] = (short)(lane + p + k + i);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(x[k]), "v"(wf[c]));
        } else {
            bf16x8 pl[3][2];
            if (MODE == 1) {
                // 3-way split of this lane's 16 fp32 elements (two k-steps of 8)
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float a = x[8 * k + 2 * j], b = x[8 * k + 2 * j + 1];
                        const unsigned h = cvt_pk(a, b);
                        a -= __uint_as_float(h << 16);
                        b -= __uint_as_float(h & 0xFFFF0000u);
                        const unsigned m = cvt_pk(a, b);
                        a -= __uint_as_float(m << 16);
                        b -= __uint_as_float(m & 0xFFFF0000u);
                        const unsigned l = cvt_pk(a, b);
                        ((unsigned*)&pl[0][k])[j] = h;
                        ((unsigned*)&pl[1][k])[j] = m;
                        ((unsigned*)&pl[2][k])[j] = l;
                    }
            } else {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int k = 0; k < 2; ++k) pl[p][k] = pre[p][k];
            }
            // hh, hm, mh, hl, lh, mm
            const int pa[6] = {0, 0, 1, 0, 2, 1}, pb[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl[pa[t]][k], wb[pb[t]], acc[c], 0, 0, 0);
        }
        // keep the inputs moving so nothing is hoisted out of the loop
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(x[i]));
    }
    const long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static void run(const char* name, long long* cyc, float* sink, const float* src) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, cyc, iters, sink, src);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, cyc, iters, sink, src);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-58s %8.1f cycles per 16-row x 64 x 64 tile (one wave per SIMD, s_memtime ticks)\n", name, s / h.size() / iters);
}

int main() {
    long long* cyc;
    float *sink, *src;
    hipMalloc(&cyc, 256 * 4 * sizeof(long long));
    hipMalloc(&sink, 256 * 256 * sizeof(float));
    hipMalloc(&src, 4096 * sizeof(float));
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 0.001f * (float)((i * 2654435761u) % 2000) - 1.0f;
    hipMemcpy(src, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    run<0>("exact fp32: 64 x v_mfma_f32_16x16x4_f32", cyc, sink, src);
    run<1>("bf16x3, X split on the fly (88 VALU) + 48 x mfma_16x16x32_bf16", cyc, sink, src);
    run<2>("bf16x3, X pre-split planes: 48 x mfma_16x16x32_bf16 alone", cyc, sink, src);
    return 0;
}
