// Probe: does v_mfma_f32_16x16x32_bf16 round or truncate when its products are small next to the
// accumulator?  N times the same k-step is added into C = c0: products that sum to +d (or -d) with d far below ulp(c0) * 2^k.
// Exact answer: c0 + N * d (computed in double).  Printed: the accumulated value's error in units of ulp(c0) for the bf16
// MFMA and for v_mfma_f32_16x16x4_f32 fed the same sums.  A rounding accumulate errs around 0 for both signs of d; a
// truncating one (two's-complement floor of the aligned addends) errs NEGATIVE for both.
//   hipcc --offload-arch=gfx950 -O2 mfma_bf16_accumulate_bias.hip -o mfma_bf16_accumulate_bias && ./mfma_bf16_accumulate_bias
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__global__ void probe(float c0, float a_val, float b_val, int n, float* out_bf16, float* out_f32) {
    const int lane = threadIdx.x;
    // bf16 operands: every A element = a_val, every B element = b_val (both exactly representable in bf16): one MFMA adds
    // 32 * a_val * b_val to every element of C
    const unsigned short ab = (unsigned short)(__float_as_uint(a_val) >> 16), bb = (unsigned short)(__float_as_uint(b_val) >> 16);
    bf16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (short)ab; B[i] = (short)bb; }
    f32x4 c = {c0, c0, c0, c0}, cf = c;
    for (int i = 0; i < n; ++i) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, c, 0, 0, 0);
        // the same 32 products as eight fp32 k-steps of four
        for (int j = 0; j < 8; ++j) cf = __builtin_amdgcn_mfma_f32_16x16x4f32(a_val, b_val, cf, 0, 0, 0);
    }
    if (lane == 0) { out_bf16[0] = c[0]; out_f32[0] = cf[0]; }
}

int main() {
    float *d1, *d2;
    hipMalloc(&d1, 4); hipMalloc(&d2, 4);
    const int n = 4096;
    printf("# c0        d = 32 a b     N     exact c0 + N d      bf16 MFMA err/ulp(c0)   fp32 MFMA err/ulp(c0)\n");
    for (float c0 : {256.0f, -256.0f, 300.5f}) {
        for (float sgn : {1.0f, -1.0f}) {
            for (int e : {-12, -16, -20}) {        // a b = 2^e / 32 * 1.25: d = 1.25 * 2^e, below ulp(256) = 2^-15 from e = -16 on
                const float a = 1.25f, b = sgn * ldexpf(1.0f, e - 5);
                const double d = 32.0 * (double)a * (double)b, exact = (double)c0 + n * d;
                probe<<<1, 64>>>(c0, a, b, n, d1, d2);
                float r1, r2;
                hipMemcpy(&r1, d1, 4, hipMemcpyDeviceToHost); hipMemcpy(&r2, d2, 4, hipMemcpyDeviceToHost);
                const double ulp = ldexp(1.0, ilogb((double)fabsf(c0)) - 23);
                printf("%8.2f  %+.4e  %5d  %18.10f   %+10.3f              %+10.3f\n", c0, d, n, exact, (r1 - exact) / ulp, (r2 - exact) / ulp);
            }
        }
    }
    return 0;
}
