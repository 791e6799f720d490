// Probe: does ds_add_f32 (LDS float atomic add) round to nearest even like v_add_f32?
// Adds n small terms to an O(1) accumulator three ways and prints the error against double.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void probe(float* out, int n, float term) {
    __shared__ float acc_atomic;
    __shared__ float acc_plain;
    if (threadIdx.x == 0) { acc_atomic = 1.0f; acc_plain = 1.0f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = 1.0f;
        for (int i = 0; i < n; ++i) {
            atomicAdd(&acc_atomic, term);
            r += term;
            volatile float* p = &acc_plain; *p = *p + term;
        }
        out[0] = acc_atomic; out[1] = r; out[2] = acc_plain;
    }
}
int main() {
    float* d; hipMalloc(&d, 16);
    const int n = 11825; const float term = 4.1234567e-5f;
    probe<<<1, 64>>>(d, n, term);
    float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    double exact = 1.0 + (double)n * (double)term;
    printf("exact %.9f  ds_add_f32 %.9f (err %.3e)  v_add_f32 %.9f (err %.3e)  lds rmw %.9f (err %.3e)\n",
           exact, h[0], h[0] - exact, h[1], h[1] - exact, h[2], h[2] - exact);
    return 0;
}
