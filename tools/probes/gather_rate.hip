// Probe: how fast can an MI355X gather random 256-byte rows (the access pattern of every kernel on the hot path: one feature row
// per edge)?  10M rows x 64 floats (2.56 GB, the headline feature matrix), 100M uniformly random row indices, each wave
// instruction loads 4 rows x 256 B (lane = 16-byte piece), 8 instructions per batch, two batches in flight -- the register
// pipeline of rgcn_dw_tile_kernel with nothing else in the loop.  Swept over waves per CU; the last lines stream the same
// number of bytes contiguously for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void fill_idx(int* idx, long n, unsigned rows, int sequential) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        idx[i] = sequential ? (int)(i % rows) : (int)(h % rows);
    }
}

template <int BATCH>
__global__ void __launch_bounds__(256) gather(const float* __restrict__ x, const int* __restrict__ idx, long n_idx, float* sink) {
    const int lane = threadIdx.x & 63, kq = lane >> 4, ml = lane & 15;
    const long waves = (long)gridDim.x * (blockDim.x >> 6), wave = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const long per = n_idx / waves / (4 * BATCH) * (4 * BATCH);
    const int* my = idx + wave * per;
    f32x4 acc = {0, 0, 0, 0};
    f32x4 a[BATCH], b[BATCH];
    auto issue = [&](f32x4 (&d)[BATCH], long o) {
#pragma unroll
        for (int s = 0; s < BATCH; ++s) d[s] = *(const f32x4*)(x + (size_t)my[o + 4 * s + kq] * 64 + 4 * ml);
    };
    auto use = [&](f32x4 (&d)[BATCH]) {
#pragma unroll
        for (int s = 0; s < BATCH; ++s) acc += d[s];
    };
    issue(a, 0);
    for (long o = 0; o + 8 * BATCH <= per; o += 8 * BATCH) {
        issue(b, o + 4 * BATCH);
        use(a);
        if (o + 12 * BATCH <= per) issue(a, o + 8 * BATCH);
        use(b);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

int main() {
    const unsigned rows = 10000000;
    const long n_idx = 100000000;
    float *x, *sink;
    int* idx;
    hipMalloc(&x, (size_t)rows * 256);
    hipMalloc(&idx, n_idx * 4);
    hipMalloc(&sink, 64);
    hipMemset(x, 0, (size_t)rows * 256);
    for (int sequential = 0; sequential < 2; ++sequential) {
        hipLaunchKernelGGL(fill_idx, dim3(4096), dim3(256), 0, 0, idx, n_idx, rows, sequential);
        for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(gather<8>, dim3(256 * wg_per_cu), dim3(256), 0, 0, x, idx, n_idx, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("%s rows, %2d waves/CU (2 x 8 loads of 4 rows in flight per wave): %.3f ms  %.2f TB/s of rows (+ %.2f TB/s of indices)\n",
                   sequential ? "sequential" : "random    ", 4 * wg_per_cu, best, n_idx * 256.0 / best / 1e9, n_idx * 4.0 / best / 1e9);
        }
    }
    return 0;
}
