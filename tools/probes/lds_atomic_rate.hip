// Probe: cycles per ds_add_f32 wave-instruction under different address patterns (4 waves per CU,
// every CU busy), against ds_write_b32 with the same addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) probe(long long* cyc, int pattern, int use_write, int iters, const int* rows) {
    __shared__ float lds[256 * 68];
    for (int i = threadIdx.x; i < 256 * 68; i += 256) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rowl = lane & 15, kq = lane >> 4;
    long long t0 = clock64();
    float v = 1.0f + lane;
    unsigned rs = rows[kq] * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        rs = rs * 1664525u + 1013904223u;                    // "random" destination row per lane group
        const int d = __builtin_amdgcn_readlane(rs >> 9, 0) + kq * 37 & 255;
        int addr;
        if (pattern == 0) addr = wave * 64 + lane;                         // lane-linear, conflict free
        else if (pattern == 1) addr = d * 64 + 16 * wave + rowl;           // current layout: 16 banks, 4-way
        else if (pattern == 2) addr = d * 65 + 16 * wave + rowl;           // row stride 65
        else if (pattern == 3) addr = d * 68 + 16 * wave + rowl;           // row stride 68
        else addr = (d & ~3) * 64 + kq * 64 + 16 * wave + rowl;            // 4 groups on 4 consecutive rows, stride 64
        if (use_write) lds[addr] = v; else atomicAdd(&lds[addr], v);
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (lds[threadIdx.x] == 12345.f) cyc[0] = 0;
}
int main() {
    long long* d; hipMalloc(&d, 256 * 8);
    std::vector<int> h(1024); unsigned s = 12345; for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (s >> 8) & 255; }
    int* rows; hipMalloc(&rows, 4096); hipMemcpy(rows, h.data(), 4096, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int w = 0; w < 2; ++w)
        for (int p = 0; p < 5; ++p) {
            probe<<<256, 256>>>(d, p, w, iters, rows);
            hipDeviceSynchronize();
            long long c[256]; hipMemcpy(c, d, sizeof(c), hipMemcpyDeviceToHost);
            double avg = 0; for (int i = 0; i < 256; ++i) avg += c[i]; avg /= 256;
            printf("%s pattern %d: %.1f clock64 ticks per wave-instruction (4 waves/CU issuing)\n", w ? "ds_write_b32" : "ds_add_f32 ", p, avg / iters);
        }
    return 0;
}
