// rgcn_dw_root.hip -- the self-loop part of the R-GCN weight gradients for gfx950 (MI355X): d_root = X^T G, d_bias = sum of
// the rows of G.  Replaces what autograd computes for RGCNConv's `root` / `bias` at /root/reference/model/modelTrainer.py:66
// (the layer's `x @ root + bias` term: PyG 2.3.1 rgcn_conv.py, called from model/layers.py:21,23).
#include "rgcn_common.h"

namespace rgcn {

// ------------------------------------------------------------------------------------------------
// root / bias gradients: d_root = X^T G, d_bias = column sums of G -- a dense [in x rows] x [rows x out] product
// ------------------------------------------------------------------------------------------------
// The "gathered" rows of the self-loop relation are the node's own: no indices, no plan.  A wave streams a contiguous
// range of rows (one MFMA k-step = 4 rows: lane (ml, kq) loads 16 bytes of x[row + kq] and 16 of g[row + kq], 1 KiB
// coalesced per instruction), two batches of kRootBatch k-steps in registers, 16 MFMAs per k-step into a 64 x 64
// accumulator in 64 VGPRs (the output tiles are strided column sets, as in rgcn_dw_direct_kernel), and writes ONE slab;
// rgcn_dw_root_reduce_kernel sums the slabs in wave order (bit-reproducible).  No LDS and few enough registers that its
// workgroups fit a CU NEXT TO a workgroup of rgcn_tile_kernel: 5 GB of streaming reads and a tenth of a launch's MFMAs,
// which the host runs on a side stream under the MFMA-bound dX launch instead of after it.
constexpr int kRootBatch = 8;                    // k-steps per register batch (two batches in flight)
#ifndef RGCN_ROOT_MAX_WAVES
#define RGCN_ROOT_MAX_WAVES 1024
#endif
constexpr int kRootMaxWaves = RGCN_ROOT_MAX_WAVES;      // 1024: one wave per SIMD of the chip
constexpr int kRootSlabFloats = 64 * 64 + 4 * 64;   // accumulator + the four row-quarters' bias sums

struct DwRootArgs {
    const float* x;
    const float* g;
    float* slabs;       // [waves][kRootSlabFloats]
    long rows;
    int ldx, ldg, din4, dout4, waves;
    int want_bias;
};

__global__ void __launch_bounds__(256, 2) rgcn_dw_root_kernel(const DwRootArgs a) {
    constexpr int B = kRootBatch;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));     // wave-uniform (scalar registers)
    if (w >= a.waves) return;
    const int ml = lane & 15, kq = lane >> 4;
    const long ksteps = (a.rows + 3) / 4;
    const long k0 = ksteps * w / a.waves, k1 = ksteps * (w + 1) / a.waves;
    // Each wave addresses ITS rows through two buffer descriptors (base = first row of the range, num_records = bytes of the
    // range): a k-step past the end of the range, a row past the end of the matrix or a 16-byte column piece beyond the
    // width is out of range and the hardware range check feeds zeros -- no branch, no select.  Per load one v_add of the
    // lane's running offset (a lane beyond the width keeps the out-of-range marker: its step is 0).
    const long r0 = 4 * k0;
    const long rcnt = (4 * k1 < a.rows ? 4 * k1 : a.rows) - r0;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (size_t)r0 * a.ldx, (unsigned)(rcnt * a.ldx * 4));
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(a.g + (size_t)r0 * a.ldg, (unsigned)(rcnt * a.ldg * 4));
    const unsigned xstep = ml < a.din4 ? 16u * (unsigned)a.ldx : 0u, gstep = ml < a.dout4 ? 16u * (unsigned)a.ldg : 0u;
    unsigned xo = ml < a.din4 ? (unsigned)(kq * a.ldx + 4 * ml) * 4u : 0xFFFFFFF0u;
    unsigned go = ml < a.dout4 ? (unsigned)(kq * a.ldg + 4 * ml) * 4u : 0xFFFFFFF0u;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    f32x4 acc[4][4];
#pragma unroll
    for (int ia = 0; ia < 4; ++ia)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ia][jb] = zero;
    f32x4 bsum = zero;

    f32x4 xa[2][B], ga[2][B];
    auto load_batch = [&](int buf) {                 // the next B k-steps of the range
#pragma unroll
        for (int s = 0; s < B; ++s) {
            xa[buf][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)xo, 0, 0));
            ga[buf][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)go, 0, 0));
            xo += xstep;
            go += gstep;
        }
    };
    auto compute_batch = [&](int buf) {
#pragma unroll
        for (int s = 0; s < B; ++s) {
            bsum += ga[buf][s];
#pragma unroll
            for (int ia = 0; ia < 4; ++ia)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb)
                    acc[ia][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[buf][s][ia], ga[buf][s][jb], acc[ia][jb], 0, 0, 0);
        }
    };
    load_batch(0);
    for (long k = k0; k < k1; k += 2 * B) {
        load_batch(1);
        compute_batch(0);
        load_batch(0);
        compute_batch(1);
    }
    // D layout of v_mfma_f32_16x16x4_f32: lane (ml, kq) holds D[m = 4 kq + r][n = ml]; m stands for x column 4 m + ia,
    // n for g column 4 ml + jb
    float* slab = a.slabs + (size_t)w * kRootSlabFloats;
#pragma unroll
    for (int ia = 0; ia < 4; ++ia)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 v;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) v[jb] = acc[ia][jb][r];
            *(f32x4*)(slab + (4 * (4 * kq + r) + ia) * 64 + 4 * ml) = v;
        }
    *(f32x4*)(slab + 64 * 64 + kq * 64 + 4 * ml) = bsum;
}

// d_root[k][n] = sum over the waves' slabs, d_bias[n] = sum over slabs and row quarters; fixed order: 16 strided partial
// sums (wave q, q + 16, ...) folded in order q = 0..15.  grid = 65 workgroups (64 rows of d_root + the bias) x 1024 threads.
__global__ void __launch_bounds__(1024) rgcn_dw_root_reduce_kernel(const float* __restrict__ slabs, int waves, int din, int dout,
                                                                   float* __restrict__ d_root, float* __restrict__ d_bias) {
    __shared__ float part[16][64];
    const int n = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int k = blockIdx.x;
    float sum = 0.f;
    if (k < 64) {
        for (int w = q; w < waves; w += 16) sum += slabs[(size_t)w * kRootSlabFloats + k * 64 + n];
    } else {
        for (int w = q; w < waves; w += 16) {
            const float* b = slabs + (size_t)w * kRootSlabFloats + 64 * 64 + n;
            sum += (b[0] + b[64]) + (b[128] + b[192]);
        }
    }
    part[q][n] = sum;
    __syncthreads();
    if (q != 0 || n >= dout) return;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += part[i][n];
    if (k < 64) {
        if (k < din && d_root != nullptr) d_root[(size_t)k * dout + n] = t;
    } else if (d_bias != nullptr) {
        d_bias[n] = t;
    }
}

}  // namespace rgcn

using namespace rgcn;

extern "C" size_t rgcn_bwd_dw_root_workspace_bytes(void) { return sizeof(float) * (size_t)kRootMaxWaves * kRootSlabFloats; }

// d_root = x^T g, d_bias = column sums of g over `rows` rows: the self-loop part of the weight gradients, plan-free
extern "C" int rgcn_bwd_dw_root(const float* x, int ldx, int din, const float* g, int ldg, int dout, long rows, void* workspace,
                                size_t workspace_bytes, float* d_root, float* d_bias, void* stream) {
    int st;
    if (!x || !g || !workspace) return RGCN_ERR_NULL;
    if (!d_root && !d_bias) return RGCN_ERR_NULL;
    if ((st = check_stride(ldx, din)) != RGCN_OK) return st;
    if ((st = check_stride(ldg, dout)) != RGCN_OK) return st;
    if (din > 64 || dout > 64) return RGCN_ERR_WIDTH;
    if (rows <= 0) return RGCN_ERR_PLAN;
    if (workspace_bytes < rgcn_bwd_dw_root_workspace_bytes()) return RGCN_ERR_WORKSPACE;
    if ((st = check_device()) != RGCN_OK) return st;
    DwRootArgs a;
    a.x = x;
    a.g = g;
    a.slabs = (float*)workspace;
    a.rows = rows;
    a.ldx = ldx;
    a.ldg = ldg;
    a.din4 = (din + 3) / 4;
    a.dout4 = (dout + 3) / 4;
    const long ksteps = (rows + 3) / 4;
    const long want = (ksteps + 2 * kRootBatch - 1) / (2 * kRootBatch);      // at least one double batch per wave
    a.waves = (int)(want < 4 ? 4 : (want > kRootMaxWaves ? kRootMaxWaves : want));
    a.want_bias = d_bias != nullptr;
    // a wave addresses its row range through 32-bit buffer offsets (the out-of-range marker sits at the top of that range)
    const long rows_per_wave = 4 * ((ksteps + a.waves - 1) / a.waves + 1);
    if ((unsigned long long)rows_per_wave * (unsigned long long)(ldx > ldg ? ldx : ldg) * 4ull >= 0xFFFFFF00ull) return RGCN_ERR_STRIDE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(rgcn_dw_root_kernel, dim3((a.waves + 3) / 4), dim3(256), 0, s, a);
    if ((st = (int)hipGetLastError()) != 0) return st;
    hipLaunchKernelGGL(rgcn_dw_root_reduce_kernel, dim3(65), dim3(1024), 0, s, a.slabs, a.waves, din, dout, d_root, d_bias);
    return (int)hipGetLastError();
}
