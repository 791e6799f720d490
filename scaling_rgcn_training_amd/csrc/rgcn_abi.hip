// rgcn_abi.hip -- the small entry points of include/rgcn_mi355x.h: version, status strings, weight packing (fp32 MFMA
// fragment order + the bf16 x 3 planes of 64 x 64 layers), activation backward.
#include "rgcn_kernels_shared.h"

namespace rgcn {

// ------------------------------------------------------------------------------------------------
// weight pack
// ------------------------------------------------------------------------------------------------
// packed[((rel*NT + s)*KT + j)*256 + lane*4 + t] = B_rel[k = 16j + 4*(lane>>4) + t][col = 16s + (lane&15)]
// which is exactly the (a, b) pairing the consumers use for v_mfma_f32_16x16x4_f32:
// lane l supplies A[row = l&15][k' = l>>4] and B[k' = l>>4][col = l&15]; MFMA step (j,t) stands for
// k = 16j + 4k' + t on both operands.
// Where the relation weights come from (reference BASELINE.json configs[2]: basis decomposition B = 30):
//   dense  weight[R', in, out];
//   basis  W_r = sum_b comp[r, b] * bases[b]   (PyG: (comp @ weight.view(B, -1)).view(R', in, out)), summed b = 0 .. B - 1 in fp32;
//   block  W_r = blockdiag(blocks[r, 0 .. nb - 1])   with blocks [R', nb, in / nb, out / nb]: zeros off the diagonal.
// The packers read W_r[k][col] through this, so a decomposed layer never materialises [R', in, out].
struct WeightSource {
    const float* weight;    // dense weights, bases, or blocks
    const float* comp;      // basis: [R', B]
    const float* root;      // [in, out] or NULL (zeros)
    int mode;               // 0 dense, 1 basis, 2 block
    int num_rel, nb, din, dout;      // nb: number of bases / blocks
    __device__ __forceinline__ float at(int rel, int k, int col) const {      // W_rel[k][col]; rel == num_rel: root
        if (rel >= num_rel) return root != nullptr ? root[(size_t)k * dout + col] : 0.f;
        if (mode == 0) return weight[((size_t)rel * din + k) * dout + col];
        if (mode == 1) {
            float v = 0.f;
            for (int b = 0; b < nb; ++b) v = fmaf(comp[(size_t)rel * nb + b], weight[((size_t)b * din + k) * dout + col], v);
            return v;
        }
        const int bi = din / nb, bo = dout / nb, blk = k / bi;
        if (col / bo != blk) return 0.f;
        return weight[(((size_t)rel * nb + blk) * bi + (k - blk * bi)) * bo + (col - blk * bo)];
    }
};

__global__ void rgcn_pack_kernel(const WeightSource src, int transpose, int KP, int NP, float* __restrict__ packed) {
    const int per_rel = KP * NP;
    const long total = (long)(src.num_rel + 1) * per_rel;
    const int KT = KP / 16;
    const int din = src.din, dout = src.dout;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int rel = (int)(idx / per_rel);
        int rem = (int)(idx % per_rel);
        const int t = rem & 3;
        const int lane = (rem >> 2) & 63;
        rem >>= 8;
        const int j = rem % KT;
        const int s = rem / KT;
        const int k = 16 * j + 4 * (lane >> 4) + t;
        const int col = 16 * s + (lane & 15);
        float v = 0.f;
        if (!transpose) {
            if (k < din && col < dout) v = src.at(rel, k, col);
        } else {
            if (k < dout && col < din) v = src.at(rel, col, k);
        }
        packed[idx] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 x 3 planes of the weights (64 x 64 layers): what rgcn_tile3p_kernel multiplies by
// ------------------------------------------------------------------------------------------------
// x W = (xh + xm + xl)(Wh + Wm + Wl) with bf16 pieces (h = bf16(v), m = bf16(v - h), l = bf16(v - h - m): 24 significant
// bits, i.e. v = h + m + l up to the last fp32 bit) and the SIX products hh, hm, mh, hl, lh, mm on v_mfma_f32_16x16x32_bf16
// (bf16 x bf16 products are exact in the fp32 accumulator; the dropped ml, lm, ll terms are below 2^-24 relative).
// W is split once, here; x is split by the producer waves of the kernel (rgcn_tile3p.hip).
constexpr int kPack3FragsPerRel = 2 * 3 * 2 * 2;        // [column half c][plane][column tile ct][k-step s]
constexpr size_t kPack3FloatsPerRel = (size_t)kPack3FragsPerRel * 64 * 4;   // 64 lanes x 16 bytes per fragment

// packed3[((((rel * 2 + c) * 3 + pl) * 2 + ct) * 2 + s) * 64 + lane] (16 bytes = 8 bf16): element j =
// plane pl of B_rel[k = 32 s + 8 (lane >> 4) + j][col = 32 c + 16 ct + (lane & 15)] -- the A operand of the Y^T product
// (A[row = column][k]) and, read the other way round, the B operand of the Y product (B[k][col]).
__device__ __forceinline__ unsigned bf16_rne(float v) {
    const unsigned u = __float_as_uint(v);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__global__ void rgcn_pack3_kernel(const WeightSource src, int transpose, uint4* __restrict__ packed) {
    const long total = (long)(src.num_rel + 1) * kPack3FragsPerRel * 64;
    const int din = src.din, dout = src.dout;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long rem = idx;
        const int lane = (int)(rem & 63); rem >>= 6;
        const int s = (int)(rem & 1); rem >>= 1;
        const int ct = (int)(rem & 1); rem >>= 1;
        const int pl = (int)(rem % 3); rem /= 3;
        const int c = (int)(rem & 1); rem >>= 1;
        const int rel = (int)rem;
        unsigned h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * s + 8 * (lane >> 4) + j, col = 32 * c + 16 * ct + (lane & 15);
            float v = 0.f;
            if (!transpose) {
                if (k < din && col < dout) v = src.at(rel, k, col);
            } else {
                if (k < dout && col < din) v = src.at(rel, col, k);
            }
            unsigned b = bf16_rne(v);
            for (int q = 0; q < pl; ++q) {
                v -= __uint_as_float(b << 16);
                b = bf16_rne(v);
            }
            h[j] = b;
        }
        packed[idx] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    }
}

// dz = da * act'(a) for an activation fused into rgcn_fwd's store (a = act(z)): relu -> (a > 0), sigmoid -> a (1 - a).
// 16 bytes per lane, grid-stride.  Used where no consumer kernel can fold the mask (rgcn_bwd_dx's `relu_of`).
__global__ void rgcn_act_backward_kernel(const float* __restrict__ av, const float* __restrict__ da, float* __restrict__ dz,
                                         long rows, int ld4, int act) {
    const long total = rows * ld4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const f32x4 y = ((const f32x4*)av)[i];
        f32x4 g = ((const f32x4*)da)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) g[c] = act == RGCN_ACT_RELU ? (y[c] > 0.f ? g[c] : 0.f) : g[c] * y[c] * (1.f - y[c]);
        ((f32x4*)dz)[i] = g;
    }
}

// ---- gradients of a decomposed layer's parameters from the dense d_W[R', in, out] the weight-gradient kernels produce --------
// basis: d_bases[b] = sum_r comp[r, b] d_W[r] (r ascending), d_comp[r, b] = <d_W[r], bases[b]> (fixed tree): Appendix A
__global__ void rgcn_basis_dbases_kernel(const float* __restrict__ dw, const float* __restrict__ comp, int R, int B, int per,
                                         float* __restrict__ d_bases) {
    const long total = (long)B * per;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / per), i = (int)(idx % per);
        float v = 0.f;
        for (int r = 0; r < R; ++r) v = fmaf(comp[(size_t)r * B + b], dw[(size_t)r * per + i], v);
        d_bases[idx] = v;
    }
}
// one wave per (r, b): lane-strided partial sums, then a fixed shuffle tree
__global__ void rgcn_basis_dcomp_kernel(const float* __restrict__ dw, const float* __restrict__ bases, int R, int B, int per,
                                        float* __restrict__ d_comp) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= (long)R * B) return;
    const int r = (int)(wave / B), b = (int)(wave % B);
    float v = 0.f;
    for (int i = lane; i < per; i += 64) v = fmaf(dw[(size_t)r * per + i], bases[(size_t)b * per + i], v);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) d_comp[wave] = v;
}
__global__ void rgcn_block_dblocks_kernel(const float* __restrict__ dw, int R, int nb, int din, int dout, float* __restrict__ d_blocks) {
    const int bi = din / nb, bo = dout / nb;
    const long total = (long)R * nb * bi * bo;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long rem = idx;
        const int o = (int)(rem % bo); rem /= bo;
        const int i = (int)(rem % bi); rem /= bi;
        const int blk = (int)(rem % nb);
        const int r = (int)(rem / nb);
        d_blocks[idx] = dw[((size_t)r * din + blk * bi + i) * dout + blk * bo + o];
    }
}

static int pack_from(const WeightSource& src, int transpose, float* packed, void* stream) {
    const int kin = transpose ? src.dout : src.din, nout = transpose ? src.din : src.dout;
    const int KP = padded_width(kin), NP = padded_width(nout);
    if (KP == 0 || NP == 0) return RGCN_ERR_WIDTH;
    const long total = (long)(src.num_rel + 1) * KP * NP;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(rgcn_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, transpose, KP, NP, packed);
    if (KP == 64 && NP == 64) {      // 64 x 64 layers also carry the bf16 x 3 planes (rgcn_tile3p_kernel), behind the fp32 fragments
        const long lanes = (long)(src.num_rel + 1) * kPack3FragsPerRel * 64;
        hipLaunchKernelGGL(rgcn_pack3_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, transpose,
                           (uint4*)(packed + total));
    }
    return (int)hipGetLastError();
}

}  // namespace rgcn

using namespace rgcn;

extern "C" int rgcn_abi_version(void) { return RGCN_ABI_VERSION; }

extern "C" const char* rgcn_status_string(int status) {
    switch (status) {
        case RGCN_OK: return "ok";
        case RGCN_ERR_NULL: return "required pointer is NULL";
        case RGCN_ERR_WIDTH: return "feature width outside 1..128";
        case RGCN_ERR_STRIDE: return "row stride must be a multiple of 4 elements and >= the width rounded up to 4";
        case RGCN_ERR_PLAN: return "inconsistent graph plan";
        case RGCN_ERR_LDS: return "plan tile too large for the 160 KiB LDS at these widths";
        case RGCN_ERR_WORKSPACE: return "workspace too small";
        case RGCN_ERR_DEVICE: return "current device is not gfx950 (MI355X)";
        case RGCN_ERR_ACT: return "unknown activation code";
        case RGCN_ERR_GRAPH: return "edge_index / edge_type value out of range";
        case RGCN_ERR_ADDRESS: return "matrix not addressable through a buffer descriptor (2^24 rows or 4 GiB and more)";
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown status";
}

extern "C" int rgcn_padded_width(int width) { return padded_width(width); }

// 64 x 64 layers also carry the bf16 x 3 split of the weights (rgcn_tile3p_kernel, rgcn_ep_transform3_kernel), behind the fp32 fragments
static size_t pack3_floats(int num_relations, int KP, int NP) {
    return (KP == 64 && NP == 64) ? (size_t)(num_relations + 1) * kPack3FloatsPerRel : 0;
}

extern "C" size_t rgcn_packed_weight_floats(int num_relations, int din, int dout) {
    const int a = padded_width(din), b = padded_width(dout);
    if (a == 0 || b == 0 || num_relations <= 0) return 0;
    return (size_t)(num_relations + 1) * a * b + pack3_floats(num_relations, a, b);
}

extern "C" int rgcn_pack_weights(const float* weight, const float* root, int num_relations, int din, int dout,
                                 int transpose, float* packed, void* stream) {
    if (!weight || !packed) return RGCN_ERR_NULL;
    if (num_relations <= 0) return RGCN_ERR_PLAN;
    return pack_from(WeightSource{weight, nullptr, root, 0, num_relations, 0, din, dout}, transpose, packed, stream);
}

extern "C" int rgcn_pack_weights_basis(const float* bases, const float* comp, const float* root, int num_relations, int num_bases,
                                       int din, int dout, int transpose, float* packed, void* stream) {
    if (!bases || !comp || !packed) return RGCN_ERR_NULL;
    if (num_relations <= 0 || num_bases <= 0) return RGCN_ERR_PLAN;
    return pack_from(WeightSource{bases, comp, root, 1, num_relations, num_bases, din, dout}, transpose, packed, stream);
}

extern "C" int rgcn_pack_weights_block(const float* blocks, const float* root, int num_relations, int num_blocks, int din, int dout,
                                       int transpose, float* packed, void* stream) {
    if (!blocks || !packed) return RGCN_ERR_NULL;
    if (num_relations <= 0 || num_blocks <= 0 || din % num_blocks != 0 || dout % num_blocks != 0) return RGCN_ERR_PLAN;
    return pack_from(WeightSource{blocks, nullptr, root, 2, num_relations, num_blocks, din, dout}, transpose, packed, stream);
}

extern "C" int rgcn_basis_backward(const float* d_w, const float* bases, const float* comp, int num_relations, int num_bases, int din,
                                   int dout, float* d_bases, float* d_comp, void* stream) {
    if (!d_w || !bases || !comp) return RGCN_ERR_NULL;
    if (num_relations <= 0 || num_bases <= 0 || din <= 0 || dout <= 0) return RGCN_ERR_PLAN;
    const int per = din * dout;
    hipStream_t s = (hipStream_t)stream;
    if (d_bases != nullptr) {
        const long total = (long)num_bases * per;
        hipLaunchKernelGGL(rgcn_basis_dbases_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_w, comp, num_relations,
                           num_bases, per, d_bases);
    }
    if (d_comp != nullptr) {
        const long waves = (long)num_relations * num_bases;
        hipLaunchKernelGGL(rgcn_basis_dcomp_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, d_w, bases, num_relations,
                           num_bases, per, d_comp);
    }
    return (int)hipGetLastError();
}

extern "C" int rgcn_block_backward(const float* d_w, int num_relations, int num_blocks, int din, int dout, float* d_blocks, void* stream) {
    if (!d_w || !d_blocks) return RGCN_ERR_NULL;
    if (num_relations <= 0 || num_blocks <= 0 || din % num_blocks != 0 || dout % num_blocks != 0) return RGCN_ERR_PLAN;
    const long total = (long)num_relations * din * dout / num_blocks;
    hipLaunchKernelGGL(rgcn_block_dblocks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_w,
                       num_relations, num_blocks, din, dout, d_blocks);
    return (int)hipGetLastError();
}

extern "C" int rgcn_act_backward(const float* a, const float* da, float* dz, long rows, int ld, int act, void* stream) {
    if (!a || !da || !dz) return RGCN_ERR_NULL;
    if (ld <= 0 || (ld % 4) != 0) return RGCN_ERR_STRIDE;
    if (act != RGCN_ACT_RELU && act != RGCN_ACT_SIGMOID) return RGCN_ERR_ACT;
    if (rows <= 0) return RGCN_OK;
    const long total = rows * (ld / 4);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(rgcn_act_backward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, da, dz, rows, ld / 4, act);
    return (int)hipGetLastError();
}

