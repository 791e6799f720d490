// rgcn_ep.hip -- the EDGE-PARALLEL path of the R-GCN layer's forward / dX for gfx950, and the entry points
// rgcn_ep_transform / rgcn_ep_segment_sum of include/rgcn_mi355x.h.
//
// Same arithmetic as rgcn_tile_kernel (torch_geometric.nn.RGCNConv.forward and its dX: reference model/layers.py:21,23,
// model/modelTrainer.py:66), cut for the graphs the reference actually trains on (model/modelTrainer.py:78,92: 89 / 45 /
// ~267 relation ids on 8k .. 1.5M nodes; hubs: graphs/AIFB/attr/sum/AIFB_sum_in.nt puts 11,825 edges on one node): there a
// (tile, relation) group of the tile-major plan holds a handful of rows, a tile kernel pays its per-chunk cost per RELATION,
// few tiles leave most CUs idle, and a hub's tile is walked by one workgroup.  Here (scaling_rgcn_training_amd/eplan.py):
//   rgcn_ep_transform_kernel    rows sorted relation-major in dense 64-slot units; a wave takes a contiguous range of units,
//                               keeps the relation's weight fragments in registers while the relation lasts, gathers the 16
//                               rows of a row tile straight into MFMA operand registers, Z^T = W_r^T X^T on
//                               v_mfma_f32_16x16x4_f32 (exact fp32; a lane ends with four consecutive columns of one row) and
//                               stores Z[slot] = w_slot * (x[src_slot] @ W_rel): no LDS, no ownership, any number of waves;
//   rgcn_ep_segment_sum_kernel  out[i] = act(bias + sum of the rows seg_idx[seg_ptr[i] .. seg_ptr[i + 1]) of Z): 16-byte
//                               pieces, a fixed order (bit-reproducible, no atomics); long segments go through levels.
// Bytes per row (in -> out): 8 + 4 in gathered, 4 out written, 4 + 4 out read again: HBM-bound when the graph is larger than
// the caches, launch-bound on the reference's datasets.
#include "rgcn_kernels_shared.h"

namespace rgcn {

struct EpArgs {
    const int* unit_rel;
    const int* unit_cnt;
    const int* slot_src;
    const float* slot_w;
    const float* x;
    const float* wp;       // rgcn_pack_weights: fp32 MFMA fragment order
    float* z;              // [n_units * 64][ldz]
    unsigned x_bytes;      // rows * ldx * 4 when x can be gathered through a buffer descriptor, else 0
    int n_rows, ldx, din4, ldz, n_units, units_per_wave;
};

// KEEP: the relation's KT x NT fragments stay in registers (at most 64 VGPRs) while consecutive units share the relation
template <int KP, int NP>
__global__ void __launch_bounds__(256) rgcn_ep_transform_kernel(const EpArgs a) {
    constexpr int KT = KP / 16, NT = NP / 16;
    constexpr bool KEEP = KT * NT <= 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int u0 = wave * a.units_per_wave;
    const int u1 = min(u0 + a.units_per_wave, a.n_units);
    if (u0 >= u1) return;
    const int row = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const f32x4* wp4 = (const f32x4*)a.wp + lane;
    f32x4 wf[KEEP ? KT * NT : 1];
    int rel_cur = -1;
    // this lane's 16-byte pieces of a gathered row: columns 16 j + 4 kq .. + 3; beyond the width: zeros
    unsigned coff[KT], rowb[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const bool in = 4 * j + kq < a.din4;
        coff[j] = in ? (unsigned)(16 * j + 4 * kq) * 4u : 0xFFFFFFF0u;
        rowb[j] = in ? (unsigned)a.ldx * 4u : 0u;
    }
    auto gather = [&](f32x4 (&a4)[KT], int src) {
        if (a.x_bytes != 0) {
#pragma unroll
            for (int j = 0; j < KT; ++j)
                a4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(__umul24((unsigned)src, rowb[j]) + coff[j]), 0, 0));
        } else {       // 64-bit pointers (2^24 rows or 4 GiB and more): padding rows and columns beyond the width read nothing
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                a4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (src < a.n_rows && 4 * j + kq < a.din4) a4[j] = *(const f32x4*)(a.x + (size_t)src * a.ldx + 16 * j + 4 * kq);
            }
        }
    };
    for (int u = u0; u < u1; ++u) {
        const int rel = ldc(a.unit_rel, u);
        const int ntile = ldc(a.unit_cnt, u) >> 4;
        if (KEEP && rel != rel_cur) {
#pragma unroll
            for (int s = 0; s < NT; ++s)
#pragma unroll
                for (int j = 0; j < KT; ++j) wf[KEEP ? j * NT + s : 0] = wp4[((size_t)(rel * NT + s) * KT + j) * 64];
            rel_cur = rel;
        }
        const size_t slot0 = (size_t)u * kChunk + row;
        int src = a.slot_src[slot0];
        float w = a.slot_w[slot0];
        f32x4 cur[KT];
        gather(cur, src);
        for (int t = 0; t < ntile; ++t) {
            // the next row tile's indices and rows are on their way while this one multiplies
            f32x4 nxt[KT];
            float w_n = 0.f;
            if (t + 1 < ntile) {
                const int src_n = a.slot_src[slot0 + 16 * (t + 1)];
                w_n = a.slot_w[slot0 + 16 * (t + 1)];
                gather(nxt, src_n);
            }
            f32x4 acc[NT];
#pragma unroll
            for (int s = 0; s < NT; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < KT; ++j)
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    const f32x4 b4 = KEEP ? wf[KEEP ? j * NT + s : 0] : wp4[((size_t)(rel * NT + s) * KT + j) * 64];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(b4[tt], cur[j][tt], acc[s], 0, 0, 0);
                }
            float* zr = a.z + (slot0 + 16 * t) * (size_t)a.ldz + 4 * kq;
#pragma unroll
            for (int s = 0; s < NT; ++s)
                if (16 * s + 4 * kq < a.ldz) *(f32x4*)(zr + 16 * s) = acc[s] * w;
            if (t + 1 < ntile) {
#pragma unroll
                for (int j = 0; j < KT; ++j) cur[j] = nxt[j];
                w = w_n;
            }
        }
    }
}

// 64 x 64 layers on large graphs: the same walk with the contraction on bf16 MFMAs over three-way split operands (the arithmetic of
// rgcn_tile3p_kernel: x W = (xh + xm + xl)(Wh + Wm + Wl), six products, fp32 accumulation from zero per row tile: fp32-equivalent).
// Here the wave that multiplies a row tile owns ALL 64 columns of it, so cutting its 16 x 64 values in registers (88 vector
// instructions) is done once per row, and the relation's three weight planes (96 VGPRs, rgcn_pack3_kernel's fragments) stay in
// registers over the hundreds of consecutive units a wave walks per relation -- the two costs that sank this form inside the tile
// kernel do not arise.  48 v_mfma_f32_16x16x32_bf16 of 16 cycles per row tile against 64
// v_mfma_f32_16x16x4_f32 of 32: the exact-fp32 form is bound by that rate at 64 x 64 (15.2 ms for 110M rows).
__global__ void __launch_bounds__(256) rgcn_ep_transform3_kernel(const EpArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int u0 = wave * a.units_per_wave;
    const int u1 = min(u0 + a.units_per_wave, a.n_units);
    if (u0 >= u1) return;
    const int row = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    // W planes: packed3[((((rel * 2 + c) * 3 + pl) * 2 + ct) * 2 + s) * 64 + lane]: A operand of Z^T = W^T X^T for columns
    // 32 c + 16 ct + (lane & 15), k = 32 s + 8 (lane >> 4) + (0..7)
    const uint4* wp4 = (const uint4*)a.wp + lane;
    bf16x8 wf[4][3][2];          // [column tile 2 c + ct][plane][k-step]
    int rel_cur = -1;
    // this lane's pieces of a gathered row: columns 32 s + 8 kq .. + 7 as two 16-byte loads; beyond the width: zeros
    unsigned coff[4], rowb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c4 = 8 * (i >> 1) + 2 * kq + (i & 1);          // 16-byte column group: (32 s + 8 kq + 4 h) / 4, i = 2 s + h
        const bool in = c4 < a.din4;
        coff[i] = in ? (unsigned)c4 * 16u : 0xFFFFFFF0u;
        rowb[i] = in ? (unsigned)a.ldx * 4u : 0u;
    }
    auto gather = [&](f32x4 (&a4)[4], int src) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a4[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(__umul24((unsigned)src, rowb[i]) + coff[i]), 0, 0));
    };
    constexpr int px[6] = {0, 0, 1, 0, 2, 1}, pw[6] = {0, 1, 0, 2, 0, 1};      // x plane, W plane: hh hm mh hl lh mm
    for (int u = u0; u < u1; ++u) {
        const int rel = ldc(a.unit_rel, u);
        const int ntile = ldc(a.unit_cnt, u) >> 4;
        if (rel != rel_cur) {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int sk = 0; sk < 2; ++sk)
                        wf[t4][pl][sk] = __builtin_bit_cast(bf16x8, wp4[((size_t)(((rel * 2 + (t4 >> 1)) * 3 + pl) * 2 + (t4 & 1)) * 2 + sk) * 64]);
            rel_cur = rel;
        }
        const size_t slot0 = (size_t)u * kChunk + row;
        int src = a.slot_src[slot0];
        float w = a.slot_w[slot0];
        f32x4 cur[4];
        gather(cur, src);
        for (int t = 0; t < ntile; ++t) {
            f32x4 nxt[4];
            float w_n = 0.f;
            if (t + 1 < ntile) {
                const int src_n = a.slot_src[slot0 + 16 * (t + 1)];
                w_n = a.slot_w[slot0 + 16 * (t + 1)];
                gather(nxt, src_n);
            }
            // three round-to-nearest bf16 pieces of the lane's 16 values: xp[plane][k-step] = 8 bf16 of k = 32 s + 8 kq + (0..7)
            u32x4 xp[3][2];
#pragma unroll
            for (int sk = 0; sk < 2; ++sk)
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    unsigned h, m, l;
                    split3_pair(cur[2 * sk + (jp >> 1)][2 * (jp & 1)], cur[2 * sk + (jp >> 1)][2 * (jp & 1) + 1], h, m, l);
                    xp[0][sk][jp] = h;
                    xp[1][sk][jp] = m;
                    xp[2][sk][jp] = l;
                }
            float* zr = a.z + (slot0 + 16 * t) * (size_t)a.ldz + 4 * kq;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sk = 0; sk < 2; ++sk)
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t4][pw[q]][sk], __builtin_bit_cast(bf16x8, xp[px[q]][sk]), acc, 0, 0, 0);
                if (16 * t4 + 4 * kq < a.ldz) *(f32x4*)(zr + 16 * t4) = acc * w;
            }
            if (t + 1 < ntile) {
#pragma unroll
                for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
                w = w_n;
            }
        }
    }
}

// out[i][c] = epilogue(sum over q in [seg_ptr[i], seg_ptr[i + 1]) of in[seg_idx ? seg_idx[q] : q][c]): G = ld4 lanes per
// segment (one 16-byte piece each), 64 / G segments per wave; the rows of a segment are added in index order.
// FINAL: + bias, activation, ReLU mask (the layer's output); else the plain sum (a level of a long segment's reduction).
struct EpSumArgs {
    const float* in;
    const int* seg_ptr;
    const int* seg_idx;
    const float* seg_w;      // optional weight of every summed row (position q): the pre-aggregation of heavy segments sums w_e x[src_e]
    const float* bias;
    const float* mask;
    float* out;
    int ldin, ldo, ldm, width, n_out, act, final_level;
};

template <int G>
__global__ void __launch_bounds__(256) rgcn_ep_segment_sum_kernel(const EpSumArgs a) {
    constexpr int SPW = 64 / G;                 // segments per wave
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long seg = wave * SPW + lane / G;
    const int piece = lane % G;
    if (seg >= a.n_out) return;
    const int q0 = a.seg_ptr[seg], q1 = a.seg_ptr[seg + 1];
    const bool col_in = 4 * piece < a.ldin;     // pieces beyond the row stride of `in` (narrow layers): nothing to read
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    auto rowp = [&](int q) { return a.in + (size_t)(a.seg_idx ? a.seg_idx[q] : q) * a.ldin + 4 * piece; };
    int q = q0;
    if (col_in && a.seg_w == nullptr) {
        // four rows in flight; the sum order ((r0 + r1) + (r2 + r3)) per group of four, groups in index order, is fixed
        for (; q + 4 <= q1; q += 4) {
            const f32x4 v0 = *(const f32x4*)rowp(q), v1 = *(const f32x4*)rowp(q + 1), v2 = *(const f32x4*)rowp(q + 2),
                        v3 = *(const f32x4*)rowp(q + 3);
            s0 += (v0 + v1) + (v2 + v3);
        }
        for (; q < q1; ++q) s1 += *(const f32x4*)rowp(q);
    } else if (col_in) {
        for (; q + 4 <= q1; q += 4) {
            const f32x4 v0 = *(const f32x4*)rowp(q) * a.seg_w[q], v1 = *(const f32x4*)rowp(q + 1) * a.seg_w[q + 1],
                        v2 = *(const f32x4*)rowp(q + 2) * a.seg_w[q + 2], v3 = *(const f32x4*)rowp(q + 3) * a.seg_w[q + 3];
            s0 += (v0 + v1) + (v2 + v3);
        }
        for (; q < q1; ++q) s1 += *(const f32x4*)rowp(q) * a.seg_w[q];
    }
    f32x4 v = s0 + s1;
    (void)s2; (void)s3;
    if (4 * piece >= a.ldo) return;
    if (a.final_level) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = 4 * piece + c;
            if (a.bias != nullptr && col < a.width) v[c] += a.bias[col];
            if (a.act == RGCN_ACT_RELU) v[c] = v[c] > 0.f ? v[c] : 0.f;
            else if (a.act == RGCN_ACT_SIGMOID) v[c] = col < a.width ? 1.f / (1.f + expf(-v[c])) : 0.f;
        }
        if (a.mask != nullptr) {
            const f32x4 m = *(const f32x4*)(a.mask + (size_t)seg * a.ldm + 4 * piece);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = m[c] > 0.f ? v[c] : 0.f;
        }
    }
    *(f32x4*)(a.out + (size_t)seg * a.ldo + 4 * piece) = v;
}

template <int KP>
static int ep_dispatch_np(int NP, const EpArgs& a, int blocks, hipStream_t s) {
    switch (NP) {
        case 16: hipLaunchKernelGGL((rgcn_ep_transform_kernel<KP, 16>), dim3(blocks), dim3(256), 0, s, a); break;
        case 32: hipLaunchKernelGGL((rgcn_ep_transform_kernel<KP, 32>), dim3(blocks), dim3(256), 0, s, a); break;
        case 64: hipLaunchKernelGGL((rgcn_ep_transform_kernel<KP, 64>), dim3(blocks), dim3(256), 0, s, a); break;
        case 128: hipLaunchKernelGGL((rgcn_ep_transform_kernel<KP, 128>), dim3(blocks), dim3(256), 0, s, a); break;
        default: return RGCN_ERR_WIDTH;
    }
    return (int)hipGetLastError();
}

}  // namespace rgcn

using namespace rgcn;

extern "C" int rgcn_ep_transform(const rgcn_edge_units_t* units, const float* x, int ldx, int din, const float* packed_w,
                                 float* z, int ldz, int dout, unsigned flags, void* stream) {
    if (!units || !x || !packed_w || !z) return RGCN_ERR_NULL;
    if (!units->unit_rel || !units->unit_cnt || !units->slot_src || !units->slot_w) return RGCN_ERR_NULL;
    if (units->n_units < 0 || units->n_nodes <= 0 || units->num_relations <= 0) return RGCN_ERR_PLAN;
    int st;
    if ((st = check_stride(ldx, din)) != RGCN_OK) return st;
    if ((st = check_stride(ldz, dout)) != RGCN_OK) return st;
    if ((st = check_device()) != RGCN_OK) return st;
    if (units->n_units == 0) return RGCN_OK;
    const int KP = padded_width(din), NP = padded_width(dout);
    EpArgs a;
    a.unit_rel = units->unit_rel;
    a.unit_cnt = units->unit_cnt;
    a.slot_src = units->slot_src;
    a.slot_w = units->slot_w;
    a.x = x;
    a.wp = packed_w;
    a.z = z;
    a.x_bytes = buffer_bytes(units->n_nodes, ldx, flags);
    a.n_rows = units->n_nodes;
    a.ldx = ldx;
    a.din4 = (din + 3) / 4;
    a.ldz = ldz;
    a.n_units = units->n_units;
    // enough waves to fill the chip (256 CUs x 16 waves) before a wave takes several units; consecutive units share a
    // relation, so a longer range reloads the weight fragments less often
    constexpr int kWaves = 256 * 16;
    a.units_per_wave = (units->n_units + kWaves - 1) / kWaves;
    const int waves = (units->n_units + a.units_per_wave - 1) / a.units_per_wave;
    const int blocks = (waves + 3) / 4;
    hipStream_t s = (hipStream_t)stream;
    // 64 x 64 on graphs large enough to be bound by the contraction: bf16 x 3 (fp32-equivalent) unless the caller pins exact fp32
    if (KP == 64 && NP == 64 && a.x_bytes != 0 && (flags & RGCN_FLAG_SPLIT_PRODUCERS) && !(flags & RGCN_FLAG_EXACT_FP32)) {
        a.wp = packed_w + (size_t)(units->num_relations + 1) * KP * NP;        // the bf16 planes behind the fp32 fragments
        hipLaunchKernelGGL(rgcn_ep_transform3_kernel, dim3(blocks), dim3(256), 0, s, a);
        return (int)hipGetLastError();
    }
    switch (KP) {
        case 16: return ep_dispatch_np<16>(NP, a, blocks, s);
        case 32: return ep_dispatch_np<32>(NP, a, blocks, s);
        case 64: return ep_dispatch_np<64>(NP, a, blocks, s);
        case 128: return ep_dispatch_np<128>(NP, a, blocks, s);
    }
    return RGCN_ERR_WIDTH;
}

extern "C" int rgcn_ep_segment_sum(const float* in, int ldin, const int32_t* seg_ptr, const int32_t* seg_idx, const float* seg_w,
                                   int n_out, int width, const float* bias, int act, const float* mask, int ldm, int final_level,
                                   float* out, int ldo, void* stream) {
    if (!in || !seg_ptr || !out) return RGCN_ERR_NULL;
    if (n_out < 0) return RGCN_ERR_PLAN;
    int st;
    if ((st = check_stride(ldin, width)) != RGCN_OK) return st;
    if ((st = check_stride(ldo, width)) != RGCN_OK) return st;
    if (mask != nullptr && (st = check_stride(ldm, width)) != RGCN_OK) return st;
    if (act != RGCN_ACT_NONE && act != RGCN_ACT_RELU && act != RGCN_ACT_SIGMOID) return RGCN_ERR_ACT;
    if ((st = check_device()) != RGCN_OK) return st;
    if (n_out == 0) return RGCN_OK;
    EpSumArgs a;
    a.in = in;
    a.seg_ptr = seg_ptr;
    a.seg_idx = seg_idx;
    a.seg_w = seg_w;
    a.bias = bias;
    a.mask = mask;
    a.out = out;
    a.ldin = ldin;
    a.ldo = ldo;
    a.ldm = ldm;
    a.width = width;
    a.n_out = n_out;
    a.act = act;
    a.final_level = final_level;
    const int ld4 = (ldo > ldin ? ldo : ldin) / 4;
    const int G = ld4 <= 4 ? 4 : (ld4 <= 8 ? 8 : (ld4 <= 16 ? 16 : 32));
    const long waves = ((long)n_out + 64 / G - 1) / (64 / G);
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
    switch (G) {
        case 4: hipLaunchKernelGGL(rgcn_ep_segment_sum_kernel<4>, dim3(blocks), dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL(rgcn_ep_segment_sum_kernel<8>, dim3(blocks), dim3(256), 0, s, a); break;
        case 16: hipLaunchKernelGGL(rgcn_ep_segment_sum_kernel<16>, dim3(blocks), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(rgcn_ep_segment_sum_kernel<32>, dim3(blocks), dim3(256), 0, s, a); break;
    }
    return (int)hipGetLastError();
}
