// rgcn_plan.hip -- the graph plan built ON THE DEVICE behind the C ABI (include/rgcn_mi355x.h:
// rgcn_edge_weights, rgcn_plan_build_begin / _finish).  gfx950 only.
//
// Input is exactly what the reference hands to the layer: the int64 COO of /root/reference/graphs/graph.py:55-69
// (edge_index rows are views of a transposed [E, 3] tensor, i.e. STRIDED; unsorted; duplicate triples kept).
// Output is the slot / chunk / tile layout the hot kernels walk (scaling_rgcn_training_amd/plan.py documents it and
// remains its test oracle: every integer array must be bit-identical, tests/test_gpu_plan_build.py).
//
//   keys      (tile, relation, row in tile, gathered node) packed into one 64-bit word per owned edge / root pseudo edge
//   sort      LSD radix sort, 8-bit digits, stable: per wave-segment digit histograms -> one exclusive scan -> scatter
//             with wave-level multi-split ranking (ballots), no cross-wave traffic inside a pass
//   merge     duplicate triples -> one slot, weights summed in float64
//   groups    (tile, relation) runs -> chunk ranges; row j of a group goes to MFMA row tile j mod nt, place j div nt
//   slots     slot_src / slot_w / slot_row / slot_acc (run-sum metadata per 16-slot row tile), chunk_* , tile_ptr
//   walk      rel_order = 64-slot units, chunks stably re-sorted by relation
//
// All byte / integer work: bound by HBM traffic (about 35 B per key and sort pass), nothing here touches MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/rgcn_mi355x.h"

namespace rgcn_planner {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int kSortItems = 32;                 // keys per lane of one wave segment
constexpr int kSegKeys = 64 * kSortItems;      // 2,048 keys per wave segment
constexpr int kSortThreads = 256;              // four independent wave segments per workgroup
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanBlock = kScanThreads * kScanItems;   // 2,048 elements per scan workgroup

__host__ __device__ inline int bits_for(u64 max_value) {   // bits needed to hold 0 .. max_value (at least 1)
    int b = 1;
    while ((max_value >> b) != 0 && b < 64) ++b;
    return b;
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of u32 (two levels: workgroup sums -> one workgroup scans the sums -> apply)
// ------------------------------------------------------------------------------------------------
__device__ inline u32 block_exclusive_scan(u32 v, u32* lds_wave_tot, u32& block_total) {
    // v: this thread's value; returns the exclusive prefix over the 256 threads of the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) lds_wave_tot[wave] = inc;
    __syncthreads();
    u32 before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        const u32 t = lds_wave_tot[w];
        if (w < wave) before += t;
        tot += t;
    }
    __syncthreads();
    block_total = tot;
    return before + inc - v;
}

__global__ void scan_reduce_kernel(const u32* __restrict__ in, u32 n, u32* __restrict__ sums) {
    __shared__ u32 wt[kScanThreads / 64];
    const size_t base = (size_t)blockIdx.x * kScanBlock + (size_t)threadIdx.x * kScanItems;
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j)
        if (base + j < n) s += in[base + j];
    u32 tot;
    block_exclusive_scan(s, wt, tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// one workgroup: sums[0 .. m) -> exclusive scan in place, grand total -> sums[m]
__global__ void scan_top_kernel(u32* __restrict__ sums, u32 m) {
    __shared__ u32 wt[kScanThreads / 64];
    u32 carry = 0;
    for (u32 c0 = 0; c0 < m; c0 += kScanBlock) {
        const u32 base = c0 + threadIdx.x * kScanItems;
        u32 v[kScanItems], s = 0;
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {
            v[j] = base + j < m ? sums[base + j] : 0u;
            s += v[j];
        }
        u32 tot;
        u32 ex = block_exclusive_scan(s, wt, tot) + carry;
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {
            if (base + j < m) sums[base + j] = ex;
            ex += v[j];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) sums[m] = carry;
}

__global__ void scan_apply_kernel(const u32* __restrict__ in, u32* __restrict__ out, u32 n, const u32* __restrict__ sums) {
    __shared__ u32 wt[kScanThreads / 64];
    const size_t base = (size_t)blockIdx.x * kScanBlock + (size_t)threadIdx.x * kScanItems;
    u32 v[kScanItems], s = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        v[j] = base + j < n ? in[base + j] : 0u;
        s += v[j];
    }
    u32 tot;
    u32 ex = block_exclusive_scan(s, wt, tot) + sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        if (base + j < n) out[base + j] = ex;
        ex += v[j];
    }
}

// out[i] = sum of in[0 .. i); the grand total lands in sums[nblocks] (device).  in may equal out.
static u32 scan_blocks(u32 n) { return (n + kScanBlock - 1) / kScanBlock; }
static void exclusive_scan(const u32* in, u32* out, u32 n, u32* sums, hipStream_t s) {
    const u32 nb = scan_blocks(n);
    if (nb == 0) {
        (void)hipMemsetAsync(sums, 0, sizeof(u32), s);
        return;
    }
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(kScanThreads), 0, s, in, n, sums);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(kScanThreads), 0, s, sums, nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(kScanThreads), 0, s, in, out, n, sums);
}

// ------------------------------------------------------------------------------------------------
// stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass
// ------------------------------------------------------------------------------------------------
__device__ inline u32 digit_of(u64 k, int shift) { return (u32)(k >> shift) & 255u; }

// hist[d * nseg + seg] = keys of wave segment `seg` whose digit is d
__global__ void radix_hist_kernel(const u64* __restrict__ keys, u32 n, int shift, u32 nseg, u32* __restrict__ hist) {
    __shared__ u32 cnt[kSortThreads / 64][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 seg = blockIdx.x * (kSortThreads / 64) + wave;
    for (int d = lane; d < 256; d += 64) cnt[wave][d] = 0;
    __syncthreads();
    if (seg < nseg) {
        const size_t base = (size_t)seg * kSegKeys;
        for (int r = 0; r < kSortItems; ++r) {
            const size_t i = base + (size_t)r * 64 + lane;
            if (i < n) atomicAdd(&cnt[wave][digit_of(keys[i], shift)], 1u);
        }
    }
    __syncthreads();
    if (seg < nseg)
        for (int d = lane; d < 256; d += 64) hist[(size_t)d * nseg + seg] = cnt[wave][d];
}

// offs = exclusive scan of hist in memory order (digit-major, segment-minor): where this segment's keys of digit d
// start in the output.  A wave ranks its 64 keys of a round by wave-level multi-split: eight ballots give every lane
// the set of lanes holding the same digit; its rank among them keeps the input order (stable), the first of them
// advances the segment's running offset of that digit in LDS.
__global__ void radix_scatter_kernel(const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout,
                                     u32* __restrict__ vout, u32 n, int shift, u32 nseg, const u32* __restrict__ offs) {
    __shared__ u32 cnt_s[kSortThreads / 64][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 seg = blockIdx.x * (kSortThreads / 64) + wave;
    volatile u32* cnt = cnt_s[wave];
    if (seg < nseg)
        for (int d = lane; d < 256; d += 64) cnt[d] = offs[(size_t)d * nseg + seg];
    __syncthreads();
    if (seg >= nseg) return;
    const size_t base = (size_t)seg * kSegKeys;
    const u64 below = (1ull << lane) - 1ull;
    for (int r = 0; r < kSortItems; ++r) {
        const size_t i = base + (size_t)r * 64 + lane;
        const bool valid = i < n;
        const u64 k = valid ? kin[i] : 0ull;
        const u32 v = valid ? vin[i] : 0u;
        const u32 dg = digit_of(k, shift);
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (dg >> b) & 1u;
            const u64 m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const u32 rank = (u32)__popcll(peers & below);
        const int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
        u32 off = 0;
        if (valid && lane == leader) {
            off = cnt[dg];
            cnt[dg] = off + (u32)__popcll(peers);
        }
        __builtin_amdgcn_wave_barrier();
        off = __shfl(off, leader);
        if (valid) {
            kout[(size_t)off + rank] = k;
            vout[(size_t)off + rank] = v;
        }
    }
}

struct SortBufs {
    u64* k[2];
    u32* v[2];
    u32* hist;   // 256 * nseg + 1
    u32* sums;   // scan_blocks(256 * nseg) + 1
};

static u32 sort_segments(u32 n) { return (n + kSegKeys - 1) / kSegKeys; }

// sorts by the low `bits` bits of the key; data starts in (k[0], v[0]); returns the index of the pair holding the result
static int radix_sort_pairs(const SortBufs& b, u32 n, int bits, hipStream_t s) {
    int cur = 0;
    if (n <= 1) return cur;
    const u32 nseg = sort_segments(n);
    const u32 nblk = (nseg + kSortThreads / 64 - 1) / (kSortThreads / 64);
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk), dim3(kSortThreads), 0, s, b.k[cur], n, shift, nseg, b.hist);
        exclusive_scan(b.hist, b.hist, 256u * nseg, b.sums, s);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk), dim3(kSortThreads), 0, s, b.k[cur], b.v[cur], b.k[cur ^ 1],
                           b.v[cur ^ 1], n, shift, nseg, b.hist);
        cur ^= 1;
    }
    return cur;
}

// ------------------------------------------------------------------------------------------------
// keys
// ------------------------------------------------------------------------------------------------
struct KeyLayout {
    int src_bits, dstl_bits, rel_bits, tile_bits;
    __host__ __device__ int gshift() const { return src_bits + dstl_bits; }
    __host__ __device__ int total() const { return src_bits + dstl_bits + rel_bits + tile_bits; }
    __host__ __device__ u64 pack(u32 tile_id, u32 rel, u32 dstl, u32 g) const {
        return ((((((u64)tile_id << rel_bits) | rel) << dstl_bits) | dstl) << src_bits) | g;
    }
};

struct Counters {       // device-resident scalars of one build
    u32 owned_edges;    // edges whose scatter node lies in the owned range
    u32 error;          // bit 0: edge_type out of range, bit 1: node id out of range
};

__global__ void root_keys_kernel(u64* __restrict__ keys, u32* __restrict__ vals, u32 n_own, u32 node_begin, u32 tile,
                                 u32 num_rel, KeyLayout kl) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_own) return;
    keys[i] = kl.pack(i / tile, num_rel, i % tile, node_begin + i);       // node gathers its own row under relation R'
    vals[i] = __float_as_uint(1.0f);
}

// one key per owned edge, appended behind the n_own root keys in arrival order (the sort makes the order irrelevant)
__global__ void edge_keys_kernel(const int64_t* __restrict__ src, int64_t src_stride, const int64_t* __restrict__ dst,
                                 int64_t dst_stride, const int64_t* __restrict__ typ, int64_t typ_stride,
                                 const float* __restrict__ w, u64 num_edges, int transposed, u32 n_nodes, u32 num_rel,
                                 u32 node_begin, u32 node_end, u32 tile, KeyLayout kl, u32 n_own, u64* __restrict__ keys,
                                 u32* __restrict__ vals, Counters* __restrict__ ctr) {
    // A wave takes kBatch x 64 consecutive edges, keeps their keys in registers and reserves room for all of them with ONE
    // atomic on the shared counter (one atomic per 64 edges was 1.6M serialised atomics on one address at 100M edges: 17.8 ms of
    // a kernel that moves 4 GB)
    constexpr int kBatch = 8;
    const int lane = threadIdx.x & 63;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 e0 = wave * (kBatch * 64); e0 < num_edges; e0 += waves * (kBatch * 64)) {
        u64 key[kBatch];
        u32 val[kBatch];
        u64 mask[kBatch];
        u32 total = 0;
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const u64 e = e0 + (u64)k * 64 + lane;
            bool own = false;
            key[k] = 0;
            val[k] = 0;
            if (e < num_edges) {
                const int64_t sv = src[e * src_stride], d = dst[e * dst_stride], t = typ[e * typ_stride];
                u32 err = 0;
                if (t < 0 || t >= (int64_t)num_rel) err |= 1u;
                if (sv < 0 || sv >= (int64_t)n_nodes || d < 0 || d >= (int64_t)n_nodes) err |= 2u;
                if (err) {
                    atomicOr(&ctr->error, err);
                } else {
                    const u32 g = (u32)(transposed ? d : sv), sc = (u32)(transposed ? sv : d);
                    if (sc >= node_begin && sc < node_end) {
                        const u32 loc = sc - node_begin;
                        own = true;
                        key[k] = kl.pack(loc / tile, (u32)t, loc % tile, g);
                        val[k] = __float_as_uint(w[e]);
                    }
                }
            }
            mask[k] = __ballot(own);
            total += (u32)__popcll(mask[k]);
        }
        if (total == 0) continue;
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&ctr->owned_edges, total);
        base = __shfl(base, 0);
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            if ((mask[k] >> lane) & 1ull) {
                const u32 pos = n_own + base + (u32)__popcll(mask[k] & ((1ull << lane) - 1ull));
                keys[pos] = key[k];
                vals[pos] = val[k];
            }
            base += (u32)__popcll(mask[k]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// duplicate merge and groups
// ------------------------------------------------------------------------------------------------
// flag[i] = 1 where element i starts a run of equal (key >> shift)
__global__ void head_flags_kernel(const u64* __restrict__ keys, u32 n, int shift, u32* __restrict__ flag) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || (keys[i] >> shift) != (keys[i - 1] >> shift)) ? 1u : 0u;
}

// ex = exclusive scan of the head flags: for the head of a run that is the run's index, for the other elements of the
// run it is the index + 1 (their head is already counted).  id[i] = ex[i] + flag[i] - 1 is the run index of EVERY element.
__global__ void run_ids_kernel(u32* __restrict__ ex, const u32* __restrict__ flag, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ex[i] = ex[i] + flag[i] - 1u;
}

// duplicate (gather, scatter, relation) triples share ONE slot whose weight is the float64 sum of theirs
__global__ void merge_kernel(const u64* __restrict__ keys, const u32* __restrict__ vals, u32 n, const u32* __restrict__ uidx,
                             u64* __restrict__ ukeys, u32* __restrict__ uvals) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 k = keys[i];
    if (i != 0 && keys[i - 1] == k) return;
    double acc = (double)__uint_as_float(vals[i]);
    u32 j = i + 1;
    bool dup = false;
    while (j < n && keys[j] == k) {
        acc += (double)__uint_as_float(vals[j]);
        dup = true;
        ++j;
    }
    const u32 u = uidx[i];
    ukeys[u] = k;
    uvals[u] = dup ? __float_as_uint((float)acc) : vals[i];
}

// group = run of equal (tile, relation); gid[i] = exclusive scan of the group head flags
__global__ void group_start_kernel(const u64* __restrict__ ukeys, u32 n, int gshift, const u32* __restrict__ gid,
                                   u32* __restrict__ gstart, u32* __restrict__ gkey, u32 n_groups) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) gstart[n_groups] = n;
    if (i >= n) return;
    const u32 gk = (u32)(ukeys[i] >> gshift);
    if (i == 0 || (u32)(ukeys[i - 1] >> gshift) != gk) {
        gstart[gid[i]] = i;
        gkey[gid[i]] = gk;
    }
}

// chunks and 64-slot units of every group.  cap: row tiles a chunk may hold (chunk / 16, or 7 for the 112-row chunks of
// rgcn_tile3p_kernel's larger tiles: the slot stride stays 128, the eighth row tile of a chunk stays free for shadow rows)
__global__ void group_sizes_kernel(const u32* __restrict__ gstart, u32 n_groups, u32 chunk, u32 cap, u32 layout, u32* __restrict__ gch,
                                   u32* __restrict__ gun) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const u32 cnt = gstart[g + 1] - gstart[g];
    const u32 nt = (cnt + 15u) / 16u;
    (void)layout;
    if (cap * 16u == chunk) {
        gch[g] = (cnt + chunk - 1u) / chunk;
        // both layouts put the rows of a chunk on contiguous row tiles from tile 0 and use ceil(cnt / 16) tiles per group
        gun[g] = (nt + 3u) / 4u;
    } else {      // layout 0 dealing over chunks of `cap` row tiles
        const u32 full = nt / cap, rem = nt % cap;
        gch[g] = full + (rem ? 1u : 0u);
        gun[g] = full * ((cap + 3u) / 4u) + (rem + 3u) / 4u;
    }
}

// ------------------------------------------------------------------------------------------------
// slots / chunks / tiles
// ------------------------------------------------------------------------------------------------
// `dstl` (row inside the tile of every slot) is scratch that lives in the caller's slot_row array until
// row_tile_kernel turns it into the final row ids: its size is the plan's slot count, which no workspace query can bound
__global__ void fill_slots_kernel(u32 n_slots, u32 n_nodes, u32 tile, int32_t* __restrict__ slot_src, float* __restrict__ slot_w,
                                  u32* __restrict__ dstl) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    slot_src[i] = (int32_t)n_nodes;     // padding gathers the row one past the end: a buffer-descriptor load returns zeros
    slot_w[i] = 0.f;
    dstl[i] = tile;                     // padding scatters into the dummy accumulator row
}

// Layout 0: row j of a group (sorted by destination) goes to MFMA row tile (j mod nt), position (j div nt), nt = ceil(n / 16):
// every tile stays sorted by destination and a run of c equal destinations is spread over c different tiles.
// Layout 1 (plan.team_placement): chunk c of the group takes its rows [128 c, 128 c + 128) on nt = ceil(n_c / 16) row tiles, cut
// at split[chunk] into part A (dealt over the first ceil(nt / 2) row tiles) and part B (the others): disjoint destinations.
__global__ void place_kernel(const u64* __restrict__ ukeys, const u32* __restrict__ uvals, u32 n, const u32* __restrict__ gid,
                             const u32* __restrict__ gstart, const u32* __restrict__ chunk_base, u32 chunk, u32 cap, u32 layout,
                             const u32* __restrict__ split, KeyLayout kl, int32_t* __restrict__ slot_src,
                             float* __restrict__ slot_w, u32* __restrict__ dstl) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 g = gid[i];
    const u32 s0 = gstart[g], cnt = gstart[g + 1] - s0;
    const u32 rank = i - s0;
    size_t slot;
    if (layout == 0) {
        const u32 nt = (cnt + 15u) / 16u, t = rank % nt;      // the group's row tile t: chunk t / cap of the group, its row tile t % cap
        slot = (size_t)(chunk_base[g] + t / cap) * chunk + (size_t)(t % cap) * 16u + rank / nt;
    } else {
        const u32 cidx = rank / 128u, jc = rank % 128u;
        const u32 left = cnt - cidx * 128u, n_c = left < 128u ? left : 128u;
        const u32 c = chunk_base[g] + cidx;
        const u32 nt = (n_c + 15u) / 16u, na = (nt + 1u) / 2u, nb = nt - na;
        const u32 sp = split[c];
        u32 in_chunk;
        if (jc < sp) {
            in_chunk = (jc % na) * 16u + jc / na;
        } else {
            const u32 j1 = jc - sp;
            in_chunk = (na + j1 % nb) * 16u + j1 / nb;
        }
        slot = (size_t)c * 128u + in_chunk;
    }
    const u64 k = ukeys[i];
    slot_src[slot] = (int32_t)(k & ((1ull << kl.src_bits) - 1ull));
    slot_w[slot] = __uint_as_float(uvals[i]);
    dstl[slot] = (u32)(k >> kl.src_bits) & ((1u << kl.dstl_bits) - 1u);
}

__global__ void chunk_meta_kernel(const u32* __restrict__ chunk_base, const u32* __restrict__ gstart, const u32* __restrict__ gkey,
                                  u32 n_groups, u32 n_chunks, u32 chunk, u32 cap, u32 layout, const u64* __restrict__ ukeys, KeyLayout kl,
                                  u32* __restrict__ split, int32_t* __restrict__ chunk_rel, int32_t* __restrict__ chunk_cnt,
                                  int32_t* __restrict__ chunk_tile, int32_t* __restrict__ chunk_flags) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    // the group whose chunk range holds c: last g with chunk_base[g] <= c
    u32 lo = 0, hi = n_groups;
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (chunk_base[mid] <= c) lo = mid; else hi = mid;
    }
    const u32 g = lo;
    const u32 cnt = gstart[g + 1] - gstart[g];
    const u32 idx = c - chunk_base[g];
    int32_t flags = 0;
    if (layout == 0) {
        const u32 nt = (cnt + 15u) / 16u, per = cap;
        const u32 left = nt - idx * per;
        chunk_cnt[c] = (int32_t)((left < per ? left : per) * 16u);
    } else {
        const u32 left = cnt - idx * 128u, n_c = left < 128u ? left : 128u;
        const u32 nt = (n_c + 15u) / 16u, na = (nt + 1u) / 2u, nb = nt - na;
        chunk_cnt[c] = (int32_t)(nt * 16u);
        if (nt <= 1u) {
            split[c] = n_c;
        } else {
            // cut at the run boundary closest to the middle of [max(1, n_c - 16 nb), min(n_c - 1, 16 na)]: mid, mid - 1, mid + 1, ...
            const u64* rows = ukeys + gstart[g] + (size_t)idx * 128u;
            const u64 dmask = (1ull << kl.dstl_bits) - 1ull;
            auto dst_of = [&](u32 j) { return (rows[j] >> kl.src_bits) & dmask; };
            const int lo_s = (int)n_c - 16 * (int)nb > 1 ? (int)n_c - 16 * (int)nb : 1;
            const int hi_s = (int)n_c - 1 < 16 * (int)na ? (int)n_c - 1 : 16 * (int)na;
            const int mid = (lo_s + hi_s + 1) / 2;
            int sp = -1;
            for (int k = 0; k <= 16 && sp < 0; ++k) {
                const int a = mid - k, b = mid + k;
                if (a >= lo_s && a <= hi_s && dst_of((u32)a - 1u) != dst_of((u32)a)) sp = a;
                else if (k > 0 && b >= lo_s && b <= hi_s && dst_of((u32)b - 1u) != dst_of((u32)b)) sp = b;
            }
            if (sp < 0) {      // one destination's run covers the whole window: the parts share it
                sp = mid;
                flags = 256;
            }
            split[c] = (u32)sp;
        }
    }
    chunk_rel[c] = (int32_t)(gkey[g] & ((1u << kl.rel_bits) - 1u));
    chunk_tile[c] = (int32_t)(gkey[g] >> kl.rel_bits);
    chunk_flags[c] = flags;
}

// per 16-slot MFMA row tile: run metadata of the forward kernel's run-sum (plan.run_metadata), the dW kernels' row ids
// (dstl aliases slot_row: every thread reads its 16 entries before it overwrites them)
__global__ void row_tile_kernel(const u32* dstl, u32 n_row_tiles, u32 chunk, u32 tile, u32 n_own,
                                const int32_t* __restrict__ chunk_tile, int32_t* __restrict__ slot_acc, int32_t* slot_row,
                                int32_t* __restrict__ chunk_flags) {
    const u32 rt = blockIdx.x * blockDim.x + threadIdx.x;
    if (rt >= n_row_tiles) return;
    const u32 per = chunk / 16u, c = rt / per;
    const u32 tbase = (u32)chunk_tile[c] * tile;
    u32 d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = dstl[(size_t)rt * 16u + i];
    u32 runend = 16;
    bool dup = false;
    int32_t acc[16];
#pragma unroll
    for (int i = 15; i >= 0; --i) {
        const bool is_end = (i == 15) || d[i] != d[i + 1];
        if (is_end) runend = (u32)i;
        if (!is_end && d[i] != tile) dup = true;        // runs of padding slots do not count
        acc[i] = (int32_t)((runend << 24) | (is_end ? d[i] : tile));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        slot_acc[(size_t)rt * 16u + i] = acc[i];
        slot_row[(size_t)rt * 16u + i] = (int32_t)(d[i] < tile ? tbase + d[i] : n_own);
    }
    if (dup) atomicOr((int*)&chunk_flags[c], 1 << (rt % per));
}

// Layout 3 = layout 0 + this pass (plan.compact_runs is its twin).  The rows of a (destination, relation) run all go through the same
// W_r into the same output row: sum_k w_k x[src_k] W = (sum_k w_k x[src_k]) W -- the producers of rgcn_tile3p_kernel add the run's
// rows in fp32 BEFORE they cut them (aggregate, then transform: the reference's own order) and the run takes ONE slot.  Chunk-local
// and only where it is simple: chunks that are a whole (tile, relation) group, runs of at most 3 rows, at most 32 runs of 2+ rows
// and 16 of 3; anything else keeps its layout-0 slots (always correct: the kernel reads the shadow counts per chunk).  New chunk:
// heads (first row of every run; runs of 3 first, then of 2, then single rows, each class in destination order) on slots 0 .. H-1;
// the second row of head h on row tile 7 - h / 16, place h % 16; the third on the row tile right below the second rows' (7 - ns1:
// row tile 6 with one tile of second rows, 5 with two; round 3 had it fixed at 5, which shut out every chunk of six head row
// tiles with a run of three -- most chunks at tiles above 224), place h -- the SAME lane of the same
// producer wave holds a head and its shadows, one row tile register apart.  chunk_cnt = 16 ceil(H / 16); chunk_flags bits 20-23 =
// the chunk's row tiles (every chunk of the plan, compacted or not), bits 16-17 =
// row tiles with second rows, bit 18 = a row tile with third rows, bit 19 = some run's rows differ in weight (the transposed
// plan: 1 / c of each edge's own destination): the producers then scale a shadow row by (its weight / its head's weight) -- the
// float in the shadow slot's slot_acc -- before they add it, and the consumers apply the head's weight as ever.  Every slot
// keeps its own weight and its run's row (a walk over all slots with a weight -- tests/plan_emulator.py -- still sums the layer).
constexpr int kCompactThreads = 32;
__global__ void __launch_bounds__(kCompactThreads) compact_runs_kernel(u32 n_chunks, u32 n_nodes, u32 tile, u32 n_own, const int32_t* __restrict__ chunk_rel,
                                    const int32_t* __restrict__ chunk_tile, int32_t* __restrict__ chunk_cnt,
                                    int32_t* __restrict__ chunk_flags, int32_t* __restrict__ slot_src, float* __restrict__ slot_w,
                                    int32_t* __restrict__ slot_row, int32_t* __restrict__ slot_acc) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const int32_t rel = chunk_rel[c], t = chunk_tile[c];
    const u32 nt = (u32)chunk_cnt[c] / 16u;
    chunk_flags[c] |= (int32_t)(nt << 20);      // every chunk: its row tiles beside its flags (one scalar word for the producers)
    if (c > 0 && chunk_rel[c - 1] == rel && chunk_tile[c - 1] == t) return;
    if (c + 1 < n_chunks && chunk_rel[c + 1] == rel && chunk_tile[c + 1] == t) return;
    if (nt == 0) return;
    const size_t base = (size_t)c * 128u;
    // the chunk's rows in sorted order: a copy per thread in LDS (32 threads x 3 x 129 words; the odd stride keeps the threads on
    // different banks) -- as private arrays they lived in scratch memory: 67 GB of traffic for a 100M-edge plan
    __shared__ int32_t s_src[kCompactThreads][129], s_d[kCompactThreads][129];
    __shared__ u32 s_w[kCompactThreads][129];
    int32_t* lsrc = s_src[threadIdx.x];
    int32_t* ld = s_d[threadIdx.x];
    u32* lw = s_w[threadIdx.x];
    u32 n = 0;
    const u32 tbase = (u32)t * tile;
    for (u32 j = 0; j < nt * 16u; ++j) {
        const size_t sl = base + (size_t)(j % nt) * 16u + j / nt;
        const int32_t sv = slot_src[sl];
        if ((u32)sv == n_nodes) break;
        lsrc[n] = sv;
        ld[n] = slot_row[sl] - (int32_t)tbase;
        lw[n] = __float_as_uint(slot_w[sl]);
        ++n;
    }
    // runs
    u32 n1 = 0, n2 = 0, n3 = 0, uneq = 0;
    for (u32 j = 0; j < n;) {
        u32 len = 1;
        while (j + len < n && ld[j + len] == ld[j]) {
            if (lw[j + len] != lw[j]) uneq = 1u;
            ++len;
        }
        if (len > 3) return;
        if (len == 1) ++n1; else if (len == 2) ++n2; else ++n3;
        j += len;
    }
    if (n2 + n3 == 0 || n3 > 16u || n2 + n3 > 32u) return;
    const u32 H = n1 + n2 + n3, nh = (H + 15u) / 16u, ns1 = (n2 + n3 + 15u) / 16u, ns2 = n3 > 0 ? 1u : 0u;
    if (nh >= nt || nh + ns1 + ns2 > 8u) return;
    // clear the chunk, then write heads and shadows
    for (u32 i = 0; i < 128u; ++i) {
        slot_src[base + i] = (int32_t)n_nodes;
        slot_w[base + i] = 0.f;
        slot_row[base + i] = (int32_t)n_own;
        slot_acc[base + i] = (int32_t)((15u << 24) | tile);
    }
    u32 h3 = 0, h2 = n3, h1 = n3 + n2;
    for (u32 j = 0; j < n;) {
        u32 len = 1;
        while (j + len < n && ld[j + len] == ld[j]) ++len;
        const u32 h = len == 3 ? h3++ : (len == 2 ? h2++ : h1++);
        const int32_t row = (int32_t)(tbase + (u32)ld[j]);
        slot_src[base + h] = lsrc[j];
        slot_w[base + h] = __uint_as_float(lw[j]);
        slot_row[base + h] = row;
        // distinct destinations inside a head tile: the run ends where it starts; padding behind the last head keeps (15, tile)
        slot_acc[base + h] = (int32_t)(((h & 15u) << 24) | (u32)ld[j]);
        if (len >= 2) {
            const size_t s1 = base + (size_t)(7u - h / 16u) * 16u + (h & 15u);
            slot_src[s1] = lsrc[j + 1];
            slot_w[s1] = __uint_as_float(lw[j + 1]);
            slot_row[s1] = row;
            slot_acc[s1] = (int32_t)__float_as_uint(__fdiv_rn(__uint_as_float(lw[j + 1]), __uint_as_float(lw[j])));
        }
        if (len == 3) {
            const size_t s2 = base + (size_t)(7u - ns1) * 16u + h;
            slot_src[s2] = lsrc[j + 2];
            slot_w[s2] = __uint_as_float(lw[j + 2]);
            slot_row[s2] = row;
            slot_acc[s2] = (int32_t)__float_as_uint(__fdiv_rn(__uint_as_float(lw[j + 2]), __uint_as_float(lw[j])));
        }
        j += len;
    }
    chunk_cnt[c] = (int32_t)(nh * 16u);
    chunk_flags[c] = (int32_t)((ns1 << 16) | (ns2 << 18) | (uneq << 19) | (nh << 20));
}

// Layout 5 = layout 0 (64-slot chunks: the tile-major weight-gradient kernel's units) + this pass (plan.dw_pairs is its twin): the
// weight gradient of a relation is a sum over slots of (w x[src])^T g[row], so two rows with ONE (destination, relation) and ONE
// weight -- a pair -- take ONE slot if x[src] + x[src2] is formed first.  Group-local, and only where it is simple: (tile, relation)
// groups of one or two chunks (128 rows: all but hubs).  The group's rows in sorted order are cut into runs of equal (row, weight),
// every run into pairs (+ one single row if its length is odd); heads = pairs + singles are dealt back over the group's chunks,
// 64 per unit, densely from slot 0.  The kernel's lane geometry wants the head of a pair on one of the first four slots of a
// 32-slot half (register 0 of the half's row pipeline): a unit of n heads has 4 such places if n <= 32, else 8; pairs beyond the
// group's places stay two single rows.  slot_src2[unit][8]: the second rows of slots 0..3 and 32..35 (padding: n_nodes).  Chunks
// the group no longer needs get chunk_cnt 0 and drop out of rel_order (the unit list is built after this pass).
constexpr int kPairThreads = 32;
__global__ void __launch_bounds__(kPairThreads) dw_pairs_kernel(u32 n_chunks, u32 n_nodes, u32 n_own, const int32_t* __restrict__ chunk_rel,
                                const int32_t* __restrict__ chunk_tile, int32_t* __restrict__ chunk_cnt, int32_t* __restrict__ slot_src,
                                float* __restrict__ slot_w, int32_t* __restrict__ slot_row, int32_t* __restrict__ slot_src2) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const int32_t rel = chunk_rel[c], t = chunk_tile[c];
    if (c > 0 && chunk_rel[c - 1] == rel && chunk_tile[c - 1] == t) return;      // not the group's first chunk
    u32 m = 1;
    while (c + m < n_chunks && chunk_rel[c + m] == rel && chunk_tile[c + m] == t) ++m;
    if (m > 2) return;                                                           // a hub's group: left as it is
    u32 nt = 0;
    for (u32 i = 0; i < m; ++i) nt += (u32)chunk_cnt[c + i] / 16u;
    if (nt == 0) return;
    __shared__ int32_t s_src[kPairThreads][129], s_row[kPairThreads][129];
    __shared__ u32 s_w[kPairThreads][129];
    int32_t* lsrc = s_src[threadIdx.x];
    int32_t* lrow = s_row[threadIdx.x];
    u32* lw = s_w[threadIdx.x];
    const size_t base = (size_t)c * 64u;
    u32 n = 0;
    for (u32 j = 0; j < nt * 16u; ++j) {      // the group's rows in sorted order: row j sits on row tile j mod nt at place j div nt
        const u32 tt = j % nt;
        const size_t sl = base + (size_t)(tt / 4u) * 64u + (size_t)(tt % 4u) * 16u + j / nt;
        const int32_t sv = slot_src[sl];
        if ((u32)sv == n_nodes) break;
        lsrc[n] = sv;
        lrow[n] = slot_row[sl];
        lw[n] = __float_as_uint(slot_w[sl]);
        ++n;
    }
    // pairs inside runs of equal (row, weight)
    u32 P = 0;
    for (u32 j = 0; j < n;) {
        u32 len = 1;
        while (j + len < n && lrow[j + len] == lrow[j] && lw[j + len] == lw[j]) ++len;
        P += len / 2u;
        j += len;
    }
    if (P == 0) return;
    // how many pairs the units have places for: demote pairs (two single heads each) until they fit.  H heads fill
    // U = ceil(H / 64) units densely from slot 0, 64 per unit; a unit of nu heads has min(4, nu) pair places on slots 0..3 and,
    // past 32 heads, min(4, nu - 32) more on slots 32..35
    u32 H = n - P;
    auto unit_heads = [&](u32 u) { return H - 64u * u < 64u ? H - 64u * u : 64u; };
    auto places = [&](u32 u) {
        const u32 nu = unit_heads(u);
        return (nu < 4u ? nu : 4u) + (nu > 32u ? (nu - 32u < 4u ? nu - 32u : 4u) : 0u);
    };
    for (;;) {
        const u32 U = (H + 63u) / 64u;
        u32 cap = 0;
        for (u32 u = 0; u < U; ++u) cap += places(u);
        if (P <= cap) break;
        --P;
        ++H;
    }
    const u32 U = (H + 63u) / 64u;          // <= 2: the group had at most 128 rows
    u32 pairs_in[2] = {0u, 0u};
    {
        u32 left = P;
        for (u32 u = 0; u < U; ++u) {
            pairs_in[u] = left < places(u) ? left : places(u);
            left -= pairs_in[u];
        }
    }
    // clear the group's chunks, then deal the heads in sorted order: the first P pairs met become pair heads on the pair places
    // (unit 0's first), every other row a single head on the next slot that is not a taken pair place
    for (u32 i = 0; i < m * 64u; ++i) {
        slot_src[base + i] = (int32_t)n_nodes;
        slot_w[base + i] = 0.f;
        slot_row[base + i] = (int32_t)n_own;
    }
    auto is_pair_place = [&](u32 u, u32 sl) {
        if (sl < 4u) return sl < pairs_in[u];
        if (sl >= 32u && sl < 36u) return sl - 28u < pairs_in[u];
        return false;
    };
    u32 pair_left = P, next_pair[2] = {0u, 0u}, next_single[2] = {0u, 0u}, single_u = 0;
    for (u32 j = 0; j < n;) {
        u32 len = 1;
        while (j + len < n && lrow[j + len] == lrow[j] && lw[j + len] == lw[j]) ++len;
        u32 k = 0;
        while (k < len) {
            const bool as_pair = k + 1 < len && pair_left > 0;
            u32 u, sl;
            if (as_pair) {
                u = next_pair[0] < pairs_in[0] ? 0u : 1u;
                const u32 kk = next_pair[u]++;
                sl = kk < 4u ? kk : 28u + kk;                  // 0..3, then 32..35
                --pair_left;
            } else {
                u = single_u;
                for (;;) {
                    while (next_single[u] < unit_heads(u) && is_pair_place(u, next_single[u])) ++next_single[u];
                    if (next_single[u] < unit_heads(u)) break;
                    single_u = ++u;
                }
                sl = next_single[u]++;
            }
            const size_t g = base + (size_t)u * 64u + sl;
            slot_src[g] = lsrc[j + k];
            slot_w[g] = __uint_as_float(lw[j + k]);
            slot_row[g] = lrow[j + k];
            if (as_pair) {
                slot_src2[(size_t)(c + u) * 8u + (sl < 4u ? sl : sl - 28u)] = lsrc[j + k + 1];
                k += 2;
            } else {
                k += 1;
            }
        }
        j += len;
    }
    for (u32 i = 0; i < m; ++i) chunk_cnt[c + i] = i < U ? (int32_t)(((unit_heads(i) + 15u) / 16u) * 16u) : 0;
}

__global__ void fill_i32_kernel(int32_t* __restrict__ p, u64 n, int32_t v) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void tile_ptr_kernel(const int32_t* __restrict__ chunk_tile, u32 n_chunks, u32 n_tiles, int32_t* __restrict__ tile_ptr) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    u32 lo = 0, hi = n_chunks;              // first chunk whose tile is >= t
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if ((u32)chunk_tile[mid] < t) lo = mid + 1; else hi = mid;
    }
    tile_ptr[t] = (int32_t)lo;
}

// ------------------------------------------------------------------------------------------------
// the weight-gradient walk
// ------------------------------------------------------------------------------------------------
__global__ void chunk_keys_kernel(const int32_t* __restrict__ chunk_rel, u32 n_chunks, u64* __restrict__ keys, u32* __restrict__ vals) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    keys[c] = (u64)(u32)chunk_rel[c];
    vals[c] = c;
}
__global__ void unit_counts_kernel(const u32* __restrict__ order, const int32_t* __restrict__ chunk_cnt, u32 n_chunks, u32* __restrict__ ucnt) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_chunks) return;
    ucnt[j] = ((u32)chunk_cnt[order[j]] + 63u) / 64u;
}
__global__ void emit_units_kernel(const u32* __restrict__ order, const int32_t* __restrict__ chunk_cnt, const u32* __restrict__ uoff,
                                  u32 n_chunks, u32 upc, int32_t* __restrict__ rel_order) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_chunks) return;
    const u32 c = order[j];
    const u32 nu = ((u32)chunk_cnt[c] + 63u) / 64u;
    for (u32 h = 0; h < nu; ++h) rel_order[uoff[j] + h] = (int32_t)(c * upc + h);
}

// ------------------------------------------------------------------------------------------------
// edge weights
// ------------------------------------------------------------------------------------------------
__global__ void weight_keys_kernel(const int64_t* __restrict__ dst, int64_t dst_stride, const int64_t* __restrict__ typ,
                                   int64_t typ_stride, u64 num_edges, u32 n_nodes, u32 num_rel, u64* __restrict__ keys,
                                   u32* __restrict__ vals, Counters* __restrict__ ctr) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= num_edges) return;
    const int64_t d = dst[e * dst_stride], t = typ[e * typ_stride];
    u32 err = 0;
    if (t < 0 || t >= (int64_t)num_rel) err |= 1u;
    if (d < 0 || d >= (int64_t)n_nodes) err |= 2u;
    if (err) atomicOr(&ctr->error, err);
    keys[e] = err ? 0ull : (u64)d * num_rel + (u64)t;
    vals[e] = (u32)e;
}
__global__ void run_start_kernel(const u64* __restrict__ keys, u32 n, const u32* __restrict__ rid, u32* __restrict__ rstart, u32 n_runs) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) rstart[n_runs] = n;
    if (i >= n) return;
    if (i == 0 || keys[i - 1] != keys[i]) rstart[rid[i]] = i;
}
// w_e = 1 / max(1, c[dst_e, rel_e]) in INPUT edge order (duplicates counted): plan.edge_weights
__global__ void weights_out_kernel(const u32* __restrict__ vals, u32 n, const u32* __restrict__ rid, const u32* __restrict__ rstart,
                                   float* __restrict__ w) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 r = rid[i];
    w[vals[i]] = 1.0f / (float)(rstart[r + 1] - rstart[r]);
}
__global__ void fill_float_kernel(float* __restrict__ p, u64 n, float v) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct Workspace {
    SortBufs sb;
    u32* scan_a;     // nmax + 1
    u32* scan_b;     // nmax + 1
    u32* gstart;     // gmax + 1
    u32* gkey;       // gmax + 1
    u32* gch;        // gmax + 1  (chunks per group -> chunk_base after the scan)
    u32* gun;        // gmax + 1  (units per group)
    Counters* ctr;
    size_t bytes;
};

// nmax: most elements any step sorts (edges + owned nodes, or edges alone for the weights)
static Workspace carve(void* base, u64 nmax, u64 gmax) {
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t nbytes) {
        void* p = base ? (char*)base + off : nullptr;
        off += align_up(nbytes);
        return p;
    };
    const u32 nseg = sort_segments((u32)nmax);
    w.ctr = (Counters*)take(sizeof(Counters));
    w.sb.k[0] = (u64*)take(nmax * 8);
    w.sb.k[1] = (u64*)take(nmax * 8);
    w.sb.v[0] = (u32*)take(nmax * 4);
    w.sb.v[1] = (u32*)take(nmax * 4);
    w.sb.hist = (u32*)take(((size_t)256 * nseg + 1) * 4);
    const u64 scan_len = nmax + 1 > (u64)256 * nseg ? nmax + 1 : (u64)256 * nseg;
    w.sb.sums = (u32*)take(((size_t)scan_blocks((u32)scan_len) + 2) * 4);
    w.scan_a = (u32*)take((nmax + 1) * 4);
    w.scan_b = (u32*)take((nmax + 1) * 4);
    w.gstart = (u32*)take((gmax + 1) * 4);
    w.gkey = (u32*)take((gmax + 1) * 4);
    w.gch = (u32*)take((gmax + 1) * 4);
    w.gun = (u32*)take((gmax + 1) * 4);
    w.bytes = off;
    return w;
}

static u32 grid_for(u64 n, int threads = 256) { return (u32)((n + threads - 1) / threads); }

static int read_u32(const u32* dev, u32* host, hipStream_t s) {
    hipError_t e = hipMemcpyAsync(host, dev, sizeof(u32), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return (int)e;
}

static int check_graph(const rgcn_graph_t* g) {
    if (g == nullptr) return RGCN_ERR_NULL;
    if (g->num_edges < 0 || g->num_nodes <= 0 || g->num_relations <= 0) return RGCN_ERR_PLAN;
    if (g->num_edges > 0 && (!g->src || !g->dst || !g->type)) return RGCN_ERR_NULL;
    if (g->num_edges >= (int64_t)0xFFFFFFFFll - g->num_nodes) return RGCN_ERR_PLAN;      // 32-bit positions
    return RGCN_OK;
}

// internal state handed from _begin to _finish (lives in rgcn_plan_sizes_t::opaque)
struct BuildState {
    u32 magic;
    u32 n_nodes, n_own, node_begin, num_rel, tile, chunk, layout;
    u32 cap;                // row tiles a chunk may hold (chunk / 16; 7: 112-row chunks at a 128-slot stride)
    u32 n_unique, n_groups, n_chunks, n_units, n_edges_owned;
    u32 ubuf;               // index of the SortBufs pair holding the merged (key, weight) arrays
    u64 nmax, gmax;
    KeyLayout kl;
};
static_assert(sizeof(BuildState) <= sizeof(((rgcn_plan_sizes_t*)nullptr)->opaque), "opaque state too small");
constexpr u32 kMagic = 0x52474350u;

}  // namespace rgcn_planner

using namespace rgcn_planner;

static u64 elems_max(int64_t num_edges, int32_t n_owned) { return (u64)num_edges + (u64)(n_owned > 0 ? n_owned : 0); }
static u64 groups_max(u64 nmax, int32_t n_owned, int32_t num_relations, int32_t tile) {
    const u64 n_tiles = ((u64)(n_owned > 0 ? n_owned : 0) + tile - 1) / (tile > 0 ? tile : 1);
    const u64 g = n_tiles * ((u64)num_relations + 1);
    return g < nmax ? g : nmax;
}

extern "C" size_t rgcn_plan_workspace_bytes(int64_t num_edges, int32_t n_owned, int32_t num_relations, int32_t tile) {
    if (num_edges < 0 || n_owned < 0 || num_relations <= 0 || tile <= 0) return 0;
    const u64 nmax = elems_max(num_edges, n_owned);
    return carve(nullptr, nmax > 0 ? nmax : 1, groups_max(nmax, n_owned, num_relations, tile) + 1).bytes;
}

extern "C" int rgcn_edge_weights(const rgcn_graph_t* g, int aggr_sum, float* w, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    int st = check_graph(g);
    if (st != RGCN_OK) return st;
    if (g->num_edges == 0) return RGCN_OK;
    if (!w || !workspace) return RGCN_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    const u64 E = (u64)g->num_edges;
    if (aggr_sum) {
        hipLaunchKernelGGL(fill_float_kernel, dim3(grid_for(E)), dim3(256), 0, s, w, E, 1.0f);
        return (int)hipGetLastError();
    }
    Workspace ws = carve(workspace, E, 1);
    if (workspace_bytes < ws.bytes) return RGCN_ERR_WORKSPACE;
    hipError_t e = hipMemsetAsync(ws.ctr, 0, sizeof(Counters), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(weight_keys_kernel, dim3(grid_for(E)), dim3(256), 0, s, g->dst, g->dst_stride, g->type, g->type_stride, E,
                       (u32)g->num_nodes, (u32)g->num_relations, ws.sb.k[0], ws.sb.v[0], ws.ctr);
    const int bits = bits_for((u64)g->num_nodes * (u64)g->num_relations - 1);
    const int cur = radix_sort_pairs(ws.sb, (u32)E, bits, s);
    hipLaunchKernelGGL(head_flags_kernel, dim3(grid_for(E)), dim3(256), 0, s, ws.sb.k[cur], (u32)E, 0, ws.scan_a);
    exclusive_scan(ws.scan_a, ws.scan_b, (u32)E, ws.sb.sums, s);
    u32 n_runs = 0, err = 0;
    if ((st = read_u32(ws.sb.sums + scan_blocks((u32)E), &n_runs, s)) != 0) return st;
    if ((st = read_u32(&ws.ctr->error, &err, s)) != 0) return st;
    if (err) return RGCN_ERR_GRAPH;
    hipLaunchKernelGGL(run_ids_kernel, dim3(grid_for(E)), dim3(256), 0, s, ws.scan_b, ws.scan_a, (u32)E);
    u32* rstart = (u32*)ws.sb.k[cur ^ 1];        // the sort's spare key buffer: E + 1 words fit into E 8-byte keys
    hipLaunchKernelGGL(run_start_kernel, dim3(grid_for(E)), dim3(256), 0, s, ws.sb.k[cur], (u32)E, ws.scan_b, rstart, n_runs);
    hipLaunchKernelGGL(weights_out_kernel, dim3(grid_for(E)), dim3(256), 0, s, ws.sb.v[cur], (u32)E, ws.scan_b, rstart, w);
    return (int)hipGetLastError();
}

extern "C" int rgcn_plan_build_begin(const rgcn_graph_t* g, const float* w, int transposed, int32_t node_begin, int32_t node_end,
                                     int32_t tile, int32_t chunk, int32_t layout, void* workspace, size_t workspace_bytes,
                                     rgcn_plan_sizes_t* sizes, void* stream) {
    int st = check_graph(g);
    if (st != RGCN_OK) return st;
    if (!sizes || !workspace || (g->num_edges > 0 && !w)) return RGCN_ERR_NULL;
    // chunk = 112: the slot stride of 128-slot chunks with at most SEVEN row tiles of rows per chunk (layouts 0 and 3) -- what
    // rgcn_tile3p_kernel's 42 KiB ring slots hold, which leaves its accumulator room for tiles up to 272
    u32 cap = (u32)chunk / 16u;
    if (chunk == 112) {
        if (layout != 0 && layout != 3) return RGCN_ERR_PLAN;
        cap = 7u;
        chunk = 128;
    }
    if (layout == 2) {
        // relation-major units (the edge-parallel path, rgcn_ep_*): ONE tile that spans the owned range makes the key order
        // (tile, relation, row, gathered node) relation-major; the rows of a relation are dealt over its row tiles as in layout 0
        if (chunk != 64 || node_end - node_begin > (1 << 24)) return RGCN_ERR_PLAN;
        tile = ((node_end - node_begin + 15) / 16) * 16;
    }
    if (tile <= 0 || (tile % 16) != 0 || (tile > 32768 && layout != 2) || (chunk != 64 && chunk != 128)) return RGCN_ERR_PLAN;
    if (layout != 0 && layout != 2 && !((layout == 1 || layout == 3) && chunk == 128) && !(layout == 5 && chunk == 64)) return RGCN_ERR_PLAN;
    // (node_begin need not be a tile multiple: tiles count from node_begin.  Callers that want a rank's tiles to BE the
    // single-rank tiles -- bit-identical outputs -- align their ranges themselves: scaling_rgcn_training_amd/dist.py)
    if (node_begin < 0 || node_end <= node_begin || node_end > g->num_nodes) return RGCN_ERR_PLAN;
    hipStream_t s = (hipStream_t)stream;
    const u32 n_own = (u32)(node_end - node_begin), R = (u32)g->num_relations;
    const u64 E = (u64)g->num_edges;
    const u32 n_tiles = (n_own + tile - 1) / tile;
    KeyLayout kl;
    kl.src_bits = bits_for((u64)g->num_nodes - 1);
    kl.dstl_bits = bits_for((u64)tile - 1);
    kl.rel_bits = bits_for((u64)R);
    kl.tile_bits = bits_for((u64)n_tiles - 1);
    if (kl.total() > 64 || kl.rel_bits + kl.tile_bits > 32) return RGCN_ERR_PLAN;
    const u64 nmax = elems_max((int64_t)E, (int32_t)n_own);
    const u64 gmax = groups_max(nmax, (int32_t)n_own, (int32_t)R, tile) + 1;
    Workspace ws = carve(workspace, nmax, gmax);
    if (workspace_bytes < ws.bytes) return RGCN_ERR_WORKSPACE;
    hipError_t e = hipMemsetAsync(ws.ctr, 0, sizeof(Counters), s);
    if (e != hipSuccess) return (int)e;
    // ---- keys: n_own root pseudo edges, then the owned edges --------------------------------------------------
    hipLaunchKernelGGL(root_keys_kernel, dim3(grid_for(n_own)), dim3(256), 0, s, ws.sb.k[0], ws.sb.v[0], n_own, (u32)node_begin,
                       (u32)tile, R, kl);
    if (E > 0) {
        const u32 blocks = grid_for(E) < 8192u ? grid_for(E) : 8192u;
        hipLaunchKernelGGL(edge_keys_kernel, dim3(blocks), dim3(256), 0, s, g->src, g->src_stride, g->dst, g->dst_stride, g->type,
                           g->type_stride, w, E, transposed, (u32)g->num_nodes, R, (u32)node_begin, (u32)node_end, (u32)tile, kl,
                           n_own, ws.sb.k[0], ws.sb.v[0], ws.ctr);
    }
    u32 owned = 0, err = 0;
    if ((st = read_u32(&ws.ctr->owned_edges, &owned, s)) != 0) return st;
    if ((st = read_u32(&ws.ctr->error, &err, s)) != 0) return st;
    if (err) return RGCN_ERR_GRAPH;
    const u32 n = n_own + owned;
    // ---- sort by (tile, relation, row in tile, gathered node), merge duplicate triples -------------------------
    const int cur = radix_sort_pairs(ws.sb, n, kl.total(), s);
    hipLaunchKernelGGL(head_flags_kernel, dim3(grid_for(n)), dim3(256), 0, s, ws.sb.k[cur], n, 0, ws.scan_a);
    exclusive_scan(ws.scan_a, ws.scan_b, n, ws.sb.sums, s);
    u32 n_unique = 0;
    if ((st = read_u32(ws.sb.sums + scan_blocks(n), &n_unique, s)) != 0) return st;
    const int ub = cur ^ 1;
    hipLaunchKernelGGL(merge_kernel, dim3(grid_for(n)), dim3(256), 0, s, ws.sb.k[cur], ws.sb.v[cur], n, ws.scan_b, ws.sb.k[ub],
                       ws.sb.v[ub]);
    // ---- groups = runs of equal (tile, relation) ------------------------------------------------------------------
    hipLaunchKernelGGL(head_flags_kernel, dim3(grid_for(n_unique)), dim3(256), 0, s, ws.sb.k[ub], n_unique, kl.gshift(), ws.scan_a);
    exclusive_scan(ws.scan_a, ws.scan_b, n_unique, ws.sb.sums, s);
    hipLaunchKernelGGL(run_ids_kernel, dim3(grid_for(n_unique)), dim3(256), 0, s, ws.scan_b, ws.scan_a, n_unique);   // scan_b = group id
    u32 n_groups = 0;
    if ((st = read_u32(ws.sb.sums + scan_blocks(n_unique), &n_groups, s)) != 0) return st;
    if ((u64)n_groups + 1 > gmax) return RGCN_ERR_PLAN;
    hipLaunchKernelGGL(group_start_kernel, dim3(grid_for(n_unique)), dim3(256), 0, s, ws.sb.k[ub], n_unique, kl.gshift(), ws.scan_b,
                       ws.gstart, ws.gkey, n_groups);
    hipLaunchKernelGGL(group_sizes_kernel, dim3(grid_for(n_groups)), dim3(256), 0, s, ws.gstart, n_groups, (u32)chunk, cap, (u32)layout, ws.gch, ws.gun);
    u32 n_chunks = 0, n_units = 0;
    exclusive_scan(ws.gun, ws.gun, n_groups, ws.sb.sums, s);
    if ((st = read_u32(ws.sb.sums + scan_blocks(n_groups), &n_units, s)) != 0) return st;
    exclusive_scan(ws.gch, ws.gch, n_groups, ws.sb.sums, s);                  // gch = chunk_base from here on
    if ((st = read_u32(ws.sb.sums + scan_blocks(n_groups), &n_chunks, s)) != 0) return st;
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    memset(sizes, 0, sizeof(*sizes));
    // (layout 2: the plan struct reports tiles of at most 32,768 nodes -- nothing walks them, the kernels' argument checks do)
    sizes->n_tiles = layout == 2 ? (int32_t)((n_own + 32767u) / 32768u) : (int32_t)n_tiles;
    sizes->n_chunks = (int32_t)n_chunks;
    sizes->n_units = (int32_t)n_units;
    sizes->n_slots = (int64_t)n_chunks * chunk;
    sizes->n_edges = (int64_t)owned;
    BuildState bs;
    memset(&bs, 0, sizeof(bs));
    bs.magic = kMagic;
    bs.n_nodes = (u32)g->num_nodes; bs.n_own = n_own; bs.node_begin = (u32)node_begin; bs.num_rel = R;
    bs.tile = (u32)tile; bs.chunk = (u32)chunk; bs.layout = (u32)layout; bs.cap = cap;
    bs.n_unique = n_unique; bs.n_groups = n_groups; bs.n_chunks = n_chunks; bs.n_units = n_units; bs.n_edges_owned = owned;
    bs.ubuf = (u32)ub; bs.nmax = nmax; bs.gmax = gmax; bs.kl = kl;
    memcpy(sizes->opaque, &bs, sizeof(bs));
    return RGCN_OK;
}

extern "C" int rgcn_plan_build_finish(const rgcn_plan_sizes_t* sizes, void* workspace, size_t workspace_bytes, rgcn_plan_t* plan,
                                      void* stream) {
    if (!sizes || !workspace || !plan) return RGCN_ERR_NULL;
    BuildState bs;
    memcpy(&bs, sizes->opaque, sizeof(bs));
    if (bs.magic != kMagic) return RGCN_ERR_PLAN;
    if (!plan->tile_ptr || !plan->chunk_rel || !plan->chunk_cnt || !plan->chunk_tile || !plan->chunk_flags || !plan->rel_order ||
        !plan->slot_src || !plan->slot_w || !plan->slot_row || !plan->slot_acc)
        return RGCN_ERR_NULL;
    Workspace ws = carve(workspace, bs.nmax, bs.gmax);
    if (workspace_bytes < ws.bytes) return RGCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const u32 n_slots = bs.n_chunks * bs.chunk, n_rt = n_slots / 16u;
    int32_t* slot_src = (int32_t*)plan->slot_src;
    float* slot_w = (float*)plan->slot_w;
    int32_t* slot_row = (int32_t*)plan->slot_row;
    int32_t* slot_acc = (int32_t*)plan->slot_acc;
    int32_t* chunk_rel = (int32_t*)plan->chunk_rel;
    int32_t* chunk_cnt = (int32_t*)plan->chunk_cnt;
    int32_t* chunk_tile = (int32_t*)plan->chunk_tile;
    int32_t* chunk_flags = (int32_t*)plan->chunk_flags;
    u32* dstl = (u32*)slot_row;   // scratch inside the output array until row_tile_kernel (see fill_slots_kernel)
    hipLaunchKernelGGL(fill_slots_kernel, dim3(grid_for(n_slots)), dim3(256), 0, s, n_slots, bs.n_nodes, bs.tile, slot_src, slot_w, dstl);
    u32* split = ws.scan_a;       // cut position of every chunk (layout 1); n_chunks <= nmax
    // relation-major units: layout 0's dealing inside the one tile; layout 3: layout 0, then compact_runs_kernel
    const u32 placement = (bs.layout == 2 || bs.layout == 3 || bs.layout == 5) ? 0u : bs.layout;
    hipLaunchKernelGGL(chunk_meta_kernel, dim3(grid_for(bs.n_chunks)), dim3(256), 0, s, ws.gch, ws.gstart, ws.gkey, bs.n_groups,
                       bs.n_chunks, bs.chunk, bs.cap, placement, ws.sb.k[bs.ubuf], bs.kl, split, chunk_rel, chunk_cnt, chunk_tile, chunk_flags);
    hipLaunchKernelGGL(place_kernel, dim3(grid_for(bs.n_unique)), dim3(256), 0, s, ws.sb.k[bs.ubuf], ws.sb.v[bs.ubuf], bs.n_unique,
                       ws.scan_b, ws.gstart, ws.gch, bs.chunk, bs.cap, placement, split, bs.kl, slot_src, slot_w, dstl);
    hipLaunchKernelGGL(row_tile_kernel, dim3(grid_for(n_rt)), dim3(256), 0, s, dstl, n_rt, bs.chunk, bs.tile, bs.n_own, chunk_tile,
                       slot_acc, slot_row, chunk_flags);
    const u32 struct_tile = bs.layout == 2 ? (bs.tile < 32768u ? bs.tile : 32768u) : bs.tile;
    const u32 n_tiles = (bs.n_own + struct_tile - 1) / struct_tile;
    // (layout 2: chunk_tile is 0 throughout, so tile_ptr comes out as [0, n_chunks, n_chunks, ...]: never walked)
    hipLaunchKernelGGL(tile_ptr_kernel, dim3(grid_for((u64)n_tiles + 1)), dim3(256), 0, s, chunk_tile, bs.n_chunks, n_tiles,
                       (int32_t*)plan->tile_ptr);
    if (bs.layout == 5) {      // pairs of rows with one (destination, relation, weight) on ONE slot: before the unit list is built
        if (!plan->slot_src2) return RGCN_ERR_NULL;
        hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for((u64)bs.n_chunks * 8u)), dim3(256), 0, s, (int32_t*)plan->slot_src2, (u64)bs.n_chunks * 8u,
                           (int32_t)bs.n_nodes);
        hipLaunchKernelGGL(dw_pairs_kernel, dim3(grid_for(bs.n_chunks, kPairThreads)), dim3(kPairThreads), 0, s, bs.n_chunks, bs.n_nodes, bs.n_own,
                           chunk_rel, chunk_tile, chunk_cnt, slot_src, slot_w, slot_row, (int32_t*)plan->slot_src2);
    }
    // ---- the weight-gradient walk: chunks stably re-sorted by relation, cut into 64-slot units ---------------------
    // (the sort reuses the pair buffers: everything read from them above is already enqueued on this stream)
    SortBufs cb = ws.sb;
    hipLaunchKernelGGL(chunk_keys_kernel, dim3(grid_for(bs.n_chunks)), dim3(256), 0, s, chunk_rel, bs.n_chunks, cb.k[0], cb.v[0]);
    const int cur = radix_sort_pairs(cb, bs.n_chunks, bs.kl.rel_bits, s);
    u32* ucnt = ws.scan_b;        // the group ids are dead by now; n_chunks <= n_unique <= nmax
    hipLaunchKernelGGL(unit_counts_kernel, dim3(grid_for(bs.n_chunks)), dim3(256), 0, s, cb.v[cur], chunk_cnt, bs.n_chunks, ucnt);
    exclusive_scan(ucnt, ucnt, bs.n_chunks, ws.sb.sums, s);
    hipLaunchKernelGGL(emit_units_kernel, dim3(grid_for(bs.n_chunks)), dim3(256), 0, s, cb.v[cur], chunk_cnt, ucnt, bs.n_chunks,
                       bs.chunk / 64u, (int32_t*)plan->rel_order);
    if (bs.layout == 3)      // last: the unit list above is layout 0's (nothing walks it on a layout-3 plan)
        hipLaunchKernelGGL(compact_runs_kernel, dim3(grid_for(bs.n_chunks, kCompactThreads)), dim3(kCompactThreads), 0, s, bs.n_chunks, bs.n_nodes, bs.tile, bs.n_own,
                           chunk_rel, chunk_tile, chunk_cnt, chunk_flags, slot_src, slot_w, slot_row, slot_acc);
    plan->n_nodes = (int32_t)bs.n_nodes;
    plan->n_owned = (int32_t)bs.n_own;
    plan->num_relations = (int32_t)bs.num_rel;
    plan->tile = (int32_t)struct_tile;
    plan->n_tiles = (int32_t)n_tiles;
    plan->n_chunks = (int32_t)bs.n_chunks;
    plan->chunk = (int32_t)bs.chunk;
    plan->n_units = (int32_t)bs.n_units;
    if (bs.layout == 5) {      // units the pairs left over: data-dependent -- this layout's _finish SYNCHRONISES the stream
        u32 nu = 0;
        const int st = read_u32(ws.sb.sums + scan_blocks(bs.n_chunks), &nu, s);
        if (st != 0) return st;
        plan->n_units = (int32_t)nu;
    }
    plan->layout = (int32_t)bs.layout;
    plan->chunk_rows = (int32_t)(bs.cap * 16u);
    return (int)hipGetLastError();
}

// ---- destination-major index over the slots of a relation-major plan (layout 2): what rgcn_ep_segment_sum walks -----------------
// seg_idx = the real slots (slot_row < n_owned) ordered by (destination row, slot index): a stable radix sort of (slot_row, slot);
// seg_ptr[d] = first position of destination d (lower bound in the sorted rows), seg_ptr[n_owned] = number of real slots.
__global__ void slot_keys_kernel(const int32_t* __restrict__ slot_row, u32 n_slots, u64* __restrict__ keys, u32* __restrict__ vals) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    keys[i] = (u64)(u32)slot_row[i];
    vals[i] = i;
}
__global__ void seg_ptr_kernel(const u64* __restrict__ keys, u32 n_slots, u32 n_owned, int32_t* __restrict__ seg_ptr) {
    const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n_owned) return;
    u32 lo = 0, hi = n_slots;               // first sorted slot whose row is >= d (padding slots carry row n_owned: they sort last)
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if ((u32)keys[mid] < d) lo = mid + 1; else hi = mid;
    }
    seg_ptr[d] = (int32_t)lo;
}
__global__ void copy_u32_kernel(const u32* __restrict__ in, u32 n, int32_t* __restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}

extern "C" int rgcn_eplan_segments(const int32_t* slot_row, int64_t n_slots, int32_t n_owned, void* workspace, size_t workspace_bytes,
                                   int32_t* seg_ptr, int32_t* seg_idx, void* stream) {
    if (!slot_row || !workspace || !seg_ptr || !seg_idx) return RGCN_ERR_NULL;
    if (n_slots <= 0 || n_slots >= (int64_t)0xFFFFFFFFll || n_owned <= 0) return RGCN_ERR_PLAN;
    Workspace ws = carve(workspace, (u64)n_slots, 1);
    if (workspace_bytes < ws.bytes) return RGCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const u32 n = (u32)n_slots;
    hipLaunchKernelGGL(slot_keys_kernel, dim3(grid_for(n)), dim3(256), 0, s, slot_row, n, ws.sb.k[0], ws.sb.v[0]);
    const int cur = radix_sort_pairs(ws.sb, n, bits_for((u64)n_owned), s);
    hipLaunchKernelGGL(seg_ptr_kernel, dim3(grid_for((u64)n_owned + 1)), dim3(256), 0, s, ws.sb.k[cur], n, (u32)n_owned, seg_ptr);
    hipLaunchKernelGGL(copy_u32_kernel, dim3(grid_for(n)), dim3(256), 0, s, ws.sb.v[cur], n, seg_idx);
    return (int)hipGetLastError();
}
