// rgcn_common.h -- device helpers shared by the gfx950 R-GCN kernels.
//
// Execution model of every hot kernel in this library (DESIGN.md "Kernels"):
//   one 512-thread workgroup per CU = 4 PRODUCER waves + 4 CONSUMER waves (one of each per SIMD);
//   producers gather feature rows by index straight into an LDS ring with LDS-DMA
//   (global_load_lds_dwordx4: per-lane source address, linear LDS destination), several ring slots
//   ahead, behind a COUNTED s_waitcnt vmcnt(N) and a raw s_barrier (a __syncthreads() would drain
//   the DMA queue); consumers read MFMA fragments from the ring with ds_read_b128/b32 and run
//   v_mfma_f32_16x16x4_f32 (exact fp32).  Row indices reach the producers through scalar loads
//   (lgkmcnt), so no ordinary vector load ever sits in the DMA queue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <cstring>
#include "../../include/rgcn_mi355x.h"

namespace rgcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));     // one operand of v_mfma_f32_16x16x32_bf16

constexpr int kChunk = 64;          // edge slots per unit of the dW walk and per ring slot of the dW kernels (== RGCN_UNIT)
constexpr int kThreads = 512;       // 8 waves: 0-3 producers, 4-7 consumers
constexpr int kProducerWaves = 4;
constexpr int kLdsBytes = 160 * 1024;

// 16 bytes of zeros: where a DMA lane points when its slot is padding or its column chunk lies
// beyond the feature width (LDS-DMA cannot be predicated per lane without changing the vmcnt count).
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void wg_barrier() {
    // consumers: make sure their LDS reads/atomics have retired; producers: nothing pending on lgkm
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// two fp32 -> two bf16 (round to nearest even), lo -> bits 0..15, hi -> bits 16..31
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// x = h + m + l with three round-to-nearest bf16 pieces (24 significant bits: exact for every finite fp32 whose bf16 rounding does
// not overflow), for two values at once: h / m / l hold the pieces of v0 in their low and of v1 in their high halves.
// 3 v_cvt_pk_bf16_f32 + 4 shift / mask + 4 subtract.  (Tried, round 3: the residuals by v_dot2c_f32_bf16 with a (-1, 0) / (0, -1)
// operand -- one instruction per element instead of two -- measured 2-3 % SLOWER in all three kernels that cut operands, and the
// packed inline constant does not mean what the builtin's vector literal says: wrong pieces.  Not kept.)
__device__ __forceinline__ void split3_pair(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(v0, v1);
    v0 -= __uint_as_float(h << 16);
    v1 -= __uint_as_float(h & 0xFFFF0000u);
    m = cvt_pk_bf16(v0, v1);
    v0 -= __uint_as_float(m << 16);
    v1 -= __uint_as_float(m & 0xFFFF0000u);
    l = cvt_pk_bf16(v0, v1);
}

// LDS-DMA: 16 B per lane from `gptr` (per lane) to lds_base + lane*16 (lds_base wave-uniform).
__device__ __forceinline__ void dma16(const float* gptr, float* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
// 4 B per lane to lds_base + lane*4.
__device__ __forceinline__ void dma4(const void* gptr, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_base, 4, 0, 0);
}

// Same, through a buffer descriptor: address = rsrc base + voffset (32-bit, per lane); an offset beyond
// num_records makes the hardware range check feed ZEROS, which is how padding rows and padding columns
// are produced without a pointer select or 64-bit address arithmetic.
// AUX: cache-policy bits of the load (0 default, 2 = nt: streamed data that is read once)
template <int AUX = 0>
__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, float* lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, voffset, 0, 0, AUX);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    // dword3 0x00020000: raw 32-bit data format (stride 0, offen addressing, range check on num_records)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// 16-byte global load that hipcc can neither sink nor wait for: the consumers prefetch the NEXT chunk's
// weight fragments with it.  As a plain C++ load the prefetch was sunk below the MFMAs it should overlap
// (LLVM moves loads towards their first use), which put an L2 round trip in front of every chunk's first
// MFMA.  Contract (cdna_hip_programming.md 5.7): the outputs are not touched before the caller's own
// s_waitcnt vmcnt(0); the consumer waves issue no other vector-memory operation in between.
template <int BYTE_OFFSET>
__device__ __forceinline__ void prefetch16(f32x4& dst, const f32x4* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(BYTE_OFFSET) : "memory");
}

// element `idx` (0..15) of sixteen wave-uniform ints held as 4 x int4 (SGPRs after s_load_dwordx16).
// A select tree, never an array: runtime-indexed arrays go to scratch.
__device__ __forceinline__ int pick16(int idx, int4 q0, int4 q1, int4 q2, int4 q3) {
    const bool b3 = idx & 8, b2 = idx & 4, b1 = idx & 2, b0 = idx & 1;
    const int ax = b3 ? (b2 ? q3.x : q2.x) : (b2 ? q1.x : q0.x);
    const int ay = b3 ? (b2 ? q3.y : q2.y) : (b2 ? q1.y : q0.y);
    const int az = b3 ? (b2 ? q3.z : q2.z) : (b2 ? q1.z : q0.z);
    const int aw = b3 ? (b2 ? q3.w : q2.w) : (b2 ? q1.w : q0.w);
    const int lo = b1 ? az : ax;
    const int hi = b1 ? aw : ay;
    return b0 ? hi : lo;
}

// XOR swizzle of the 16-byte column position inside a ring row (applied on the DMA SOURCE address and
// on the read; the LDS image itself stays lane-linear as LDS-DMA requires).
//   kRowRead  : rows are read 16 at a time with ds_read_b128 (A operand of H @ W)        -> f(r) = r
//   kColRead  : 4 consecutive rows are read per ds_read_b32 (both operands of H^T @ G)   -> f(r) = rot2(r)
//   kLinear   : no swizzle (whole rows read by 16 consecutive lanes with ds_read_b128: conflict free as is)
enum SwizzleMode { kRowRead = 0, kColRead = 1, kLinear = 2 };
template <int MODE, int LPR>
__device__ __forceinline__ int swizzle(int row) {
    constexpr int mask = (LPR - 1) < 15 ? (LPR - 1) : 15;
    if (MODE == kLinear) return 0;
    if (MODE == kRowRead) return row & mask;
    return (((row & 3) << 2) | ((row >> 2) & 3)) & mask;
}

// Wave-uniform metadata loads.  Reading through the CONSTANT address space makes hipcc emit scalar
// loads (s_load_*, lgkmcnt) even after the kernel has issued LDS-DMA or stores; a plain global read
// would be re-classified as clobberable, become a vector load and put a compiler-inserted
// s_waitcnt vmcnt(0) in front of its first use -- draining the DMA ring the producers keep in flight.
// Valid because the plan arrays are never written while a kernel that reads them runs.
typedef const __attribute__((address_space(4))) int c_int;
typedef const __attribute__((address_space(4))) i32x4 c_i32x4;
__device__ __forceinline__ int ldc(const int* p, long idx) { return ((c_int*)(uintptr_t)p)[idx]; }
__device__ __forceinline__ int4 ldc4(const int* p, long idx4) {
    const i32x4 v = ((c_i32x4*)(uintptr_t)p)[idx4];
    return make_int4(v[0], v[1], v[2], v[3]);
}

// The same loads through a SCALAR BUFFER descriptor: s_buffer_load_dword takes its byte offset from one SGPR, so a loop that
// walks several per-chunk arrays in step pays ONE s_add per iteration for all of them (ldc's 64-bit index arithmetic is 4-5
// scalar instructions per load, ~25 of a consumer wave's ~70 per chunk in rgcn_tile3p_kernel), and an offset past num_records
// reads 0 -- no clamp, no branch for "one past the end".  Inline asm: the compiler does not know the result arrives later, so
// NOTHING may touch the destination (not even a copy) before sbuf_wait names it.
__device__ __forceinline__ i32x4 make_srsrc(const void* base, long bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xFFFFull));
    r[2] = __builtin_amdgcn_readfirstlane((int)(bytes < 0 ? 0 : (bytes > 0x7FFFFFFCl ? 0x7FFFFFFCl : bytes)));
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void sbuf_load(int& dst, i32x4 rsrc, unsigned byte_off) {
    // (readfirstlane: where the optimiser has rewritten the running offset in terms of a loop counter it keeps in a vector
    // register, the operand would otherwise reach the asm as a VGPR -- "illegal VGPR to SGPR copy"; folded away where it is scalar)
    const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane((int)byte_off);
    asm volatile("s_buffer_load_dword %0, %1, %2" : "=s"(dst) : "s"(rsrc), "s"(off) : "memory");
}
// everything this wave has in flight on lgkmcnt (scalar loads return out of order: there is no counted wait for them) has
// landed; the loaded values are operands so that no use of them can be scheduled above the wait
__device__ __forceinline__ void sbuf_wait(int& a, int& b, int& c) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c)::"memory");
}
__device__ __forceinline__ void sbuf_wait(int& a) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a)::"memory");
}

__host__ __device__ inline int padded_width(int w) {
    if (w < 1 || w > 128) return 0;
    return w <= 16 ? 16 : (w <= 32 ? 32 : (w <= 64 ? 64 : 128));
}


// ---- host-side argument checks shared by the translation units of the library ----------------------------------
// The library is gfx950 code only: any other device (or none) is RGCN_ERR_DEVICE.  Looked up once per device.
inline int check_device() {
    static std::atomic<int> state[64];          // 0 unknown, 1 gfx950, 2 other
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return RGCN_ERR_DEVICE;
    if (dev >= 64) return RGCN_ERR_DEVICE;
    int st = state[dev].load(std::memory_order_relaxed);
    if (st == 0) {
        hipDeviceProp_t prop;
        st = (hipGetDeviceProperties(&prop, dev) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ? 1 : 2;
        state[dev].store(st, std::memory_order_relaxed);
    }
    return st == 1 ? RGCN_OK : RGCN_ERR_DEVICE;
}

inline int check_stride(int ld, int width) {
    if (width < 1 || width > RGCN_MAX_WIDTH) return RGCN_ERR_WIDTH;
    if ((ld % 4) != 0 || ld < ((width + 3) / 4) * 4) return RGCN_ERR_STRIDE;
    return RGCN_OK;
}


}  // namespace rgcn
