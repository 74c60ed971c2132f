// Helpers shared by the layer kernels (gfx950): typed 8-element loads, LDS tile stores, dropout hash.
#pragma once
#include <algorithm>
#include <type_traits>

#include "gcnpt_common.h"

namespace gcnpt {

// ---------------------------------------------------------------------------------------------------
// element access: 8 consecutive elements <-> 8 floats
// ---------------------------------------------------------------------------------------------------
template <typename T> struct io;

template <> struct io<float> {
    static __device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ float load1(const float* p) { return *p; }
    static __device__ __forceinline__ void store1(float* p, float v) { *p = v; }
};

template <> struct io<bf16_t> {
    static __device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
        v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
    }
    static __device__ __forceinline__ float load1(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void store1(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 8 elements of row `row` starting at column k0 of a dense [*, K] matrix; columns >= K read as 0
template <typename T>
__device__ __forceinline__ void load_row8(const T* base, size_t row, int K, int k0, bool vec, float (&v)[8]) {
    const T* p = base + row * (size_t)K + k0;
    if (vec && k0 + 8 <= K) {
        io<T>::load8(p, v);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (k0 + j < K) ? io<T>::load1(p + j) : 0.0f;
    }
}

// LDS tile element type for each compute type, and how 8 floats are parked in it
template <typename CT> struct tile;
template <> struct tile<bf16_t> {
    static __device__ __forceinline__ void put8(bf16_t* p, const float (&v)[8]) {
        uint4 u;
        u.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        u.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        u.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
        u.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
        *reinterpret_cast<uint4*>(p) = u;
    }
};
template <> struct tile<float> {
    static __device__ __forceinline__ void put8(float* p, const float (&v)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};

// LDS row stride in dwords: >= payload, == 8 (mod 16) so the 16 rows x 4 k-groups that one
// ds_read_b128 wave-instruction touches fall on distinct banks (bank = dword % 64, 16-lane groups)
__host__ __device__ inline int lds_stride_dw(int payload_dw) {
    int s = round_up(payload_dw, 4);
    while ((s & 15) != 8) s += 4;
    return s;
}

// counter-based dropout: one 32-bit hash of (seed, row, column pair) decides the two columns of the pair with
// 16 bits each, so the mask is a pure function of (seed, row, col) whatever the kernel's tiling is
__device__ __forceinline__ unsigned drop_hash(uint64_t seed, unsigned row, unsigned col_pair) {
    unsigned x = row * 0x9E3779B1u + col_pair * 0x85EBCA77u + (unsigned)seed;
    x ^= x >> 15; x *= 0x2C1B3C6Du;
    x += (unsigned)(seed >> 32);
    x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return x;
}
__device__ __forceinline__ bool drop_keep(unsigned hash, unsigned col, unsigned thresh16) {
    return ((col & 1u) ? (hash >> 16) : (hash & 0xffffu)) >= thresh16;
}

// v / d for d = deg + 1 given r = 1/d: one Newton step on the quotient gives the correctly rounded result
// in all but rare half-ulp cases (<= 1 ulp then), at 3 FMAs instead of a full IEEE division sequence
__device__ __forceinline__ float div_by(float v, float d, float r) {
    const float q = v * r;
    return __builtin_fmaf(__builtin_fmaf(-q, d, v), r, q);
}

template <typename IT> struct raw8 { uint4 a, b; };      // 8 elements as loaded (bf16: a only)

// 8 elements of row `row` from column k0.  NO load here is behind a condition: hipcc puts every conditional
// load in its own basic block with an s_waitcnt vmcnt(0) in front, which turns a batch of loads into a chain
// of round trips.  Callers pass a (row, k0) that is valid memory (clamped) and zero the result if it was not wanted.
template <typename IT, bool VEC>
__device__ __forceinline__ void issue8(const IT* base, size_t row, int K, int k0, raw8<IT>& r) {
    if constexpr (VEC) {                                    // K % 8 == 0, 16-byte aligned base, k0 + 8 <= K
        const IT* p = base + row * (size_t)K + k0;
        r.a = *reinterpret_cast<const uint4*>(p);
        if constexpr (sizeof(IT) == 4) r.b = *reinterpret_cast<const uint4*>(p + 4);
    } else {                                                // any K / alignment: 8 clamped element loads
        const IT* p = base + row * (size_t)K;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = io<IT>::load1(p + min(k0 + j, K - 1));
            v[j] = (k0 + j < K) ? x : 0.0f;
        }
        if constexpr (sizeof(IT) == 4) {
            r.a = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
            r.b = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
        } else {
            r.a.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            r.a.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            r.a.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
            r.a.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
        }
    }
}

// The same for rows that are only 8-byte (bf16) / 16-byte (f32) aligned: K % 4 == 0 (hidden = 300 is the case in point).  Two half
// loads, the upper one clamped to the row's last 4 columns: in the row's last chunk it then returns real, finite values of the row
// in the 4 slots past its end, which is harmless -- those columns meet zero weights (the packed images are zero padded past K) and
// fall outside what the weight gradient stores.  k0 <= K - 4.
template <typename IT>
__device__ __forceinline__ void issue8_half(const IT* base, size_t row, int K, int k0, raw8<IT>& r) {
    const IT* p = base + row * (size_t)K;
    const int k1 = min(k0 + 4, K - 4);
    if constexpr (sizeof(IT) == 4) {
        r.a = *reinterpret_cast<const uint4*>(p + k0);
        r.b = *reinterpret_cast<const uint4*>(p + k1);
    } else {
        const uint2 lo = *reinterpret_cast<const uint2*>(p + k0);
        const uint2 hi = *reinterpret_cast<const uint2*>(p + k1);
        r.a = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}

// the 8 floats of a raw8, all zero when !live
template <typename IT>
__device__ __forceinline__ void unpack8(const raw8<IT>& r, bool live, float (&v)[8]) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    const uint4 a = live ? r.a : z;
    if constexpr (sizeof(IT) == 4) {
        const uint4 b = live ? r.b : z;
        v[0] = __uint_as_float(a.x); v[1] = __uint_as_float(a.y); v[2] = __uint_as_float(a.z); v[3] = __uint_as_float(a.w);
        v[4] = __uint_as_float(b.x); v[5] = __uint_as_float(b.y); v[6] = __uint_as_float(b.z); v[7] = __uint_as_float(b.w);
    } else {
        v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
        v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
        v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
        v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
    }
}

// out-tile row stride in dwords: 16-byte aligned rows, == 4 (mod 8) to spread the 4 row groups of an
// accumulator store over the banks
__host__ __device__ inline int out_stride_dw(int payload_dw) {
    int s = round_up(payload_dw, 4);
    while ((s & 7) != 4) s += 4;
    return s;
}

// CPL consecutive columns of a row <-> floats (the element-wise kernels: diag_kernels.hip, pool_kernels.hip)
template <typename T, int CPL> struct dgio;
template <> struct dgio<float, 4> {
    static __device__ __forceinline__ void ld(const float* p, int, float (&v)[4]) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    }
    static __device__ __forceinline__ void st(float* p, int, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct dgio<bf16_t, 4> {
    static __device__ __forceinline__ void ld(const bf16_t* p, int, float (&v)[4]) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void st(bf16_t* p, int, const float (&v)[4]) {
        uint2 u;
        u.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        u.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<uint2*>(p) = u;
    }
};
template <typename T> struct dgio<T, 1> {    // any H / alignment: `n` = 1 when the column exists, 0 past the end (pointer clamped by the caller)
    static __device__ __forceinline__ void ld(const T* p, int n, float (&v)[1]) { const float x = io<T>::load1(p); v[0] = n ? x : 0.0f; }
    static __device__ __forceinline__ void st(T* p, int n, const float (&v)[1]) { if (n) io<T>::store1(p, v[0]); }
};

// the tile's 32 rows of an LDS bf16 tile X (stride in elements) as a fragment image (include/gcnpt.h): row-contraction
// operand of the weight gradient, 8 consecutive rows per lane, read transposed with ds_read_b64_tr_b16
__device__ __forceinline__ void emit_frag_image(uint4* F, const bf16_t* X, int stride, int width, int wave, int n_waves, int lane, size_t nks, size_t blk) {
    const int w_tiles = ceil_div(width, 16);
    const int i = lane & 15, g = lane >> 4, q4 = i >> 2, pp = i & 3;
    for (int t = wave; t < w_tiles; t += n_waves) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(X + (size_t)(8 * g + q4) * stride + 16 * t + 4 * pp));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(X + (size_t)(8 * g + 4 + q4) * stride + 16 * t + 4 * pp));
        uint4 u;
        u.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
        u.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
        u.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
        u.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
        F[((size_t)t * nks + blk) * 64 + lane] = u;
    }
}

// host-side argument helpers of the C-ABI wrappers
// rows of a batch: B sentences padded to T tokens, or -- T = 0 -- B token-packed rows whose pattern has absolute columns (include/gcnpt.h)
static inline long long rows_of(int B, int T) { return T > 0 ? (long long)B * T : (long long)B; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int kstep_of(int dtype) { return dtype == GCNPT_BF16 ? 32 : 16; }
static inline size_t esize(int dtype) { return dtype == GCNPT_BF16 ? 2 : 4; }
static inline bool dtype_ok(int d) { return d == GCNPT_F32 || d == GCNPT_BF16; }

}  // namespace gcnpt
