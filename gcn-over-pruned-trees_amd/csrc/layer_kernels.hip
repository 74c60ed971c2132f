// GCN layer kernels for gfx950 (CDNA4): reference model/gcn.py:269-271, 390-393 and their autograd.
//
//   forward        out = dropout(relu((((A+I) h) W^T + 2 b) / (deg + 1)))
//   backward-data  dh  = ((A+I)^T dZ) W            with dZ = dY * 1[Y>0] * scale / (deg + 1)
//   backward-weight dW += dZ^T ((A+I) h),  db += 2 sum_r dZ
//
// forward and backward-data are the SAME row-tile kernel: a workgroup owns ROWS consecutive token rows,
// (1) stages their CSR extents in LDS, (2) gathers "self + neighbours" of the source rows in fp32 and
// parks the tile in LDS in the MFMA operand type, (3) streams the pre-packed weight fragments
// (gcnpt_pack_weights) as the MFMA B operand, (4) applies the epilogue on the accumulators.
// The dense [B,T,T] bmm of the reference (gcn.py:269) never exists: the adjacency has <= 3 entries
// per kept token, so aggregation is a gather, and only the W contraction runs on the matrix cores.
#include "gcnpt_common.h"

namespace gcnpt {

constexpr int ROWS = 32;             // token rows per workgroup (two 16-row MFMA tiles)
constexpr int LAYER_THREADS = 256;   // 4 waves; each owns every 4th 16-column output tile
constexpr int NTW = 4;               // output tiles per wave per pass (256 columns per pass)

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

// counter-based dropout decision for output element e: uniform 24-bit value from (seed, e)
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t e, unsigned thresh24) {
    unsigned x = (unsigned)e ^ (unsigned)seed;
    unsigned y = (unsigned)(e >> 32) ^ (unsigned)(seed >> 32) ^ 0x9E3779B9u;
    x *= 0x85EBCA6Bu; x ^= x >> 15; x += y * 0xC2B2AE35u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (x >> 8) >= thresh24;
}

struct RowTileParams {
    const void* src;        // fwd: h [N,K]      bwd: dY [N,K]
    const void* yref;       // bwd: Y [N,K] (stored layer output)
    const void* wfrag;      // packed B operand, gcnpt_pack_weights
    const float* bias;      // fwd: [NOUT]
    const int32_t* g_row_ptr;   // pattern gathered over (fwd: A, bwd: A^T)
    const int32_t* g_col_idx;
    const int32_t* d_row_ptr;   // pattern whose row length gives deg (always A)
    void* out;              // [N,NOUT]
    int N, T, K, NOUT, Kpad;
    int vec_in;             // source rows may be read 8 elements at a time
    float scale;            // bwd: 1/(1-p) of the dropout applied to Y
    float drop_p;           // fwd
    unsigned drop_thresh24;
    uint64_t seed;
};

template <typename CT, typename IT, typename OT, bool BWD>
__global__ __launch_bounds__(LAYER_THREADS) void rowtile_kernel(const RowTileParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int KSTEP = sizeof(CT) == 2 ? 32 : 16;            // K consumed per MFMA group
    const int stride = lds_stride_dw(p.Kpad * (int)sizeof(CT) / 4) * 4 / (int)sizeof(CT);   // in CT elements
    CT* S = reinterpret_cast<CT*>(smem_raw);
    int* rbeg = reinterpret_cast<int*>(smem_raw + (size_t)ROWS * stride * sizeof(CT));
    int* rend = rbeg + ROWS;
    float* rdenom = reinterpret_cast<float*>(rend + ROWS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * ROWS;
    const IT* src = static_cast<const IT*>(p.src);
    const IT* yref = static_cast<const IT*>(p.yref);

    // (1) CSR extents of the tile's rows -> LDS
    if (tid < ROWS) {
        const int r = r0 + tid;
        int beg = 0, end = 0;
        float dn = 1.0f;
        if (r < p.N) {
            const int b = r / p.T, i = r - b * p.T;
            const size_t q = (size_t)b * (p.T + 1) + i;
            beg = p.g_row_ptr[q]; end = p.g_row_ptr[q + 1];
            dn = (float)(p.d_row_ptr[q + 1] - p.d_row_ptr[q] + 1);      // gcn.py:261
        }
        rbeg[tid] = beg; rend[tid] = end; rdenom[tid] = dn;
    }
    __syncthreads();

    // (2) gather: S[row,:] = x[row,:] + sum_{c in pattern row} x[c,:]   (fp32), parked as CT
    const int nchunk = p.Kpad / 8;
    for (int it = tid; it < ROWS * nchunk; it += LAYER_THREADS) {
        const int row = it / nchunk, k0 = (it - row * nchunk) * 8;
        const int r = r0 + row;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < p.N && k0 < p.K) {
            const int b = r / p.T;
            const size_t rbase = (size_t)b * p.T;
            auto add_row = [&](size_t c) {
                float v[8];
                load_row8<IT>(src, c, p.K, k0, p.vec_in, v);
                if (BWD) {
                    float y[8];
                    load_row8<IT>(yref, c, p.K, k0, p.vec_in, y);
                    const int cb = (int)(c / p.T);
                    const size_t q = (size_t)cb * (p.T + 1) + (c - (size_t)cb * p.T);
                    const float inv = p.scale / (float)(p.d_row_ptr[q + 1] - p.d_row_ptr[q] + 1);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += (y[j] > 0.0f) ? v[j] * inv : 0.0f;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += v[j];
                }
            };
            add_row((size_t)r);                                          // the explicit W(h) term, gcn.py:271
            for (int e = rbeg[row]; e < rend[row]; ++e) add_row(rbase + p.g_col_idx[e]);   // gcn.py:269
        }
        tile<CT>::put8(S + (size_t)row * stride + k0, acc);
    }
    __syncthreads();

    // (3) + (4) tile x weights on the matrix cores, epilogue on the accumulators
    const int n_tiles = ceil_div(p.NOUT, 16);
    const int ksteps = p.Kpad / KSTEP;
    const uint4* wfrag = static_cast<const uint4*>(p.wfrag);
    OT* out = static_cast<OT*>(p.out);
    const int arow = lane & 15, kgrp = lane >> 4;

    for (int pass = 0; pass * 4 * NTW < n_tiles; ++pass) {
        f32x4_t acc[2][NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) { acc[0][j] = (f32x4_t){0, 0, 0, 0}; acc[1][j] = (f32x4_t){0, 0, 0, 0}; }
        const int tile0 = pass * 4 * NTW + wave;

        for (int ks = 0; ks < ksteps; ++ks) {
            if constexpr (sizeof(CT) == 2) {
                const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(S + (size_t)arow * stride + ks * 32 + kgrp * 8);
                const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(S + (size_t)(arow + 16) * stride + ks * 32 + kgrp * 8);
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const int tl = tile0 + j * 4;
                    if (tl < n_tiles) {
                        const uint4 w = wfrag[((size_t)tl * ksteps + ks) * 64 + lane];
                        const bf16x8_t bq = __builtin_bit_cast(bf16x8_t, w);
                        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bq, acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bq, acc[1][j], 0, 0, 0);
                    }
                }
            } else {
                const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(S + (size_t)arow * stride + ks * 16 + kgrp * 4);
                const f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(S + (size_t)(arow + 16) * stride + ks * 16 + kgrp * 4);
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const int tl = tile0 + j * 4;
                    if (tl < n_tiles) {
                        const uint4 w = wfrag[((size_t)tl * ksteps + ks) * 64 + lane];
                        const f32x4_t bq = __builtin_bit_cast(f32x4_t, w);
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], bq[s], acc[0][j], 0, 0, 0);
                            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bq[s], acc[1][j], 0, 0, 0);
                        }
                    }
                }
            }
        }

#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int tl = tile0 + j * 4;
            if (tl >= n_tiles) continue;
            const int col = tl * 16 + (lane & 15);
            if (col >= p.NOUT) continue;
            const float b2 = BWD ? 0.0f : 2.0f * p.bias[col];            // bias enters twice, gcn.py:270-271
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = mt * 16 + (lane >> 4) * 4 + g;
                    const int r = r0 + row;
                    if (r >= p.N) continue;
                    float v = acc[mt][j][g];
                    if (!BWD) {
                        v = (v + b2) / rdenom[row];                      // gcn.py:390
                        v = v > 0.0f ? v : 0.0f;                         // gcn.py:392
                        if (p.drop_p > 0.0f) {                            // gcn.py:393
                            const uint64_t e = (uint64_t)r * (uint64_t)p.NOUT + (uint64_t)col;
                            v = drop_keep(p.seed, e, p.drop_thresh24) ? v * p.scale : 0.0f;
                        }
                    }
                    io<OT>::store1(out + (size_t)r * p.NOUT + col, v);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// nn.Linear weight [H,Din] fp32 -> MFMA B-operand fragments
//   fragment (tile, kstep, lane) = 16 bytes:
//     bf16: 8 values  B[k = 32 kstep + 8 (lane>>4) + j][n = 16 tile + (lane&15)],  j = 0..7
//     f32 : 4 values  B[k = 16 kstep + 4 (lane>>4) + s][n = 16 tile + (lane&15)],  s = 0..3
//   forward image : B[k][n] = W[n][k]  (n over H,   k over Din)
//   backward image: B[k][n] = W[k][n]  (n over Din, k over H)
// ---------------------------------------------------------------------------------------------------
template <typename CT>
__global__ void pack_weights_kernel(const float* __restrict__ W, int H, int Din, uint4* __restrict__ wf,
                                    uint4* __restrict__ wb) {
    constexpr int KSTEP = sizeof(CT) == 2 ? 32 : 16;
    constexpr int PER = sizeof(CT) == 2 ? 8 : 4;
    const int ksf = round_up(Din, KSTEP) / KSTEP, ntf = ceil_div(H, 16);
    const int ksb = round_up(H, KSTEP) / KSTEP, ntb = ceil_div(Din, 16);
    const long long nf = wf ? (long long)ntf * ksf * 64 : 0;
    const long long nb = wb ? (long long)ntb * ksb * 64 : 0;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < nf + nb;
         id += (long long)gridDim.x * blockDim.x) {
        const bool bwd = id >= nf;
        const long long f = bwd ? id - nf : id;
        const int ks_n = bwd ? ksb : ksf;
        const int lane = (int)(f & 63);
        const int ks = (int)((f >> 6) % ks_n), tl = (int)((f >> 6) / ks_n);
        const int n = tl * 16 + (lane & 15);
        const int kb = ks * KSTEP + (lane >> 4) * PER;
        float v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int k = kb + j;
            float x = 0.0f;
            if (!bwd) { if (n < H && k < Din) x = W[(size_t)n * Din + k]; }
            else      { if (k < H && n < Din) x = W[(size_t)k * Din + n]; }
            v[j] = x;
        }
        uint4 u;
        if constexpr (sizeof(CT) == 2) {
            u.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            u.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            u.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
            u.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
        } else {
            u.x = __float_as_uint(v[0]); u.y = __float_as_uint(v[1]); u.z = __float_as_uint(v[2]); u.w = __float_as_uint(v[3]);
        }
        (bwd ? wb : wf)[f] = u;
    }
}

// ---------------------------------------------------------------------------------------------------
// backward-weight: dW[H,Din] += dZ^T S over a slice of the rows, S = (A+I) h recomputed by gather.
// Workgroup = (64 x 96) block of dW x one K-slice of rows; 4 waves as 2 (m) x 2 (n), each 2 x 3 tiles.
// Per 32-row chunk the dZ and S tiles are parked row-major in LDS and read TRANSPOSED
// (ds_read_b64_tr_b16 for bf16; plain ds_read_b32 for f32) because the contraction index is the row.
// ---------------------------------------------------------------------------------------------------
constexpr int WB_M = 64, WB_N = 96, WB_K = 32;

struct WeightGradParams {
    const void* dY; const void* Y; const void* h;
    const int32_t* row_ptr; const int32_t* col_idx; const int32_t* d_row_ptr;
    float* dW; float* db;
    int N, T, Din, H, rows_per_slice;
    int vec_g, vec_h;
    float scale;
};

template <typename CT, typename GT, typename HT>
__global__ __launch_bounds__(LAYER_THREADS) void weight_grad_kernel(const WeightGradParams p) {
    constexpr int ZS = sizeof(CT) == 2 ? (WB_M + 8) : (WB_M + 4);    // LDS row strides (elements), 8-byte aligned rows
    constexpr int SS = sizeof(CT) == 2 ? (WB_N + 8) : (WB_N + 4);
    __shared__ __attribute__((aligned(16))) CT Zt[WB_K * ZS];
    __shared__ __attribute__((aligned(16))) CT St[WB_K * SS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_base = blockIdx.x * WB_M, n_base = blockIdx.y * WB_N;
    const int k_lo = blockIdx.z * p.rows_per_slice, k_hi = min(p.N, k_lo + p.rows_per_slice);
    const GT* dY = static_cast<const GT*>(p.dY);
    const GT* Y = static_cast<const GT*>(p.Y);
    const HT* h = static_cast<const HT*>(p.h);

    f32x4_t acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4_t){0, 0, 0, 0};
    float dbacc = 0.0f;

    for (int kc = k_lo; kc < k_hi; kc += WB_K) {
        // dZ chunk: 32 rows x 64 columns = 256 items of 8
        {
            const int row = tid >> 3, c0 = (tid & 7) * 8;
            const int r = kc + row, col = m_base + c0;
            float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (r < k_hi && col < p.H) {
                float g[8], y[8];
                load_row8<GT>(dY, (size_t)r, p.H, col, p.vec_g, g);
                load_row8<GT>(Y, (size_t)r, p.H, col, p.vec_g, y);
                const int b = r / p.T;
                const size_t q = (size_t)b * (p.T + 1) + (r - b * p.T);
                const float inv = p.scale / (float)(p.d_row_ptr[q + 1] - p.d_row_ptr[q] + 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = (y[j] > 0.0f) ? g[j] * inv : 0.0f;
            }
            tile<CT>::put8(Zt + row * ZS + c0, z);
        }
        // S chunk: 32 rows x 96 columns = 384 items of 8
        for (int it = tid; it < WB_K * (WB_N / 8); it += LAYER_THREADS) {
            const int row = it / (WB_N / 8), c0 = (it - row * (WB_N / 8)) * 8;
            const int r = kc + row, col = n_base + c0;
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (r < k_hi && col < p.Din) {
                const int b = r / p.T;
                const size_t q = (size_t)b * (p.T + 1) + (r - b * p.T);
                float v[8];
                load_row8<HT>(h, (size_t)r, p.Din, col, p.vec_h, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] = v[j];
                const int beg = p.row_ptr[q], end = p.row_ptr[q + 1];
                for (int e = beg; e < end; ++e) {
                    load_row8<HT>(h, (size_t)b * p.T + p.col_idx[e], p.Din, col, p.vec_h, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) s[j] += v[j];
                }
            }
            tile<CT>::put8(St + row * SS + c0, s);
        }
        __syncthreads();

        if (blockIdx.y == 0 && tid < WB_M) {                     // bias gradient: column sums of dZ
            float sum = 0.0f;
            for (int row = 0; row < WB_K; ++row) {
                if constexpr (sizeof(CT) == 2) sum += bf16_to_f32(Zt[row * ZS + tid]);
                else sum += Zt[row * ZS + tid];
            }
            dbacc += sum;
        }

        if constexpr (sizeof(CT) == 2) {
            // lane (i = lane&15, g = lane>>4) needs X[k = 8g + j][c0 + i], j = 0..7: two transposed 4x16 reads
            const int i = lane & 15, g = lane >> 4, q4 = i >> 2, pp = i & 3;
            bf16x8_t a[2], bq[3];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int c0 = (wm * 2 + mt) * 16 + 4 * pp;
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t*)(Zt + (8 * g + q4) * ZS + c0));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t*)(Zt + (8 * g + 4 + q4) * ZS + c0));
                typedef __attribute__((ext_vector_type(8))) short s16x8_t;
                const s16x8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                a[mt] = __builtin_bit_cast(bf16x8_t, both);
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const int c0 = (wn * 3 + nt) * 16 + 4 * pp;
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t*)(St + (8 * g + q4) * SS + c0));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t*)(St + (8 * g + 4 + q4) * SS + c0));
                typedef __attribute__((ext_vector_type(8))) short s16x8_t;
                const s16x8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bq[nt] = __builtin_bit_cast(bf16x8_t, both);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], bq[nt], acc[mt][nt], 0, 0, 0);
        } else {
            const int i = lane & 15, g = lane >> 4;
#pragma unroll
            for (int k4 = 0; k4 < WB_K / 4; ++k4) {
                float a[2], bq[3];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) a[mt] = Zt[(k4 * 4 + g) * ZS + (wm * 2 + mt) * 16 + i];
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) bq[nt] = St[(k4 * 4 + g) * SS + (wn * 3 + nt) * 16 + i];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bq[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // reduce the K-slices with float atomics (dW/db were zeroed by the launcher or the caller)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = m_base + (wm * 2 + mt) * 16 + (lane >> 4) * 4 + g;
                const int n = n_base + (wn * 3 + nt) * 16 + (lane & 15);
                if (m < p.H && n < p.Din) atomicAdd(p.dW + (size_t)m * p.Din + n, acc[mt][nt][g]);
            }
    if (blockIdx.y == 0 && tid < WB_M && m_base + tid < p.H) atomicAdd(p.db + m_base + tid, 2.0f * dbacc);
}

}  // namespace gcnpt

// =====================================================================================================
// C-ABI
// =====================================================================================================
using namespace gcnpt;

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int kstep_of(int dtype) { return dtype == GCNPT_BF16 ? 32 : 16; }
static inline size_t esize(int dtype) { return dtype == GCNPT_BF16 ? 2 : 4; }

extern "C" size_t gcnpt_packed_bytes(int n_out, int k_in, int dtype) {
    if (n_out <= 0 || k_in <= 0 || (dtype != GCNPT_F32 && dtype != GCNPT_BF16)) return 0;
    const int ks = round_up(k_in, kstep_of(dtype)) / kstep_of(dtype);
    return (size_t)ceil_div(n_out, 16) * ks * 64 * 16;
}

extern "C" int gcnpt_pack_weights(void* stream, const float* W, int H, int Din, int dtype, void* w_fwd, void* w_bwd) {
    GCNPT_REQUIRE(W && (w_fwd || w_bwd), "pack_weights: null pointer");
    GCNPT_REQUIRE(H > 0 && Din > 0, "pack_weights: H and Din must be positive");
    GCNPT_REQUIRE(dtype == GCNPT_F32 || dtype == GCNPT_BF16, "pack_weights: dtype %d", dtype);
    const size_t frags = (w_fwd ? gcnpt_packed_bytes(H, Din, dtype) : 0) / 16 + (w_bwd ? gcnpt_packed_bytes(Din, H, dtype) : 0) / 16;
    const int grid = (int)((frags + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GCNPT_BF16)
        hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, W, H, Din, (uint4*)w_fwd, (uint4*)w_bwd);
    else
        hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(grid), dim3(256), 0, s, W, H, Din, (uint4*)w_fwd, (uint4*)w_bwd);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

template <typename CT, typename IT, typename OT, bool BWD>
static int launch_rowtile(hipStream_t s, const RowTileParams& p) {
    const int stride = lds_stride_dw(p.Kpad * (int)sizeof(CT) / 4) * 4 / (int)sizeof(CT);
    const size_t lds = (size_t)ROWS * stride * sizeof(CT) + ROWS * (2 * sizeof(int) + sizeof(float));
    if (lds > 160 * 1024) return fail(GCNPT_E_UNSUPPORTED, "layer: K=%d needs %zu B of LDS per workgroup", p.K, lds);
    auto kern = rowtile_kernel<CT, IT, OT, BWD>;
    if (lds > 64 * 1024)
        GCNPT_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(ceil_div(p.N, ROWS)), dim3(LAYER_THREADS), lds, s, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

template <bool BWD>
static int dispatch_rowtile(hipStream_t s, const RowTileParams& p, int in_dtype, int out_dtype, int compute) {
    if (compute == GCNPT_F32) {
        if (in_dtype != GCNPT_F32 || out_dtype != GCNPT_F32)
            return fail(GCNPT_E_UNSUPPORTED, "compute_dtype f32 needs f32 activations");
        return launch_rowtile<float, float, float, BWD>(s, p);
    }
    if (in_dtype == GCNPT_F32 && out_dtype == GCNPT_F32) return launch_rowtile<bf16_t, float, float, BWD>(s, p);
    if (in_dtype == GCNPT_F32 && out_dtype == GCNPT_BF16) return launch_rowtile<bf16_t, float, bf16_t, BWD>(s, p);
    if (in_dtype == GCNPT_BF16 && out_dtype == GCNPT_F32) return launch_rowtile<bf16_t, bf16_t, float, BWD>(s, p);
    return launch_rowtile<bf16_t, bf16_t, bf16_t, BWD>(s, p);
}

static inline bool dtype_ok(int d) { return d == GCNPT_F32 || d == GCNPT_BF16; }

extern "C" int gcnpt_layer_fwd(void* stream, const void* h, int h_dtype, const void* w_fwd, const float* bias,
                               const int32_t* row_ptr, const int32_t* col_idx, const int32_t* deg_row_ptr, int B, int T,
                               int Din, int H, void* out, int out_dtype, int compute_dtype, float drop_p, uint64_t seed) {
    GCNPT_REQUIRE(h && w_fwd && bias && row_ptr && col_idx && out, "layer_fwd: null pointer");
    GCNPT_REQUIRE(B > 0 && T > 0 && Din > 0 && H > 0, "layer_fwd: sizes must be positive");
    GCNPT_REQUIRE(dtype_ok(h_dtype) && dtype_ok(out_dtype) && dtype_ok(compute_dtype), "layer_fwd: bad dtype");
    GCNPT_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "layer_fwd: drop_p=%f outside [0,1)", (double)drop_p);
    if ((long long)B * T > 0x7fffffffLL / 2) return fail(GCNPT_E_UNSUPPORTED, "layer_fwd: B*T too large");
    RowTileParams p{};
    p.src = h; p.yref = nullptr; p.wfrag = w_fwd; p.bias = bias;
    p.g_row_ptr = row_ptr; p.g_col_idx = col_idx; p.d_row_ptr = deg_row_ptr ? deg_row_ptr : row_ptr; p.out = out;
    p.N = B * T; p.T = T; p.K = Din; p.NOUT = H; p.Kpad = round_up(Din, kstep_of(compute_dtype));
    p.vec_in = (Din % 8 == 0) && aligned16(h);
    p.drop_p = drop_p; p.scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    p.drop_thresh24 = (unsigned)((double)drop_p * 16777216.0);
    p.seed = seed;
    return dispatch_rowtile<false>((hipStream_t)stream, p, h_dtype, out_dtype, compute_dtype);
}

extern "C" int gcnpt_layer_bwd_data(void* stream, const void* dY, const void* Y, int g_dtype, const void* w_bwd,
                                    const int32_t* row_ptr, const int32_t* rowT_ptr, const int32_t* colT_idx, int B,
                                    int T, int Din, int H, void* dh, int dh_dtype, int compute_dtype, float scale) {
    GCNPT_REQUIRE(dY && Y && w_bwd && row_ptr && rowT_ptr && colT_idx && dh, "layer_bwd_data: null pointer");
    GCNPT_REQUIRE(B > 0 && T > 0 && Din > 0 && H > 0, "layer_bwd_data: sizes must be positive");
    GCNPT_REQUIRE(dtype_ok(g_dtype) && dtype_ok(dh_dtype) && dtype_ok(compute_dtype), "layer_bwd_data: bad dtype");
    RowTileParams p{};
    p.src = dY; p.yref = Y; p.wfrag = w_bwd; p.bias = nullptr;
    p.g_row_ptr = rowT_ptr; p.g_col_idx = colT_idx; p.d_row_ptr = row_ptr; p.out = dh;
    p.N = B * T; p.T = T; p.K = H; p.NOUT = Din; p.Kpad = round_up(H, kstep_of(compute_dtype));
    p.vec_in = (H % 8 == 0) && aligned16(dY) && aligned16(Y);
    p.scale = scale; p.drop_p = 0.0f;
    return dispatch_rowtile<true>((hipStream_t)stream, p, g_dtype, dh_dtype, compute_dtype);
}

template <typename CT, typename GT, typename HT>
static int launch_weight_grad(hipStream_t s, const WeightGradParams& p, dim3 grid) {
    hipLaunchKernelGGL((weight_grad_kernel<CT, GT, HT>), grid, dim3(LAYER_THREADS), 0, s, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_layer_bwd_weight(void* stream, const void* dY, const void* Y, int g_dtype, const void* h,
                                      int h_dtype, const int32_t* row_ptr, const int32_t* col_idx,
                                      const int32_t* deg_row_ptr, int B, int T, int Din, int H, float* dW, float* db,
                                      int compute_dtype, float scale, int zero_first) {
    GCNPT_REQUIRE(dY && Y && h && row_ptr && col_idx && dW && db, "layer_bwd_weight: null pointer");
    GCNPT_REQUIRE(B > 0 && T > 0 && Din > 0 && H > 0, "layer_bwd_weight: sizes must be positive");
    GCNPT_REQUIRE(dtype_ok(g_dtype) && dtype_ok(h_dtype) && dtype_ok(compute_dtype), "layer_bwd_weight: bad dtype");
    if (compute_dtype == GCNPT_F32 && (g_dtype != GCNPT_F32 || h_dtype != GCNPT_F32))
        return fail(GCNPT_E_UNSUPPORTED, "compute_dtype f32 needs f32 activations");
    hipStream_t s = (hipStream_t)stream;
    if (zero_first) {
        GCNPT_HIP_CHECK(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)H * Din, s));
        GCNPT_HIP_CHECK(hipMemsetAsync(db, 0, sizeof(float) * (size_t)H, s));
    }
    WeightGradParams p{};
    p.dY = dY; p.Y = Y; p.h = h; p.row_ptr = row_ptr; p.col_idx = col_idx; p.dW = dW; p.db = db;
    p.d_row_ptr = deg_row_ptr ? deg_row_ptr : row_ptr;
    p.N = B * T; p.T = T; p.Din = Din; p.H = H; p.scale = scale;
    p.vec_g = (H % 8 == 0) && aligned16(dY) && aligned16(Y);
    p.vec_h = (Din % 8 == 0) && aligned16(h);
    const int mb = ceil_div(H, WB_M), nb = ceil_div(Din, WB_N);
    int slices = max(1, min(ceil_div(p.N, WB_K), 256 / (mb * nb) > 0 ? 256 / (mb * nb) : 1));
    p.rows_per_slice = round_up(ceil_div(p.N, slices), WB_K);
    slices = ceil_div(p.N, p.rows_per_slice);
    const dim3 grid(mb, nb, slices);
    if (compute_dtype == GCNPT_F32) return launch_weight_grad<float, float, float>(s, p, grid);
    if (g_dtype == GCNPT_F32 && h_dtype == GCNPT_F32) return launch_weight_grad<bf16_t, float, float>(s, p, grid);
    if (g_dtype == GCNPT_F32 && h_dtype == GCNPT_BF16) return launch_weight_grad<bf16_t, float, bf16_t>(s, p, grid);
    if (g_dtype == GCNPT_BF16 && h_dtype == GCNPT_F32) return launch_weight_grad<bf16_t, bf16_t, float>(s, p, grid);
    return launch_weight_grad<bf16_t, bf16_t, bf16_t>(s, p, grid);
}
