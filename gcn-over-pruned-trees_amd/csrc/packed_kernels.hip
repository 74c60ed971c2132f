// Token-packed layout (gfx950): variable-length batches as [sum(len), width] rows + cu_seqlens instead of the reference's
// [B, T, width] padded to the longest sentence (data/loader.py:109-121, model/gcn.py:96-97,106).
//
// The reference's adjacency is block diagonal (one [T,T] block per sentence, model/tree.py:167-204), so a batch of pruned trees
// is ONE sparse matrix over the packed token rows: row r = cu_seqlens[b] + i for token i of sentence b, columns are packed row
// numbers too.  gcnpt_pack_trees rewrites the arrays gcnpt_prune_to_csr / gcnpt_gather_trees produced for [B, T] into that
// form (entries and their order unchanged); the layer kernels run on it unchanged with T = 0 ("columns are absolute"), on
// sum(len) rows instead of B*T.  gcnpt_pack_rows / gcnpt_unpack_rows move activations between the two layouts at the module
// boundary, so GCN.forward still returns [B, T, H] (model/gcn.py:395).
#include "layer_common.h"

namespace gcnpt {

constexpr int PK_THREADS = 256;

struct PackSrc {
    const int32_t *row_ptr, *col_idx, *label, *rowT_ptr, *colT_idx, *ell, *ellT;
    const uint8_t* pool_mask;
    const int32_t* len;
};
struct PackDst {
    int32_t *cu, *row_ptr, *col_idx, *label, *rowT_ptr, *colT_idx, *ell, *ellT;
    uint8_t* pool_mask;
    int32_t* row_sent;
    int32_t* status;
};

// sums over j < n of three per-sentence quantities at once (one round of loads, one pair of barriers); every thread gets the results
template <typename F>
__device__ __forceinline__ void block_prefix3(int n, F f, int (*scratch)[3], int (&out)[3]) {
    int v[3] = {0, 0, 0};
    for (int j = threadIdx.x; j < n; j += PK_THREADS) {
        int x[3];
        f(j, x);
        v[0] += x[0]; v[1] += x[1]; v[2] += x[2];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[q] += __shfl_xor(v[q], d);
    if ((threadIdx.x & 63) == 0) { scratch[threadIdx.x >> 6][0] = v[0]; scratch[threadIdx.x >> 6][1] = v[1]; scratch[threadIdx.x >> 6][2] = v[2]; }
    __syncthreads();
    out[0] = out[1] = out[2] = 0;
#pragma unroll
    for (int w = 0; w < PK_THREADS / WAVE; ++w) { out[0] += scratch[w][0]; out[1] += scratch[w][1]; out[2] += scratch[w][2]; }
}

// one workgroup per sentence: its row offset cu[b] and entry offsets are prefix sums over the sentences before it (B is a few
// hundred at most: every workgroup sums for itself, no second pass), then rows, ELL heads and entries are copied with the
// columns shifted by cu[b]
__global__ __launch_bounds__(PK_THREADS) void pack_trees_kernel(const PackSrc src, int B, int T, int cap, const PackDst dst, int n_rows, int nnz_cap) {
    __shared__ int scratch[PK_THREADS / WAVE][3];
    const int b = blockIdx.x, t = threadIdx.x;
    auto nnz_of = [&](const int32_t* rp, int j) { return rp[(size_t)j * (T + 1) + T] - rp[(size_t)j * (T + 1)]; };
    int pre[3];
    block_prefix3(b, [&](int j, int (&x)[3]) {
        x[0] = min(src.len[j], T);
        x[1] = nnz_of(src.row_ptr, j);
        x[2] = src.rowT_ptr ? nnz_of(src.rowT_ptr, j) : 0;
    }, scratch, pre);
    const int cu = pre[0], eo = pre[1], eoT = pre[2];
    const int len = min(src.len[b], T);
    const int nnz = nnz_of(src.row_ptr, b), nnzT = src.rowT_ptr ? nnz_of(src.rowT_ptr, b) : 0;
    const bool fits = cu + len <= n_rows && eo + nnz <= nnz_cap && eoT + nnzT <= nnz_cap;
    if (t == 0) {
        dst.cu[b] = cu;
        if (!fits) dst.status[0] = GCNPT_E_CAPACITY;                 // (status[0] was cleared by the launcher)
        if (b == B - 1) {
            dst.cu[B] = cu + len;
            dst.status[1] = cu + len;
            if (fits) {
                dst.row_ptr[cu + len] = eo + nnz;
                if (dst.rowT_ptr) dst.rowT_ptr[cu + len] = eoT + nnzT;
            }
        }
    }
    if (!fits) return;
    const int rp0 = src.row_ptr[(size_t)b * (T + 1)], rpT0 = src.rowT_ptr ? src.rowT_ptr[(size_t)b * (T + 1)] : 0;
    for (int i = t; i < len; i += PK_THREADS) {
        dst.row_ptr[cu + i] = eo + src.row_ptr[(size_t)b * (T + 1) + i] - rp0;
        if (dst.rowT_ptr) dst.rowT_ptr[cu + i] = eoT + src.rowT_ptr[(size_t)b * (T + 1) + i] - rpT0;
        dst.pool_mask[cu + i] = src.pool_mask[(size_t)b * T + i];
        dst.row_sent[cu + i] = b;
    }
    for (int k = t; k < nnz; k += PK_THREADS) {
        dst.col_idx[eo + k] = src.col_idx[rp0 + k] + cu;
        if (dst.label) dst.label[eo + k] = src.label[rp0 + k];
    }
    if (dst.colT_idx)
        for (int k = t; k < nnzT; k += PK_THREADS) dst.colT_idx[eoT + k] = src.colT_idx[rpT0 + k] + cu;
    for (int q = t; q < 2 * len; q += PK_THREADS) {                 // ELL heads, 16 bytes at a time: [count, c0..c2] / [c3..c6]
        const int i = q >> 1, half = q & 1;
        const size_t o = ((size_t)b * T + i) * 2 + half;
        const int cnt = src.ell[((size_t)b * T + i) * 8];
        int4 v = reinterpret_cast<const int4*>(src.ell)[o];
        int* e = reinterpret_cast<int*>(&v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int slot = half * 4 + j - 1;                       // entry number of this word (-1 = the count)
            if (slot >= 0) e[j] = slot < cnt ? e[j] + cu : 0;
        }
        reinterpret_cast<int4*>(dst.ell)[((size_t)(cu + i)) * 2 + half] = v;
        if (dst.ellT) {
            const int cntT = src.ellT[((size_t)b * T + i) * 8];
            int4 w = reinterpret_cast<const int4*>(src.ellT)[o];
            int* f = reinterpret_cast<int*>(&w);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int slot = half * 4 + j - 1;
                if (slot >= 0) f[j] = slot < cntT ? f[j] + cu : 0;
            }
            reinterpret_cast<int4*>(dst.ellT)[((size_t)(cu + i)) * 2 + half] = w;
        }
    }
}

// rows of a [B, T, W] tensor <-> packed [N, W]; VB = bytes per copy piece (16 or the element size); one workgroup per 8 token slots
template <int VB, bool UNPACK>
__global__ __launch_bounds__(PK_THREADS) void move_rows_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                             const int32_t* __restrict__ cu, int B, int T, size_t row_bytes) {
    const int b = blockIdx.y, i0 = blockIdx.x * 8;
    const int base = cu[b], len = cu[b + 1] - base;
    const size_t pieces = row_bytes / VB;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));  // (first-class vector: a select between uint4 structs went through scratch memory)
    typedef typename std::conditional<VB == 16, u32x4_t, typename std::conditional<VB == 4, unsigned, unsigned short>::type>::type V;
    for (int i = i0; i < min(i0 + 8, T); ++i) {
        const size_t padded = ((size_t)b * T + i) * row_bytes, packed = ((size_t)base + i) * row_bytes;
        if (UNPACK) {
            V z{};
            for (size_t q = threadIdx.x; q < pieces; q += PK_THREADS)
                *reinterpret_cast<V*>(dst + padded + q * VB) = i < len ? *reinterpret_cast<const V*>(src + packed + q * VB) : z;
        } else if (i < len) {
            for (size_t q = threadIdx.x; q < pieces; q += PK_THREADS)
                *reinterpret_cast<V*>(dst + packed + q * VB) = *reinterpret_cast<const V*>(src + padded + q * VB);
        }
    }
}

}  // namespace gcnpt

using namespace gcnpt;

extern "C" int gcnpt_pack_trees(void* stream, const int32_t* src_row_ptr, const int32_t* src_col_idx, const int32_t* src_label,
                                const int32_t* src_rowT_ptr, const int32_t* src_colT_idx, const int32_t* src_ell, const int32_t* src_ellT,
                                const uint8_t* src_pool_mask, const int32_t* len, int B, int T, int cap, int32_t* cu_seqlens,
                                int32_t* row_ptr, int32_t* col_idx, int32_t* label, int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell,
                                int32_t* ellT, uint8_t* pool_mask, int32_t* row_sent, int n_rows, int nnz_cap, int32_t* status) {
    GCNPT_REQUIRE(src_row_ptr && src_col_idx && src_ell && src_pool_mask && len, "pack_trees: null source pointer");
    GCNPT_REQUIRE(cu_seqlens && row_ptr && col_idx && ell && pool_mask && row_sent && status, "pack_trees: null destination pointer");
    GCNPT_REQUIRE(B > 0 && T > 0 && cap > 0 && n_rows > 0 && nnz_cap > 0, "pack_trees: sizes must be positive");
    GCNPT_REQUIRE(!label || src_label, "pack_trees: labels wanted but the source has none");
    GCNPT_REQUIRE((!rowT_ptr && !colT_idx && !ellT) || (rowT_ptr && colT_idx && ellT && src_rowT_ptr && src_colT_idx && src_ellT),
                  "pack_trees: the transposed pattern is all-or-nothing");
    hipStream_t s = (hipStream_t)stream;
    GCNPT_HIP_CHECK(hipMemsetAsync(status, 0, 2 * sizeof(int32_t), s));
    PackSrc a{src_row_ptr, src_col_idx, src_label, src_rowT_ptr, src_colT_idx, src_ell, src_ellT, src_pool_mask, len};
    PackDst d{cu_seqlens, row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, row_sent, status};
    hipLaunchKernelGGL(pack_trees_kernel, dim3(B), dim3(PK_THREADS), 0, s, a, B, T, cap, d, n_rows, nnz_cap);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

template <bool UNPACK>
static int move_rows(void* stream, const void* src, int dtype, const int32_t* cu, int B, int T, int W, void* dst, const char* what) {
    GCNPT_REQUIRE(src && dst && cu, "%s: null pointer", what);
    GCNPT_REQUIRE(B > 0 && T > 0 && W > 0, "%s: sizes must be positive", what);
    GCNPT_REQUIRE(dtype_ok(dtype), "%s: bad dtype", what);
    const size_t rb = (size_t)W * esize(dtype);
    const unsigned char* a = static_cast<const unsigned char*>(src);
    unsigned char* d = static_cast<unsigned char*>(dst);
    const dim3 grid(ceil_div(T, 8), B);
    hipStream_t s = (hipStream_t)stream;
    if (rb % 16 == 0 && aligned16(src) && aligned16(dst)) hipLaunchKernelGGL((move_rows_kernel<16, UNPACK>), grid, dim3(PK_THREADS), 0, s, a, d, cu, B, T, rb);
    else if (dtype == GCNPT_F32) hipLaunchKernelGGL((move_rows_kernel<4, UNPACK>), grid, dim3(PK_THREADS), 0, s, a, d, cu, B, T, rb);
    else hipLaunchKernelGGL((move_rows_kernel<2, UNPACK>), grid, dim3(PK_THREADS), 0, s, a, d, cu, B, T, rb);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_pack_rows(void* stream, const void* src, int dtype, const int32_t* cu_seqlens, int B, int T, int W, void* dst) {
    return move_rows<false>(stream, src, dtype, cu_seqlens, B, T, W, dst, "pack_rows");
}
extern "C" int gcnpt_unpack_rows(void* stream, const void* src, int dtype, const int32_t* cu_seqlens, int B, int T, int W, void* dst) {
    return move_rows<true>(stream, src, dtype, cu_seqlens, B, T, W, dst, "unpack_rows");
}
