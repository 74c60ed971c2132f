// The sentence-slice layer kernel (forward and backward-data), transform-first: reference model/gcn.py:269-271, 390-393.
//
//   forward        out = dropout(relu(((A+I)(h W^T) + 2 b) / (deg + 1)))          -- the same sum as ((A+I)h) W^T, reassociated
//   backward-data  dh  = (A+I)^T (dZ W);  side outputs for the weight gradient dW = G^T h, db = 2 sum dZ:  the rows of
//                  G = (A+I)^T dZ and dbpart = column sums of dZ (a slice stages its share of the dZ columns of all its rows in LDS)
//
// A workgroup owns (a run of WHOLE sentences, <= 128 rows) x (a slice of <= 4 of the output's 16-column tiles):
//   (1) every load it will ever need leaves at entry, nothing waits for the adjacency:
//       - the rows, as MFMA operand fragments straight into registers: wave w = (row-tile pair w & 3, k-half w >> 2) loads the 16-byte
//         pieces of ITS two row tiles for ITS half of the k-steps -- every row byte crosses the CU's L1 exactly once, no LDS copy;
//       - the weight fragments of the column slice, once, into LDS (they are the operand every wave needs);
//       - the rows' ELL heads, bias / hand-over rows;
//   (2) per column tile a wave reads the tile's fragments from LDS (conflict-free 1-KiB reads) and runs them against its register-resident
//       row fragments on the matrix cores; the two k-halves' partial sums P0, P1 (fp32) are parked in LDS;
//   (3) the aggregation over the pruned tree is a gather of P rows FROM LDS (aggregation is column-separable and sentence-local),
//       fused with the epilogue (bias twice, degree normalisation, ReLU, dropout / the hand-over to the layer below) and the row stores.
// Compared with the row-tile kernel (rowtile_body.h): no dependent HBM round trip behind the ELL heads, the weight slice is reused
// over all rows of the sentences instead of 32, and no fragment image of (A+I)h is written.  Values differ from the row-tile form by
// reassociation only (3.9e-7).  K beyond 14 k-steps is taken in chunks (the partial sums then accumulate through LDS).
#pragma once
#include "layer_common.h"

namespace gcnpt {

constexpr int SS_WAVES = 8, SS_THREADS = SS_WAVES * WAVE;
constexpr int SS_NB_INLINE = 7;      // neighbours per row that the ELL head carries (include/gcnpt.h)
constexpr int SS_CTW = 4;            // most column tiles a slice has
constexpr int SS_RTMAX = 8;          // most 16-row tiles a group has
constexpr int SS_PSTRIDE = SS_CTW * 16 + 4;      // P row stride, floats

struct SentParams {
    const void* src;        // fwd: h [N,K]      bwd: dY or dZ [N,K]
    const void* yref;       // bwd, MODE 1: Y [N,K] (stored layer output)
    const void* wfrag;      // packed B operand for [NOUT x K], gcnpt_pack_weights
    const float* bias;      // fwd: [NOUT]
    const int32_t* g_row_ptr;   // pattern gathered over (fwd: A, bwd: A^T): CSR, only read for rows with > 7 entries
    const int32_t* g_col_idx;
    const int32_t* g_ell;       // its ELL head
    const int32_t* d_ell;       // ELL head whose [8r] gives deg (always the forward pattern A)
    void* out;              // [N,NOUT]; NULL = only the side outputs below are wanted
    void* g_out;            // bwd: NULL or rows of G = (A+I)^T dZ, [N,K] of the compute type
    float* dbpart;          // bwd: NULL or [n_groups][K] column sums of dZ over each group's rows
    float* zero_p[4];       // NULL or accumulators to clear for the weight gradient that follows
    int zero_n[4];
    const void* relu_src;   // bwd: NULL, or this layer's INPUT rows [N,NOUT]: the result leaves as the layer below's dZ
    float next_scale;
    int N, T, K, NOUT;
    int R, rtn;             // rows per group (whole sentences), 16-row tiles per group
    int n_groups, n_slices, n_ct;           // column tiles of the whole output
    int ksteps, kc, ksh, n_chunks;          // k-steps in all, per chunk, per k-half of a chunk, chunks
    int p_off, z_off, meta_off;             // LDS byte offsets of P0 (P1 follows it), of the dZ share tile (bwd) and of the adjacency tables
    int zq;                 // bwd: 8-column pieces of K a slice takes for the side outputs (its share is pieces [slice * zq, ...))
    unsigned t_magic;       // ceil(2^32 / T)
    int vec_out;            // bytes per row-store piece: 16 / 8 (rows only 8-byte aligned)
    int vec_k;              // the same for the rows of G
    float scale;            // bwd: 1/(1-p) of the dropout applied to Y; fwd: 1/(1-drop_p)
    float drop_p;
    unsigned drop_thresh16;
    uint64_t seed;
    const uint64_t* seed_dev;
    unsigned long long* stamps;   // diagnostic builds only
    int knob;
};

// 8 values as bf16: 16 bytes
__device__ __forceinline__ uint4 ss_pack_bf16(const float (&v)[8]) {
    uint4 o;
    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    o.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
    o.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
    return o;
}

// 8 consecutive values of an LDS row of the compute type as floats
template <typename CT>
__device__ __forceinline__ void ss_load8(const CT* xp, float (&g)[8]) {
    if constexpr (sizeof(CT) == 2) {
        const uint4 u4 = *reinterpret_cast<const uint4*>(xp);
        const unsigned w[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) { g[2 * t] = __uint_as_float(w[t] << 16); g[2 * t + 1] = __uint_as_float(w[t] & 0xffff0000u); }
    } else {
        const float4 a = *reinterpret_cast<const float4*>(xp), b = *reinterpret_cast<const float4*>(xp + 4);
        g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = b.x; g[5] = b.y; g[6] = b.z; g[7] = b.w;
    }
}

// 8 consecutive columns of a global row: 16-byte / 8-byte pieces, or element-wise at the row's ragged end (n_valid < 8)
template <typename T>
__device__ __forceinline__ void ss_store8(T* dst, const float (&v)[8], int n_valid, int vec) {
    if (n_valid >= 8) {
        if constexpr (sizeof(T) == 2) {
            const uint4 o = ss_pack_bf16(v);
            if (vec == 16) *reinterpret_cast<uint4*>(dst) = o;
            else { reinterpret_cast<uint2*>(dst)[0] = make_uint2(o.x, o.y); reinterpret_cast<uint2*>(dst)[1] = make_uint2(o.z, o.w); }
        } else {
            reinterpret_cast<float4*>(dst)[0] = make_float4(v[0], v[1], v[2], v[3]);
            reinterpret_cast<float4*>(dst)[1] = make_float4(v[4], v[5], v[6], v[7]);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < n_valid) io<T>::store1(dst + t, v[t]);
    }
}

// MODE 0 = forward, 1 = backward-data deriving dZ = dY * 1[Y>0] * scale / (deg+1) in the loader, 2 = backward-data on ready-made dZ rows.
// VEC: 8 = rows read 16 bytes at a time (K % 8 == 0, 16-byte aligned), 4 = in 8-byte (bf16) / 16-byte (f32) halves (K % 4 == 0).
// KSH: most k-steps a wave keeps in registers (its half of a chunk).
template <typename CT, typename IT, typename OT, int MODE, int VEC, int KSH>
__global__ __launch_bounds__(SS_THREADS, 2) void sent_kernel(const SentParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool BWD = MODE != 0, MASKED = MODE == 1;
    constexpr int KSTEP = sizeof(CT) == 2 ? 32 : 16;
    constexpr int AW = 16 / (int)sizeof(CT);                    // CT elements per 16-byte fragment (bf16: 8, f32: 4)
    constexpr bool WIDE = sizeof(IT) * AW == 32;                // f32 rows feeding bf16 fragments: 32 bytes per lane and k-step
    constexpr int NWMAX = SS_CTW * 2 * KSH * 64 / SS_THREADS;   // 16-byte pieces of weight fragments a thread stages per chunk
    constexpr int EPMAX = SS_RTMAX * 16 / 64;                   // epilogue rounds: 64 rows x 8 column pieces per round

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one), each with its own L2: every slice of a group
    // runs on ONE XCD, so the group's rows cross the fabric once and the other slices read them from that L2; the same group of the
    // next layer's launch lands on the same XCD and finds the rows this one wrote (speed only; any placement gives the same values)
    const int xcd = (int)blockIdx.x & 7, bj = (int)blockIdx.x >> 3;
    const int slice = bj % p.n_slices, group = (bj / p.n_slices) * 8 + xcd;
    if (group >= p.n_groups) return;
    const int r0 = group * p.R;
    const int nrows = min(p.R, p.N - r0);
    const int ct0 = slice * p.n_ct / p.n_slices, ctn = (slice + 1) * p.n_ct / p.n_slices - ct0;
    const int col0 = ct0 * 16;
    const int RP = p.rtn * 16;
    const int rp = wave & 3, kh = wave >> 2;                    // row-tile pair, k-half
    const int arow = lane & 15, kgrp = lane >> 4;

    uint4* Wl = reinterpret_cast<uint4*>(smem);                          // [ctn][kc][64] the slice's weight fragments of the current chunk
    float* P0 = reinterpret_cast<float*>(smem + p.p_off);                // [2][RP][SS_PSTRIDE]
    CT* Zs = reinterpret_cast<CT*>(smem + p.z_off);                      // bwd: [RP][zq * 8 + 8] this slice's share of the dZ columns
    int* rell = reinterpret_cast<int*>(smem + p.meta_off);               // [RP][8] ELL heads (count, 7 sentence-local columns)
    float* rden = reinterpret_cast<float*>(rell + 8 * RP);               // [RP] deg + 1
    int* rsb = reinterpret_cast<int*>(rden + RP);                        // [RP] first row of the row's sentence, group-local
    float* sbias = reinterpret_cast<float*>(rsb + RP);                   // [64] fwd: the bias of this slice's columns

    const IT* src = static_cast<const IT*>(p.src);
    const IT* yref = static_cast<const IT*>(p.yref);
    const uint4* wfrag = static_cast<const uint4*>(p.wfrag);
    uint64_t seed_off = 0;
    if (!BWD && p.seed_dev) seed_off = *p.seed_dev;             // scalar load, consumed in the epilogue
    GCNPT_STAMP_REAL(p.stamps);
    GCNPT_STAMP(p.stamps, 0);

    // ---- (1) every load of the workgroup.  NO load is behind a condition (hipcc would put a full s_waitcnt in front of it):
    //      addresses are clamped and unwanted values dropped when they are used
    // row fragments: lane (row arow of the tile, k-group kgrp) takes AW consecutive columns of its row per k-step.  A column past the
    // row's end is clamped into the row (finite values of the row): it meets a zero weight (the packed images are zero padded past K)
    const int kmax = VEC == 8 ? p.K - AW : p.K - AW / 2;
    auto ldx = [&](const IT* base, size_t row, int k0, uint4& lo, uint4& hi) {
        const IT* q = base + row * (size_t)p.K;
        if constexpr (VEC == 8) {
            const int kc0 = min(k0, kmax);
            lo = *reinterpret_cast<const uint4*>(q + kc0);
            if constexpr (WIDE) hi = *reinterpret_cast<const uint4*>(q + kc0 + 4);
        } else {
            const int ka = min(k0, kmax), kb = min(k0 + AW / 2, kmax);
            if constexpr (WIDE) {                                 // 2 x 16 bytes
                lo = *reinterpret_cast<const uint4*>(q + ka);
                hi = *reinterpret_cast<const uint4*>(q + kb);
            } else {                                              // 2 x 8 bytes
                const uint2 u = *reinterpret_cast<const uint2*>(q + ka), v = *reinterpret_cast<const uint2*>(q + kb);
                lo = make_uint4(u.x, u.y, v.x, v.y);
            }
        }
    };
    uint4 xlo[2][KSH], xhi[WIDE ? 2 : 1][WIDE ? KSH : 1];
    uint4 ylo[MASKED ? 2 : 1][MASKED ? KSH : 1], yhi[(MASKED && WIDE) ? 2 : 1][(MASKED && WIDE) ? KSH : 1];
    int xdeg[2] = {0, 0};
    size_t xrow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) xrow[i] = (size_t)min(r0 + min(2 * rp + i, p.rtn - 1) * 16 + arow, p.N - 1);
    auto issue_x = [&](int ch) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks) {
                const int k0 = (ch * p.kc + kh * p.ksh + ks) * KSTEP + kgrp * AW;
                uint4 dummy;
                if constexpr (WIDE) ldx(src, xrow[i], k0, xlo[i][ks], xhi[i][ks]);
                else ldx(src, xrow[i], k0, xlo[i][ks], dummy);
                if constexpr (MASKED) {
                    if constexpr (WIDE) ldx(yref, xrow[i], k0, ylo[i][ks], yhi[i][ks]);
                    else ldx(yref, xrow[i], k0, ylo[i][ks], dummy);
                }
            }
    };
    issue_x(0);
    if constexpr (MASKED) {
#pragma unroll
        for (int i = 0; i < 2; ++i) xdeg[i] = p.d_ell[xrow[i] * 8];
    }

    // the slice's weight fragments of chunk ch: ctn tiles x kc k-steps x 1 KiB, contiguous per tile in the packed image
    uint4 wst[NWMAX];
    auto issue_w = [&](int ch) {
        const int nk = min(p.kc, p.ksteps - ch * p.kc);
#pragma unroll
        for (int u = 0; u < NWMAX; ++u) {
            const int f = u * SS_THREADS + tid;                   // piece f = (tile, k-step, lane) of the chunk
            const int t = (f >> 6) / p.kc, ks = (f >> 6) - t * p.kc;
            wst[u] = wfrag[((size_t)(ct0 + min(t, ctn - 1)) * p.ksteps + ch * p.kc + min(ks, nk - 1)) * 64 + (f & 63)];
        }
    };
    issue_w(0);

    int4 ellv;
    int degv;
    {
        const size_t r = (size_t)min(r0 + (tid >> 1), p.N - 1);
        ellv = reinterpret_cast<const int4*>(p.g_ell)[r * 2 + (tid & 1)];
        degv = p.d_ell[r * 8];                                                        // gcn.py:261
    }
    float bias_v = 0.0f;
    if constexpr (!BWD) bias_v = p.bias[min(col0 + (tid & 63), p.NOUT - 1)];
    // the epilogue's map: a thread owns 8 columns (piece ep_p8) of row ep_row0 + 64 u
    const int ep_p8 = tid & 7, ep_row0 = tid >> 3;
    const int ep_col = col0 + ep_p8 * 8;
    // bwd hand-over: this thread's pieces of the layer's input rows, in 8-byte halves so that rows that are only 8-byte aligned
    // (bf16 rows of 300 columns) take the same path
    constexpr int HQ = sizeof(OT) == 2 ? 2 : 4;                 // 8-byte halves per 8-column piece
    uint2 hin[BWD ? EPMAX : 1][HQ];
    // bwd side outputs: this slice's share of the dZ columns, all rows of the group: piece (row ep_row0 + 64 u, ep_p8)
    raw8<IT> zs[BWD ? EPMAX : 1], zsy[MASKED ? EPMAX : 1];
    int zsdeg[MASKED ? EPMAX : 1];
    const int q_lo = slice * p.zq, q_n = max(0, min(p.zq, ceil_div(p.K, 8) - q_lo));     // this slice's 8-column pieces of K
    const bool side = BWD && (p.g_out || p.dbpart);
    if constexpr (BWD) {
        const OT* relu = static_cast<const OT*>(p.relu_src);
        const OT* rbase = relu ? relu : static_cast<const OT*>(p.src);
#pragma unroll
        for (int u = 0; u < EPMAX; ++u) {
            const size_t r = (size_t)min(r0 + ep_row0 + 64 * u, p.N - 1);
            const size_t off = relu ? r * (size_t)p.NOUT + (size_t)min(ep_col, p.NOUT - 8) : 0;
#pragma unroll
            for (int q = 0; q < HQ; ++q) hin[u][q] = *reinterpret_cast<const uint2*>(rbase + off + q * (8 / HQ));
            const int k0 = min((q_lo + min(ep_p8, max(q_n - 1, 0))) * 8, VEC == 8 ? p.K - 8 : p.K - 4);
            if constexpr (VEC == 8) issue8<IT, true>(src, r, p.K, k0, zs[u]);
            else issue8_half<IT>(src, r, p.K, k0, zs[u]);
            if constexpr (MASKED) {
                if constexpr (VEC == 8) issue8<IT, true>(yref, r, p.K, k0, zsy[u]);
                else issue8_half<IT>(yref, r, p.K, k0, zsy[u]);
                zsdeg[u] = p.d_ell[r * 8];
            }
        }
    }
    GCNPT_STAMP(p.stamps, 1);

    // ---- park the adjacency: every row's ELL head, denominator and sentence base
    {
        const int row = tid >> 1, half = tid & 1;
        if (row < RP) {
            const int e0 = (half == 0 && row >= nrows) ? 0 : ellv.x;                 // rows past the group's end aggregate nothing
            reinterpret_cast<int4*>(rell)[row * 2 + half] = make_int4(e0, ellv.y, ellv.z, ellv.w);
            rden[row] = (float)(degv + 1);                    // both halves write it: a use under `half == 0` only would let hipcc sink the load
            rsb[row] = (int)__umulhi((unsigned)row, p.t_magic) * p.T;
        }
    }
    if constexpr (!BWD) {
        if (tid < 64) sbias[tid] = bias_v;
    }
    // bwd: the share tile (and dZ itself for the weight gradient when the loader derived it)
    const int zstride = p.zq * 8 + 8;
    if constexpr (BWD) {
        if (side && ep_p8 < q_n) {
#pragma unroll
            for (int u = 0; u < EPMAX; ++u) {
                const int row = ep_row0 + 64 * u;
                if (row >= RP) continue;
                float v[8];
                unpack8<IT>(zs[u], row < nrows, v);
                if constexpr (MASKED) {
                    float y[8];
                    unpack8<IT>(zsy[u], row < nrows, y);
                    const float inv = p.scale / (float)(zsdeg[u] + 1);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = (y[q] > 0.0f) ? v[q] * inv : 0.0f;
                }
                const int kcol = (q_lo + ep_p8) * 8;
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (kcol + q < p.K) ? v[q] : 0.0f;      // (rows read in halves: the clamped upper half)
                tile<CT>::put8(Zs + (size_t)row * zstride + ep_p8 * 8, v);
            }
        }
    }

    const bool want_out = p.out != nullptr;
    float* Ph = P0 + (kh ? (size_t)RP * SS_PSTRIDE : 0) + (size_t)(2 * rp * 16 + arow) * SS_PSTRIDE + kgrp * 4;    // this lane's slot of its k-half's sums
    const bool first = 2 * rp < p.rtn, second = 2 * rp + 1 < p.rtn;

    for (int ch = 0; ch < p.n_chunks; ++ch) {
        // ---- (2a) the chunk's weight fragments -> LDS, fragment order
        if (ch > 0) __syncthreads();                             // every wave has read the previous chunk's fragments
#pragma unroll
        for (int u = 0; u < NWMAX; ++u) {
            const int f = u * SS_THREADS + tid;
            if (f < ctn * p.kc * 64) Wl[f] = wst[u];
        }
        __syncthreads();
        if (ch == 0) GCNPT_STAMP(p.stamps, 2);
        if (ch + 1 < p.n_chunks) issue_w(ch + 1);

        // ---- (2b) P += rows . W^T[:, slice] on the matrix cores.  Swapped operands (weights as A): a lane ends up with 4 CONSECUTIVE
        //      output columns of one row
        if (want_out && first) {
            const int nkh = min(min(p.kc, p.ksteps - ch * p.kc) - kh * p.ksh, p.ksh);     // k-steps of this wave's half (may be <= 0)
            // the row fragments in the MFMA operand type; k-steps past the wave's half become zeros, so the loop below has no branch
            uint4 xf[2][KSH];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks) {
                    uint4 f;
                    if constexpr (!MASKED && sizeof(IT) == sizeof(CT)) {
                        f = xlo[i][ks];
                    } else {
                        float v[8], y[8];
                        raw8<IT> rx, ry;
                        rx.a = xlo[i][ks];
                        if constexpr (WIDE) rx.b = xhi[i][ks];
                        unpack8<IT>(rx, true, v);                 // (f32 fragments use the first 4)
                        if constexpr (MASKED) {
                            ry.a = ylo[i][ks];
                            if constexpr (WIDE) ry.b = yhi[i][ks];
                            unpack8<IT>(ry, true, y);
                            const float inv = p.scale / (float)(xdeg[i] + 1);
#pragma unroll
                            for (int q = 0; q < AW; ++q) v[q] = (y[q] > 0.0f) ? v[q] * inv : 0.0f;
                        }
                        if constexpr (sizeof(CT) == 2) f = ss_pack_bf16(v);
                        else f = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
                    }
                    xf[i][ks] = ks < nkh ? f : make_uint4(0, 0, 0, 0);
                }
            if (ch + 1 < p.n_chunks) issue_x(ch + 1);
            const uint4* wk = Wl + (size_t)(kh * p.ksh) * 64 + lane;
            auto read_w = [&](int c, uint4 (&dst)[KSH]) {
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks) dst[ks] = wk[((size_t)c * p.kc + min(ks, max(nkh - 1, 0))) * 64];
            };
            auto col_tile = [&](int c, const uint4 (&cur)[KSH], uint4 (&nxt)[KSH], auto first_chunk) {
                read_w(min(c + 1, ctn - 1), nxt);
                f32x4_t acc[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[i] = (f32x4_t){0, 0, 0, 0};
                    if constexpr (!decltype(first_chunk)::value) {
                        const float4 o = *reinterpret_cast<const float4*>(Ph + (size_t)i * 16 * SS_PSTRIDE + min(c, ctn - 1) * 16);
                        acc[i] = (f32x4_t){o.x, o.y, o.z, o.w};
                    }
                }
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if constexpr (sizeof(CT) == 2) {
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, cur[ks]),
                                                                             __builtin_bit_cast(bf16x8_t, xf[i][ks]), acc[i], 0, 0, 0);
                        } else {
                            const f32x4_t bq = __builtin_bit_cast(f32x4_t, cur[ks]), aq = __builtin_bit_cast(f32x4_t, xf[i][ks]);
#pragma unroll
                            for (int s = 0; s < 4; ++s) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[s], aq[s], acc[i], 0, 0, 0);
                        }
                    }
                if (c < ctn) {                  // (the second tile of the last pair may be a clamped duplicate: not stored)
                    *reinterpret_cast<float4*>(Ph + c * 16) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
                    if (second) *reinterpret_cast<float4*>(Ph + (size_t)16 * SS_PSTRIDE + c * 16) = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
                }
            };
            uint4 a[KSH], b[KSH];
            read_w(0, a);
            if (ch == 0) {
                for (int c = 0; c < ctn; c += 2) { col_tile(c, a, b, std::true_type{}); col_tile(c + 1, b, a, std::true_type{}); }
            } else {
                for (int c = 0; c < ctn; c += 2) { col_tile(c, a, b, std::false_type{}); col_tile(c + 1, b, a, std::false_type{}); }
            }
        }
    }
    GCNPT_STAMP(p.stamps, 3);

#pragma unroll
    for (int z = 0; z < 4; ++z)
        if (p.zero_p[z])
            for (int i = (int)blockIdx.x * SS_THREADS + tid; i < p.zero_n[z]; i += (int)gridDim.x * SS_THREADS) p.zero_p[z][i] = 0.0f;
    __syncthreads();
    GCNPT_STAMP(p.stamps, 4);

    auto csr_begin = [&](int row) {                                 // > 7 entries: the row continues in the CSR
        const size_t rg = (size_t)(r0 + row);
        return p.g_row_ptr[(rg / p.T) * (p.T + 1) + (rg - rg / p.T * p.T)];
    };
    auto nbr_of = [&](int row, int e, int beg) {                    // entry e of a row, e < its count
        return e < SS_NB_INLINE ? rell[row * 8 + 1 + e] : p.g_col_idx[beg + e];
    };

    // ---- bwd side outputs from the share tile: G = dZ + the sum over the row's entries of dZ, column sums of dZ
    if constexpr (BWD) {
        if (side) {
            if (p.g_out && ep_p8 < q_n) {
                CT* G = static_cast<CT*>(p.g_out);
                for (int row = ep_row0; row < nrows; row += 64) {
                    float g[8];
                    ss_load8<CT>(Zs + (size_t)row * zstride + ep_p8 * 8, g);
                    const int n = rell[row * 8];
                    if (n > 0) {
                        const int sb = rsb[row], beg = n > SS_NB_INLINE ? csr_begin(row) : 0;
                        for (int e = 0; e < n; ++e) {
                            float w[8];
                            ss_load8<CT>(Zs + (size_t)(sb + nbr_of(row, e, beg)) * zstride + ep_p8 * 8, w);
#pragma unroll
                            for (int t = 0; t < 8; ++t) g[t] += w[t];
                        }
                    }
                    const int kcol = (q_lo + ep_p8) * 8;
                    ss_store8<CT>(G + (size_t)(r0 + row) * p.K + kcol, g, p.K - kcol, p.vec_k);
                }
            }
            if (p.dbpart) {
                for (int c = tid; c < q_n * 8; c += SS_THREADS) {
                    float s = 0.0f;
                    for (int row = 0; row < nrows; ++row) s += io<CT>::load1(Zs + (size_t)row * zstride + c);
                    if (q_lo * 8 + c < p.K) p.dbpart[(size_t)group * p.K + q_lo * 8 + c] = s;
                }
            }
        }
    }
    if (!want_out) return;

    // ---- (3) aggregation from LDS + epilogue + row stores
    OT* out = static_cast<OT*>(p.out);
    const size_t p1_off = (size_t)RP * SS_PSTRIDE;
    const bool col_live = ep_p8 < ctn * 2 && ep_col < p.NOUT;
#pragma unroll
    for (int u = 0; u < EPMAX; ++u) {
        const int row = ep_row0 + 64 * u;
        if (row >= nrows || !col_live) continue;
        const float* q = P0 + (size_t)row * SS_PSTRIDE + ep_p8 * 8;
        float v[8];
        {
            const float4 a0 = *reinterpret_cast<const float4*>(q), a1 = *reinterpret_cast<const float4*>(q + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(q + p1_off), b1 = *reinterpret_cast<const float4*>(q + p1_off + 4);
            v[0] = a0.x + b0.x; v[1] = a0.y + b0.y; v[2] = a0.z + b0.z; v[3] = a0.w + b0.w;      // the explicit W(h) term, gcn.py:271
            v[4] = a1.x + b1.x; v[5] = a1.y + b1.y; v[6] = a1.z + b1.z; v[7] = a1.w + b1.w;
        }
        const int n = rell[row * 8];
        if (n > 0) {                                                                    // gcn.py:269 as a gather: ~1 row in 8 of a pruned tree
            const int sb = rsb[row], beg = n > SS_NB_INLINE ? csr_begin(row) : 0;
            for (int e = 0; e < n; ++e) {
                const float* qn = P0 + (size_t)(sb + nbr_of(row, e, beg)) * SS_PSTRIDE + ep_p8 * 8;
                const float4 a0 = *reinterpret_cast<const float4*>(qn), a1 = *reinterpret_cast<const float4*>(qn + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(qn + p1_off), b1 = *reinterpret_cast<const float4*>(qn + p1_off + 4);
                v[0] += a0.x + b0.x; v[1] += a0.y + b0.y; v[2] += a0.z + b0.z; v[3] += a0.w + b0.w;
                v[4] += a1.x + b1.x; v[5] += a1.y + b1.y; v[6] += a1.z + b1.z; v[7] += a1.w + b1.w;
            }
        }
        const float den = rden[row];
        if constexpr (!BWD) {
            const float inv = 1.0f / den;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float x = div_by(v[k] + 2.0f * sbias[ep_p8 * 8 + k], den, inv);  // gcn.py:270-271 (the bias enters twice), 390
                v[k] = x > 0.0f ? x : 0.0f;                                            // gcn.py:392
            }
            if (p.drop_p > 0.0f) {                                                    // gcn.py:393: one hash per column pair
#pragma unroll
                for (int h2 = 0; h2 < 4; ++h2) {
                    const unsigned dh = drop_hash(p.seed + seed_off, (unsigned)(r0 + row), (unsigned)(ep_col >> 1) + h2);
                    v[2 * h2] = drop_keep(dh, 0u, p.drop_thresh16) ? v[2 * h2] * p.scale : 0.0f;
                    v[2 * h2 + 1] = drop_keep(dh, 1u, p.drop_thresh16) ? v[2 * h2 + 1] * p.scale : 0.0f;
                }
            }
        } else if (p.relu_src) {
            // hand-over to the layer below: its dZ instead of dh (gcn.py:390-393 differentiated where the rows are at hand)
            const float f = p.next_scale / den;
            if (ep_col + 8 <= p.NOUT) {
                float hx[8];
                if constexpr (sizeof(OT) == 2) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        hx[4 * h] = __uint_as_float(hin[u][h].x << 16); hx[4 * h + 1] = __uint_as_float(hin[u][h].x & 0xffff0000u);
                        hx[4 * h + 2] = __uint_as_float(hin[u][h].y << 16); hx[4 * h + 3] = __uint_as_float(hin[u][h].y & 0xffff0000u);
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < 4; ++h) { hx[2 * h] = __uint_as_float(hin[u][h].x); hx[2 * h + 1] = __uint_as_float(hin[u][h].y); }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = hx[k] > 0.0f ? v[k] * f : 0.0f;
            } else {                                        // the row's last, partial piece (its prefetch was clamped): element-wise
                const OT* relu = static_cast<const OT*>(p.relu_src);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float hv = io<OT>::load1(relu + (size_t)(r0 + row) * p.NOUT + min(ep_col + k, p.NOUT - 1));
                    v[k] = hv > 0.0f ? v[k] * f : 0.0f;
                }
            }
        }
        ss_store8<OT>(out + (size_t)(r0 + row) * p.NOUT + ep_col, v, p.NOUT - ep_col, p.vec_out);
    }
    GCNPT_STAMP(p.stamps, 5);
}

}  // namespace gcnpt
