// Sentence-resident GCN stack for gfx950: ALL layers of reference model/gcn.py:266-393 in one launch per direction.
//
// Nothing in the layer loop crosses a sentence (block-diagonal adjacency, per-sentence degrees), so a workgroup that
// owns a WHOLE sentence (T <= 112 token rows; TACRED's longest sentence has 96) runs layer after layer with the
// inter-layer activations resident in LDS and no grid-wide synchronisation.  Aggregation uses linearity,
//      ((A+I) h) W^T = (A+I) (h W^T),
// so the matrix cores run on the raw rows first (nothing depends on the adjacency), the fp32 product tile P is parked
// in LDS, and the neighbour sum happens LDS -> LDS in the epilogue: no kernel here ever gathers rows from HBM.
//   forward  per layer: A-tile (x or h_l, bf16, LDS) --MFMA--> P (fp32, LDS) --epilogue: sum over row pattern, +2b,
//            /(deg+1), ReLU, dropout--> h_{l+1} (HBM copy for backward + LDS tile for the next layer)
//   backward per layer: Z = dZ_l (bf16, LDS) --LDS gather over the transposed pattern--> G = (A+I)^T dZ_l --MFMA with W-->
//            dh_l (accumulators) --elementwise--> dZ_{l-1} straight back into the Z tile (dh never touches HBM)
// The weight gradient uses dW = dZ^T (A+I) h = G^T h: its two operands are the raw input tile (saved by forward) and
// the gathered tile G (saved by backward), both as fragment images with per-sentence k-steps; db = 2 sum dZ is added
// by the backward kernel itself.
// bf16 MFMA operands / fp32 accumulate only (fp32 tiles of 112 rows do not fit LDS); the per-layer kernels of
// rowtile_kernels.hip remain the general path (long sentences, fp32, wide layers).
#include "layer_common.h"

namespace gcnpt {

constexpr int ST_THREADS = 512, ST_WAVES = 8, ST_MT = 7, ST_ROWS = ST_MT * 16, ST_MAXL = 4, ST_RING = 4, ST_NB = 7;
constexpr int ST_RPW = ST_ROWS / ST_WAVES;          // 14 rows per wave in the row-wise phases

// LDS row strides: A tiles are read as MFMA fragments (conflict-free ds_read_b128); P is fp32 with 16-byte rows
__host__ __device__ inline int st_astride(int kpad) { return lds_stride_dw(kpad / 2) * 2; }          // bf16 elements
__host__ __device__ inline int st_pstride(int width) { return round_up(width, 16) + 4; }             // floats

struct StackFwdParams {
    int L, B, T, Din, H;
    const void* x;                       // [B*T, Din]
    const uint4* wf[ST_MAXL];            // packed forward weights (gcnpt_pack_weights)
    const float* bias[ST_MAXL];
    void* h_out[ST_MAXL];                // h_{l+1} [B*T, H]: bf16 for l < L-1, out dtype for the last
    uint4* h_frag[ST_MAXL];              // NULL or fragment image of the layer INPUT h_l (per-sentence k-steps)
    float* zero[2 * ST_MAXL];            // NULL or accumulators (dW_l, db_l) cleared for the backward kernels
    int zero_n[2 * ST_MAXL];
    const int32_t* g_ell; const int32_t* d_ell; const int32_t* row_ptr; const int32_t* col_idx;
    float drop_p[ST_MAXL], drop_scale[ST_MAXL];
    unsigned thresh16[ST_MAXL];
    uint64_t seed[ST_MAXL];
    const uint64_t* seed_dev;   // NULL or a device word added to every seed
    int vec_h, vec_out;
    unsigned long long* stamps;   // diagnostic builds only
};

struct StackBwdParams {
    int L, B, T, Din, H;
    const void* dY;                      // [B*T, H] gradient of the stack output
    const void* Y[ST_MAXL];              // h_{l+1}: bf16 for l < L-1, g dtype for the last
    const uint4* wb[ST_MAXL];            // packed backward weights
    void* dx;                            // NULL or [B*T, Din]
    uint4* g_frag[ST_MAXL];              // NULL or fragment image of G_l = (A+I)^T dZ_l
    float* db[ST_MAXL];                  // NULL or bias gradients (added into; cleared by stack_fwd)
    const int32_t* d_ell; const int32_t* gT_ell; const int32_t* rowT_ptr; const int32_t* colT_idx;
    float scale[ST_MAXL];                // 1/(1-p) of the dropout applied to h_{l+1}
    int vec_g, vec_y;
    unsigned long long* stamps;   // diagnostic builds only
};

// fragment image (include/gcnpt.h) of the sentence's tile X [ST_ROWS][stride] bf16 (rows >= T are zero; rows >= ST_ROWS
// do not exist and read as zero): k-step index = b * kss + s, 32 rows each
__device__ __forceinline__ void st_emit_frags(const bf16_t* X, int stride, int width, uint4* F, int b, int kss, int nks,
                                              int lane, int wave) {
    const int w_tiles = ceil_div(width, 16);
    const int i = lane & 15, g = lane >> 4, q4 = i >> 2, pp = i & 3;
    for (int ts = wave; ts < w_tiles * kss; ts += ST_WAVES) {
        const int t = ts / kss, s = ts - t * kss;
        const int rlo = 32 * s + 8 * g + q4, rhi = rlo + 4;
        const bool in_lo = 32 * s + 8 * g < ST_ROWS, in_hi = 32 * s + 8 * g + 4 < ST_ROWS;   // whole 4-row blocks are in or out
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(X + (size_t)min(rlo, ST_ROWS - 1) * stride + 16 * t + 4 * pp));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(X + (size_t)min(rhi, ST_ROWS - 1) * stride + 16 * t + 4 * pp));
        uint4 u;
        u.x = in_lo ? ((unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16)) : 0u;
        u.y = in_lo ? ((unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16)) : 0u;
        u.z = in_hi ? ((unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16)) : 0u;
        u.w = in_hi ? ((unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16)) : 0u;
        F[((size_t)t * nks + (size_t)b * kss + s) * 64 + lane] = u;
    }
}

// weights of the wave's two output tiles stream through a 4-deep register ring (prefetch distance: 4 k-steps = 56 MFMAs)
struct WRing { uint4 r[ST_RING][2]; };

__device__ __forceinline__ void st_ring_load(WRing& w, const uint4* wf, int ksteps, int tl0, int tl1, int lane) {
#pragma unroll
    for (int s = 0; s < ST_RING; ++s) {
        const int kk = min(s, ksteps - 1);
        w.r[s][0] = wf[((size_t)tl0 * ksteps + kk) * 64 + lane];
        w.r[s][1] = wf[((size_t)tl1 * ksteps + kk) * 64 + lane];
    }
}

// acc[mt][j] += A[112 x Kpad] * Wfrag.  Weights are the MFMA "A" operand, so lane (i = lane & 15, q = lane >> 4) ends
// up with row 16 mt + i and the 4 CONSECUTIVE columns 16 tile + 4q .. 4q+3 of the product.
__device__ __forceinline__ void st_mfma(const bf16_t* A, int stride, int ksteps, const uint4* wf, int tl0, int tl1, int lane,
                                        WRing& w, f32x4_t (&acc)[ST_MT][2]) {
    const int arow = lane & 15, kgrp = lane >> 4;
    uint4 a_cur[ST_MT], a_nxt[ST_MT];
    auto read_a = [&](int kk, uint4 (&dst)[ST_MT]) {
#pragma unroll
        for (int mt = 0; mt < ST_MT; ++mt)
            dst[mt] = *reinterpret_cast<const uint4*>(A + (size_t)(mt * 16 + arow) * stride + kk * 32 + kgrp * 8);
    };
    read_a(0, a_cur);
    for (int kg = 0; kg < ksteps; kg += ST_RING) {
#pragma unroll
        for (int s = 0; s < ST_RING; ++s) {
            if (kg + s < ksteps) {                                   // wave-uniform; no global load is decided by it
                read_a(min(kg + s + 1, ksteps - 1), a_nxt);
                const bf16x8_t b0 = __builtin_bit_cast(bf16x8_t, w.r[s][0]), b1 = __builtin_bit_cast(bf16x8_t, w.r[s][1]);
#pragma unroll
                for (int mt = 0; mt < ST_MT; ++mt) {
                    const bf16x8_t a = __builtin_bit_cast(bf16x8_t, a_cur[mt]);
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, a, acc[mt][0], 0, 0, 0);
                    acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, a, acc[mt][1], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < ST_MT; ++mt) a_cur[mt] = a_nxt[mt];
            }
            const int kk = min(kg + s + ST_RING, ksteps - 1);         // refill the slot for the next round: clamped, unconditional
            w.r[s][0] = wf[((size_t)tl0 * ksteps + kk) * 64 + lane];
            w.r[s][1] = wf[((size_t)tl1 * ksteps + kk) * 64 + lane];
        }
    }
}

// rows of a [B*T, K] matrix -> bf16 A-tile in LDS (rows >= T and columns >= K are zero).  Wave w owns rows w, w+8, ...;
// lane = 16-byte chunk; every load of a batch is in flight before the first is used.  f(v, row) post-processes a row.
template <typename XT, bool VEC, typename F>
__device__ __forceinline__ void st_stage_rows(const XT* x, size_t row0, int T, int K, int kpad, bf16_t* A, int stride, int lane,
                                              int wave, F f) {
    constexpr int RB = sizeof(XT) == 2 ? ST_RPW : ST_RPW / 2;       // rows in flight per lane (4 / 8 VGPRs each)
    const int nchunk = kpad / 8;
    for (int c0 = 0; c0 < nchunk; c0 += WAVE) {                      // one trip unless K > 512
        const int k0 = (c0 + lane) * 8;
        const bool in = k0 < kpad, live = k0 < K;
        const int kc = min(k0, VEC ? K - 8 : K - 1);
        for (int j0 = 0; j0 < ST_RPW; j0 += RB) {
            raw8<XT> raw[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int row = wave + ST_WAVES * (j0 + j);
                issue8<XT, VEC>(x, row0 + min(row, T - 1), K, kc, raw[j]);
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int row = wave + ST_WAVES * (j0 + j);
                float v[8];
                unpack8<XT>(raw[j], live && row < T, v);
                f(v, row, k0);
                if (in) tile<bf16_t>::put8(A + (size_t)row * stride + k0, v);
            }
        }
    }
}

// ELL heads of the sentence's rows (count + 7 columns) -> LDS, rows >= T read as "no entries"
__device__ __forceinline__ void st_stage_ell(const int32_t* ell, size_t row0, int T, int* dst, int tid) {
    for (int i = tid; i < ST_ROWS * 2; i += ST_THREADS) {
        const int row = i >> 1, half = i & 1;
        const int4 e = reinterpret_cast<const int4*>(ell)[(row0 + min(row, T - 1)) * 2 + half];
        reinterpret_cast<int4*>(dst)[row * 2 + half] = make_int4((half == 0 && row >= T) ? 0 : e.x, e.y, e.z, e.w);
    }
}

// =====================================================================================================
// forward
// =====================================================================================================
template <typename IT, typename OT, bool VECX>
__global__ __launch_bounds__(ST_THREADS, 2) void stack_fwd_kernel(const StackFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, T = p.T, H = p.H;
    const size_t row0 = (size_t)b * T;
    const int hpad = round_up(H, 32), kpad0 = round_up(p.Din, 32);
    const int astride0 = st_astride(kpad0), hstride = st_astride(hpad), pstride = st_pstride(H);
    const size_t r0_bytes = max((size_t)ST_ROWS * astride0 * sizeof(bf16_t), (size_t)ST_ROWS * pstride * sizeof(float));
    bf16_t* X = reinterpret_cast<bf16_t*>(smem_raw);                       // region 0: layer-0 input tile, then P
    float* P = reinterpret_cast<float*>(smem_raw);
    bf16_t* Ht = reinterpret_cast<bf16_t*>(smem_raw + r0_bytes);           // region 1: input tile of layers >= 1
    int* rell = reinterpret_cast<int*>(Ht + (size_t)ST_ROWS * hstride);    // [ST_ROWS][8]
    float* rden = reinterpret_cast<float*>(rell + ST_ROWS * 8);            // [ST_ROWS] deg + 1
    const int n_tiles = ceil_div(H, 16);
    const int tl0 = min(wave, n_tiles - 1), tl1 = min(wave + ST_WAVES, n_tiles - 1);
    const int kss = ceil_div(T, 32), nks = p.B * kss;

    GCNPT_STAMP_REAL(p.stamps);
    GCNPT_STAMP(p.stamps, 0);
    WRing ring;
    st_ring_load(ring, p.wf[0], kpad0 / 32, tl0, tl1, lane);              // nothing below depends on these: issued first
    st_stage_ell(p.g_ell, row0, T, rell, tid);
    for (int row = tid; row < ST_ROWS; row += ST_THREADS)
        rden[row] = (float)(p.d_ell[(row0 + min(row, T - 1)) * 8] + 1);   // gcn.py:261
    st_stage_rows<IT, VECX>(static_cast<const IT*>(p.x), row0, T, p.Din, kpad0, X, astride0, lane, wave,
                            [](float (&)[8], int, int) {});
    for (int z = 0; z < 2 * ST_MAXL; ++z)                                  // accumulators of the backward kernels
        if (p.zero[z])
            for (int i = b * ST_THREADS + tid; i < p.zero_n[z]; i += gridDim.x * ST_THREADS) p.zero[z][i] = 0.0f;
    GCNPT_STAMP(p.stamps, 1);
    __syncthreads();
    GCNPT_STAMP(p.stamps, 2);

    for (int l = 0; l < p.L; ++l) {
        const bool last = l == p.L - 1;
        const int K = l == 0 ? p.Din : H;
        const int kpad = round_up(K, 32), ksteps = kpad / 32;
        const bf16_t* A = l == 0 ? X : Ht;
        const int astride = l == 0 ? astride0 : hstride;

        if (p.h_frag[l]) st_emit_frags(A, astride, K, p.h_frag[l], b, kss, nks, lane, wave);
        GCNPT_STAMP(p.stamps, 3 + 4 * l);

        f32x4_t acc[ST_MT][2];
#pragma unroll
        for (int mt = 0; mt < ST_MT; ++mt) { acc[mt][0] = (f32x4_t){0, 0, 0, 0}; acc[mt][1] = (f32x4_t){0, 0, 0, 0}; }
        st_mfma(A, astride, ksteps, p.wf[l], tl0, tl1, lane, ring, acc);
        if (!last) st_ring_load(ring, p.wf[l + 1], hpad / 32, tl0, tl1, lane);      // next layer's weights fly during the epilogue
        GCNPT_STAMP(p.stamps, 4 + 4 * l);
        __syncthreads();                                                             // everyone is done reading region 0 / Ht
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (wave + j * ST_WAVES < n_tiles) {
                const int col0 = (wave + j * ST_WAVES) * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int mt = 0; mt < ST_MT; ++mt)
                    *reinterpret_cast<f32x4_t*>(P + (size_t)(mt * 16 + (lane & 15)) * pstride + col0) = acc[mt][j];
            }
        }
        __syncthreads();
        GCNPT_STAMP(p.stamps, 5 + 4 * l);

        // ---- epilogue: out[r,:] = dropout(relu((sum_{c in row r} P[c,:] + P[r,:] + 2 b) / (deg + 1)))   gcn.py:269-271, 390-393
        // half-wave = one row, lane = 8 consecutive columns: stores to HBM are whole rows, 16/32 bytes per lane
        const int sub = lane >> 5, k0 = (lane & 31) * 8;
        const bool col_live = k0 < H;
        float b2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) b2[q] = 2.0f * p.bias[l][min(k0 + q, H - 1)];    // bias enters twice, gcn.py:270-271
        for (int j0 = 0; j0 < ST_RPW; j0 += 2) {
            const int row = wave + ST_WAVES * (j0 + sub), other = wave + ST_WAVES * (j0 + 1 - sub);
            const bool row_live = row < T;
            const int n = row_live ? rell[row * 8] : 0;
            const int nmax = max(n, other < T ? rell[other * 8] : 0);
            const int kc = min(k0, round_up(H, 16) - 8);
            float s[8];
            {
                const float4 u = *reinterpret_cast<const float4*>(P + (size_t)row * pstride + kc);
                const float4 v = *reinterpret_cast<const float4*>(P + (size_t)row * pstride + kc + 4);
                s[0] = u.x; s[1] = u.y; s[2] = u.z; s[3] = u.w; s[4] = v.x; s[5] = v.y; s[6] = v.z; s[7] = v.w;   // the explicit W(h) term
            }
            for (int e = 0; e < nmax; ++e) {
                const bool on = e < n;
                int col = rell[row * 8 + 1 + min(e, ST_NB - 1)];
                if (e >= ST_NB) col = p.col_idx[p.row_ptr[(size_t)b * (T + 1) + min(row, T - 1)] + (on ? e : 0)];   // hubs: continue in the CSR
                const int c = on ? col : row;
                const float4 u = *reinterpret_cast<const float4*>(P + (size_t)c * pstride + kc);
                const float4 v = *reinterpret_cast<const float4*>(P + (size_t)c * pstride + kc + 4);
                if (on) { s[0] += u.x; s[1] += u.y; s[2] += u.z; s[3] += u.w; s[4] += v.x; s[5] += v.y; s[6] += v.z; s[7] += v.w; }
            }
            const float den = rden[row], inv = 1.0f / den;
            float o[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float xv = div_by(s[q] + b2[q], den, inv);           // gcn.py:390
                o[q] = (xv > 0.0f && col_live && k0 + q < H && row_live) ? xv : 0.0f;    // gcn.py:392; pad rows / columns stay 0
            }
            if (p.drop_p[l] > 0.0f) {                                       // gcn.py:393
#pragma unroll
                for (int h2 = 0; h2 < 4; ++h2) {
                    const unsigned dh = drop_hash(p.seed[l] + (p.seed_dev ? *p.seed_dev : 0ull), (unsigned)(row0 + row), (unsigned)(k0 >> 1) + h2);
                    o[2 * h2] = drop_keep(dh, 0u, p.thresh16[l]) ? o[2 * h2] * p.drop_scale[l] : 0.0f;
                    o[2 * h2 + 1] = drop_keep(dh, 1u, p.thresh16[l]) ? o[2 * h2 + 1] * p.drop_scale[l] : 0.0f;
                }
            }
            if (!last) {
                if (k0 < hpad) tile<bf16_t>::put8(Ht + (size_t)row * hstride + k0, o);       // next layer's input tile (zeros in the padding)
                bf16_t* out = static_cast<bf16_t*>(p.h_out[l]) + (row0 + row) * H + k0;
                if (row_live && col_live) {
                    if (p.vec_h) tile<bf16_t>::put8(out, o);
                    else for (int q = 0; q < 8; ++q) if (k0 + q < H) out[q] = f32_to_bf16(o[q]);
                }
            } else if (row_live && col_live) {
                OT* out = static_cast<OT*>(p.h_out[l]) + (row0 + row) * H + k0;
                if (p.vec_out) tile<OT>::put8(out, o);
                else for (int q = 0; q < 8; ++q) if (k0 + q < H) io<OT>::store1(out + q, o[q]);
            }
        }
        GCNPT_STAMP(p.stamps, 6 + 4 * l);
        __syncthreads();                                                     // Ht complete, P dead
    }
}

// =====================================================================================================
// backward
// =====================================================================================================
template <typename GT, typename XT, bool VECG>
__global__ __launch_bounds__(ST_THREADS, 2) void stack_bwd_kernel(const StackBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, T = p.T, H = p.H;
    const size_t row0 = (size_t)b * T;
    const int hpad = round_up(H, 32), zstride = st_astride(hpad), hsteps = hpad / 32;
    bf16_t* Z = reinterpret_cast<bf16_t*>(smem_raw);                       // dZ_l of the sentence
    bf16_t* G = Z + (size_t)ST_ROWS * zstride;                             // (A+I)^T dZ_l
    int* rellT = reinterpret_cast<int*>(G + (size_t)ST_ROWS * zstride);    // [ST_ROWS][8] transposed pattern
    float* rden = reinterpret_cast<float*>(rellT + ST_ROWS * 8);           // [ST_ROWS] deg + 1
    const int kss = ceil_div(T, 32), nks = p.B * kss;
    const int Ltop = p.L - 1;

    GCNPT_STAMP_REAL(p.stamps);
    GCNPT_STAMP(p.stamps, 0);
    WRing ring;
    {
        const int nt = ceil_div(Ltop == 0 ? p.Din : H, 16);
        st_ring_load(ring, p.wb[Ltop], hsteps, min(wave, nt - 1), min(wave + ST_WAVES, nt - 1), lane);
    }
    st_stage_ell(p.gT_ell, row0, T, rellT, tid);
    for (int row = tid; row < ST_ROWS; row += ST_THREADS)
        rden[row] = (float)(p.d_ell[(row0 + min(row, T - 1)) * 8] + 1);
    // dZ_{L-1} = dY * 1[Y > 0] * scale / (deg + 1): dY and Y rows are read straight, deg per row from the ELL head
    {
        const GT* dY = static_cast<const GT*>(p.dY);
        const GT* Y = static_cast<const GT*>(p.Y[Ltop]);
        constexpr int RB = sizeof(GT) == 2 ? ST_RPW / 2 : 3;               // two streams per row
        const int k0 = lane * 8;
        const bool in = k0 < hpad, live = k0 < H;
        const int kc = min(k0, VECG ? H - 8 : H - 1);
        for (int j0 = 0; j0 < ST_RPW; j0 += RB) {
            raw8<GT> rg[RB], ry[RB];
            float inv[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int row = wave + ST_WAVES * min(j0 + j, ST_RPW - 1);
                const size_t r = row0 + min(row, T - 1);
                issue8<GT, VECG>(dY, r, H, kc, rg[j]);
                issue8<GT, VECG>(Y, r, H, kc, ry[j]);
                inv[j] = p.scale[Ltop] / (float)(p.d_ell[r * 8] + 1);
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                if (j0 + j >= ST_RPW) continue;
                const int row = wave + ST_WAVES * (j0 + j);
                float g[8], y[8];
                unpack8<GT>(rg[j], live && row < T, g);
                unpack8<GT>(ry[j], live && row < T, y);
#pragma unroll
                for (int q = 0; q < 8; ++q) g[q] = y[q] > 0.0f ? g[q] * inv[j] : 0.0f;
                if (in) tile<bf16_t>::put8(Z + (size_t)row * zstride + k0, g);
            }
        }
    }
    GCNPT_STAMP(p.stamps, 1);
    __syncthreads();

    for (int l = Ltop; l >= 0; --l) {
        const int K = l == 0 ? p.Din : H;                                   // width of dh_l
        GCNPT_STAMP(p.stamps, 2 + 4 * (Ltop - l));
        const int n_tiles = ceil_div(K, 16);

        // ---- db_l += 2 sum_r dZ_l[r,:]   (bias enters twice)
        if (p.db[l])
            for (int c = tid; c < H; c += ST_THREADS) {
                float s = 0.0f;
                for (int r = 0; r < T; ++r) s += bf16_to_f32(Z[(size_t)r * zstride + c]);
                atomicAdd(p.db[l] + c, 2.0f * s);
            }

        GCNPT_STAMP(p.stamps, 3 + 4 * (Ltop - l));
        // ---- G = (A+I)^T dZ_l : LDS -> LDS over the transposed pattern (half-wave = row, lane = 8 columns)
        {
            const int sub = lane >> 5, k0 = (lane & 31) * 8;
            for (int j0 = 0; j0 < ST_RPW; j0 += 2) {
                const int row = wave + ST_WAVES * (j0 + sub), other = wave + ST_WAVES * (j0 + 1 - sub);
                const int n = row < T ? rellT[row * 8] : 0;
                const int nmax = max(n, other < T ? rellT[other * 8] : 0);
                const int kc = min(k0, hpad - 8);
                float s[8], v[8];
                io<bf16_t>::load8(Z + (size_t)row * zstride + kc, s);
                for (int e = 0; e < nmax; ++e) {
                    const bool on = e < n;
                    int col = rellT[row * 8 + 1 + min(e, ST_NB - 1)];
                    if (e >= ST_NB) col = p.colT_idx[p.rowT_ptr[(size_t)b * (T + 1) + min(row, T - 1)] + (on ? e : 0)];
                    io<bf16_t>::load8(Z + (size_t)(on ? col : row) * zstride + kc, v);
                    if (on) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) s[q] += v[q];
                    }
                }
                if (k0 < hpad) tile<bf16_t>::put8(G + (size_t)row * zstride + k0, s);
            }
        }
        __syncthreads();
        GCNPT_STAMP(p.stamps, 4 + 4 * (Ltop - l));
        if (p.g_frag[l]) st_emit_frags(G, zstride, H, p.g_frag[l], b, kss, nks, lane, wave);
        if (l == 0 && !p.dx) break;                                          // the stack input needs no gradient

        // ---- dh_l = G W_l, 16 output tiles per pass
        for (int pass = 0; pass * 2 * ST_WAVES < n_tiles; ++pass) {
            const int t0 = pass * 2 * ST_WAVES + wave, t1 = t0 + ST_WAVES;
            const int tl0 = min(t0, n_tiles - 1), tl1 = min(t1, n_tiles - 1);
            if (pass > 0) st_ring_load(ring, p.wb[l], hsteps, tl0, tl1, lane);
            f32x4_t acc[ST_MT][2];
#pragma unroll
            for (int mt = 0; mt < ST_MT; ++mt) { acc[mt][0] = (f32x4_t){0, 0, 0, 0}; acc[mt][1] = (f32x4_t){0, 0, 0, 0}; }
            st_mfma(G, zstride, hsteps, p.wb[l], tl0, tl1, lane, ring, acc);
            if (l > 0) {
                // dZ_{l-1} = dh_l * 1[h_l > 0] * scale / (deg + 1) goes straight back into the Z tile
                const bf16_t* Yl = static_cast<const bf16_t*>(p.Y[l - 1]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int tl = j == 0 ? t0 : t1;
                    if (tl >= n_tiles) continue;
                    const int col0 = tl * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int mt = 0; mt < ST_MT; ++mt) {
                        const int row = mt * 16 + (lane & 15);
                        const size_t r = row0 + min(row, T - 1);
                        float y[4];
                        if (p.vec_y && col0 + 4 <= H) {
                            const uint2 u = *reinterpret_cast<const uint2*>(Yl + r * H + col0);
                            y[0] = __uint_as_float(u.x << 16); y[1] = __uint_as_float(u.x & 0xffff0000u);
                            y[2] = __uint_as_float(u.y << 16); y[3] = __uint_as_float(u.y & 0xffff0000u);
                        } else {
#pragma unroll
                            for (int g = 0; g < 4; ++g) y[g] = bf16_to_f32(Yl[r * H + min(col0 + g, H - 1)]);
                        }
                        const float inv = p.scale[l - 1] / rden[row];
                        uint2 pk;
                        float z[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) z[g] = (y[g] > 0.0f && row < T && col0 + g < H) ? acc[mt][j][g] * inv : 0.0f;
                        pk.x = (unsigned)f32_to_bf16(z[0]) | ((unsigned)f32_to_bf16(z[1]) << 16);
                        pk.y = (unsigned)f32_to_bf16(z[2]) | ((unsigned)f32_to_bf16(z[3]) << 16);
                        *reinterpret_cast<uint2*>(Z + (size_t)row * zstride + col0) = pk;       // Z is dead since the gather barrier
                    }
                }
                // next layer down: its weights fly during the db / gather phases
                const int ntn = ceil_div(l - 1 == 0 ? p.Din : H, 16);
                st_ring_load(ring, p.wb[l - 1], hsteps, min(wave, ntn - 1), min(wave + ST_WAVES, ntn - 1), lane);
            } else {
                XT* dx = static_cast<XT*>(p.dx);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int tl = j == 0 ? t0 : t1;
                    if (tl >= n_tiles) continue;
                    const int col0 = tl * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int mt = 0; mt < ST_MT; ++mt) {
                        const int row = mt * 16 + (lane & 15);
                        if (row >= T) continue;
                        XT* dst = dx + (row0 + row) * p.Din + col0;
                        if (col0 + 4 <= p.Din && (p.Din & 3) == 0) {
                            if constexpr (sizeof(XT) == 2) {
                                uint2 pk;
                                pk.x = (unsigned)f32_to_bf16(acc[mt][j][0]) | ((unsigned)f32_to_bf16(acc[mt][j][1]) << 16);
                                pk.y = (unsigned)f32_to_bf16(acc[mt][j][2]) | ((unsigned)f32_to_bf16(acc[mt][j][3]) << 16);
                                *reinterpret_cast<uint2*>(dst) = pk;
                            } else {
                                *reinterpret_cast<float4*>(dst) = make_float4(acc[mt][j][0], acc[mt][j][1], acc[mt][j][2], acc[mt][j][3]);
                            }
                        } else {
#pragma unroll
                            for (int g = 0; g < 4; ++g) if (col0 + g < p.Din) io<XT>::store1(dst + g, acc[mt][j][g]);
                        }
                    }
                }
            }
        }
        GCNPT_STAMP(p.stamps, 5 + 4 * (Ltop - l));
        __syncthreads();                                                     // Z = dZ_{l-1} complete; G dead
    }
}

}  // namespace gcnpt

// =====================================================================================================
// C-ABI
// =====================================================================================================
using namespace gcnpt;

static size_t stack_fwd_lds(int Din, int H) {
    const int hpad = round_up(H, 32), kpad0 = round_up(Din, 32);
    const size_t r0 = std::max((size_t)ST_ROWS * st_astride(kpad0) * sizeof(bf16_t), (size_t)ST_ROWS * st_pstride(H) * sizeof(float));
    return r0 + (size_t)ST_ROWS * st_astride(hpad) * sizeof(bf16_t) + (size_t)ST_ROWS * 9 * sizeof(int);
}
static size_t stack_bwd_lds(int H) {
    return (size_t)2 * ST_ROWS * st_astride(round_up(H, 32)) * sizeof(bf16_t) + (size_t)ST_ROWS * 9 * sizeof(int);
}

extern "C" int gcnpt_stack_supported(int T, int Din, int H, int n_layers, int compute_dtype) {
    if (!(compute_dtype == GCNPT_BF16 && T >= 1 && T <= ST_ROWS && H >= 8 && H <= 256 && Din >= 8 && Din <= 1024 &&
          n_layers >= 1 && n_layers <= ST_MAXL))
        return 0;
    return stack_fwd_lds(Din, H) <= 160 * 1024 && stack_bwd_lds(H) <= 160 * 1024;     // the sentence's tiles must fit the CU's LDS
}

extern "C" size_t gcnpt_stack_frag_bytes(int B, int T, int width) {
    if (B <= 0 || T <= 0 || width <= 0) return 0;
    return (size_t)ceil_div(width, 16) * (size_t)B * ceil_div(T, 32) * 64 * 16;
}

static int stack_check(const char* who, int T, int Din, int H, int L) {
    if (!gcnpt_stack_supported(T, Din, H, L, GCNPT_BF16))
        return fail(GCNPT_E_UNSUPPORTED, "%s: T=%d Din=%d H=%d L=%d outside the sentence-resident kernels' limits "
                    "(T <= %d, 8 <= H <= 256, L <= %d, bf16, tiles within 160 KB of LDS): use the per-layer entry points", who, T, Din, H, L,
                    ST_ROWS, ST_MAXL);
    return GCNPT_OK;
}

template <typename IT, typename OT, bool VECX>
static int launch_stack_fwd(hipStream_t s, const StackFwdParams& p) {
    const size_t lds = stack_fwd_lds(p.Din, p.H);
    auto kern = stack_fwd_kernel<IT, OT, VECX>;
    GCNPT_LDS_ATTR_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(p.B), dim3(ST_THREADS), lds, s, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_stack_fwd(void* stream, int n_layers, const void* x, int x_dtype, const void* const* w_fwd,
                               const float* const* bias, const int32_t* row_ptr, const int32_t* col_idx, const int32_t* ell,
                               const int32_t* deg_ell, int B, int T, int Din, int H, void* const* h_out, int out_dtype,
                               const float* drop_p, const uint64_t* seed, void* const* h_frag, float* const* zero_dW,
                               float* const* zero_db, const uint64_t* seed_dev) {
    GCNPT_REQUIRE(x && w_fwd && bias && row_ptr && col_idx && ell && h_out && drop_p && seed, "stack_fwd: null pointer");
    GCNPT_REQUIRE(B > 0 && dtype_ok(x_dtype) && dtype_ok(out_dtype), "stack_fwd: bad argument");
    if (int rc = stack_check("stack_fwd", T, Din, H, n_layers)) return rc;
    StackFwdParams p{};
    p.stamps = static_cast<unsigned long long*>(g_debug_stamps);
    p.L = n_layers; p.B = B; p.T = T; p.Din = Din; p.H = H; p.x = x; p.seed_dev = seed_dev;
    p.g_ell = ell; p.d_ell = deg_ell ? deg_ell : ell; p.row_ptr = row_ptr; p.col_idx = col_idx;
    p.vec_h = (H % 8 == 0);
    for (int l = 0; l < n_layers; ++l) {
        GCNPT_REQUIRE(w_fwd[l] && bias[l] && h_out[l], "stack_fwd: null pointer (layer %d)", l);
        GCNPT_REQUIRE(drop_p[l] >= 0.0f && drop_p[l] < 1.0f, "stack_fwd: drop_p outside [0,1)");
        p.wf[l] = static_cast<const uint4*>(w_fwd[l]); p.bias[l] = bias[l]; p.h_out[l] = h_out[l];
        p.h_frag[l] = h_frag ? static_cast<uint4*>(h_frag[l]) : nullptr;
        p.drop_p[l] = drop_p[l]; p.drop_scale[l] = drop_p[l] > 0.0f ? 1.0f / (1.0f - drop_p[l]) : 1.0f;
        p.thresh16[l] = (unsigned)((double)drop_p[l] * 65536.0); p.seed[l] = seed[l];
        const int K = l == 0 ? Din : H;
        p.zero[2 * l] = zero_dW ? zero_dW[l] : nullptr; p.zero_n[2 * l] = H * K;
        p.zero[2 * l + 1] = zero_db ? zero_db[l] : nullptr; p.zero_n[2 * l + 1] = H;
        if (l + 1 < n_layers) p.vec_h = p.vec_h && aligned16(h_out[l]);
    }
    p.vec_out = ((H * esize(out_dtype)) % 16 == 0) && (H % 8 == 0) && aligned16(h_out[n_layers - 1]);
    const bool vecx = (Din % 8 == 0) && aligned16(x);
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == GCNPT_F32 && out_dtype == GCNPT_F32)
        return vecx ? launch_stack_fwd<float, float, true>(s, p) : launch_stack_fwd<float, float, false>(s, p);
    if (x_dtype == GCNPT_F32)
        return vecx ? launch_stack_fwd<float, bf16_t, true>(s, p) : launch_stack_fwd<float, bf16_t, false>(s, p);
    if (out_dtype == GCNPT_F32)
        return vecx ? launch_stack_fwd<bf16_t, float, true>(s, p) : launch_stack_fwd<bf16_t, float, false>(s, p);
    return vecx ? launch_stack_fwd<bf16_t, bf16_t, true>(s, p) : launch_stack_fwd<bf16_t, bf16_t, false>(s, p);
}

template <typename GT, typename XT, bool VECG>
static int launch_stack_bwd(hipStream_t s, const StackBwdParams& p) {
    const size_t lds = stack_bwd_lds(p.H);
    auto kern = stack_bwd_kernel<GT, XT, VECG>;
    GCNPT_LDS_ATTR_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(p.B), dim3(ST_THREADS), lds, s, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_stack_bwd(void* stream, int n_layers, const void* dY, const void* const* Y, int g_dtype,
                               const void* const* w_bwd, const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                               const int32_t* ellT, int B, int T, int Din, int H, void* dx, int dx_dtype, const float* scale,
                               void* const* g_frag, float* const* db) {
    GCNPT_REQUIRE(dY && Y && w_bwd && ell && rowT_ptr && colT_idx && ellT && scale, "stack_bwd: null pointer");
    GCNPT_REQUIRE(B > 0 && dtype_ok(g_dtype) && dtype_ok(dx_dtype), "stack_bwd: bad argument");
    if (int rc = stack_check("stack_bwd", T, Din, H, n_layers)) return rc;
    StackBwdParams p{};
    p.stamps = static_cast<unsigned long long*>(g_debug_stamps);
    p.L = n_layers; p.B = B; p.T = T; p.Din = Din; p.H = H; p.dY = dY; p.dx = dx;
    p.d_ell = ell; p.gT_ell = ellT; p.rowT_ptr = rowT_ptr; p.colT_idx = colT_idx;
    p.vec_y = (H % 4 == 0);
    for (int l = 0; l < n_layers; ++l) {
        GCNPT_REQUIRE(Y[l] && w_bwd[l], "stack_bwd: null pointer (layer %d)", l);
        p.Y[l] = Y[l]; p.wb[l] = static_cast<const uint4*>(w_bwd[l]); p.scale[l] = scale[l];
        p.g_frag[l] = g_frag ? static_cast<uint4*>(g_frag[l]) : nullptr;
        p.db[l] = db ? db[l] : nullptr;
        if (l + 1 < n_layers) p.vec_y = p.vec_y && ((reinterpret_cast<uintptr_t>(Y[l]) & 7) == 0);
    }
    const bool vecg = (H % 8 == 0) && aligned16(dY) && aligned16(Y[n_layers - 1]);
    hipStream_t s = (hipStream_t)stream;
    if (g_dtype == GCNPT_F32 && dx_dtype == GCNPT_F32)
        return vecg ? launch_stack_bwd<float, float, true>(s, p) : launch_stack_bwd<float, float, false>(s, p);
    if (g_dtype == GCNPT_F32)
        return vecg ? launch_stack_bwd<float, bf16_t, true>(s, p) : launch_stack_bwd<float, bf16_t, false>(s, p);
    if (dx_dtype == GCNPT_F32)
        return vecg ? launch_stack_bwd<bf16_t, float, true>(s, p) : launch_stack_bwd<bf16_t, float, false>(s, p);
    return vecg ? launch_stack_bwd<bf16_t, bf16_t, true>(s, p) : launch_stack_bwd<bf16_t, bf16_t, false>(s, p);
}
