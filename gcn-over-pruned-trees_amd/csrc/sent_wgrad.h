// Weight gradient of the sentence-slice form: dW[H,Din] += G^T h, db[H] += 2 sum_r dZ[r,:]  (reference model/gcn.py:270-271, autograd),
// from ROWS: G = (A+I)^T dZ as the backward-data launch left it ([N,H], compute type) and the layer's own input rows h ([N,Din]).
// No fragment images: the forward writes nothing for this kernel.  The contraction runs over the ROW index, so both MFMA operands need 8
// consecutive rows per lane; a wave stages the 32 rows x (64 + 48) columns of a k-step in its own LDS tile (coalesced 16-byte row
// pieces) and reads them back transposed with ds_read_b64_tr_b16 -- wave-private, no workgroup barrier.  Workgroup = a (4 x 3)-tile
// block of dW x one slice of the rows; its waves take every 8th k-step of the slice, meet in LDS, slices are combined with float atomics
// (the plan and the XCD-aware block map are the fragment-image kernel's, wgrad_common.h).
#pragma once
#include "layer_common.h"
#include "wgrad_common.h"

namespace gcnpt {

constexpr int SW_WAVES = 8, SW_KB = 2;                 // waves per workgroup, k-steps a wave stages per batch
constexpr int SW_TS = WG_MT * 16 + WG_NT * 16 + 8;     // staging tile row stride, elements

struct SentWgradParams {
    const void* g;          // [N,H] rows of G, compute type
    const void* h;          // [N,Din] the layer's input rows
    const float* dbpart;    // [n_groups][H] column sums of dZ per sentence group
    float* dW; float* db;
    int N, H, Din, n_groups;
    int m_tiles, n_tiles, nks, ks_per_wg, mb, nb, slices;
    unsigned long long* stamps;
    int knob;
};

struct SentWgradMulti {
    SentWgradParams l[WG_MAX_LAYERS];
    int first[WG_MAX_LAYERS + 1];
    int n;
};

// VB: bytes per lane-load of a row piece (16, or 8 for rows that are only 8-byte aligned)
template <typename CT, typename HT, int VB>
__device__ __forceinline__ void sent_wgrad_body(const SentWgradParams& p, const int id, unsigned char* smem) {
    constexpr int RK = sizeof(CT) == 2 ? 32 : 16;                 // rows per k-step
    constexpr int GW = WG_MT * 16, HW = WG_NT * 16;               // columns of the two operand tiles
    constexpr int GE = VB / (int)sizeof(CT), HE = VB / (int)sizeof(HT);       // elements per lane-load
    constexpr int GL = RK * (GW / GE) / WAVE, HL = (RK * (HW / HE) + WAVE - 1) / WAVE;   // wave-loads per k-step
    constexpr int RT = WG_MT * WG_NT;
    typedef f32x4_t RedTile[RT][WAVE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xg = id & 7, rest = id >> 3;
    int slice, blk;
    if ((p.slices & 7) == 0) { const int sp = p.slices >> 3; slice = xg * sp + rest % sp; blk = rest / sp; }
    else                     { const int gp = 8 / p.slices;  slice = xg / gp;             blk = rest * gp + xg % gp; }
    if (blk >= p.mb * p.nb) return;
    const int bm = blk % p.mb, bn = blk / p.mb;
    const int m0 = bm * WG_MT, n0 = bn * WG_NT;
    const int ks_lo = slice * p.ks_per_wg, ks_hi = min(p.nks, ks_lo + p.ks_per_wg);
    GCNPT_STAMP_REAL(p.stamps);
    GCNPT_STAMP(p.stamps, 11);

    const CT* G = static_cast<const CT*>(p.g);
    const HT* Hin = static_cast<const HT*>(p.h);
    CT* tile0 = reinterpret_cast<CT*>(smem) + (size_t)wave * SW_KB * RK * SW_TS;      // this wave's staging tiles

    f32x4_t acc[WG_MT][WG_NT];
#pragma unroll
    for (int i = 0; i < WG_MT; ++i)
#pragma unroll
        for (int j = 0; j < WG_NT; ++j) acc[i][j] = (f32x4_t){0, 0, 0, 0};

    typedef typename std::conditional<VB == 16, uint4, uint2>::type V;
    for (int base = ks_lo + wave; base < ks_hi; base += SW_WAVES * SW_KB) {
        // every load of the batch (unconditional, clamped), then the tiles, then the matrix cores
        V gq[SW_KB][GL], hq[SW_KB][HL];
#pragma unroll
        for (int u = 0; u < SW_KB; ++u) {
            const int ks = min(base + SW_WAVES * u, p.nks - 1);
#pragma unroll
            for (int w = 0; w < GL; ++w) {
                const int idx = w * WAVE + lane, row = idx / (GW / GE), pc = idx - row * (GW / GE);
                const size_t r = (size_t)min(ks * RK + row, p.N - 1);
                const int col = min(m0 * 16 + pc * GE, p.H - GE);
                gq[u][w] = *reinterpret_cast<const V*>(G + r * p.H + col);
            }
#pragma unroll
            for (int w = 0; w < HL; ++w) {
                const int idx = w * WAVE + lane, row = min(idx / (HW / HE), RK - 1), pc = idx - (idx / (HW / HE)) * (HW / HE);
                const size_t r = (size_t)min(ks * RK + row, p.N - 1);
                const int col = min(n0 * 16 + pc * HE, p.Din - HE);
                hq[u][w] = *reinterpret_cast<const V*>(Hin + r * p.Din + col);
            }
        }
#pragma unroll
        for (int u = 0; u < SW_KB; ++u) {
            const int ks = base + SW_WAVES * u;
            const bool live = ks < ks_hi;                                          // past the slice: contributes zeros
            CT* tl = tile0 + (size_t)u * RK * SW_TS;
            const V zero = V{};
#pragma unroll
            for (int w = 0; w < GL; ++w) {
                const int idx = w * WAVE + lane, row = idx / (GW / GE), pc = idx - row * (GW / GE);
                const bool ok = live && ks * RK + row < p.N && m0 * 16 + pc * GE < p.H;
                *reinterpret_cast<V*>(tl + (size_t)row * SW_TS + pc * GE) = ok ? gq[u][w] : zero;
            }
#pragma unroll
            for (int w = 0; w < HL; ++w) {
                const int idx = w * WAVE + lane, row = idx / (HW / HE), pc = idx - row * (HW / HE);
                if (row >= RK) continue;
                const bool ok = live && ks * RK + row < p.N && n0 * 16 + pc * HE < p.Din;
                const V x = ok ? hq[u][w] : zero;
                CT* dst = tl + (size_t)row * SW_TS + GW + pc * HE;
                if constexpr (sizeof(HT) == sizeof(CT)) {
                    *reinterpret_cast<V*>(dst) = x;
                } else {                                                           // f32 rows -> bf16 operand
                    const unsigned* xw = reinterpret_cast<const unsigned*>(&x);
#pragma unroll
                    for (int e = 0; e < HE; e += 2)
                        *reinterpret_cast<unsigned*>(dst + e) = (unsigned)f32_to_bf16(__uint_as_float(xw[e])) | ((unsigned)f32_to_bf16(__uint_as_float(xw[e + 1])) << 16);
                }
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int u = 0; u < SW_KB; ++u) {
            const CT* tl = tile0 + (size_t)u * RK * SW_TS;
            uint4 fr[WG_MT + WG_NT];
            const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
            for (int t = 0; t < WG_MT + WG_NT; ++t) {
                if constexpr (sizeof(CT) == 2) {
                    const int q4 = i16 >> 2, pp = i16 & 3;
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(tl + (size_t)(8 * g + q4) * SW_TS + 16 * t + 4 * pp));
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(tl + (size_t)(8 * g + 4 + q4) * SW_TS + 16 * t + 4 * pp));
                    fr[t].x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
                    fr[t].y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
                    fr[t].z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
                    fr[t].w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
                } else {
                    fr[t].x = __float_as_uint(tl[(size_t)(4 * g + 0) * SW_TS + 16 * t + i16]);
                    fr[t].y = __float_as_uint(tl[(size_t)(4 * g + 1) * SW_TS + 16 * t + i16]);
                    fr[t].z = __float_as_uint(tl[(size_t)(4 * g + 2) * SW_TS + 16 * t + i16]);
                    fr[t].w = __float_as_uint(tl[(size_t)(4 * g + 3) * SW_TS + 16 * t + i16]);
                }
            }
#pragma unroll
            for (int i = 0; i < WG_MT; ++i)
#pragma unroll
                for (int j = 0; j < WG_NT; ++j) {
                    if constexpr (sizeof(CT) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fr[i]),
                                                                             __builtin_bit_cast(bf16x8_t, fr[WG_MT + j]), acc[i][j], 0, 0, 0);
                    } else {
                        const f32x4_t af = __builtin_bit_cast(f32x4_t, fr[i]), bf = __builtin_bit_cast(f32x4_t, fr[WG_MT + j]);
#pragma unroll
                        for (int s = 0; s < 4; ++s) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], acc[i][j], 0, 0, 0);
                    }
                }
        }
        wave_lds_fence();                                         // the tiles are rewritten by the next batch
    }
    GCNPT_STAMP(p.stamps, 12);

    // waves meet in LDS (over the staging tiles, once every wave is done with its own); wave w then owns tiles w, w + 8, ...
    __syncthreads();
    RedTile* red = reinterpret_cast<RedTile*>(smem);
#pragma unroll
    for (int i = 0; i < WG_MT; ++i)
#pragma unroll
        for (int j = 0; j < WG_NT; ++j) red[wave][i * WG_NT + j][lane] = acc[i][j];
    __syncthreads();
    GCNPT_STAMP(p.stamps, 13);
    for (int tt = wave; tt < RT; tt += SW_WAVES) {
        const int i = tt / WG_NT, j = tt - i * WG_NT;
        if (m0 + i >= p.m_tiles || n0 + j >= p.n_tiles) continue;
        f32x4_t v = red[0][tt][lane];
#pragma unroll
        for (int w = 1; w < SW_WAVES; ++w) v += red[w][tt][lane];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int m = (m0 + i) * 16 + (lane >> 4) * 4 + g;
            const int n = (n0 + j) * 16 + (lane & 15);
            if (m < p.H && n < p.Din) atomicAdd(p.dW + (size_t)m * p.Din + n, v[g]);
        }
    }
    // db = 2 sum dZ from the per-group column sums the backward-data launch left: slice s takes the groups s, s + slices, ...
    if (bn == 0 && p.db && tid < GW) {
        const int m = m0 * 16 + tid;
        if (m < p.H) {
            float sdb = 0.0f;
            for (int gq = slice; gq < p.n_groups; gq += p.slices) sdb += p.dbpart[(size_t)gq * p.H + m];
            atomicAdd(p.db + m, 2.0f * sdb);                         // bias enters twice
        }
    }
    GCNPT_STAMP(p.stamps, 14);
}

template <typename CT, typename HT, int VB>
__global__ __launch_bounds__(SW_WAVES * WAVE, 2) void sent_wgrad_kernel(const SentWgradMulti mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sw_smem[];
    int layer = 0;
#pragma unroll
    for (int i = 1; i < WG_MAX_LAYERS; ++i) layer += (i < mp.n && (int)blockIdx.x >= mp.first[i]) ? 1 : 0;
    sent_wgrad_body<CT, HT, VB>(mp.l[layer], (int)blockIdx.x - mp.first[layer], sw_smem);
}

// dynamic LDS: the waves' staging tiles, reused for the reduction
inline size_t sent_wgrad_lds(size_t ct_size) {
    const size_t stage = (size_t)SW_WAVES * SW_KB * (ct_size == 2 ? 32 : 16) * SW_TS * ct_size;
    const size_t red = sizeof(f32x4_t) * WG_MT * WG_NT * WAVE * SW_WAVES;
    return stage > red ? stage : red;
}

}  // namespace gcnpt
