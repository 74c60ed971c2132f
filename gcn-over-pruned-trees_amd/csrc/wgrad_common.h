// Weight gradient dW += dZ^T S from the two fragment images: the device code of gcnpt_layer_bwd_weight, shared with the backward-data
// launch that carries the layer above's weight gradient as a side job (rowtile_kernels.hip).
#pragma once
#include "layer_common.h"

namespace gcnpt {

// ---------------------------------------------------------------------------------------------------
// backward-weight: dW[H,Din] += dZ^T S,  db[H] += 2 sum_r dZ[r,:]      (S = (A+I) h)
//
// Both operands arrive as fragment images (include/gcnpt.h) written by the row-tile kernels, so every
// operand fetch is one fully coalesced 1-KiB wave load straight into registers: no LDS, no transposes,
// no CSR.  A workgroup owns a (4 x 3)-tile block of dW and one slice of the contraction (row) range;
// its 4 waves take every 4th k-step of the slice, issue up to WG_KB k-steps of loads at once, and meet
// in LDS at the end; the slices of different workgroups are combined with float atomics.
// ---------------------------------------------------------------------------------------------------
constexpr int WG_MT = 4, WG_NT = 3, WG_KB = 5;

struct WeightGradParams {
    const uint4* zf;     // fragment image of dZ  [m_tiles][nks][64]
    const uint4* sf;     // fragment image of S   [n_tiles][nks][64]
    float* dW; float* db;
    int H, Din, m_tiles, n_tiles, nks, ks_per_wg;
    int mb, nb, slices;  // block grid; the launch grid is 1-D so that the block -> (m, n, slice) map can follow the XCDs
    unsigned long long* stamps;   // diagnostic builds only
    int knob;
};

template <typename CT>
__device__ __forceinline__ float frag_sum(const uint4& u) {
    if constexpr (sizeof(CT) == 2) {
        return (__uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u)) + (__uint_as_float(u.y << 16) + __uint_as_float(u.y & 0xffff0000u)) +
               (__uint_as_float(u.z << 16) + __uint_as_float(u.z & 0xffff0000u)) + (__uint_as_float(u.w << 16) + __uint_as_float(u.w & 0xffff0000u));
    } else {
        return (__uint_as_float(u.x) + __uint_as_float(u.y)) + (__uint_as_float(u.z) + __uint_as_float(u.w));
    }
}

// One launch serves the weight gradients of several layers (they all become computable at the end of the backward
// sweep and each alone fills at most one workgroup per CU): layer l owns blocks [first[l], first[l+1]), first[l] % 8 == 0.
constexpr int WG_MAX_LAYERS = 8;
struct WeightGradMulti {
    WeightGradParams l[WG_MAX_LAYERS];
    int first[WG_MAX_LAYERS + 1];
    int n;
};

// WG_WAVES waves split a workgroup's k-steps: 4 when a layer has the launch to itself (its slices are short), 8 when several
// layers share the CUs (twice the k-steps per workgroup: 8 waves still take them in ONE batch of loads each)
// Block shape: (WG_MT x NT) output tiles per workgroup, KB k-steps of loads in flight per wave.  (4 x 3, 5) for small batches (one
// latency chain per CU: as many loads in flight as the registers hold); (4 x 6, 2) for big ones, where the fragment stream through
// the CU's L1 path is the bound: 10 fragment loads feed 24 MFMAs instead of 7 feeding 12.
template <typename CT, int WG_WAVES, int NT, int KB>
__device__ __forceinline__ void weight_grad_body(const WeightGradParams& p, const int id, unsigned char* wg_smem) {
    constexpr int RT = WG_MT * NT, RH = RT > 12 ? RT / 2 : RT;                             // tiles reduced per LDS round (12 KiB per wave each)
    typedef f32x4_t RedTile[RH][WAVE];
    RedTile* red = reinterpret_cast<RedTile*>(wg_smem);                                   // [WG_WAVES] per-wave partial tiles
    typedef float DbTile[WG_MT][16];
    DbTile* dbred = reinterpret_cast<DbTile*>(wg_smem + sizeof(RedTile) * WG_WAVES);     // [WG_WAVES]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups are dealt round-robin over the 8 XCDs (id % 8 labels the XCD group), each with its own L2.
    // All blocks of one contraction slice read the same rows of both images, so a slice is pinned to one
    // XCD group: its rows cross the fabric once and every other read hits that XCD's L2.  (Speed only.)
    const int xg = id & 7, rest = id >> 3;
    int slice, blk;
    // Which slices an XCD takes follows the row-tile kernels: XCD x wrote the x-th eighth of the row tiles (contiguous runs, see
    // rowtile_kernels.hip), i.e. of the k-steps, and those lines are still in ITS L2 when the images are read back.
    if ((p.slices & 7) == 0) { const int sp = p.slices >> 3; slice = xg * sp + rest % sp; blk = rest / sp; }
    else                     { const int gp = 8 / p.slices;  slice = xg / gp;             blk = rest * gp + xg % gp; }
    if (blk >= p.mb * p.nb) return;
    const int bm = blk % p.mb, bn = blk / p.mb;
    const int m0 = bm * WG_MT, n0 = bn * NT;
    const int ks_lo = slice * p.ks_per_wg, ks_hi = min(p.nks, ks_lo + p.ks_per_wg);
    const bool want_db = bn == 0 && p.db != nullptr;
    GCNPT_STAMP_REAL(p.stamps);
    GCNPT_STAMP(p.stamps, 11);

    f32x4_t acc[WG_MT][NT];
    float dbp[WG_MT];
#pragma unroll
    for (int i = 0; i < WG_MT; ++i) {
        dbp[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0, 0, 0, 0};
    }

    // Every load below is unconditional (clamped indices): a load behind a runtime condition gets its own basic
    // block and an s_waitcnt vmcnt(0) from hipcc, which would turn this batch into 35 serial round trips.
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    for (int base = ks_lo + wave; base < ks_hi; base += WG_WAVES * KB) {
        uint4 a[KB][WG_MT], b[KB][NT];
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            int ks = min(base + WG_WAVES * u, p.nks - 1);
#ifdef GCNPT_STAMPS
            if (p.knob & 2) ks = 0;                                       // experiment: every load hits the same lines
#endif
#pragma unroll
            for (int i = 0; i < WG_MT; ++i) a[u][i] = p.zf[((size_t)min(m0 + i, p.m_tiles - 1) * p.nks + ks) * 64 + lane];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[u][j] = p.sf[((size_t)min(n0 + j, p.n_tiles - 1) * p.nks + ks) * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const bool live = base + WG_WAVES * u < ks_hi;                       // past the slice: contributes zeros
#pragma unroll
            for (int i = 0; i < WG_MT; ++i) {
                const uint4 av = live ? a[u][i] : zero4;
                // db on the VALU, in every block.  Tried and measured slower: the column sums as one more MFMA against a fragment
                // of ones (+3.4 us: hipcc then schedules for occupancy and no longer keeps the batch of loads in flight), and
                // summing only in the blocks that store db (a branch here makes every wait in the batch a full drain, +1.8 us)
                dbp[i] += frag_sum<CT>(av);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (sizeof(CT) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av),
                                                                             __builtin_bit_cast(bf16x8_t, b[u][j]), acc[i][j], 0, 0, 0);
                    } else {
                        const f32x4_t af = __builtin_bit_cast(f32x4_t, av), bf = __builtin_bit_cast(f32x4_t, b[u][j]);
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    }

    GCNPT_STAMP(p.stamps, 12);
    // waves meet in LDS (RH tiles per round); wave w then owns tiles w, w + WG_WAVES, ... of the round
    if (want_db) {
#pragma unroll
        for (int i = 0; i < WG_MT; ++i) {
            float v = dbp[i];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) dbred[wave][i][lane] = v;
        }
    }
#pragma unroll
    for (int r0t = 0; r0t < RT; r0t += RH) {
        if (r0t > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < WG_MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int tt = i * NT + j;                                   // compile-time after unrolling
                if (tt >= r0t && tt < r0t + RH) red[wave][tt - r0t][lane] = acc[i][j];
            }
        __syncthreads();
        if (r0t == 0) GCNPT_STAMP(p.stamps, 13);
        for (int tl = wave; tl < RH; tl += WG_WAVES) {
            const int tt = r0t + tl;
            const int i = tt / NT, j = tt - i * NT;
            if (m0 + i >= p.m_tiles || n0 + j >= p.n_tiles) continue;
            f32x4_t v = red[0][tl][lane];
#pragma unroll
            for (int w = 1; w < WG_WAVES; ++w) v += red[w][tl][lane];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = (m0 + i) * 16 + (lane >> 4) * 4 + g;
                const int n = (n0 + j) * 16 + (lane & 15);
#ifdef GCNPT_STAMPS
                if (p.knob & 4) { if (m < p.H && n < p.Din) p.dW[(size_t)m * p.Din + n] = v[g]; continue; }     // experiment: stores for atomics
#endif
                if (m < p.H && n < p.Din) atomicAdd(p.dW + (size_t)m * p.Din + n, v[g]);
            }
        }
    }
    if (want_db && tid < WG_MT * 16) {
        const int i = tid >> 4, c = tid & 15;
        const int m = (m0 + i) * 16 + c;
        if (m0 + i < p.m_tiles && m < p.H)
        {
            float sdb = dbred[0][i][c];
#pragma unroll
            for (int w = 1; w < WG_WAVES; ++w) sdb += dbred[w][i][c];
            atomicAdd(p.db + m, 2.0f * sdb);                                 // bias enters twice
        }
    }
    GCNPT_STAMP(p.stamps, 14);
}


// one launch of its own: layer l owns blocks [first[l], first[l+1]), first[l] % 8 == 0
template <typename CT, int WG_WAVES, int NT, int KB>
__global__ __launch_bounds__(WG_WAVES * WAVE, 2) void weight_grad_kernel(const WeightGradMulti mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
    int layer = 0;
#pragma unroll
    for (int i = 1; i < WG_MAX_LAYERS; ++i) layer += (i < mp.n && (int)blockIdx.x >= mp.first[i]) ? 1 : 0;
    weight_grad_body<CT, WG_WAVES, NT, KB>(mp.l[layer], (int)blockIdx.x - mp.first[layer], wg_smem);
}

// host side (weight_kernels.hip): fills p for one layer; returns the number of workgroups it takes (a multiple of 8)
int plan_weight_grad(WeightGradParams& p, const void* z_frag, const void* s_frag, int nks, int Din, int H, float* dW, float* db,
                     int blocks_in_launch, int waves, int wg_budget, int nt);
// dynamic LDS of a weight-gradient workgroup of NW waves
inline size_t weight_grad_lds(int nw) { return (sizeof(f32x4_t) * 12 * WAVE + sizeof(float) * WG_MT * 16) * (size_t)nw; }

}  // namespace gcnpt
