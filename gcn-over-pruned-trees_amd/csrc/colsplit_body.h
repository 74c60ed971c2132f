// The row-tile layer kernel for SMALL batches of WIDE layers: every 32-row tile is given to `col_split` workgroups that gather the same rows
// and each produce `tiles_pp` of the layer's output column tiles.
//
// Why (EXPERIMENTS.md): the one-tile-per-workgroup kernel of rowtile_body.h pulls ALL weight fragments of the layer through its CU's
// L1 path (~20 B per clock and CU) -- 360 KB at the C5 input layer, next to 50 KB of rows -- and when a batch has a few dozen row tiles
// (BASELINE configs[4] on 8 GPUs: 16 sentences per GPU, token-packed: 25 tiles) nine CUs in ten have no tile at all while the others
// wait for their weights.  Split 3 ... 8 ways by output COLUMNS a workgroup takes in 45 ... 120 KB of weights; the rows are gathered
// col_split times (from the same XCD's L2: the workgroups of a tile sit next to each other), which idle CUs do for free.  At the C2 widths
// (156 KB of weights) the split buys nothing and costs the backward launch its passenger (the weight gradient that rides on idle CUs):
// the dispatcher takes this form only from 170 KB of weight fragments on and for at most 128 row tiles.
//
// Structure: the one-shot kernel's, with two differences.  (1) A wave whose slot lies past the workgroup's share of the column tiles
// requests no fragments and skips the matrix phase: the kernel body exists in two straight-line copies, chosen per wave at the top, so that
// neither copy has a load behind a condition (round 3 had one copy and the owning waves' fragments in front of everything else: their ELL
// heads then landed behind 19 ... 24 KB of weights).  (2) All k-steps of a wave's tiles are resident (six register-budgeted configurations, no spills), the fragment image of S / dZ is dealt over the
// workgroups that share the tile.  Every barrier is `s_waitcnt lgkmcnt(0)` + `s_barrier`.
//
// Values: bit-identical to rowtile_body.h's (same gather order, same k order on the matrix cores, same epilogue) -- the tests compare
// the two forms exactly.
#pragma once
#include "rowtile_body.h"

namespace gcnpt {

__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS traffic has landed; global loads stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// PI: 8-column chunks of the tile's own rows a thread requests in one batch (3 cover K <= 384, 5 cover K <= 640)
template <typename CT, typename IT, typename OT, bool BWD, int VEC, int NTW, int KS, bool DZIN, int PI>
__global__ __launch_bounds__(RT_THREADS, 2) void rowtile_colsplit_kernel(const RowTileParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    static_assert(sizeof(CT) == 2, "the column-split form exists for bf16 MFMA operands");
    static_assert(BWD || !DZIN, "DZIN is a backward mode");
    constexpr int RTT = RT_THREADS, RTW = RT_WAVES;
    constexpr bool MASKED = BWD && !DZIN;
    constexpr int KSTEP = 32;
    constexpr bool WIDE = MASKED && sizeof(IT) == 4;
#ifndef GCNPT_CS_NBU
#define GCNPT_CS_NBU 4               // neighbour rows an item requests together when the rows are ready-made (A/B: 4 / 7 = the whole ELL head)
#endif
    constexpr int NBU = MASKED ? (WIDE ? 2 : 4) : GCNPT_CS_NBU;
    const int stride = lds_stride_dw(p.Kpad * (int)sizeof(CT) / 4) * 4 / (int)sizeof(CT);
    const int ncols_pass = p.tiles_pp * 16;
    const int ostride = out_stride_dw(min(round_up(p.NOUT, 16), ncols_pass) * (int)sizeof(OT) / 4) * 4 / (int)sizeof(OT);
    const size_t s_bytes = (size_t)ROWS * stride * sizeof(CT);
    const size_t o_bytes = (size_t)ROWS * ostride * sizeof(OT);
    // LDS: S | Z (backward: the tile's own dZ rows before aggregation, read by the image emission) | O | meta | bias
    CT* const Sw = reinterpret_cast<CT*>(smem_raw);
    CT* const Zw = reinterpret_cast<CT*>(smem_raw + s_bytes);
    OT* O = reinterpret_cast<OT*>(smem_raw + (BWD ? 2 : 1) * s_bytes);
    int* meta0 = reinterpret_cast<int*>(smem_raw + (BWD ? 2 : 1) * s_bytes + o_bytes);
    float* sbias = reinterpret_cast<float*>(meta0 + TILE_META_INTS);       // [tiles_pp * 16] fwd: the bias of this workgroup's columns
    const TileMeta m = tile_meta(meta0);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = (int)blockIdx.x, nwg = (int)gridDim.x;              // nwg % 8 == 0

    // accumulators of the weight gradients that follow: cleared by everyone, before anyone may leave
#pragma unroll
    for (int z = 0; z < 4; ++z)
        if (p.zero_p[z])
            for (int i = id * RTT + tid; i < p.zero_n[z]; i += nwg * RTT) p.zero_p[z][i] = 0.0f;

    // Which tile: XCD x (workgroups id % 8 == x) takes the x-th contiguous eighth of the row tiles, like the one-shot kernel, so a tile's
    // neighbour rows are rows the same L2 serves anyway; the `col_split` workgroups of a tile are neighbours inside their XCD.
    // (grid = 8 x ceil(tiles / 8) x col_split: an XCD with one tile less leaves col_split workgroups without work)
    const int C = p.col_split;
    const int xg = id & 7, jx = id >> 3;
    const int cpass = jx % C, gl = jx / C;
    const int n_rt = ceil_div(p.N, ROWS);
    const int xq = n_rt >> 3, xr = n_rt & 7;
    const int t_lo = xg * xq + min(xg, xr), t_cnt = xq + (xg < xr ? 1 : 0);
    if (gl >= t_cnt) return;
    const int tile_id = t_lo + gl, r0 = tile_id * ROWS;

    const uint4* wfrag = static_cast<const uint4*>(p.wfrag);
    const int n_ctiles = ceil_div(p.NOUT, 16);
    const int ksteps = p.Kpad / KSTEP;
    uint64_t seed_off = 0;
    if (!BWD && p.seed_dev) seed_off = *p.seed_dev;
    static_assert(VEC == 8 || VEC == 4, "rows are read in 16- or 8-byte pieces");

    // Two roles, two copies of the code (a wave-uniform branch at the top; both copies run the same barriers): a wave that owns column
    // tiles requests their fragments in pieces BETWEEN the gather's phases, like the one-shot kernel (a wave that requests faster than the
    // CU's L2 -> L1 path delivers stalls at the request, rowtile_body.h) -- heads and own rows first in the queue, a third of the fragments
    // behind them, a third behind the neighbour requests, the rest behind the parked rows; a wave whose slot lies past the workgroup's
    // share requests none and skips the matrix phase.  Straight-line code in each copy: every wait counts exactly what was requested.
    bool has_tile[NTW];
    bool any_tile = false;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int tloc = j * RTW + wave;
        has_tile[j] = tloc < p.tiles_pp && cpass * p.tiles_pp + tloc < n_ctiles;
        any_tile = any_tile || has_tile[j];
    }
    GCNPT_STAMP_REAL(p.stamps);
    GCNPT_STAMP(p.stamps, 0);
    const int arow = lane & 15, kgrp = lane >> 4;
    const int c_lo = cpass * ncols_pass, c_hi = min(p.NOUT, c_lo + ncols_pass);

    auto body = [&](auto role) {
        constexpr bool HASW = decltype(role)::value;
        uint4 wreg[KS][NTW];
        auto load_w = [&](int ks_lo, int ks_hi) {                       // (an absent second slot of an owning wave: a duplicate of the last tile, not used)
            if constexpr (HASW) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        if (ks < ks_lo || ks >= ks_hi) continue;            // compile-time after unrolling
                        const int tl = min(cpass * p.tiles_pp + j * RTW + wave, n_ctiles - 1);
                        wreg[ks][j] = wfrag[((size_t)tl * ksteps + min(ks, ksteps - 1)) * 64 + lane];
                    }
            }
        };
        // ---- (1) the tile's adjacency (ELL heads, degrees) -> LDS; (2) its rows and its aggregating rows' neighbours -> S: the phases of
        //      the one-shot kernel (rowtile_phases.h), PI chunks of the tile's own rows per thread in the first batch
        const TileHeads heads = load_tile_heads(p, r0, lane);
        float bias_v = 0.0f;
        if constexpr (!BWD) bias_v = p.bias[min(cpass * ncols_pass + min(tid, ncols_pass - 1), p.NOUT - 1)];
        const TileGather<CT, IT, MASKED, VEC, NBU> G{p, m, Sw, stride, r0};
        const int n_items = ROWS * (p.Kpad / 8);
        raw8<IT> self[PI], selfy[PI];
#pragma unroll
        for (int u = 0; u < PI; ++u) G.issue_self(u * RTT + tid, self[u], selfy[u]);
        constexpr int KS_A = KS / 3, KS_B = 2 * KS / 3;
        load_w(0, KS_A);
        park_tile_heads<BWD>(p, m, heads, r0, lane, true);
        if constexpr (!BWD) { if (tid < ncols_pass) sbias[tid] = bias_v; }
        wave_lds_fence();
        GCNPT_STAMP(p.stamps, 1);
        const int n_g = *m.gcount * (p.Kpad / 8);
        GatherItem<IT, NBU> g0;
        G.issue(n_g, tid, g0);
        load_w(KS_A, KS_B);
        GCNPT_STAMP(p.stamps, 2);
#pragma unroll
        for (int u = 0; u < PI; ++u) G.template copy_item<BWD>(u * RTT + tid, self[u], selfy[u], Zw);
        GCNPT_STAMP(p.stamps, 9);
        load_w(KS_B, KS);
        GCNPT_STAMP(p.stamps, 10);
        if (wave * WAVE < n_g) G.finish(n_g, tid, g0);
        GCNPT_STAMP(p.stamps, 11);
        for (int base = RTT; base < n_g; base += RTT) {                  // more than 512 (aggregating row, chunk) items: further rounds
            GatherItem<IT, NBU> g;
            G.issue(n_g, base + tid, g);
            G.finish(n_g, base + tid, g);
        }
        for (int first = PI * RTT; first < n_items; first += RTT) {       // K wider than PI covers: further batches
            raw8<IT> s, sy;
            G.issue_self(first + tid, s, sy);
            G.template copy_item<BWD>(first + tid, s, sy, Zw);
        }
        GCNPT_STAMP(p.stamps, 3);
        lds_barrier();                                                      // S (and Z) complete
        GCNPT_STAMP(p.stamps, 4);

        // side output: the tile in MFMA fragment order for the weight gradient; the column tiles of the image are dealt over the workgroups
        // that share this row tile
        if (p.frag_out) emit_tile_image(static_cast<uint4*>(p.frag_out), BWD ? Zw : Sw, stride, wave + RTW * cpass, RTW * C, ceil_div(p.K, 16), lane,
                                        (size_t)n_rt, tile_id, BWD);

        // (3) the tile meets the resident weights (a wave without a column tile reads no operand either: LDS bandwidth)
        f32x4_t acc[2][NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) { acc[0][j] = (f32x4_t){0, 0, 0, 0}; acc[1][j] = (f32x4_t){0, 0, 0, 0}; }
        if constexpr (HASW) {
            constexpr int AH = GCNPT_A_AHEAD;
            uint4 a_st[AH + 1][2];                                       // [0] = the k-step on the matrix cores, [d] = d k-steps ahead
            auto read_a = [&](int kk, uint4 (&dst)[2]) {
                dst[0] = *reinterpret_cast<const uint4*>(Sw + (size_t)arow * stride + kk * KSTEP + kgrp * 8);
                dst[1] = *reinterpret_cast<const uint4*>(Sw + (size_t)(arow + 16) * stride + kk * KSTEP + kgrp * 8);
            };
#pragma unroll
            for (int d = 0; d < AH; ++d) read_a(min(d, ksteps - 1), a_st[d]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks < ksteps) {                                       // wave-uniform, no global load inside
                    read_a(min(ks + AH, ksteps - 1), a_st[AH]);
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        const bf16x8_t bq = __builtin_bit_cast(bf16x8_t, wreg[ks][j]);
                        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq, __builtin_bit_cast(bf16x8_t, a_st[0][0]), acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq, __builtin_bit_cast(bf16x8_t, a_st[0][1]), acc[1][j], 0, 0, 0);
                    }
#pragma unroll
                    for (int d = 0; d < AH; ++d) { a_st[d][0] = a_st[d + 1][0]; a_st[d][1] = a_st[d + 1][1]; }
                }
            }
        }
        GCNPT_STAMP(p.stamps, 5);

        // (4) epilogue on the accumulators -> O, then whole rows leave in 16-byte pieces (rowtile_phases.h)
        if constexpr (HASW) {
            float den[2], inv[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) { den[mt] = m.rden[mt * 16 + (lane & 15)]; inv[mt] = m.rinv[mt * 16 + (lane & 15)]; }
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                if (!has_tile[j]) continue;
                const int col0 = (cpass * p.tiles_pp + j * RTW + wave) * 16 + (lane >> 4) * 4;
                const f32x4_t aj[2] = {acc[0][j], acc[1][j]};
                epilogue_tile<OT, BWD>(p, aj, col0, col0 - c_lo, r0, den, inv, sbias, O, ostride, seed_off, lane);
            }
        }
    };
    if (any_tile) body(std::true_type{});
    else body(std::false_type{});
    GCNPT_STAMP(p.stamps, 6);
    lds_barrier();                                                      // O complete
    GCNPT_STAMP(p.stamps, 7);
    store_tile_rows<OT, BWD, RTT>(p, O, ostride, m.rden, r0, c_lo, c_hi - c_lo, tid);
    GCNPT_STAMP(p.stamps, 8);
}

#ifdef GCNPT_RT_PART
template <typename CT, typename IT, typename OT, bool BWD, int VEC, int NTW, int KS, bool DZIN, int PI>
static inline int launch_colsplit_cfg(hipStream_t s, RowTileParams p, int col_split) {
    const int n_ctiles = ceil_div(p.NOUT, 16);
    p.tiles_pp = ceil_div(n_ctiles, col_split);
    p.col_split = ceil_div(n_ctiles, p.tiles_pp);                        // no workgroup without a column tile of its own
    const int stride = lds_stride_dw(p.Kpad * (int)sizeof(CT) / 4) * 4 / (int)sizeof(CT);
    const int ncols_pass = p.tiles_pp * 16;
    const int ostride = out_stride_dw(std::min(round_up(p.NOUT, 16), ncols_pass) * (int)sizeof(OT) / 4) * 4 / (int)sizeof(OT);
    const size_t s_bytes = (size_t)ROWS * stride * sizeof(CT), o_bytes = (size_t)ROWS * ostride * sizeof(OT);
    const size_t lds = (BWD ? 2 : 1) * s_bytes + o_bytes + (size_t)ROWS * 13 * sizeof(int) + (size_t)ncols_pass * sizeof(float);
    if (lds > 160 * 1024) return GCNPT_NOT_TAKEN;
    const int grid = 8 * ceil_div(ceil_div(p.N, ROWS), 8) * p.col_split;
    auto kern = rowtile_colsplit_kernel<CT, IT, OT, BWD, VEC, NTW, KS, DZIN, PI>;
    GCNPT_LDS_ATTR_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(RT_THREADS), lds, s, p);
    note_launch(grid, RT_THREADS, lds, sizeof(p));
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

// the dispatcher: colsplit_plan (rowtile_body.h) says whether and how; GCNPT_NOT_TAKEN leaves the launch to the one-shot form
template <typename CT, typename IT, typename OT, bool BWD, int VEC, bool DZIN>
static inline int try_colsplit(hipStream_t s, const RowTileParams& p) {
    if constexpr (sizeof(CT) != 2 || VEC == 0) {
        return GCNPT_NOT_TAKEN;
    } else {
        constexpr bool MASKED = BWD && !DZIN;
        int c = 0;
        switch (colsplit_plan(p.N, p.Kpad, p.NOUT, (int)sizeof(CT), VEC, MASKED, p.out != nullptr, &c)) {
            case 0: if constexpr (!MASKED) return launch_colsplit_cfg<CT, IT, OT, BWD, VEC, 2, 12, DZIN, 3>(s, p, c); break;
            case 1: return launch_colsplit_cfg<CT, IT, OT, BWD, VEC, 2, 7, DZIN, 2>(s, p, c);
            case 2: if constexpr (!MASKED) return launch_colsplit_cfg<CT, IT, OT, BWD, VEC, 3, 7, DZIN, 2>(s, p, c); break;
            case 3: if constexpr (!MASKED) return launch_colsplit_cfg<CT, IT, OT, BWD, VEC, 1, 20, DZIN, 5>(s, p, c); break;
            case 4: if constexpr (!MASKED) return launch_colsplit_cfg<CT, IT, OT, BWD, VEC, 2, 10, DZIN, 3>(s, p, c); break;
            case 5: return launch_colsplit_cfg<CT, IT, OT, BWD, VEC, 1, 10, DZIN, 3>(s, p, c);
            default: break;
        }
        return GCNPT_NOT_TAKEN;
    }
}
#endif

}  // namespace gcnpt
