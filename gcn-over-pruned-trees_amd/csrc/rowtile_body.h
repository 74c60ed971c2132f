// The row-tile layer kernel (forward and backward-data) and its launch configuration templates.  Included by rowtile_part.hip, which is
// compiled once per (precision combination, direction) so that the ~300 instantiations build in parallel, and by rowtile_kernels.hip
// (the C-ABI), which only needs the parameter structs.
#pragma once
#include "rowtile_phases.h"
#include "wgrad_common.h"

namespace gcnpt {

constexpr int RT_THREADS = 512;      // 8 waves; wave w owns output tiles w, w+8, ...
constexpr int RT_WAVES = RT_THREADS / WAVE;
static_assert(RT_THREADS == 16 * ROWS, "the row store loop gives every row 16 threads (two rounds in the 4-wave form)");
#ifndef GCNPT_A_AHEAD
#define GCNPT_A_AHEAD 2              // k-steps the tile's MFMA operand is read ahead of its use (measured 1..4 at the C2 shape: 51.7 / 51.1 / 51.3 / 51.7 us per step)
#endif
#ifndef GCNPT_W_STAGGER
#define GCNPT_W_STAGGER 3            // n > 0: the fragments behind the early quarter are requested in n pieces (<= 5) between the gather's phases
#endif
#ifndef GCNPT_W_EARLY_NUM
#define GCNPT_W_EARLY_NUM 1          // quarters of a wave's weight fragments requested before the adjacency is known (0..4 measured: 1 is best)
#endif

// DZIN (bwd only): `src` already holds dZ (the layer above wrote it, see relu_src), so the row loader is the forward's plain
// gather -- one load per neighbour instead of three (dY, Y, degree).
// (a device function: the launch of its own below, and the backward launch that also carries a weight gradient, share it; block_id /
// n_blocks: this workgroup's tile number and the number of tiles, which that launch does not read off blockIdx / gridDim)
// NWV: waves per workgroup.  8 for the headline batches (one workgroup per CU, the shortest chain); 4 for big batches, where two
// workgroups share a CU (same registers per wave, half the waves each) and one's memory waits overlap the other's arithmetic.
template <typename CT, typename IT, typename OT, bool BWD, int VEC, int NTW, int KSMAX, bool DZIN = false, int NWV = 8>
__device__ __forceinline__ void rowtile_body(const RowTileParams& p, const int block_id, const int n_blocks, unsigned char* smem_raw) {
    constexpr int RTT = NWV * WAVE, RTW = NWV;                  // threads and waves of this workgroup
    static_assert(BWD || !DZIN, "DZIN is a backward mode");
    constexpr bool MASKED = BWD && !DZIN;                       // the loader computes dZ = dY * 1[Y>0] * scale / (deg+1) itself
    constexpr int KSTEP = sizeof(CT) == 2 ? 32 : 16;            // K consumed per fragment
    constexpr bool WIDE = MASKED && sizeof(IT) == 4;            // two fp32 streams per row: fewer rows in flight per thread
    constexpr int ITEMS = WIDE ? 2 : 3;                         // 8-element chunks a thread gathers per batch
    constexpr int NBU = WIDE ? 2 : 4;                           // neighbour rows fetched together
    const int stride = lds_stride_dw(p.Kpad * (int)sizeof(CT) / 4) * 4 / (int)sizeof(CT);     // S row stride, CT elements
    const int ncols_pass = RTW * NTW * 16;
    const int ostride = out_stride_dw(min(round_up(p.NOUT, 16), ncols_pass) * (int)sizeof(OT) / 4) * 4 / (int)sizeof(OT);
    const size_t s_bytes = (size_t)ROWS * stride * sizeof(CT);
    CT* S = reinterpret_cast<CT*>(smem_raw);
    // bwd: Z (the tile's own dZ rows, before aggregation) is only read by the fragment-image emission, the out tile O only
    // written from the epilogue on: they share one region, with a barrier between the two uses
    const size_t o_bytes = (size_t)ROWS * ostride * sizeof(OT);
    const size_t zo_bytes = BWD ? (s_bytes > o_bytes ? s_bytes : o_bytes) : o_bytes;
    CT* Z = reinterpret_cast<CT*>(smem_raw + s_bytes);
    OT* O = reinterpret_cast<OT*>(smem_raw + s_bytes);
    int* meta = reinterpret_cast<int*>(smem_raw + s_bytes + zo_bytes);
    const TileMeta m = tile_meta(meta);                         // ELL heads, denominators, the list of aggregating rows (rowtile_phases.h)
    float* sbias = reinterpret_cast<float*>(meta + TILE_META_INTS);   // [max(RTT, columns of a pass)] fwd: the bias of this pass's columns

    // `wave` through readfirstlane: the compiler then knows it is uniform and does every wave-dependent address in SALU
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one), each with its own L2, and a row's neighbours
    // sit in its own sentence, i.e. in the adjacent tiles: XCD x takes a CONTIGUOUS run of tiles, so that the neighbour rows a tile
    // gathers are rows the same L2 serves to the tiles next to it (speed only; any placement gives the same values)
    const int xg = block_id & 7, xq = n_blocks >> 3, xr = n_blocks & 7;
    const int tile_id = xg * xq + min(xg, xr) + (block_id >> 3);
    const int r0 = tile_id * ROWS;
    const uint4* wfrag = static_cast<const uint4*>(p.wfrag);
    const int n_tiles = ceil_div(p.NOUT, 16);
    const int ksteps = p.Kpad / KSTEP;
    uint64_t seed_off = 0;                                  // scalar load, consumed in the epilogue
    if (!BWD && p.seed_dev) seed_off = *p.seed_dev;
    GCNPT_STAMP_REAL(p.stamps);
    GCNPT_STAMP(p.stamps, 0);

    // (1) the tile's adjacency: the 32 ELL heads (1 KiB) and the degrees for the denominators.  EVERY wave loads all of
    //     them (64 lanes x 16 bytes; waves 1..7 hit wave 0's lines) and keeps its own copy of the derived tables: the load
    //     is unconditional and first in the queue, and no wave waits for another one before it can start gathering.
    const TileHeads heads = load_tile_heads(p, r0, lane);
    float bias_v = 0.0f;                           // fwd: one bias element per thread, parked in LDS with the heads (registers are
    if constexpr (!BWD) bias_v = p.bias[min(tid, p.NOUT - 1)];      // too scarce to carry 4 per tile through the whole kernel)
    float bias_w = 0.0f;                           // (4 waves x 5 tiles: 320 columns per pass, a second element for the first 64 threads)
    if constexpr (!BWD && RTW * NTW * 16 > RTT) bias_w = p.bias[min(RTT + tid, p.NOUT - 1)];

    // own rows of the first batch (everyone)
    const TileGather<CT, IT, MASKED, VEC, NBU> G{p, m, S, stride, r0};
    const int nchunk = p.Kpad / 8;
    const int n_items = ROWS * nchunk;
    raw8<IT> self[ITEMS], selfy[ITEMS];
    auto issue_self = [&](int batch) {
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) G.issue_self((batch * ITEMS + u) * RTT + tid, self[u], selfy[u]);
    };
    issue_self(0);

    // (0) this wave's weight fragments.  The vector-memory counter retires IN ORDER: whatever is issued before the
    //     neighbour loads of step (2) has to land before they can be consumed.  So only the first KS_EARLY k-steps go
    //     out now (they drain while the ELL heads are on their way); the rest follows the neighbour loads.
    constexpr int KS_EARLY = KSMAX * GCNPT_W_EARLY_NUM / 4;
    uint4 wreg[KSMAX][NTW];
    auto load_w = [&](int pass, int kc0, int ks_lo, int ks_hi) {
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                if (ks < ks_lo || ks >= ks_hi) continue;                    // compile-time after unrolling
                const int tl = min(pass * RTW * NTW + j * RTW + wave, n_tiles - 1);
                const int kk = min(kc0 + ks, ksteps - 1);
#ifdef GCNPT_STAMPS
                if (p.knob & 1) { wreg[ks][j] = wfrag[lane]; continue; }          // experiment: no weight traffic
                if ((p.knob & 32) && ks >= (KSMAX * 2 + 4) / 5) { wreg[ks][j] = wfrag[lane]; continue; }   // experiment: 40 % of the weight traffic (a 64-row x 1/3-column tiling's share)
#endif
                wreg[ks][j] = wfrag[((size_t)tl * ksteps + kk) * 64 + lane];
#ifdef GCNPT_STAMPS
                if (p.knob & 16) {      // experiment (VERDICT item 2b by proxy): twice the weight bytes through the CU, as an fp32 source would need
                    const uint4 extra = wfrag[((size_t)(n_tiles - 1 - tl) * ksteps + (ksteps - 1 - kk)) * 64 + lane];
                    asm volatile("" ::"v"(extra.x), "v"(extra.y), "v"(extra.z), "v"(extra.w));
                }
#endif
            }
    };
    // (The MFMAs run with swapped operands, weights as A, so a lane ends up with 4 CONSECUTIVE output columns of one row:
    // columns 16 tile + 4 (lane>>4) + g.)
    load_w(0, 0, 0, KS_EARLY);              // unconditional even when only the side outputs are wanted: a branch here would
                                            // make every later wait assume the shorter queue

    // park the heads in LDS and compact the rows that aggregate anything (a pruned tree keeps ~1 token in 8).  All
    // waves write the same values to the same places; each reads back only after its own writes (wave_lds_fence).
    park_tile_heads<BWD>(p, m, heads, r0, lane, p.out != nullptr);
    if constexpr (!BWD) {
        sbias[tid] = bias_v;
        if constexpr (RTW * NTW * 16 > RTT) { if (tid < RTW * NTW * 16 - RTT) sbias[RTT + tid] = bias_w; }
    }
    GCNPT_STAMP(p.stamps, 1);
    wave_lds_fence();
    GCNPT_STAMP(p.stamps, 2);

    // (2) gcn.py:269 as a gather, S[row,:] = x[row,:] + sum_{c in pattern row} x[c,:] (TileGather, rowtile_phases.h): the rows that
    //     aggregate something as (row, chunk) items over ALL waves, every other row a plain copy of the loads issued at the top
    typedef GatherItem<IT, NBU> GItem;
    const int n_g = *m.gcount * nchunk;

    GItem g0;
    G.issue(n_g, tid, g0);
    // a wave that requests faster than the CU's L2 -> L1 path delivers (~30 B per clock) stalls AT the request: with everything asked
    // for here, the phases below only start when the last request has left
    constexpr int NP = GCNPT_W_STAGGER ? GCNPT_W_STAGGER : 1;   // pieces the rest is requested in, one per site below
    auto load_piece = [&](auto site) {
        constexpr int i = decltype(site)::value;
        if constexpr (i < NP) load_w(0, 0, KS_EARLY + (KSMAX - KS_EARLY) * i / NP, KS_EARLY + (KSMAX - KS_EARLY) * (i + 1) / NP);
    };
    load_piece(std::integral_constant<int, 0>{});               // 156 KB per workgroup at Din=360, H=200
    GCNPT_STAMP(p.stamps, 3);

    // (2b)
    const int n_batches = ceil_div(n_items, ITEMS * RTT);
    auto copy_batch = [&](int batch) {
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) G.template copy_item<BWD>((batch * ITEMS + u) * RTT + tid, self[u], selfy[u], Z);
    };
    copy_batch(0);
    load_piece(std::integral_constant<int, 1>{});
    GCNPT_STAMP(p.stamps, 4);
    // (2a) -- only the waves that own an item: the sums cost a wave ~120 VALU instructions whether its lanes are live or not
    if (wave * WAVE < n_g) G.finish(n_g, tid, g0);
    load_piece(std::integral_constant<int, 2>{});
    for (int base = RTT; base < n_g; base += RTT) {      // tiles with more than 512 / (K/8) aggregating rows
        GItem g;
        G.issue(n_g, base + tid, g);
        G.finish(n_g, base + tid, g);
    }
    for (int batch = 1; batch < n_batches; ++batch) {                  // K > 384: the tile's own rows take several batches
        issue_self(batch);
        copy_batch(batch);
    }
    GCNPT_STAMP(p.stamps, 5);
    __syncthreads();
    load_piece(std::integral_constant<int, 3>{});
    GCNPT_STAMP(p.stamps, 6);

    // side outputs: the tile in MFMA fragment order for the weight gradient (rows are its contraction index),
    // and cleared accumulators for the kernel that follows
    if (p.frag_out) emit_tile_image(static_cast<uint4*>(p.frag_out), BWD ? Z : S, stride, wave, RTW, ceil_div(p.K, 16), lane, (size_t)n_blocks, tile_id, BWD);
    load_piece(std::integral_constant<int, 4>{});
    if constexpr (BWD) {
        if (p.frag_out) __syncthreads();                         // every wave has read its share of Z: the region becomes O
    }
#pragma unroll
    for (int z = 0; z < 4; ++z)
        if (p.zero_p[z])
            for (int i = block_id * RTT + tid; i < p.zero_n[z]; i += n_blocks * RTT) p.zero_p[z][i] = 0.0f;
    GCNPT_STAMP(p.stamps, 7);
    if (!p.out) return;

    // (3) + (4)
    const int arow = lane & 15, kgrp = lane >> 4;
    const int n_pass = ceil_div(n_tiles, RTW * NTW);

    for (int pass = 0; pass < n_pass; ++pass) {
        f32x4_t acc[2][NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) { acc[0][j] = (f32x4_t){0, 0, 0, 0}; acc[1][j] = (f32x4_t){0, 0, 0, 0}; }
        const int tile0 = pass * RTW * NTW + wave;

        for (int kc0 = 0; kc0 < ksteps; kc0 += KSMAX) {
            if (pass > 0 || kc0 > 0) load_w(pass, kc0, 0, KSMAX);
            // A fragments are read GCNPT_A_AHEAD k-steps ahead of the MFMAs that use them
            constexpr int AW = sizeof(CT) == 2 ? 8 : 4;                  // CT elements per lane per k-step (16 bytes)
            constexpr int AH = GCNPT_A_AHEAD;
            uint4 a_st[AH + 1][2];                                       // [0] = the k-step on the matrix cores, [d] = d k-steps ahead
            auto read_a = [&](int kk, uint4 (&dst)[2]) {
                dst[0] = *reinterpret_cast<const uint4*>(S + (size_t)arow * stride + kk * KSTEP + kgrp * AW);
                dst[1] = *reinterpret_cast<const uint4*>(S + (size_t)(arow + 16) * stride + kk * KSTEP + kgrp * AW);
            };
#pragma unroll
            for (int d = 0; d < AH; ++d) read_a(min(kc0 + d, ksteps - 1), a_st[d]);
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks) {
                if (kc0 + ks < ksteps) {                                 // wave-uniform, no global load inside
                    read_a(min(kc0 + ks + AH, ksteps - 1), a_st[AH]);
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        if constexpr (sizeof(CT) == 2) {
                            const bf16x8_t bq = __builtin_bit_cast(bf16x8_t, wreg[ks][j]);
                            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq, __builtin_bit_cast(bf16x8_t, a_st[0][0]), acc[0][j], 0, 0, 0);
                            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq, __builtin_bit_cast(bf16x8_t, a_st[0][1]), acc[1][j], 0, 0, 0);
                        } else {
                            const f32x4_t bq = __builtin_bit_cast(f32x4_t, wreg[ks][j]);
                            const f32x4_t a0 = __builtin_bit_cast(f32x4_t, a_st[0][0]), a1 = __builtin_bit_cast(f32x4_t, a_st[0][1]);
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[s], a0[s], acc[0][j], 0, 0, 0);
                                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[s], a1[s], acc[1][j], 0, 0, 0);
                            }
                        }
                    }
#pragma unroll
                    for (int d = 0; d < AH; ++d) { a_st[d][0] = a_st[d + 1][0]; a_st[d][1] = a_st[d + 1][1]; }
                }
            }
        }

        GCNPT_STAMP(p.stamps, 8);
        // epilogue on the accumulators -> LDS out tile (tiles past the last real one hold duplicates: not stored).
        // Lane (i = lane & 15, q = lane >> 4) holds, per 16x16 tile, row i and the 4 consecutive columns 4q..4q+3.
        if (pass > 0) {
            __syncthreads();                                             // previous pass's rows have left O
            if constexpr (!BWD) {                                        // (NOUT > 512 only) this pass's bias
                for (int c = tid; c < ncols_pass; c += RTT) sbias[c] = p.bias[min(pass * ncols_pass + c, p.NOUT - 1)];
                __syncthreads();
            }
        }
        float den[2], inv[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) { den[mt] = m.rden[mt * 16 + (lane & 15)]; inv[mt] = m.rinv[mt * 16 + (lane & 15)]; }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int tl = tile0 + j * RTW;
            if (tl >= n_tiles) continue;
            const int col0 = tl * 16 + (lane >> 4) * 4;
            const f32x4_t aj[2] = {acc[0][j], acc[1][j]};
            epilogue_tile<OT, BWD>(p, aj, col0, col0 - pass * ncols_pass, r0, den, inv, sbias, O, ostride, seed_off, lane);
        }
        __syncthreads();
        GCNPT_STAMP(p.stamps, 9);

        // whole rows leave in 16-byte pieces; bwd with relu_src: as the layer below's dZ (store_tile_rows, rowtile_phases.h)
        const int c_lo = pass * ncols_pass, c_hi = min(p.NOUT, c_lo + ncols_pass);
        store_tile_rows<OT, BWD, RTT>(p, O, ostride, m.rden, r0, c_lo, c_hi - c_lo, tid);
    }
    GCNPT_STAMP(p.stamps, 10);
}


template <typename CT, typename IT, typename OT, bool BWD, int VEC, int NTW, int KSMAX, bool DZIN = false, int NWV = 8>
__global__ __launch_bounds__(NWV * WAVE, 2) void rowtile_kernel(const RowTileParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    rowtile_body<CT, IT, OT, BWD, VEC, NTW, KSMAX, DZIN, NWV>(p, (int)blockIdx.x, (int)gridDim.x, smem_raw);
}

// A backward-data launch with the WEIGHT GRADIENT OF THE LAYER ABOVE as a side job: that gradient only needs the two fragment images
// earlier launches have left (dZ_{l+1}, S_{l+1}), and a batch of <= ~6 k rows leaves a third of the CUs without a row tile.  Workgroups [0, n_tiles) are row tiles, workgroups
// [wg_first, gridDim.x) (wg_first = n_tiles rounded up to 8, so that the weight gradient's block -> XCD map holds) contract one slice of
// one output block each.  The last launch of the sweep is then the bottom layer's weight gradient alone: 7.5 us instead of 13.0 us for
// both layers (EXPERIMENTS.md).
// The weight gradient that rides: ONE layer's, planned for the CUs without a row tile (weight_grad_body of wgrad_common.h, (4 x 3)-tile
// blocks, XCD-aware slice map).  Measured in round 3 and NOT adopted (EXPERIMENTS.md, profiles/r03_riders_*): both layers' gradients in
// the bottom layer's launch (the dZ image of the layer below written by the hand-over epilogue) -- 22 us for that launch with (4 x 3)
// blocks, 19 with (4 x 6), 20.5 with every CU sharing the units, 29 with an LDS-staged rows-form kernel -- against 10.5 + 7.5 us for this
// launch plus the bottom layer's own: a CU retires ~1 float atomic per clock and a workgroup's fragment stream ~25 B per clock, so a
// third of the chip cannot take both gradients in the time the row tiles need.
struct SideWgrads {
    WeightGradParams l;
    int blocks;                         // passenger workgroups (a multiple of 8)
};

template <typename CT, typename IT, typename OT, int VEC, int NTW, int KSMAX>
__global__ __launch_bounds__(RT_THREADS, 2) void rowtile_wgrad_kernel(const RowTileParams p, const SideWgrads sw, const int n_tiles,
                                                                      const int wg_first) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if ((int)blockIdx.x < n_tiles) rowtile_body<CT, IT, OT, true, VEC, NTW, KSMAX, true>(p, (int)blockIdx.x, n_tiles, smem_raw);
    else if ((int)blockIdx.x >= wg_first) weight_grad_body<CT, RT_WAVES, WG_NT, WG_KB>(sw.l, (int)blockIdx.x - wg_first, smem_raw);
}

}  // namespace gcnpt

namespace gcnpt {

// Whether a layer launch takes the column-split form (colsplit_body.h), and with which configuration and split: ONE host function for
// the dispatcher (try_colsplit) and for the backward sweep's prediction of which launch carries a weight gradient (a column-split
// launch never does).  Configurations (column tiles per wave, resident k-steps, chunks of own rows per thread), each sized to stay
// inside 256 registers without spilling; a layer takes the one that holds all its k-steps and its own rows with the fewest registers;
// the backward from dY and Y (three loads per gathered row) only takes the two smallest:
//   0 (2,12,3) K <= 384      1 (2,7,2) K <= 224      2 (3,7,2) K <= 224, wide output      3 (1,20,5) K <= 640: the C5 input layer
//   4 (2,10,3) K <= 320: the C5 hidden layers      5 (1,10,3): their backward from dY
// By itself (GCNPT_OPT_COL_SPLIT = -1) the form is taken for at most 128 row tiles (so that 2 ... 8 workgroups per tile fit the 256 CUs)
// of a layer with at least 170 KB of weight fragments; measured (EXPERIMENTS.md): 71 -> 62 us per step for the per-GPU shard of
// BASELINE configs[4] (16 packed sentences, 600 -> 300 -> 300), nothing at the C2 widths (156 KB), where it also costs the backward launch its
// passenger.  n >= 1 forces it with at least n workgroups per tile whatever the batch (tests).
struct ColSplitCfg { int ntw, ks, pi; bool masked_ok; };
constexpr ColSplitCfg COLSPLIT_CFGS[6] = {{2, 12, 3, false}, {2, 7, 2, true}, {3, 7, 2, false}, {1, 20, 5, false}, {2, 10, 3, false}, {1, 10, 3, true}};
// ct_size: bytes of the MFMA operand type; vec: elements per row load (8 / 4 / 0); masked: backward deriving dZ from dY and Y.
// Returns the configuration index (>= 0) and sets *split, or -1: the launch takes the one-shot form.
inline int colsplit_plan(int N, int Kpad, int NOUT, int ct_size, int vec, bool masked, bool have_out, int* split) {
    if (ct_size != 2 || vec == 0) return -1;
    const int forced = option(GCNPT_OPT_COL_SPLIT);
    const int n_rt8 = ceil_div(ceil_div(N, ROWS), 8);
    const int ksteps = Kpad / 32, n_ctiles = ceil_div(NOUT, 16), chunks_pt = ceil_div(ROWS * (Kpad / 8), RT_THREADS);
    const int c_max = forced > 0 ? 8 : std::min(8, 32 / n_rt8);
    if (forced == 0 || !have_out) return -1;
    if (forced < 0 && (c_max < 2 || option(GCNPT_OPT_FOUR_WAVES) >= 0 || (size_t)ksteps * n_ctiles * 1024 < (size_t)170 * 1024)) return -1;
    int best = -1, best_c = 0, best_regs = 1 << 30;
    for (int i = 0; i < 6; ++i) {
        const ColSplitCfg& g = COLSPLIT_CFGS[i];
        if (ksteps > g.ks || chunks_pt > g.pi || (masked && !g.masked_ok)) continue;
        const int c_min = ceil_div(n_ctiles, RT_WAVES * g.ntw);           // the fewest workgroups per row tile this configuration allows
        const int want = forced > 0 ? forced : std::min(c_max, ceil_div(n_ctiles, 2));      // as many as there are idle CUs for
        const int c = std::max(c_min, std::min(want, n_ctiles)), regs = g.ntw * g.ks + g.pi;
        if (c > c_max && c > c_min) continue;
        if (forced < 0 && c > c_max) continue;
        if (regs < best_regs) { best = i; best_c = c; best_regs = regs; }
    }
    if (best < 0 || (forced < 0 && best_c < 2)) return -1;
    *split = best_c;
    return best;
}

// Weight gradients the next backward-data launch should carry (layers_bwd_impl sets it around that one call; thread-local because it
// is only an argument that skips four levels of dispatch templates, not state: it never outlives the call that set it)
struct SideWgrad { const SideWgrads* sw = nullptr; bool carried = false; };
extern thread_local SideWgrad t_side;      // defined in rowtile_kernels.hip


// one precision combination x one direction of the row-tile kernel (rowtile_part.hip): combo 0 = exact f32; 1..4 = bf16 MFMA operands
// with (in, out) activations (f32, f32), (f32, bf16), (bf16, f32), (bf16, bf16); mode 0 = forward, 1 = backward-data deriving dZ from
// dY and Y, 2 = backward-data on ready-made dZ rows
int rowtile_launch(int combo, int mode, hipStream_t s, const RowTileParams& p);

#ifdef GCNPT_RT_PART
template <typename CT, typename IT, typename OT, bool BWD, int VEC, int NTW, int KSMAX, bool DZIN = false, int NWV = 8>
static inline int launch_rowtile_cfg(hipStream_t s, const RowTileParams& p) {
    const int stride = lds_stride_dw(p.Kpad * (int)sizeof(CT) / 4) * 4 / (int)sizeof(CT);
    const int ncols_pass = NWV * NTW * 16;
    const int ostride = out_stride_dw(std::min(round_up(p.NOUT, 16), ncols_pass) * (int)sizeof(OT) / 4) * 4 / (int)sizeof(OT);
    const size_t s_bytes = (size_t)ROWS * stride * sizeof(CT), o_bytes = (size_t)ROWS * ostride * sizeof(OT);
    const size_t lds = s_bytes + (BWD ? std::max(s_bytes, o_bytes) : o_bytes) +
                       (size_t)ROWS * 13 * sizeof(int) + (size_t)std::max(NWV * WAVE, ncols_pass) * sizeof(float);
    if (lds > 160 * 1024) return fail(GCNPT_E_UNSUPPORTED, "layer: K=%d needs %zu B of LDS per workgroup", p.K, lds);
    const int n_tiles = ceil_div(p.N, ROWS);
    // the uniform-precision instantiations can carry the layer above's weight gradient on the CUs that have no row tile
    if constexpr (NWV == 8 && BWD && DZIN && std::is_same<IT, OT>::value && sizeof(CT) == sizeof(IT)) {
        if (t_side.sw && t_side.sw->blocks > 0) {
            auto kern = rowtile_wgrad_kernel<CT, IT, OT, VEC, NTW, KSMAX>;
            GCNPT_LDS_ATTR_ONCE(kern, 160 * 1024);
            const int wg_first = round_up(n_tiles, 8);
            const int grid = wg_first + t_side.sw->blocks;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(RT_THREADS), std::max(lds, weight_grad_lds(RT_WAVES)), s, p, *t_side.sw, n_tiles, wg_first);
            note_launch(grid, RT_THREADS, std::max(lds, weight_grad_lds(RT_WAVES)), sizeof(p) + sizeof(SideWgrads) + 8);
            GCNPT_HIP_CHECK(hipGetLastError());
            t_side.carried = true;
            return GCNPT_OK;
        }
    }
    auto kern = rowtile_kernel<CT, IT, OT, BWD, VEC, NTW, KSMAX, DZIN, NWV>;
    GCNPT_LDS_ATTR_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(NWV * WAVE), lds, s, p);
    note_launch(n_tiles, NWV * WAVE, lds, sizeof(p));
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

// Big batches (more row tiles than CUs): workgroups of 4 waves, two or three per CU, so that one workgroup's memory waits overlap another's
// arithmetic (the 8-wave form keeps ~240 registers per wave: one workgroup per CU, nothing to overlap with).  Measured (C2 widths,
// B = 64 ... 1024; C5 widths): slower below ~256 tiles (B = 80: +20 %), 3 % faster at B = 100, 9-12 % faster from B = 256 on and at the
// C5 shape; not when the narrower workgroup needs one more column pass than the wide one (360 columns: 23 tiles = 20 + 3).
// gcnpt_set_option(GCNPT_OPT_FOUR_WAVES, 0 / 1) forces the choice (A/B, tests).
static inline bool use_four_waves(const RowTileParams& p) {
    const int forced = option(GCNPT_OPT_FOUR_WAVES);
    if (forced == 0 || forced == 1) return forced == 1;
    if (ceil_div(p.N, ROWS) <= 256) return false;
    const int n_tiles = ceil_div(p.NOUT, 16);
    const int pass8 = n_tiles <= 24 ? 1 : ceil_div(n_tiles, 32), pass4 = n_tiles <= 24 ? 1 : ceil_div(n_tiles, 20);
    return pass4 <= pass8;
}

// the column-split form for small batches of wide layers (colsplit_body.h): GCNPT_OK / an error when it took the launch, GCNPT_NOT_TAKEN when it does not apply
constexpr int GCNPT_NOT_TAKEN = 1;
template <typename CT, typename IT, typename OT, bool BWD, int VEC, bool DZIN>
static inline int try_colsplit(hipStream_t s, const RowTileParams& p);

// output tiles per wave: the smallest of {2,3,4} that covers NOUT in one pass (8 waves x NTW x 16 columns),
// with as many K-steps of weight fragments resident in registers as ~128 VGPRs allow.  Rows that cannot be
// read 16 bytes at a time (width not a multiple of 8, unaligned base) take the element-load instantiation.
template <typename CT, typename IT, typename OT, bool BWD, int VEC, bool DZIN>
static inline int launch_rowtile_vec(hipStream_t s, const RowTileParams& p) {
    const int n_tiles = ceil_div(p.NOUT, 16);
    {
        const int rc = try_colsplit<CT, IT, OT, BWD, VEC, DZIN>(s, p);
        if (rc != GCNPT_NOT_TAKEN) return rc;
    }
    if (use_four_waves(p)) {                                  // 4 waves cover 8 / 12 / 16 column tiles per pass
        if (n_tiles <= 4 * 2) return launch_rowtile_cfg<CT, IT, OT, BWD, VEC, 2, 12, DZIN, 4>(s, p);
        if (n_tiles <= 4 * 3) return launch_rowtile_cfg<CT, IT, OT, BWD, VEC, 3, 7, DZIN, 4>(s, p);
        if (n_tiles <= 4 * 4) return launch_rowtile_cfg<CT, IT, OT, BWD, VEC, 4, 5, DZIN, 4>(s, p);
        if (n_tiles > 20 && n_tiles <= 24) return launch_rowtile_cfg<CT, IT, OT, BWD, VEC, 6, 3, DZIN, 4>(s, p);      // 360 columns in one pass
        return launch_rowtile_cfg<CT, IT, OT, BWD, VEC, 5, 4, DZIN, 4>(s, p);      // 300 / 600 columns: one / two passes of 20 tiles
    }
    if (n_tiles <= RT_WAVES * 2) {
        // 13 k-steps = the C-GCN input width (2 x 200 BiLSTM states): one more resident k-step instead of a second load phase
        if (p.Kpad / (sizeof(CT) == 2 ? 32 : 16) == 13) return launch_rowtile_cfg<CT, IT, OT, BWD, VEC, 2, 13, DZIN>(s, p);
        return launch_rowtile_cfg<CT, IT, OT, BWD, VEC, 2, 12, DZIN>(s, p);
    }
    if (n_tiles <= RT_WAVES * 3) return launch_rowtile_cfg<CT, IT, OT, BWD, VEC, 3, 7, DZIN>(s, p);
    return launch_rowtile_cfg<CT, IT, OT, BWD, VEC, 4, 5, DZIN>(s, p);
}

template <typename CT, typename IT, typename OT, bool BWD, bool DZIN = false>
static inline int launch_rowtile(hipStream_t s, const RowTileParams& p) {
    if (p.vec_in == 8) return launch_rowtile_vec<CT, IT, OT, BWD, 8, DZIN>(s, p);
    if (p.vec_in == 4) return launch_rowtile_vec<CT, IT, OT, BWD, 4, DZIN>(s, p);
    return launch_rowtile_cfg<CT, IT, OT, BWD, 0, 4, BWD ? 3 : 4, DZIN>(s, p);
}

#endif

}  // namespace gcnpt
