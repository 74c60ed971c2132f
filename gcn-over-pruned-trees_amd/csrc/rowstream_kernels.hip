// GCN layer forward / backward-data for BIG batches on gfx950 (CDNA4): reference model/gcn.py:269-271, 390-393 and autograd.
// Same arithmetic as rowtile_kernels.hip (bf16 MFMA operands, fp32 accumulation):
//   forward        out = dropout(relu((((A+I) h) W^T + 2 b) / (deg + 1)))
//   backward-data  dh  = ((A+I)^T dZ) W     (dZ given; optional hand-over: dh * 1[layer input > 0] * scale / (deg + 1) = dZ of the layer below)
//
// rowtile_kernels.hip is built for a batch that gives every CU ONE 32-row tile: one latency chain, everything in flight at once.
// With thousands of tiles (B=128, T=300: 1200) that chain repeats tile after tile with nothing overlapped -- 28 k cycles per tile of
// which 7.7 k are MFMA -- and every tile re-streams the whole weight image through its CU (360 KB at 600 -> 300).  This kernel is the
// streaming form of the same layer:
//   * a workgroup is PERSISTENT (grid = min(tiles, CUs)) and owns 64-row tiles t, t + grid, ...: the weight stream per row is halved;
//   * its 12 waves have two roles.  Waves 0-3 (loaders) gather tile i+1 -- own row + the row's entries, fp32 sums, bf16 -> LDS -- while
//     waves 4-11 (matrix waves) run tile i on the matrix cores out of the other LDS buffer, the weight fragments streaming through a
//     rotating register window (L2-resident image, 3-4 k-steps ahead).  Three workgroup barriers per tile, no other synchronisation;
//   * the tile's adjacency (ELL heads, degrees) is fetched two tiles ahead, so the gather never waits for it.
// STATUS: parity-green (tests/test_gpu_stream.py) but SLOWER than the row-tile kernel at the one shape it was built for, so it is
// opt-in (GCNPT_ROWSTREAM=1, see rowstream_wanted below).
// Rows are addressed as in rowtile_kernels.hip (padded [B,T] or token-packed with T = 0), the fragment image of S = (A+I)h is
// written by the loaders from the LDS tile (forward), the epilogue and the row stores are the row-tile kernel's.
#include <cstdlib>

#include "layer_common.h"

namespace gcnpt {

#ifdef GCNPT_STAMPS
#define RS_STAMP(buf, who, slot)                                                                         \
    do {                                                                                                 \
        if ((buf) && (int)threadIdx.x == (who)) {                                                        \
            unsigned long long _t;                                                                       \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                  \
            (buf)[(size_t)blockIdx.x * 16 + (slot)] = _t;                                                \
        }                                                                                                \
    } while (0)
#else
#define RS_STAMP(buf, who, slot) do {} while (0)
#endif

constexpr int RS_ROWS = 64;
constexpr int RS_LD_WAVES = 4, RS_MM_WAVES = 8;   // 12 waves = 3 per SIMD: up to 168 VGPRs each (16 waves at 128 spilled > 100 registers)
constexpr int RS_THREADS = (RS_LD_WAVES + RS_MM_WAVES) * 64;
constexpr int RS_LD_THREADS = RS_LD_WAVES * 64;
constexpr int RS_NTW = 3;                   // output tiles per matrix wave and pass: 8 x 3 x 16 = 384 columns per pass
constexpr int RS_PASS_COLS = RS_MM_WAVES * RS_NTW * 16;
constexpr int RS_KU = 3;                    // weight k-steps in flight per matrix wave
constexpr int RS_NBU = 7;                   // entries fetched with a row in ONE round trip: all an ELL head carries (a later trip per
                                            // extra entry, item after item, made the gather 30 k cycles per tile)
constexpr int RS_NB_INLINE = 7;
constexpr int RS_META_INTS = RS_ROWS * 12 + 4;  // per tile: ELL heads [64][8], deg+1 [64] (float), sentence base [64], the rows that
                                                // aggregate something / that are plain copies, compacted [64] + [64], and their two counts

struct StreamParams {
    const void* src;            // fwd: h [N,K]    bwd: dZ [N,K]
    const uint4* wfrag;         // packed weights (gcnpt_pack_weights image for this direction)
    const float* bias;          // fwd
    const int32_t *g_row_ptr, *g_col_idx, *g_ell, *d_ell;
    void* out;                  // [N,NOUT]
    uint4* frag_out;            // fwd: NULL or fragment image of S = (A+I)h
    float *zero_a, *zero_b;     // bwd: NULL or accumulators to clear
    int zero_a_n, zero_b_n;
    const void* relu_src;       // bwd: NULL or this layer's input rows [N,NOUT] (hand-over, see gcnpt_layer_bwd_data)
    float next_scale;
    int N, T, K, NOUT, Kpad, n_tiles_rows;
    unsigned chunk_magic;
    int strideS, ostride;       // element strides of the S tiles (bf16) and of the out tile (OT)
    int o_off;                  // byte offset of the out tile in LDS, or -1: it aliases the S buffer the tile was computed from
    int meta_off, bias_off;     // byte offsets
    int vec_out;
    float scale, drop_p;
    unsigned drop_thresh16;
    uint64_t seed;
    const uint64_t* seed_dev;
    unsigned long long* stamps;     // diagnostic builds only
};

template <typename IT, typename OT, bool BWD, int VEC>
__global__ __launch_bounds__(RS_THREADS) void rowstream_kernel(const StreamParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    const size_t s_bytes = (size_t)RS_ROWS * p.strideS * 2;
    // (S buffers are addressed as rsm + offset everywhere: an array of two pointers indexed by i & 1 loses the LDS address space and
    //  every tile read becomes a flat_load that also waits for the global weight loads)
    auto s_buf = [&](int i) { return reinterpret_cast<bf16_t*>(rsm + (size_t)(i & 1) * s_bytes); };
    int* meta = reinterpret_cast<int*>(rsm + p.meta_off);                 // [3][RS_META_INTS]
    float* sbias = reinterpret_cast<float*>(rsm + p.bias_off);            // fwd: [round_up(NOUT, 16)]

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave < RS_LD_WAVES;
    const IT* src = static_cast<const IT*>(p.src);
    const int nchunk = p.Kpad / 8, ksteps = p.Kpad / 32;
    const int n_tiles = ceil_div(p.NOUT, 16), n_pass = ceil_div(n_tiles, RS_MM_WAVES * RS_NTW);
    const int kmax8 = VEC == 8 ? p.K - 8 : p.K - 4;
    const int my_tiles = p.n_tiles_rows > (int)blockIdx.x ? (p.n_tiles_rows - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    auto tile_of = [&](int i) { return (int)blockIdx.x + i * (int)gridDim.x; };
    uint64_t seed_off = 0;
    if (!BWD && p.seed_dev) seed_off = *p.seed_dev;

    auto ld8 = [&](size_t row, int k0c, raw8<IT>& dst) {
        if constexpr (VEC == 8) issue8<IT, true>(src, row, p.K, k0c, dst);
        else issue8_half<IT>(src, row, p.K, k0c, dst);
    };
    // a tile's adjacency -> LDS (loader threads): ELL heads, deg + 1, sentence base of each row
    auto load_meta = [&](int i) {
        if (i >= my_tiles) return;
        int* m = meta + (i % 3) * RS_META_INTS;
        const int r0 = tile_of(i) * RS_ROWS;
        if (tid < 2 * RS_ROWS) {
            const size_t er = (size_t)min(r0 + (tid >> 1), p.N - 1);
            int4 v = reinterpret_cast<const int4*>(p.g_ell)[er * 2 + (tid & 1)];
            if ((tid & 1) == 0 && r0 + (tid >> 1) >= p.N) v.x = 0;                        // rows past the end aggregate nothing
            reinterpret_cast<int4*>(m)[tid] = v;
        } else if (tid < 3 * RS_ROWS) {
            const int row = tid - 2 * RS_ROWS;
            const size_t er = (size_t)min(r0 + row, p.N - 1);
            reinterpret_cast<float*>(m)[8 * RS_ROWS + row] = (float)(p.d_ell[er * 8] + 1);   // gcn.py:261
            m[9 * RS_ROWS + row] = p.T ? (int)er / p.T * p.T : 0;
        } else if (tid < 4 * RS_ROWS) {
            // one wave splits the tile's rows into those that aggregate entries and those that are plain copies of their own row
            // (a pruned tree keeps a minority of the tokens): the gather handles the two kinds in separate, dense item lists
            const int row = tid - 3 * RS_ROWS;
            const bool agg = r0 + row < p.N && p.g_ell[(size_t)min(r0 + row, p.N - 1) * 8] > 0;
            const unsigned long long ma = __ballot(agg), below = (1ull << row) - 1ull;
            if (agg) m[10 * RS_ROWS + __popcll(ma & below)] = row;
            else m[11 * RS_ROWS + __popcll(~ma & below)] = row;
            if (row == 0) { m[12 * RS_ROWS] = __popcll(ma); m[12 * RS_ROWS + 1] = RS_ROWS - __popcll(ma); }
        }
    };
    auto div_chunk = [&](int x) { return (int)__umulhi((unsigned)x, p.chunk_magic); };

    // S[row] = src[row] + sum over the row's entries (gcn.py:269 + the explicit W(h) term of gcn.py:271), 64 rows, by the loaders.
    // Two dense item lists (item = one 8-column chunk of one row): plain rows are raw copies, ITA of them in flight per thread; rows
    // that aggregate fetch their own chunk and all (<= 7) ELL entries in ONE round trip, ITB items in flight per thread.
    constexpr int ITA = sizeof(IT) == 2 ? 4 : 2;
    constexpr int ITB = sizeof(IT) == 2 ? 3 : 1;
    // (a) plain rows: by the MATRIX waves, right after their MFMA phase (they finish a tile long before the loaders have gathered the
    //     next one, and raw copies need few registers); nthr threads numbered t
    auto gather_plain = [&](int i, int t, int nthr) {
        const int* m = meta + (i % 3) * RS_META_INTS;
        bf16_t* S = s_buf(i);
        const int r0 = tile_of(i) * RS_ROWS;
        const int n_plain = m[12 * RS_ROWS + 1];
        for (int it0 = t; it0 < n_plain * nchunk; it0 += nthr * ITA) {
            raw8<IT> own[ITA];
#pragma unroll
            for (int u = 0; u < ITA; ++u) {
                const int it = min(it0 + u * nthr, n_plain * nchunk - 1);
                const int li = div_chunk(it), row = m[11 * RS_ROWS + li];
                ld8((size_t)min(r0 + row, p.N - 1), min((it - li * nchunk) * 8, kmax8), own[u]);
            }
#pragma unroll
            for (int u = 0; u < ITA; ++u) {
                const int it = it0 + u * nthr;
                if (it >= n_plain * nchunk) continue;
                const int li = div_chunk(it), row = m[11 * RS_ROWS + li], k0 = (it - li * nchunk) * 8;
                const bool live = r0 + row < p.N && k0 < p.K;
                if constexpr (sizeof(IT) == 2) {
                    *reinterpret_cast<uint4*>(S + (size_t)row * p.strideS + k0) = live ? own[u].a : make_uint4(0, 0, 0, 0);
                } else {
                    float acc[8];
                    unpack8<IT>(own[u], live, acc);
                    tile<bf16_t>::put8(S + (size_t)row * p.strideS + k0, acc);
                }
            }
        }
    };
    // (b) rows that aggregate: by the loader waves
    auto gather = [&](int i, int tid) {
        const int* m = meta + (i % 3) * RS_META_INTS;
        bf16_t* S = s_buf(i);
        const int r0 = tile_of(i) * RS_ROWS;
        const int n_items = m[12 * RS_ROWS] * nchunk;
        for (int it0 = tid; it0 < n_items; it0 += RS_LD_THREADS * ITB) {
            raw8<IT> own[ITB], nb[ITB][RS_NBU];
#pragma unroll
            for (int u = 0; u < ITB; ++u) {
                const int it = min(it0 + u * RS_LD_THREADS, n_items - 1);
                const int li = div_chunk(it), row = m[10 * RS_ROWS + li], k0c = min((it - li * nchunk) * 8, kmax8);
                const size_t gr = (size_t)(r0 + row);
                const int n_ell = min(m[row * 8], RS_NB_INLINE), base = m[9 * RS_ROWS + row];
                ld8(gr, k0c, own[u]);
#pragma unroll
                for (int e = 0; e < RS_NBU; ++e) {
                    if (__ballot(e < n_ell) == 0ull) { nb[u][e] = own[u]; continue; }        // wave-uniform: no lane of the wave has an e-th entry
                    const size_t c = e < n_ell ? (size_t)(base + m[row * 8 + 1 + e]) : gr;   // no e-th entry: the row itself (same lines), dropped below
                    ld8(c, k0c, nb[u][e]);
                }
            }
#pragma unroll
            for (int u = 0; u < ITB; ++u) {
                const int it = it0 + u * RS_LD_THREADS;
                if (it >= n_items) continue;
                const int li = div_chunk(it), row = m[10 * RS_ROWS + li], k0 = (it - li * nchunk) * 8, k0c = min(k0, kmax8);
                const int grow = r0 + row;
                const bool live = k0 < p.K;
                const int cnt = m[row * 8], n_ell = min(cnt, RS_NB_INLINE), base = m[9 * RS_ROWS + row];
                float acc[8];
                unpack8<IT>(own[u], live, acc);
#pragma unroll
                for (int e = 0; e < RS_NBU; ++e) {
                    if (__ballot(e < n_ell) == 0ull) continue;
                    float v[8];
                    unpack8<IT>(nb[u][e], live && e < n_ell, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += v[j];
                }
                if (cnt > RS_NB_INLINE) {                                    // > 7 entries: the rest from the CSR
                    const int beg = p.T ? p.g_row_ptr[(size_t)(base / p.T) * (p.T + 1) + (grow - base)] : p.g_row_ptr[grow];
                    for (int e = RS_NB_INLINE; e < cnt; ++e) {
                        raw8<IT> x;
                        ld8((size_t)(base + p.g_col_idx[beg + e]), k0c, x);
                        float v[8];
                        unpack8<IT>(x, live, v);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] += v[j];
                    }
                }
                tile<bf16_t>::put8(S + (size_t)row * p.strideS + k0, acc);
            }
        }
    };

    // ---- prologue: adjacency of the first two tiles, bias, the first tile's gather
    if (loader) {
        __builtin_amdgcn_s_setprio(1);          // the gather is the longer side of every tile: the loaders win the issue arbitration
        load_meta(0);
        load_meta(1);
    } else if (!BWD) {
        for (int c = tid - RS_LD_THREADS; c < round_up(p.NOUT, 16); c += RS_THREADS - RS_LD_THREADS) sbias[c] = p.bias[min(c, p.NOUT - 1)];
    }
    __syncthreads();
    if (my_tiles > 0) {
        if (loader) gather(0, tid);
        else gather_plain(0, tid - RS_LD_THREADS, RS_THREADS - RS_LD_THREADS);
    }
    __syncthreads();

    const int mw = wave - RS_LD_WAVES;
    OT* out = static_cast<OT*>(p.out);
    const int tid0 = tid;
    for (int i = 0; i < my_tiles; ++i) {
        // The thread index is made opaque once per tile: otherwise hipcc hoists every per-thread address of the loop body (LDS rows of the
        // MFMA operands, the twelve out-tile slots, the store rows, the gather's item decode) out of the tile loop, runs out of registers
        // and reloads them from scratch at every use -- a memory round trip each (8-10 k cycles per phase in the stamps)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        int lane = tid & 63, arow = lane & 15, kgrp = lane >> 4;
        const int r0 = tile_of(i) * RS_ROWS;
        const bf16_t* S = s_buf(i);
        const int* m = meta + (i % 3) * RS_META_INTS;
        const float* rden = reinterpret_cast<const float*>(m) + 8 * RS_ROWS;
        OT* O = reinterpret_cast<OT*>(p.o_off >= 0 ? rsm + p.o_off : rsm + (size_t)(i & 1) * s_bytes);
        for (int pass = 0; pass < n_pass; ++pass) {
            f32x4_t acc[4][RS_NTW];
            const bool st = i == 1 && pass == 0;
            if (st) { RS_STAMP(p.stamps, 0, 0); RS_STAMP(p.stamps, RS_LD_THREADS, 8); }
            if (loader) {
                if (pass == 0) {
                    if (!BWD && p.frag_out) {                                    // the tile in MFMA fragment order for the weight gradient
                        const size_t nks = (size_t)ceil_div(p.N, 32);
#pragma unroll
                        for (int half = 0; half < 2; ++half)
                            if ((size_t)(2 * tile_of(i) + half) < nks)
                                emit_frag_image(p.frag_out, S + (size_t)half * 32 * p.strideS, p.strideS, p.K, wave, RS_LD_WAVES, lane, nks,
                                                (size_t)(2 * tile_of(i) + half));
                    }
                    if (st) RS_STAMP(p.stamps, 0, 1);
                    if (i + 1 < my_tiles) gather(i + 1, tid);
                    if (st) RS_STAMP(p.stamps, 0, 2);
                    load_meta(i + 2);
                }
            } else {
                // matrix waves: swapped operands (weights as A), so a lane ends up with 4 consecutive output columns of one row
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int j = 0; j < RS_NTW; ++j) acc[mt][j] = (f32x4_t){0, 0, 0, 0};
                const uint4* wb[RS_NTW];
#pragma unroll
                for (int j = 0; j < RS_NTW; ++j)
                    wb[j] = p.wfrag + (size_t)min(pass * RS_MM_WAVES * RS_NTW + j * RS_MM_WAVES + mw, n_tiles - 1) * ksteps * 64 + lane;
                uint4 w[RS_KU][RS_NTW];
#pragma unroll
                for (int u = 0; u < RS_KU; ++u)
#pragma unroll
                    for (int j = 0; j < RS_NTW; ++j) w[u][j] = wb[j][(size_t)min(u, ksteps - 1) * 64];
                for (int ks0 = 0; ks0 < ksteps; ks0 += RS_KU) {
#pragma unroll
                    for (int u = 0; u < RS_KU; ++u) {
                        const int ks = ks0 + u;
                        if (ks < ksteps) {
                            uint4 a[4];
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt)
                                a[mt] = *reinterpret_cast<const uint4*>(S + (size_t)(arow + 16 * mt) * p.strideS + ks * 32 + kgrp * 8);
#pragma unroll
                            for (int j = 0; j < RS_NTW; ++j) {
                                const bf16x8_t bq = __builtin_bit_cast(bf16x8_t, w[u][j]);
#pragma unroll
                                for (int mt = 0; mt < 4; ++mt)
                                    acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq, __builtin_bit_cast(bf16x8_t, a[mt]), acc[mt][j], 0, 0, 0);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < RS_NTW; ++j) w[u][j] = wb[j][(size_t)min(ks + RS_KU, ksteps - 1) * 64];   // the slot just used: k-step ks + KU
                    }
                }
            }
            if (st) RS_STAMP(p.stamps, RS_LD_THREADS, 9);
            if (!loader && pass == 0 && i + 1 < my_tiles) gather_plain(i + 1, tid - RS_LD_THREADS, RS_THREADS - RS_LD_THREADS);
            if (st) RS_STAMP(p.stamps, RS_LD_THREADS, 7);
            __syncthreads();                                                     // S fully read (the out tile may alias it), next tile gathered
            if (st) { RS_STAMP(p.stamps, 0, 3); RS_STAMP(p.stamps, RS_LD_THREADS, 10); }
            asm volatile("" : "+v"(tid));                                        // (a fresh thread index per phase, see above)
            lane = tid & 63; arow = lane & 15; kgrp = lane >> 4;
            if (!loader) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int row = 16 * mt + arow;
                    const float den = rden[row], inv = 1.0f / den;
#pragma unroll
                    for (int j = 0; j < RS_NTW; ++j) {
                        const int tl = pass * RS_MM_WAVES * RS_NTW + j * RS_MM_WAVES + mw;
                        if (tl >= n_tiles) continue;
                        const int col0 = tl * 16 + kgrp * 4;
                        float v[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) v[g] = acc[mt][j][g];
                        if constexpr (!BWD) {
                            const float4 bv = *reinterpret_cast<const float4*>(sbias + col0);
                            const float bq[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const float t = div_by(v[g] + 2.0f * bq[g], den, inv);   // gcn.py:270-271 (the bias enters twice), 390
                                v[g] = t > 0.0f ? t : 0.0f;                             // gcn.py:392
                            }
                            if (p.drop_p > 0.0f) {                                      // gcn.py:393: one hash per column pair
#pragma unroll
                                for (int h2 = 0; h2 < 2; ++h2) {
                                    const unsigned dh = drop_hash(p.seed + seed_off, (unsigned)(r0 + row), (unsigned)(col0 >> 1) + h2);
                                    v[2 * h2] = drop_keep(dh, 0u, p.drop_thresh16) ? v[2 * h2] * p.scale : 0.0f;
                                    v[2 * h2 + 1] = drop_keep(dh, 1u, p.drop_thresh16) ? v[2 * h2 + 1] * p.scale : 0.0f;
                                }
                            }
                        }
                        OT* dst = O + (size_t)row * p.ostride + (col0 - pass * RS_PASS_COLS);
                        if constexpr (sizeof(OT) == 2) {
                            uint2 pk;
                            pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                            pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                            *reinterpret_cast<uint2*>(dst) = pk;
                        } else {
                            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                    }
                }
            }
            if (st) RS_STAMP(p.stamps, RS_LD_THREADS, 11);
            __syncthreads();
            if (st) { RS_STAMP(p.stamps, 0, 4); RS_STAMP(p.stamps, RS_LD_THREADS, 12); }
            asm volatile("" : "+v"(tid));
            // ---- whole rows leave (all waves, 16 threads per row)
            {
                const int c_lo = pass * RS_PASS_COLS, width = min(p.NOUT, c_lo + RS_PASS_COLS) - c_lo;
                const OT* relu = BWD ? static_cast<const OT*>(p.relu_src) : nullptr;
                auto store_rows = [&](auto vtag) {
                    using V = decltype(vtag);
                    constexpr int PER = (int)sizeof(V) / (int)sizeof(OT);
                    constexpr int NWD = (int)sizeof(V) / 4;
                    const int pieces = width / PER;
                    for (int row = tid >> 4; row < RS_ROWS; row += RS_THREADS / 16) {
                        const int r = r0 + row;
                        if (r >= p.N) continue;
                        const float f = BWD && relu ? p.next_scale / rden[row] : 1.0f;
                        for (int pc = tid & 15; pc < pieces; pc += 16) {
                            V o = *reinterpret_cast<const V*>(O + (size_t)row * p.ostride + pc * PER);
                            if (BWD && relu) {                                   // hand-over: dZ of the layer below instead of dh
                                const V hin = *reinterpret_cast<const V*>(relu + (size_t)r * p.NOUT + c_lo + pc * PER);
                                if constexpr (sizeof(OT) == 2) {
                                    unsigned* ow = reinterpret_cast<unsigned*>(&o);
                                    const unsigned* hw = reinterpret_cast<const unsigned*>(&hin);
#pragma unroll
                                    for (int q = 0; q < NWD; ++q) {
                                        const float lo = bf16_to_f32((bf16_t)(hw[q] & 0xffffu)) > 0.0f ? bf16_to_f32((bf16_t)(ow[q] & 0xffffu)) * f : 0.0f;
                                        const float hi = bf16_to_f32((bf16_t)(hw[q] >> 16)) > 0.0f ? bf16_to_f32((bf16_t)(ow[q] >> 16)) * f : 0.0f;
                                        ow[q] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                                    }
                                } else {
                                    float* ow = reinterpret_cast<float*>(&o);
                                    const float* hw = reinterpret_cast<const float*>(&hin);
#pragma unroll
                                    for (int q = 0; q < NWD; ++q) ow[q] = hw[q] > 0.0f ? ow[q] * f : 0.0f;
                                }
                            }
                            *reinterpret_cast<V*>(out + (size_t)r * p.NOUT + c_lo + pc * PER) = o;
                        }
                    }
                };
                if (p.vec_out == 16) store_rows(uint4{});
                else store_rows(uint2{});
            }
            if (i == 0 && pass == 0) {                                           // cleared accumulators for the weight gradient that follows
                if (p.zero_a)
                    for (int q = blockIdx.x * RS_THREADS + tid; q < p.zero_a_n; q += gridDim.x * RS_THREADS) p.zero_a[q] = 0.0f;
                if (p.zero_b)
                    for (int q = blockIdx.x * RS_THREADS + tid; q < p.zero_b_n; q += gridDim.x * RS_THREADS) p.zero_b[q] = 0.0f;
            }
            if (st) { RS_STAMP(p.stamps, 0, 5); RS_STAMP(p.stamps, RS_LD_THREADS, 13); }
            __syncthreads();                                                     // the out tile has left: its LDS may be gathered into again
            if (st) { RS_STAMP(p.stamps, 0, 6); RS_STAMP(p.stamps, RS_LD_THREADS, 14); }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// dZ = dY * 1[Y > 0] * scale / (deg + 1) as rows + fragment image (the top layer of a big-batch backward sweep), or -- MASK = false --
// only the fragment image of rows that already are dZ.  One workgroup per 32 rows.
// ---------------------------------------------------------------------------------------------------
constexpr int DZ_THREADS = 256;
struct DzParams {
    const void *dy, *y;
    const int32_t* d_ell;
    void* dz;                   // MASK: [N,H] written
    uint4* frag;                // fragment image of dZ (may be NULL when MASK)
    int N, H, Hpad, stride, vec;
    float scale;
};

template <typename IT, bool MASK>
__global__ __launch_bounds__(DZ_THREADS) void dz_rows_kernel(const DzParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dzm[];
    bf16_t* Z = reinterpret_cast<bf16_t*>(dzm);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * 32;
    const IT* dy = static_cast<const IT*>(p.dy);
    const IT* y = static_cast<const IT*>(p.y);
    IT* dz = static_cast<IT*>(p.dz);
    const int nchunk = p.Hpad / 8;
    for (int it = tid; it < 32 * nchunk; it += DZ_THREADS) {
        const int row = it / nchunk, k0 = (it - row * nchunk) * 8;
        const int r = r0 + row;
        const bool live = r < p.N && k0 < p.H;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.0f;
        if (live) {
            const float f = MASK ? p.scale / (float)(p.d_ell[(size_t)r * 8] + 1) : 1.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (k0 + j < p.H) {
                    const float g = io<IT>::load1(dy + (size_t)r * p.H + k0 + j);
                    if constexpr (MASK) v[j] = io<IT>::load1(y + (size_t)r * p.H + k0 + j) > 0.0f ? g * f : 0.0f;
                    else v[j] = g;
                }
            }
            if constexpr (MASK) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (k0 + j < p.H) io<IT>::store1(dz + (size_t)r * p.H + k0 + j, v[j]);
            }
        }
        tile<bf16_t>::put8(Z + (size_t)row * p.stride + k0, v);
    }
    __syncthreads();
    if (p.frag) emit_frag_image(p.frag, Z, p.stride, p.H, wave, DZ_THREADS / 64, lane, (size_t)gridDim.x, (size_t)blockIdx.x);
}

}  // namespace gcnpt

// =====================================================================================================
// host side
// =====================================================================================================
using namespace gcnpt;

// Whether the streaming kernel takes a layer: bf16 operands, rows readable in (half) chunks, enough tiles that a persistent
// workgroup gets several, and the two S buffers (+ a separate out tile when the output needs several column passes) fit the LDS.
static bool rowstream_plan(StreamParams& p, int N, int K, int NOUT, size_t es_out, size_t* lds_out) {
    const int Kpad = round_up(K, 32);
    p.strideS = lds_stride_dw(Kpad / 2) * 2;
    const size_t s_bytes = (size_t)RS_ROWS * p.strideS * 2;
    const int n_tiles = ceil_div(NOUT, 16), n_pass = ceil_div(n_tiles, RS_MM_WAVES * RS_NTW);
    const int ocols = std::min(round_up(NOUT, 16), RS_PASS_COLS);
    p.ostride = (int)(out_stride_dw((int)(ocols * es_out / 4)) * 4 / es_out);
    const size_t o_bytes = (size_t)RS_ROWS * p.ostride * es_out;
    size_t off = 2 * s_bytes;
    if (n_pass == 1 && o_bytes <= s_bytes) p.o_off = -1;
    else { p.o_off = (int)off; off += round_up((int)o_bytes, 16); }
    p.meta_off = (int)off; off += (size_t)3 * RS_META_INTS * 4;
    p.bias_off = (int)off; off += (size_t)round_up(NOUT, 16) * 4;
    *lds_out = off;
    return off <= 160 * 1024;
}

// EXPERIMENTAL, opt-in (environment variable GCNPT_ROWSTREAM=1, read at every call: no latched state): measured on MI355X at
// B=128, T=300, 300 -> 300 / 300 -> 600: forward 59 us against 45 us for the row-tile kernel, backward-data 96
// against 77 us.  The matrix waves finish a 64-row tile in ~7 k cycles, but the four loader waves need ~31 k cycles to gather the next
// one (3-item batches, one memory round trip + ~2.5 k cycles of unpack/add per batch, sharing their SIMDs' issue slots with the
// matrix waves), and a workgroup only gets 2-3 tiles, so the un-overlapped first gather is a quarter of its time.
static bool rowstream_wanted(int N) {
    const char* e = std::getenv("GCNPT_ROWSTREAM");
    return N >= 16384 && e && e[0] == '1';      // >= 256 tiles of 64 rows: every CU has a tile, most have several
}

bool rowstream_enabled() { return rowstream_wanted(1 << 30); }

template <typename IT, typename OT, bool BWD>
static int launch_rowstream(hipStream_t s, const StreamParams& p, size_t lds, int vec_in) {
    // as many workgroups as give every one the same number of tiles (600 tiles: 200 workgroups of 3, not 256 of 2-3)
    const int grid = ceil_div(p.n_tiles_rows, ceil_div(p.n_tiles_rows, 256));
    if (vec_in == 8) {
        auto kern = rowstream_kernel<IT, OT, BWD, 8>;
        GCNPT_LDS_ATTR_ONCE(kern, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(RS_THREADS), lds, s, p);
    } else {
        auto kern = rowstream_kernel<IT, OT, BWD, 4>;
        GCNPT_LDS_ATTR_ONCE(kern, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(RS_THREADS), lds, s, p);
    }
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

template <bool BWD>
static int dispatch_rowstream(hipStream_t s, const StreamParams& p, size_t lds, int vec_in, int in_dtype, int out_dtype) {
    if (in_dtype == GCNPT_F32 && out_dtype == GCNPT_F32) return launch_rowstream<float, float, BWD>(s, p, lds, vec_in);
    if (in_dtype == GCNPT_F32 && out_dtype == GCNPT_BF16) return launch_rowstream<float, bf16_t, BWD>(s, p, lds, vec_in);
    if (in_dtype == GCNPT_BF16 && out_dtype == GCNPT_F32) return launch_rowstream<bf16_t, float, BWD>(s, p, lds, vec_in);
    return launch_rowstream<bf16_t, bf16_t, BWD>(s, p, lds, vec_in);
}

// Called by gcnpt_layer_fwd / gcnpt_layer_bwd_data (rowtile_kernels.hip) before they launch the row-tile kernel: returns 1 when the
// streaming kernel took the layer, 0 when it does not apply, < 0 on error.
int rowstream_try_fwd(hipStream_t s, const void* h, int h_dtype, const void* w_fwd, const float* bias, const int32_t* row_ptr,
                      const int32_t* col_idx, const int32_t* ell, const int32_t* deg_ell, int N, int T, int Din, int H, void* out,
                      int out_dtype, float drop_p, uint64_t seed, void* s_frag, const uint64_t* seed_dev, int vec_in, int vec_out) {
    if (!rowstream_wanted(N) || vec_in < 4 || vec_out < 8) return 0;
    StreamParams p{};
    size_t lds = 0;
    if (!rowstream_plan(p, N, Din, H, esize(out_dtype), &lds)) return 0;
    p.src = h; p.wfrag = static_cast<const uint4*>(w_fwd); p.bias = bias;
    p.g_row_ptr = row_ptr; p.g_col_idx = col_idx; p.g_ell = ell; p.d_ell = deg_ell ? deg_ell : ell;
    p.out = out; p.frag_out = static_cast<uint4*>(s_frag);
    p.N = N; p.T = T; p.K = Din; p.NOUT = H; p.Kpad = round_up(Din, 32); p.n_tiles_rows = ceil_div(N, RS_ROWS);
    p.chunk_magic = 0xffffffffu / (unsigned)(p.Kpad / 8) + 1u;
    p.vec_out = vec_out;
    p.drop_p = drop_p; p.scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    p.drop_thresh16 = (unsigned)((double)drop_p * 65536.0);
    p.seed = seed; p.seed_dev = seed_dev;
    p.stamps = static_cast<unsigned long long*>(g_debug_stamps);
    const int rc = dispatch_rowstream<false>(s, p, lds, vec_in, h_dtype, out_dtype);
    return rc == GCNPT_OK ? 1 : rc;
}

int rowstream_try_bwd(hipStream_t s, const void* dZ, int g_dtype, const void* w_bwd, const int32_t* ell, const int32_t* rowT_ptr,
                      const int32_t* colT_idx, const int32_t* ellT, int N, int T, int Din, int H, void* dh, int dh_dtype, float* zero_dW,
                      float* zero_db, const void* relu_src, float next_scale, int vec_in, int vec_out) {
    if (!rowstream_wanted(N) || vec_in < 4 || vec_out < 8 || !dh) return 0;
    StreamParams p{};
    size_t lds = 0;
    if (!rowstream_plan(p, N, H, Din, esize(dh_dtype), &lds)) return 0;
    p.src = dZ; p.wfrag = static_cast<const uint4*>(w_bwd);
    p.g_row_ptr = rowT_ptr; p.g_col_idx = colT_idx; p.g_ell = ellT; p.d_ell = ell;
    p.out = dh;
    p.zero_a = zero_dW; p.zero_a_n = H * Din; p.zero_b = zero_db; p.zero_b_n = H;
    p.relu_src = relu_src; p.next_scale = next_scale;
    p.N = N; p.T = T; p.K = H; p.NOUT = Din; p.Kpad = round_up(H, 32); p.n_tiles_rows = ceil_div(N, RS_ROWS);
    p.chunk_magic = 0xffffffffu / (unsigned)(p.Kpad / 8) + 1u;
    p.vec_out = vec_out;
    p.stamps = static_cast<unsigned long long*>(g_debug_stamps);
    const int rc = dispatch_rowstream<true>(s, p, lds, vec_in, g_dtype, dh_dtype);
    return rc == GCNPT_OK ? 1 : rc;
}

// dZ rows (mask = 1) and / or their fragment image for N rows of width H
int launch_dz_rows(hipStream_t s, const void* dY, const void* Y, int dtype, const int32_t* ell, int N, int H, float scale, void* dz, void* z_frag,
                   int mask) {
    DzParams p{};
    p.dy = dY; p.y = Y; p.d_ell = ell; p.dz = dz; p.frag = static_cast<uint4*>(z_frag);
    p.N = N; p.H = H; p.Hpad = round_up(H, 32); p.stride = lds_stride_dw(p.Hpad / 2) * 2; p.scale = scale;
    const size_t lds = (size_t)32 * p.stride * 2;
    if (lds > 64 * 1024) return fail(GCNPT_E_UNSUPPORTED, "dz_rows: width %d needs %zu B of LDS", H, lds);
    const dim3 grid(ceil_div(N, 32));
    if (dtype == GCNPT_BF16) {
        if (mask) hipLaunchKernelGGL((dz_rows_kernel<bf16_t, true>), grid, dim3(DZ_THREADS), lds, s, p);
        else hipLaunchKernelGGL((dz_rows_kernel<bf16_t, false>), grid, dim3(DZ_THREADS), lds, s, p);
    } else {
        if (mask) hipLaunchKernelGGL((dz_rows_kernel<float, true>), grid, dim3(DZ_THREADS), lds, s, p);
        else hipLaunchKernelGGL((dz_rows_kernel<float, false>), grid, dim3(DZ_THREADS), lds, s, p);
    }
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}
