// Big batches (>= 16 k token rows): a GCN layer as TWO launches instead of one row-tile kernel -- reference model/gcn.py:269-271, 390-393.
//
// The row-tile kernel (rowtile_kernels.hip) is built for the headline batch (157 workgroups, one per CU, everything in flight at once).
// With thousands of tiles it runs them one after the other per CU, and every 32-row tile pulls the WHOLE weight matrix through its
// CU's vector-memory path again (360 KB per tile at 600 -> 300): 7.7 k cycles of a tile's ~28 k are MFMA, the rest is latency nobody
// hides.  Here the two halves of the layer get the launch shape each one wants:
//
//   rowprep_kernel  (memory side, 4 workgroups of 4 waves per CU):  S = (A + I) X as a gather.  Writes (a) the rows that aggregate at
//       least one neighbour (~1 in 8 in a pruned tree) into a row-major workspace -- every other row of S IS the input row --, and
//       (b) the 32-row tile as the fragment image the weight gradient contracts over (forward: of S; backward: of the raw dZ rows).
//   rowgemm_kernel  (matrix side, one workgroup of 8 waves per CU): out = epilogue(S W) for 128 / 160 rows per workgroup, so the weight
//       fragments pulled through the CU are shared by 8-10 row tiles instead of 2.  2 x 4 waves: a wave owns RTW row tiles x NTW column
//       tiles of accumulators and loads both operands straight from global memory into registers, one k-step ahead (rows: 16 B per
//       lane from the input or the workspace, chosen per row; weights: the packed fragment image, gcnpt_pack_weights).  No LDS.
//
// Same arithmetic, same order as the row-tile kernel (fp32 neighbour sums rounded to bf16 once, k-steps accumulated in order, the same
// epilogue expressions): outputs and fragment images are bit-identical to it (tests/test_gpu_split.py).  bf16 compute only.
#include "layer_common.h"

namespace gcnpt {

constexpr int RP_ROWS = 32, RP_THREADS = 256, RP_WAVES = RP_THREADS / WAVE;
constexpr int RP_NB = 4;            // neighbour rows fetched together (the row-tile kernel's NBU: same order of additions)
constexpr int RP_ITEMS = 4;         // 8-element chunks a thread copies per batch
constexpr int RP_INLINE = 7;        // neighbours per row that the ELL head carries (include/gcnpt.h)

struct RowPrepParams {
    const void* src;                // [N,K] rows (IT)
    const int32_t* g_row_ptr;       // pattern gathered over (fwd: A, bwd: A^T)
    const int32_t* g_col_idx;
    const int32_t* g_ell;
    bf16_t* agg;                    // [N,K] workspace: the aggregated rows (all rows when all_rows)
    void* frag_out;                 // NULL or fragment image of the tile
    int frag_of_agg;                // 1: image of S (forward), 0: image of the raw rows (backward: dZ)
    int all_rows;                   // also copy the rows that aggregate nothing (IT != bf16: the matrix kernel reads bf16 only)
    int N, T, K, Kpad;
    unsigned chunk_magic;
    unsigned long long* stamps;     // diagnostic builds only
};

template <typename IT, int VEC>
__global__ __launch_bounds__(RP_THREADS) void rowprep_kernel(const RowPrepParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rp_smem[];
    const int stride = lds_stride_dw(p.Kpad / 2) * 2;                     // bf16 elements
    bf16_t* X = reinterpret_cast<bf16_t*>(rp_smem);
    int* meta = reinterpret_cast<int*>(rp_smem + (size_t)RP_ROWS * stride * sizeof(bf16_t));
    int* rell = meta;                       // [32][8]
    int* rsb = meta + 8 * RP_ROWS;          // [32] first row of the row's sentence
    int* glist = meta + 9 * RP_ROWS;        // [32] rows that aggregate, compacted
    int* gcount = meta + 10 * RP_ROWS;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD x takes a contiguous run of tiles (as the row-tile kernel does): a tile's neighbour rows are its sentence's
    const int xg = blockIdx.x & 7, xq = gridDim.x >> 3, xr = gridDim.x & 7;
    const int tile_id = xg * xq + min(xg, xr) + (blockIdx.x >> 3);
    const int r0 = tile_id * RP_ROWS;
    const IT* src = static_cast<const IT*>(p.src);
    GCNPT_STAMP_REAL(p.stamps);
    GCNPT_STAMP(p.stamps, 0);

    // every wave loads all 32 heads and keeps its own copy of the tables (no barrier before the gather can start)
    const int erow = lane >> 1, ehalf = lane & 1;
    const size_t er = (size_t)min(r0 + erow, p.N - 1);
    const int4 ell_v = reinterpret_cast<const int4*>(p.g_ell)[er * 2 + ehalf];
    const int sb_v = p.T ? (int)er / p.T * p.T : 0;
    {
        const bool first = ehalf == 0;
        const int e0 = (first && r0 + erow >= p.N) ? 0 : ell_v.x;
        reinterpret_cast<int4*>(rell)[erow * 2 + ehalf] = make_int4(e0, ell_v.y, ell_v.z, ell_v.w);
        rsb[erow] = sb_v;
        const bool agg = first && e0 > 0;
        const unsigned long long m = __ballot(agg);
        if (agg) glist[__popcll(m & ((1ull << lane) - 1ull))] = erow;
        if (lane == 0) *gcount = __popcll(m);
    }
    wave_lds_fence();
    GCNPT_STAMP(p.stamps, 1);

    const int nchunk = p.Kpad / 8;
    auto div_chunk = [&](int x) { return (int)__umulhi((unsigned)x, p.chunk_magic); };
    const int kmax8 = VEC == 8 ? p.K - 8 : p.K - 4;
    auto ld8 = [&](size_t row, int k0c, raw8<IT>& dst) {
        if constexpr (VEC == 8) issue8<IT, true>(src, row, p.K, k0c, dst);
        else issue8_half<IT>(src, row, p.K, k0c, dst);
    };
    // 8 bf16 of a chunk -> workspace row (8-byte halves when rows are only 8-byte aligned; nothing past column K)
    auto to_ws = [&](int row, int k0, const uint4& u) {
        if (r0 + row >= p.N || k0 >= p.K) return;
        bf16_t* d = p.agg + (size_t)(r0 + row) * p.K + k0;
        if constexpr (VEC == 8) {
            *reinterpret_cast<uint4*>(d) = u;
        } else {
            *reinterpret_cast<uint2*>(d) = make_uint2(u.x, u.y);
            if (k0 + 8 <= p.K) *reinterpret_cast<uint2*>(d + 4) = make_uint2(u.z, u.w);
        }
    };
    auto pack8 = [&](const float (&v)[8]) {
        uint4 u;
        u.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        u.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        u.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
        u.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
        return u;
    };
    const bool want_tile = p.frag_out != nullptr;

    // (a) rows that aggregate something, compacted: item = (list slot, 8-column chunk); self + <= 4 neighbours leave in one round trip
    const int n_g = *gcount * nchunk;
    for (int base = 0; base < n_g; base += RP_THREADS) {
        const int gi = base + tid;
        const bool has = gi < n_g;
        const int li = has ? div_chunk(gi) : 0;
        const int row = has ? glist[li] : 0;
        const int k0 = has ? (gi - li * nchunk) * 8 : 0;
        const bool live = has && k0 < p.K;
        const int n = live ? rell[row * 8] : 0;
        const size_t r = (size_t)min(r0 + row, p.N - 1);
        const int sbase = rsb[row];
        const int k0c = min(k0, kmax8);
        raw8<IT> s, nb[RP_NB];
        ld8(r, k0c, s);
        const int n_ell = min(n, RP_INLINE);
#pragma unroll
        for (int e = 0; e < RP_NB; ++e) {
            const bool on = e < n_ell;
            const size_t c = on ? (size_t)(sbase + rell[row * 8 + 1 + e]) : (size_t)r0;
            ld8(c, on ? k0c : 0, nb[e]);
        }
        float acc[8];
        unpack8<IT>(s, live, acc);                                          // the explicit W(h) term, gcn.py:271
#pragma unroll
        for (int e = 0; e < RP_NB; ++e) {
            float v[8];
            unpack8<IT>(nb[e], e < n_ell, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
        auto round = [&](int e0, int lim, auto from_lds) {
            raw8<IT> q[RP_NB];
#pragma unroll
            for (int e = 0; e < RP_NB; ++e) {
                const bool on = e0 + e < lim;
                size_t c;
                if constexpr (decltype(from_lds)::value) {
                    c = (size_t)(sbase + rell[row * 8 + 1 + min(e0 + e, RP_INLINE - 1)]);
                } else {                                                    // > 7 entries: continue in the CSR
                    const int beg = p.T ? p.g_row_ptr[(size_t)(sbase / p.T) * (p.T + 1) + (r - sbase)] : p.g_row_ptr[r];
                    c = (size_t)(sbase + p.g_col_idx[on ? beg + e0 + e : beg]);
                }
                c = on ? c : r;
                ld8(c, k0c, q[e]);
            }
#pragma unroll
            for (int e = 0; e < RP_NB; ++e) {
                float v[8];
                unpack8<IT>(q[e], e0 + e < lim, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
        };
        for (int e0 = RP_NB; e0 < n_ell; e0 += RP_NB) round(e0, n_ell, std::true_type{});
        for (int e0 = RP_INLINE; e0 < n; e0 += RP_NB) round(e0, n, std::false_type{});
        if (has) {
            const uint4 u = pack8(acc);
            to_ws(row, k0, u);
            if (want_tile && p.frag_of_agg) *reinterpret_cast<uint4*>(X + (size_t)row * stride + k0) = u;
        }
    }

    GCNPT_STAMP(p.stamps, 2);
    // (b) the tile's own rows: into the LDS tile for the fragment image (raw mode: all of them; S mode: those that (a) did not write),
    //     and into the workspace when the matrix kernel cannot read the input rows as they are
    if (want_tile || p.all_rows) {
        const int n_items = RP_ROWS * nchunk;
        for (int it0 = 0; it0 < n_items; it0 += RP_ITEMS * RP_THREADS) {
            raw8<IT> self[RP_ITEMS];
#pragma unroll
            for (int u = 0; u < RP_ITEMS; ++u) {
                const int it = min(it0 + u * RP_THREADS + tid, n_items - 1);
                const int row = div_chunk(it), k0 = (it - row * nchunk) * 8;
                ld8((size_t)min(r0 + row, p.N - 1), min(k0, kmax8), self[u]);
            }
#pragma unroll
            for (int u = 0; u < RP_ITEMS; ++u) {
                const int it = it0 + u * RP_THREADS + tid;
                if (it >= n_items) continue;
                const int row = div_chunk(it), k0 = (it - row * nchunk) * 8;
                const bool live = r0 + row < p.N && k0 < p.K;
                uint4 w;
                if constexpr (sizeof(IT) == 2) {
                    w = live ? self[u].a : make_uint4(0, 0, 0, 0);                 // bf16 rows: the 16 bytes as they are
                } else {
                    float v[8];
                    unpack8<IT>(self[u], live, v);
                    w = pack8(v);
                }
                const bool is_agg = rell[row * 8] > 0;
                if (want_tile && !(p.frag_of_agg && is_agg)) *reinterpret_cast<uint4*>(X + (size_t)row * stride + k0) = w;
                if (p.all_rows && !is_agg) to_ws(row, k0, w);
            }
        }
    }
    GCNPT_STAMP(p.stamps, 3);
    if (!want_tile) return;
    __syncthreads();
    GCNPT_STAMP(p.stamps, 4);
    emit_frag_image(static_cast<uint4*>(p.frag_out), X, stride, p.K, wave, RP_WAVES, lane, (size_t)gridDim.x, (size_t)tile_id);
    GCNPT_STAMP(p.stamps, 5);
}

// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int RG_THREADS = 512, RG_WR = 2, RG_WC = 4;      // 8 waves: 2 row groups x 4 column groups

struct RowGemmParams {
    const bf16_t* rows_plain;       // [N,K] rows that aggregate nothing (the layer's input, or the workspace when it holds every row)
    const bf16_t* rows_agg;         // [N,K] workspace: rows that aggregate (sel_ell[8r] > 0)
    const int32_t* sel_ell;         // ELL head of the pattern rowprep gathered over
    const int32_t* d_ell;           // ELL head of the forward pattern: [8r] = deg
    const uint4* wfrag;             // packed weights (A operand: output columns on the MFMA's M side)
    const float* bias;              // fwd
    void* out;                      // [N,NOUT]
    const void* relu_src;           // bwd: NULL or the layer's input rows [N,NOUT] (hand-over: the result leaves as dZ of the layer below)
    float* zero_a; float* zero_b;   // accumulators cleared for the weight gradient that follows
    int zero_a_n, zero_b_n;
    int N, K, NOUT, ksteps, n_tiles, tiles_per_pass, passes, row_blocks;
    int bwd, out_f32;
    int ostride_b, n_phase, store16;    // epilogue: LDS out-tile row stride (bytes), row groups per phase, 16- or 8-byte row pieces
    float scale, drop_p, next_scale;
    unsigned drop_thresh16;
    uint64_t seed;
    const uint64_t* seed_dev;
    unsigned long long* stamps;     // diagnostic builds only
    int knob;
};

constexpr int RG_WSTAGES = 2;       // weight fragments (L2-resident): two register sets, loaded one k-step ahead
constexpr int RG_BSTAGES = 3;       // row fragments (HBM): global -> registers three k-steps ahead -> LDS ring (2 stages) one k-step ahead
constexpr int RG_LDS_STAGES = 2;

template <int RTW>
__host__ __device__ constexpr int rg_bslots() { return (RG_WR * RTW + 7) / 8; }     // row tiles a wave stages per k-step

template <int VEC, int RTW, int NTW>
__global__ __launch_bounds__(RG_THREADS) void rowgemm_kernel(const RowGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rg_smem[];
    constexpr int RT_ALL = RG_WR * RTW;                  // row tiles of the workgroup
    constexpr int NBL = rg_bslots<RTW>();
    uint4* ring = reinterpret_cast<uint4*>(rg_smem);     // [RG_LDS_STAGES][RT_ALL][64] row fragments (MFMA B operand order)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / RG_WC, wc = wave % RG_WC;
    if (p.zero_a)
        for (int i = blockIdx.x * RG_THREADS + tid; i < p.zero_a_n; i += gridDim.x * RG_THREADS) p.zero_a[i] = 0.0f;
    if (p.zero_b)
        for (int i = blockIdx.x * RG_THREADS + tid; i < p.zero_b_n; i += gridDim.x * RG_THREADS) p.zero_b[i] = 0.0f;
    // the column passes of one row block run on the same XCD (blocks b, b + 8, ... share one): its rows cross the fabric once
    const int xg = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int rb = (rest / p.passes) * 8 + xg, pass = rest % p.passes;
    if (rb >= p.row_blocks) return;
    GCNPT_STAMP_REAL(p.stamps);
    GCNPT_STAMP(p.stamps, 0);
    const int rw0 = rb * RT_ALL * 16;                    // first row of the workgroup
    const int arow = lane & 15, kgrp = lane >> 4;
    const int tile0 = pass * p.tiles_per_pass + wc * NTW;
    const int tile_hi = min(p.n_tiles, (pass + 1) * p.tiles_per_pass);
    uint64_t seed_off = 0;
    if (!p.bwd && p.seed_dev) seed_off = *p.seed_dev;

    // weight fragments: a wave's column tiles are wave-uniform, so each load is a scalar base + one lane offset per k-step
    const uint4* wbase[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) wbase[j] = p.wfrag + (size_t)min(tile0 + j, p.n_tiles - 1) * p.ksteps * 64;
    auto load_w = [&](int ks, uint4 (&dst)[NTW]) {
        int off = min(ks, p.ksteps - 1) * 64 + lane;
#ifdef GCNPT_STAMPS
        if (p.knob & 1) off = lane;                                       // experiment: every weight load hits the same lines
#endif
#pragma unroll
        for (int j = 0; j < NTW; ++j) dst[j] = wbase[j][off];
    };
    uint4 wq[RG_WSTAGES][NTW];
    load_w(0, wq[0]);
    // the row tiles this wave stages for the workgroup: tile slot*8 + wave (a duplicate of the last one past the end)
    const bf16_t* rowp[NBL];
    int stile[NBL];
#pragma unroll
    for (int sl = 0; sl < NBL; ++sl) {
        stile[sl] = min(sl * 8 + wave, RT_ALL - 1);
        const size_t r = (size_t)min(rw0 + stile[sl] * 16 + arow, p.N - 1);
        rowp[sl] = (p.sel_ell[r * 8] > 0 ? p.rows_agg : p.rows_plain) + r * (size_t)p.K;
    }
    const int kmax = VEC == 8 ? p.K - 8 : p.K - 4;
    auto load_b = [&](int ks, uint4 (&dst)[NBL]) {
        int kc = min(ks, p.ksteps - 1) * 32 + kgrp * 8;
#ifdef GCNPT_STAMPS
        if (p.knob & 2) kc = kgrp * 8;                                    // experiment: every row load hits the same lines
#endif
        // columns past K: real values of the row (clamped); they meet zero weights (the packed image is zero padded)
#pragma unroll
        for (int sl = 0; sl < NBL; ++sl) {
            if constexpr (VEC == 8) {
                const uint4 t = *reinterpret_cast<const uint4*>(rowp[sl] + min(kc, kmax));
                dst[sl] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                const uint2 lo = *reinterpret_cast<const uint2*>(rowp[sl] + min(kc, kmax));
                const uint2 hi = *reinterpret_cast<const uint2*>(rowp[sl] + min(kc + 4, kmax));
                dst[sl] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    };
    auto park_b = [&](int ks, const uint4 (&src)[NBL]) {                      // staged fragments -> LDS ring, in MFMA operand order
        uint4* st = ring + (size_t)(ks % RG_LDS_STAGES) * RT_ALL * 64;
#pragma unroll
        for (int sl = 0; sl < NBL; ++sl) st[stile[sl] * 64 + lane] = src[sl];
    };
    uint4 bq[RG_BSTAGES][NBL];
    load_b(0, bq[0]);
    load_b(1, bq[1]);
    load_b(2, bq[2]);
    GCNPT_STAMP(p.stamps, 1);

    f32x4_t acc[RTW][NTW];
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[rt][j] = (f32x4_t){0, 0, 0, 0};

    park_b(0, bq[0]);
    __syncthreads();
    // k-step ks: read its row fragments from the ring, park k-step ks+1, request rows ks+3 and weights ks+1, multiply, barrier
    auto step = [&](int ks, uint4 (&w_cur)[NTW], uint4 (&w_nxt)[NTW], uint4 (&b_park)[NBL], uint4 (&b_load)[NBL]) {
        const uint4* st = ring + (size_t)(ks % RG_LDS_STAGES) * RT_ALL * 64;
        load_w(ks + 1, w_nxt);
        park_b(ks + 1, b_park);
        load_b(ks + 3, b_load);
        // row fragments one row tile ahead of the MFMAs that use them (two live instead of RTW)
        uint4 bf_cur = st[(wr * RTW) * 64 + lane];
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt) {
            uint4 bf_nxt = bf_cur;
            if (rt + 1 < RTW) bf_nxt = st[(wr * RTW + rt + 1) * 64 + lane];
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w_cur[j]), __builtin_bit_cast(bf16x8_t, bf_cur),
                                                                     acc[rt][j], 0, 0, 0);
            bf_cur = bf_nxt;
        }
#ifdef GCNPT_STAMPS
        if (p.knob & 4) return;                                           // experiment: no barrier (the values become wrong)
#endif
        __syncthreads();
    };
    // (register sets of two and three, unrolled by six so that every array index is a constant; whole groups of six run without a
    // branch between their loads, the last k-steps follow one by one)
    int ks = 0;
    for (; ks + 6 <= p.ksteps; ks += 6) {
        step(ks, wq[0], wq[1], bq[1], bq[0]);
        step(ks + 1, wq[1], wq[0], bq[2], bq[1]);
        step(ks + 2, wq[0], wq[1], bq[0], bq[2]);
        step(ks + 3, wq[1], wq[0], bq[1], bq[0]);
        step(ks + 4, wq[0], wq[1], bq[2], bq[1]);
        step(ks + 5, wq[1], wq[0], bq[0], bq[2]);
    }
    if (ks < p.ksteps) step(ks, wq[0], wq[1], bq[1], bq[0]);
    if (ks + 1 < p.ksteps) step(ks + 1, wq[1], wq[0], bq[2], bq[1]);
    if (ks + 2 < p.ksteps) step(ks + 2, wq[0], wq[1], bq[0], bq[2]);
    if (ks + 3 < p.ksteps) step(ks + 3, wq[1], wq[0], bq[1], bq[0]);
    if (ks + 4 < p.ksteps) step(ks + 4, wq[0], wq[1], bq[2], bq[1]);
    GCNPT_STAMP(p.stamps, 2);

    // epilogue on the accumulators -> LDS out tile -> whole rows (lane (i = lane & 15, q = lane >> 4) holds row i and the 4 consecutive
    // columns 16 tile + 4q .. + 3).  f32 rows of the whole workgroup do not fit the LDS: one row group per phase then.
    const int c_lo = pass * p.tiles_per_pass * 16, c_hi = min(p.NOUT, tile_hi * 16);
    const int width = c_hi - c_lo;
    const int oes = p.out_f32 ? 4 : 2;
    const int ostride_b = p.ostride_b;                                          // out-tile row stride in bytes
    unsigned char* O = rg_smem;
    const int n_phase = p.n_phase;
    // bias and degrees in one batch of loads (clamped: a load behind a condition would cost a full round trip each)
    float4 bias4[NTW];
    float den_r[RTW];
    const float* bsrc = p.bwd ? reinterpret_cast<const float*>(p.wfrag) : p.bias;    // (bwd: any readable floats, the values are not used)
#pragma unroll
    for (int j = 0; j < NTW; ++j) bias4[j] = *reinterpret_cast<const float4*>(bsrc + min((tile0 + j) * 16 + kgrp * 4, p.NOUT - 4));
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt)
        den_r[rt] = (float)(p.d_ell[(size_t)min(rw0 + (wr * RTW + rt) * 16 + arow, p.N - 1) * 8] + 1);             // gcn.py:261
    for (int ph = 0; ph < n_phase; ++ph) {
        if (n_phase == 1 || wr == ph) {
            const int lrow0 = n_phase == 1 ? wr * RTW * 16 : 0;
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) {
                const int r = rw0 + (wr * RTW + rt) * 16 + arow;
                const float den = den_r[rt];
                const float inv = 1.0f / den;
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const int tl = tile0 + j;
                    const int col0 = tl * 16 + kgrp * 4;
                    if (tl >= tile_hi || col0 >= p.NOUT) continue;
                    float v[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) v[g] = acc[rt][j][g];
                    if (!p.bwd) {
                        const float bq4[4] = {bias4[j].x, bias4[j].y, bias4[j].z, bias4[j].w};
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float x = div_by(v[g] + 2.0f * bq4[g], den, inv);       // gcn.py:270-271 (the bias enters twice), 390
                            v[g] = x > 0.0f ? x : 0.0f;                                   // gcn.py:392
                        }
                        if (p.drop_p > 0.0f) {                                            // gcn.py:393
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) {
                                const unsigned dh = drop_hash(p.seed + seed_off, (unsigned)r, (unsigned)(col0 >> 1) + h2);
                                v[2 * h2] = drop_keep(dh, 0u, p.drop_thresh16) ? v[2 * h2] * p.scale : 0.0f;
                                v[2 * h2 + 1] = drop_keep(dh, 1u, p.drop_thresh16) ? v[2 * h2 + 1] * p.scale : 0.0f;
                            }
                        }
                    }
                    unsigned char* dst = O + (size_t)(lrow0 + rt * 16 + arow) * ostride_b + (size_t)(col0 - c_lo) * oes;
                    if (p.out_f32) {
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        uint2 pk;
                        pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                        pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<uint2*>(dst) = pk;
                    }
                }
            }
        }
        __syncthreads();
        // whole rows leave in 16-byte pieces (8-byte ones when the row width only allows those), 16 threads per row; the hand-over to the
        // layer below (its dZ instead of dh, gcn.py:390-393 differentiated) masks and scales them while they leave the LDS
        const int rows_ph = n_phase == 1 ? RT_ALL * 16 : RTW * 16;
        const int row_base = rw0 + (n_phase == 1 ? 0 : ph * RTW * 16);
        auto store_rows = [&](auto vtag) {
            using V = decltype(vtag);
            constexpr int NW = (int)sizeof(V) / 4;
            constexpr int RP = 5;                                                // pieces of a row per thread and batch (320 columns: all)
            const int per = (int)sizeof(V) / oes;
            const int pieces = width / per;
            const bool mask = p.bwd && p.relu_src;
            for (int lrow = tid >> 4; lrow < rows_ph; lrow += RG_THREADS / 16) {
                const int r = row_base + lrow;
                if (r >= p.N) break;
                const size_t go = ((size_t)r * p.NOUT + c_lo) * oes;
                const unsigned char* orow = O + (size_t)lrow * ostride_b;
                float f = 1.0f;
                if (mask) f = p.next_scale / (float)(p.d_ell[(size_t)r * 8] + 1);
                for (int pc0 = tid & 15; pc0 < pieces; pc0 += 16 * RP) {
                    V h[RP], o[RP];
                    if (mask) {
#pragma unroll
                        for (int u = 0; u < RP; ++u)
                            h[u] = *reinterpret_cast<const V*>(static_cast<const unsigned char*>(p.relu_src) + go +
                                                               (size_t)min(pc0 + 16 * u, pieces - 1) * sizeof(V));
                    }
#pragma unroll
                    for (int u = 0; u < RP; ++u) o[u] = *reinterpret_cast<const V*>(orow + (size_t)min(pc0 + 16 * u, pieces - 1) * sizeof(V));
#pragma unroll
                    for (int u = 0; u < RP; ++u) {
                        const int pc = pc0 + 16 * u;
                        if (pc >= pieces) continue;
                        V ov = o[u];
                        if (mask) {
                            unsigned ow[NW], hw[NW];
                            __builtin_memcpy(ow, &ov, sizeof(V));
                            __builtin_memcpy(hw, &h[u], sizeof(V));
                            if (p.out_f32) {
#pragma unroll
                                for (int q = 0; q < NW; ++q) ow[q] = __uint_as_float(hw[q]) > 0.0f ? __float_as_uint(__uint_as_float(ow[q]) * f) : 0u;
                            } else {
#pragma unroll
                                for (int q = 0; q < NW; ++q) {
                                    const float lo = bf16_to_f32((bf16_t)(hw[q] & 0xffffu)) > 0.0f ? bf16_to_f32((bf16_t)(ow[q] & 0xffffu)) * f : 0.0f;
                                    const float hi = bf16_to_f32((bf16_t)(hw[q] >> 16)) > 0.0f ? bf16_to_f32((bf16_t)(ow[q] >> 16)) * f : 0.0f;
                                    ow[q] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                                }
                            }
                            __builtin_memcpy(&ov, ow, sizeof(V));
                        }
                        *reinterpret_cast<V*>(static_cast<unsigned char*>(p.out) + go + (size_t)pc * sizeof(V)) = ov;
                    }
                }
            }
        };
        if (p.store16) store_rows(uint4{});
        else store_rows(uint2{});
        if (ph + 1 < n_phase) __syncthreads();
    }
    GCNPT_STAMP(p.stamps, 3);
}

}  // namespace gcnpt

using namespace gcnpt;

// ---- host side: called from gcnpt_layer_fwd / gcnpt_layer_bwd_data when the caller passed a workspace (rowtile_kernels.hip) ----
namespace {

template <typename IT>
int launch_prep_it(hipStream_t s, const RowPrepParams& p, int vec) {
    const int stride = lds_stride_dw(p.Kpad / 2) * 2;
    const size_t lds = (size_t)RP_ROWS * stride * sizeof(bf16_t) + (size_t)(10 * RP_ROWS + 4) * sizeof(int);
    if (lds > 64 * 1024) return 0;
    const dim3 grid(ceil_div(p.N, RP_ROWS));
    if (vec == 8) hipLaunchKernelGGL((rowprep_kernel<IT, 8>), grid, dim3(RP_THREADS), lds, s, p);
    else hipLaunchKernelGGL((rowprep_kernel<IT, 4>), grid, dim3(RP_THREADS), lds, s, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    return 1;
}

template <int VEC, int RTW, int NTW>
int launch_gemm_cfg(hipStream_t s, RowGemmParams& p, dim3 grid) {
    // LDS: the ring of row fragments during the k loop, then the out tile of the epilogue (all rows of the workgroup, or one row
    // group per phase when those do not fit)
    const int oes = p.out_f32 ? 4 : 2;
    const int wcols = std::min(p.tiles_per_pass * 16, round_up(p.NOUT, 16));
    p.ostride_b = out_stride_dw(wcols * oes / 4) * 4;
    const size_t ring = (size_t)RG_LDS_STAGES * RG_WR * RTW * 1024;
    size_t otile = (size_t)RG_WR * RTW * 16 * p.ostride_b;
    p.n_phase = 1;
    if (otile > 150 * 1024) { p.n_phase = RG_WR; otile /= RG_WR; }
    if (otile > 150 * 1024) return 0;
    const size_t lds = std::max(ring, otile);
    auto kern = rowgemm_kernel<VEC, RTW, NTW>;
    GCNPT_LDS_ATTR_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(RG_THREADS), lds, s, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    return 1;
}

template <int VEC, int RTW>
int launch_gemm_ntw(hipStream_t s, RowGemmParams& p, int ntw, dim3 grid) {
    if (ntw <= 3) return launch_gemm_cfg<VEC, RTW, 3>(s, p, grid);
    if (ntw == 4) return launch_gemm_cfg<VEC, RTW, 4>(s, p, grid);
    return launch_gemm_cfg<VEC, RTW, 5>(s, p, grid);
}

}  // namespace

size_t rowsplit_agg_bytes(long long rows, int width) { return (size_t)((rows * width * 2 + 255) / 256 * 256); }

bool rowsplit_wanted(long long rows) {
    // opt-in (GCNPT_ROWSPLIT=1): measured no faster than the row-tile kernels at the shapes tried
    if (rows < 16384) return false;
    const char* e = getenv("GCNPT_ROWSPLIT");
    return e && e[0] == '1';
}

// 1 = taken, 0 = does not apply (the caller falls back to the row-tile kernel), < 0 = error
int rowsplit_layer(hipStream_t s, bool bwd, const void* src, int src_dtype, const void* wfrag, const float* bias, const int32_t* g_row_ptr,
                   const int32_t* g_col_idx, const int32_t* g_ell, const int32_t* d_ell, int N, int T, int K, int NOUT, void* out, int out_dtype,
                   float drop_p, uint64_t seed, const uint64_t* seed_dev, void* frag_out, float* zero_a, int zero_a_n, float* zero_b, int zero_b_n,
                   const void* relu_src, float next_scale, void* ws, size_t ws_bytes) {
    if (!ws || !out || !rowsplit_wanted(N) || ws_bytes < rowsplit_agg_bytes(N, K) || !aligned16(ws)) return 0;
    const size_t es = esize(src_dtype);
    const bool al16 = aligned16(src), al_half = (reinterpret_cast<uintptr_t>(src) % (4 * es)) == 0;
    const int vec = (K % 8 == 0 && al16) ? 8 : ((K % 4 == 0 && K >= 8 && al_half) ? 4 : 0);
    if (vec == 0) return 0;
    // the matrix kernel stores 4 columns per lane: bf16 8 bytes, f32 16 bytes
    const size_t oes = esize(out_dtype);
    if (NOUT % 4 != 0 || (reinterpret_cast<uintptr_t>(out) % (4 * oes)) != 0 || (relu_src && (reinterpret_cast<uintptr_t>(relu_src) % (4 * oes)) != 0)) return 0;
    if (!bwd && (reinterpret_cast<uintptr_t>(bias) % 16) != 0) return 0;

    RowPrepParams q{};
    q.src = src; q.g_row_ptr = g_row_ptr; q.g_col_idx = g_col_idx; q.g_ell = g_ell; q.agg = static_cast<bf16_t*>(ws);
    q.frag_out = frag_out; q.frag_of_agg = bwd ? 0 : 1; q.all_rows = src_dtype != GCNPT_BF16;
    // (diagnostic builds: knob bit 3 stamps the gather launch, otherwise the matrix launch -- their workgroup indices overlap)
    q.stamps = (g_debug_knob & 8) ? static_cast<unsigned long long*>(g_debug_stamps) : nullptr;
    q.N = N; q.T = T; q.K = K; q.Kpad = round_up(K, 32); q.chunk_magic = 0xffffffffu / (unsigned)(q.Kpad / 8) + 1u;
    const int rc = src_dtype == GCNPT_BF16 ? launch_prep_it<bf16_t>(s, q, vec) : launch_prep_it<float>(s, q, vec);
    if (rc <= 0) return rc;

    RowGemmParams p{};
    p.stamps = (g_debug_knob & 8) ? nullptr : static_cast<unsigned long long*>(g_debug_stamps); p.knob = g_debug_knob;
    p.rows_plain = q.all_rows ? q.agg : static_cast<const bf16_t*>(src); p.rows_agg = q.agg;
    p.sel_ell = g_ell; p.d_ell = d_ell; p.wfrag = static_cast<const uint4*>(wfrag); p.bias = bias; p.out = out;
    p.relu_src = relu_src; p.next_scale = next_scale;
    p.zero_a = zero_a; p.zero_a_n = zero_a_n; p.zero_b = zero_b; p.zero_b_n = zero_b_n;
    p.N = N; p.K = K; p.NOUT = NOUT; p.ksteps = q.Kpad / 32; p.n_tiles = ceil_div(NOUT, 16);
    p.passes = ceil_div(p.n_tiles, RG_WC * 5);
    p.tiles_per_pass = ceil_div(p.n_tiles, p.passes);
    const int ntw = ceil_div(p.tiles_per_pass, RG_WC);
    p.tiles_per_pass = std::min(p.tiles_per_pass, ntw * RG_WC);
    p.bwd = bwd ? 1 : 0; p.out_f32 = out_dtype == GCNPT_F32 ? 1 : 0;
    // 16-byte row pieces when every pass starts and ends on one (f32: always; bf16: widths and pass offsets that are multiples of 8)
    p.store16 = (p.out_f32 || (NOUT % 8 == 0 && (p.tiles_per_pass * 16) % 8 == 0 && aligned16(out) && (!relu_src || aligned16(relu_src)))) ? 1 : 0;
    p.drop_p = drop_p; p.scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    p.drop_thresh16 = (unsigned)((double)drop_p * 65536.0);
    p.seed = seed; p.seed_dev = seed_dev;
    // rows per workgroup: 160 or 128, whichever needs fewer row-rounds over the 256 CUs
    const int gvec = q.all_rows ? (K % 8 == 0 ? 8 : 4) : vec;
    auto cost = [&](int rtw) { const long long blocks = (long long)ceil_div(N, RG_WR * rtw * 16) * p.passes; return (blocks + 255) / 256 * rtw; };
    const int rtw = cost(5) <= cost(4) ? 5 : 4;
    p.row_blocks = ceil_div(N, RG_WR * rtw * 16);
    const dim3 grid(8 * ceil_div(p.row_blocks, 8) * p.passes);
    if (gvec == 8) return rtw == 5 ? launch_gemm_ntw<8, 5>(s, p, ntw, grid) : launch_gemm_ntw<8, 4>(s, p, ntw, grid);
    return rtw == 5 ? launch_gemm_ntw<4, 5>(s, p, ntw, grid) : launch_gemm_ntw<4, 4>(s, p, ntw, grid);
}
