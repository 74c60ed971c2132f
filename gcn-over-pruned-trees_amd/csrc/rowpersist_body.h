// The row-tile layer kernel for BIG batches (more row tiles than CUs): one persistent 8-wave workgroup per CU that keeps its weight
// fragments in registers for the whole launch and walks over its share of the 32-row tiles, with the next tile's rows on their way
// while the current tile is on the matrix cores.
//
// Why ("big batches"): the one-tile-per-workgroup kernel of rowtile_body.h pulls ALL weight fragments through its
// CU's L1 path for every 32 rows -- 156 KB of weights against 30 KB of rows at the C2 widths, 360 KB against 50 KB at the C5 widths --
// and a launch of 1 200 ... 3 200 tiles runs at the rate the L2s can re-serve those weights (~9 TB/s), not at anything the rows need.
// Here a workgroup loads its fragments ONCE.  What does not fit 8 waves x ~30 fragments (the register budget next to the gather's
// prefetch) is split by output COLUMNS over `col_split` workgroups that gather the same rows (from the same XCD's L2) and each produce
// `tiles_pp` column tiles of them; the split costs the gather col_split times, which is still less than the weights cost before.
//
// One iteration of a workgroup (tile k on the matrix cores; S, and the backward's Z, are double-buffered by the tile's parity, the ELL heads
// and the tables derived from them triple-buffered):
//   matrix(k)    S[k] x resident weights, epilogue -> O                                                        barrier
//   finish(k+1)  the rows requested one iteration ago have arrived: sums in fp32, S[k+1] <- (A+I) h in the MFMA operand type
//   leave(k)     fragment image of S[k] (or dZ) for the weight gradient; whole rows leave from O                 [stores]
//   request(k+2) the heads requested one iteration ago -> meta[k+2]; request the heads of tile k+3 and the rows of tile k+2   [loads]
//                                                                                                              barrier
// The order inside an iteration is what keeps the prefetch alive.  The vector-memory counter retires loads AND stores in issue order and
// hipcc counts only the loads it is sure of, so a wait for prefetched rows also waits for every store issued after them: the requests
// therefore go out LAST in an iteration (behind its stores) and are consumed in the next one BEFORE that one's stores -- the only
// stores a wait can then fall behind are a whole iteration old.  (First version, requests before the matrix phase and stores after it:
// 3.3 k cycles per tile in `finish`, waiting for the previous tile's stores to be acknowledged.)
// Every barrier is `s_waitcnt lgkmcnt(0)` + `s_barrier`: __syncthreads() would drain the vector-memory queue (it is a fence over global
// memory too) and with it the prefetch that the whole arrangement exists for.  No load sits behind a condition (a tile past the end is
// requested at clamped addresses and dropped).
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

// PI: 8-column chunks of the tile's own rows a thread keeps in flight across the matrix phase (3 cover K <= 384, 5 cover K <= 640)
template <typename CT, typename IT, typename OT, bool BWD, int VEC, int NTW, int KS, bool DZIN, int PI>
__global__ __launch_bounds__(RT_THREADS, 2) void rowpersist_kernel(const RowTileParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    static_assert(sizeof(CT) == 2, "the persistent form exists for bf16 MFMA operands");
    static_assert(BWD || !DZIN, "DZIN is a backward mode");
    constexpr int RTT = RT_THREADS, RTW = RT_WAVES;
    constexpr bool MASKED = BWD && !DZIN;
    constexpr int KSTEP = 32;
    constexpr bool WIDE = MASKED && sizeof(IT) == 4;
    constexpr int NBU = WIDE ? 2 : 4;
    constexpr int META_INTS = 13 * ROWS;

    const int stride = lds_stride_dw(p.Kpad * (int)sizeof(CT) / 4) * 4 / (int)sizeof(CT);
    const int ncols_pass = p.tiles_pp * 16;
    const int ostride = out_stride_dw(min(round_up(p.NOUT, 16), ncols_pass) * (int)sizeof(OT) / 4) * 4 / (int)sizeof(OT);
    const size_t s_bytes = (size_t)ROWS * stride * sizeof(CT);
    const size_t o_bytes = (size_t)ROWS * ostride * sizeof(OT);
    // LDS: S[2] | Z[2] (backward: the tile's own dZ rows before aggregation, read by the image emission) | O | meta[3] | bias
    auto S_of = [&](int b) { return reinterpret_cast<CT*>(smem_raw + (size_t)b * s_bytes); };
    auto Z_of = [&](int b) { return reinterpret_cast<CT*>(smem_raw + (size_t)(2 + b) * s_bytes); };
    OT* O = reinterpret_cast<OT*>(smem_raw + (BWD ? 4 : 2) * s_bytes);
    int* meta0 = reinterpret_cast<int*>(smem_raw + (BWD ? 4 : 2) * s_bytes + o_bytes);
    float* sbias = reinterpret_cast<float*>(meta0 + 3 * META_INTS);        // [tiles_pp * 16] fwd: the bias of this workgroup's columns
    CT* Sw = S_of(0);                                                       // where finish_rows puts a tile (set per call)
    CT* Zw = Z_of(0);
    struct Meta { int* rell; float* rinv; float* rden; int* glist; int* rsb; int* gcount; };
    auto meta_of = [&](int b) {
        int* m = meta0 + b * META_INTS;
        return Meta{m, reinterpret_cast<float*>(m + 8 * ROWS), reinterpret_cast<float*>(m + 9 * ROWS), m + 10 * ROWS, m + 11 * ROWS, m + 12 * ROWS};
    };

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = (int)blockIdx.x, nwg = (int)gridDim.x;              // nwg % 8 == 0

    // accumulators of the weight gradients that follow: cleared by everyone, before anyone may leave
#pragma unroll
    for (int z = 0; z < 4; ++z)
        if (p.zero_p[z])
            for (int i = id * RTT + tid; i < p.zero_n[z]; i += nwg * RTT) p.zero_p[z][i] = 0.0f;

    // Which tiles: XCD x (workgroups id % 8 == x) takes the x-th contiguous eighth of the row tiles, like the one-shot kernel, so a tile's
    // neighbour rows are rows the same L2 serves anyway.  Inside an XCD the `col_split` workgroups of one row group sit next to each other and
    // walk the same tiles at the same time.
    const int C = p.col_split;
    const int xg = id & 7, jx = id >> 3, per_x = nwg >> 3;
    const int G = per_x / C;                                          // row groups per XCD
    const int cpass = jx % C, gl = jx / C;
    const int n_rt = ceil_div(p.N, ROWS);
    const int xq = n_rt >> 3, xr = n_rt & 7;
    const int t_lo = xg * xq + min(xg, xr), t_cnt = xq + (xg < xr ? 1 : 0);
    if (gl >= G || gl >= t_cnt) return;
    const int my_n = (t_cnt - gl + G - 1) / G;                         // tiles t_lo + gl + k G, k < my_n
    auto tile_of = [&](int k) { return t_lo + gl + min(k, my_n - 1) * G; };   // (past the end: the last one again, requested and dropped)

    const IT* src = static_cast<const IT*>(p.src);
    const IT* yref = static_cast<const IT*>(p.yref);
    const uint4* wfrag = static_cast<const uint4*>(p.wfrag);
    const int n_ctiles = ceil_div(p.NOUT, 16);
    const int ksteps = p.Kpad / KSTEP;
    uint64_t seed_off = 0;
    if (!BWD && p.seed_dev) seed_off = *p.seed_dev;

    const int erow = lane >> 1, ehalf = lane & 1;
    struct Heads { int4 ell; int deg; };
    auto load_heads = [&](int tile) {
        const size_t er = (size_t)min(tile * ROWS + erow, p.N - 1);
        Heads h;
        h.ell = reinterpret_cast<const int4*>(p.g_ell)[er * 2 + ehalf];
        h.deg = p.d_ell[er * 8];                                                            // gcn.py:261
        return h;
    };
    // every wave parks the same values in the same places and reads back only its own writes (wave_lds_fence)
    auto park_heads = [&](const Heads& h, int tile, const Meta& m) {
        const int r0 = tile * ROWS;
        const int er = min(r0 + erow, p.N - 1);
        const bool first = ehalf == 0;
        const int e0 = (first && r0 + erow >= p.N) ? 0 : h.ell.x;
        reinterpret_cast<int4*>(m.rell)[erow * 2 + ehalf] = make_int4(e0, h.ell.y, h.ell.z, h.ell.w);
        const float dn = (float)(h.deg + 1);
        m.rsb[erow] = p.T ? er / p.T * p.T : 0;
        m.rinv[erow] = (BWD ? p.scale : 1.0f) / dn;
        m.rden[erow] = dn;
        const bool agg = first && e0 > 0;
        const unsigned long long mk = __ballot(agg);
        if (agg) m.glist[__popcll(mk & ((1ull << lane) - 1ull))] = erow;
        if (lane == 0) *m.gcount = __popcll(mk);
    };

    const int nchunk = p.Kpad / 8;
    auto div_chunk = [&](int x) { return (int)__umulhi((unsigned)x, p.chunk_magic); };
    const int n_items = ROWS * nchunk;
    const int kmax8 = VEC == 8 ? p.K - 8 : p.K - 4;
    auto ld8 = [&](const IT* base, size_t row, int k0c, raw8<IT>& dst) {
        if constexpr (VEC == 8) issue8<IT, true>(base, row, p.K, k0c, dst);
        else issue8_half<IT>(base, row, p.K, k0c, dst);
    };
    static_assert(VEC == 8 || VEC == 4, "rows are read in 16- or 8-byte pieces");

    struct GItem { raw8<IT> s, sy, nb[NBU], nby[NBU]; int dcnt[NBU]; };
    struct Rows { raw8<IT> self[PI], selfy[PI]; GItem g; };

    auto issue_self = [&](int r0, int first_item, raw8<IT>& s, raw8<IT>& sy) {
        const int it = first_item + tid;
        const int row = div_chunk(it), k0 = (it - row * nchunk) * 8;
        const size_t r = (size_t)min(r0 + row, p.N - 1);
        ld8(src, r, min(k0, kmax8), s);
        if (MASKED) ld8(yref, r, min(k0, kmax8), sy);
    };
    auto g_decode = [&](const Meta& m, int n_g, int gi, int& row, int& k0, int& n) {
        const bool has = gi < n_g;
        const int li = has ? div_chunk(gi) : 0;
        row = has ? m.glist[li] : 0;
        k0 = has ? (gi - li * nchunk) * 8 : 0;
        n = (has && k0 < p.K) ? m.rell[row * 8] : 0;
        return has;
    };
    auto g_issue = [&](const Meta& m, int r0, int n_g, int gi, GItem& g) {
        int row, k0, n;
        g_decode(m, n_g, gi, row, k0, n);
        const size_t r = (size_t)min(r0 + row, p.N - 1);
        const int sbase = m.rsb[row];
        const int k0c = min(k0, kmax8);
        ld8(src, r, k0c, g.s);
        if (MASKED) ld8(yref, r, k0c, g.sy);
#pragma unroll
        for (int e = 0; e < NBU; ++e) {
            const bool on = e < min(n, NB_INLINE);
            const size_t c = on ? (size_t)(sbase + m.rell[row * 8 + 1 + e]) : (size_t)min(r0, p.N - 1);
            const int kc = on ? k0c : 0;
            ld8(src, c, kc, g.nb[e]);
            if (MASKED) {
                ld8(yref, c, kc, g.nby[e]);
                g.dcnt[e] = p.d_ell[c * 8];
            }
        }
    };
    auto g_finish = [&](const Meta& m, int r0, int n_g, int gi, const GItem& g) {
        int row, k0, n;
        const bool has = g_decode(m, n_g, gi, row, k0, n);
        const bool live = has && k0 < p.K;
        const size_t rc = (size_t)min(r0 + row, p.N - 1);
        const int sbase = m.rsb[row];
        const int k0c = min(k0, kmax8);
        float acc[8];
        unpack8<IT>(g.s, live, acc);                                    // the explicit W(h) term, gcn.py:271
        if (MASKED) {
            float y[8];
            unpack8<IT>(g.sy, live, y);
            const float inv = m.rinv[row];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = (y[j] > 0.0f) ? acc[j] * inv : 0.0f;
        }
        const int n_ell = min(n, NB_INLINE);
#pragma unroll
        for (int e = 0; e < NBU; ++e) {
            const bool on = e < n_ell;
            float v[8];
            unpack8<IT>(g.nb[e], on, v);
            if (MASKED) {
                float y[8];
                unpack8<IT>(g.nby[e], on, y);
                const float ninv = p.scale / (float)(g.dcnt[e] + 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (y[j] > 0.0f) ? v[j] * ninv : 0.0f;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
        }
        // rows with more than NBU entries: further round trips (not prefetched)
        auto round = [&](int e0, int lim, auto from_lds) {
            raw8<IT> nb[NBU], nby[NBU];
            float ninv[NBU];
#pragma unroll
            for (int e = 0; e < NBU; ++e) {
                const bool on = e0 + e < lim;
                size_t c;
                if constexpr (decltype(from_lds)::value) {
                    c = (size_t)(sbase + m.rell[row * 8 + 1 + min(e0 + e, NB_INLINE - 1)]);
                } else {
                    const int beg = p.T ? p.g_row_ptr[(size_t)(sbase / p.T) * (p.T + 1) + (rc - sbase)] : p.g_row_ptr[rc];
                    c = (size_t)(sbase + p.g_col_idx[on ? beg + e0 + e : beg]);
                }
                c = on ? c : rc;
                ld8(src, c, k0c, nb[e]);
                if (MASKED) {
                    ld8(yref, c, k0c, nby[e]);
                    ninv[e] = p.scale / (float)(p.d_ell[c * 8] + 1);
                }
            }
#pragma unroll
            for (int e = 0; e < NBU; ++e) {
                const bool on = e0 + e < lim;
                float v[8];
                unpack8<IT>(nb[e], on, v);
                if (MASKED) {
                    float y[8];
                    unpack8<IT>(nby[e], on, y);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += (y[j] > 0.0f) ? v[j] * ninv[e] : 0.0f;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += v[j];
                }
            }
        };
        for (int e0 = NBU; e0 < n_ell; e0 += NBU) round(e0, n_ell, std::true_type{});
        for (int e0 = NB_INLINE; e0 < n; e0 += NBU) round(e0, n, std::false_type{});
        if (has) tile<CT>::put8(Sw + (size_t)row * stride + k0, acc);
    };
    // a chunk of the tile's own rows: into S unless the row aggregates (g_finish writes those), into Z for the backward's image
    auto copy_item = [&](const Meta& m, int r0, int it, const raw8<IT>& s, const raw8<IT>& sy) {
        if (it >= n_items) return;
        const int row = div_chunk(it), k0 = (it - row * nchunk) * 8;
        const bool live = r0 + row < p.N && k0 < p.K;
        if constexpr (!BWD && sizeof(IT) == 2) {
            if (!(m.rell[row * 8] > 0)) *reinterpret_cast<uint4*>(Sw + (size_t)row * stride + k0) = live ? s.a : make_uint4(0, 0, 0, 0);
            return;
        }
        float acc[8];
        unpack8<IT>(s, live, acc);
        if (MASKED) {
            float y[8];
            unpack8<IT>(sy, live, y);
            const float inv = m.rinv[row];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = (y[j] > 0.0f) ? acc[j] * inv : 0.0f;
        }
        if (BWD) {
            if (p.frag_out) tile<CT>::put8(Zw + (size_t)row * stride + k0, acc);
        }
        if (!(m.rell[row * 8] > 0)) tile<CT>::put8(Sw + (size_t)row * stride + k0, acc);
    };

    auto issue_rows = [&](const Meta& m, int tile, Rows& R) {
        const int r0 = tile * ROWS;
#pragma unroll
        for (int u = 0; u < PI; ++u) issue_self(r0, u * RTT, R.self[u], R.selfy[u]);
        g_issue(m, r0, *m.gcount * nchunk, tid, R.g);
    };
    auto finish_rows = [&](const Meta& m, int tile, const Rows& R) {
        const int r0 = tile * ROWS;
        const int n_g = *m.gcount * nchunk;
#pragma unroll
        for (int u = 0; u < PI; ++u) copy_item(m, r0, u * RTT + tid, R.self[u], R.selfy[u]);
        if (wave * WAVE < n_g) g_finish(m, r0, n_g, tid, R.g);
        for (int base = RTT; base < n_g; base += RTT) {              // more than 512 (aggregating row, chunk) items: not prefetched
            GItem g;
            g_issue(m, r0, n_g, base + tid, g);
            g_finish(m, r0, n_g, base + tid, g);
        }
        for (int first = PI * RTT; first < n_items; first += RTT) {   // K wider than PI covers: not prefetched
            raw8<IT> s, sy;
            issue_self(r0, first, s, sy);
            copy_item(m, r0, first + tid, s, sy);
        }
    };

    // ---- prologue: heads of the first three tiles, rows of the first two, the resident weights, S[0] -------------------------------------
    Heads h_nxt = load_heads(tile_of(0));
    float bias_v = 0.0f;
    if constexpr (!BWD) bias_v = p.bias[min(cpass * ncols_pass + min(tid, ncols_pass - 1), p.NOUT - 1)];
    park_heads(h_nxt, tile_of(0), meta_of(0));
    if constexpr (!BWD) { if (tid < ncols_pass) sbias[tid] = bias_v; }
    wave_lds_fence();
    h_nxt = load_heads(tile_of(1));
    Rows R;
    issue_rows(meta_of(0), tile_of(0), R);

    // wave w owns column tiles cpass * tiles_pp + w, + 8, ... (tiles past this workgroup's share: duplicates of the last, never stored)
    uint4 wreg[KS][NTW];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int tl = min(cpass * p.tiles_pp + min(j * RTW + wave, p.tiles_pp - 1), n_ctiles - 1);
            wreg[ks][j] = wfrag[((size_t)tl * ksteps + min(ks, ksteps - 1)) * 64 + lane];
        }

    OT* out = static_cast<OT*>(p.out);
    const int arow = lane & 15, kgrp = lane >> 4;
    const int c_lo = cpass * ncols_pass, c_hi = min(p.NOUT, c_lo + ncols_pass);
    const int width = c_hi - c_lo;
    const OT* relu = BWD ? static_cast<const OT*>(p.relu_src) : nullptr;

    Sw = S_of(0); Zw = Z_of(0);
    finish_rows(meta_of(0), tile_of(0), R);
    park_heads(h_nxt, tile_of(1), meta_of(1));
    wave_lds_fence();
    h_nxt = load_heads(tile_of(2));
    issue_rows(meta_of(1), tile_of(1), R);
    lds_barrier();

    int mb = 0;                                                         // k % 3
    for (int k = 0; k < my_n; ++k) {
        const int b = k & 1;
        const int mb1 = mb == 2 ? 0 : mb + 1, mb2 = mb1 == 2 ? 0 : mb1 + 1;
        const Meta m = meta_of(mb);
        const CT* Sr = S_of(b);
        const int tile_id = tile_of(k), r0 = tile_id * ROWS;
#ifdef GCNPT_STAMPS
        unsigned long long* const stamps = k == (p.knob >> 8) ? p.stamps : nullptr;     // one iteration's phases (diagnostic builds)
#endif
        GCNPT_STAMP(stamps, 0);

        // matrix(k): the tile meets the resident weights
        f32x4_t acc[2][NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) { acc[0][j] = (f32x4_t){0, 0, 0, 0}; acc[1][j] = (f32x4_t){0, 0, 0, 0}; }
        {
            uint4 a_cur[2], a_nxt[2];
            auto read_a = [&](int kk, uint4 (&dst)[2]) {
                dst[0] = *reinterpret_cast<const uint4*>(Sr + (size_t)arow * stride + kk * KSTEP + kgrp * 8);
                dst[1] = *reinterpret_cast<const uint4*>(Sr + (size_t)(arow + 16) * stride + kk * KSTEP + kgrp * 8);
            };
            read_a(0, a_cur);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks < ksteps) {                                       // wave-uniform, no global load inside
                    read_a(min(ks + 1, ksteps - 1), a_nxt);
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        const bf16x8_t bq = __builtin_bit_cast(bf16x8_t, wreg[ks][j]);
                        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq, __builtin_bit_cast(bf16x8_t, a_cur[0]), acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq, __builtin_bit_cast(bf16x8_t, a_cur[1]), acc[1][j], 0, 0, 0);
                    }
                    a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1];
                }
            }
        }
        GCNPT_STAMP(stamps, 1);

        // epilogue on the accumulators -> O.  Lane (i = lane & 15, q = lane >> 4) holds row i and columns 4q..4q+3 of a 16x16 tile.
        {
            float den[2], inv[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) { den[mt] = m.rden[mt * 16 + (lane & 15)]; inv[mt] = m.rinv[mt * 16 + (lane & 15)]; }
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int tloc = j * RTW + wave;
                const int tl = cpass * p.tiles_pp + tloc;
                if (tloc >= p.tiles_pp || tl >= n_ctiles) continue;
                const int col0 = tl * 16 + (lane >> 4) * 4;
                const int lcol0 = col0 - c_lo;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int row = mt * 16 + (lane & 15);
                    float v[4];
                    float bq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if constexpr (!BWD) {
                        const float4 bv = *reinterpret_cast<const float4*>(sbias + lcol0);
                        bq[0] = bv.x; bq[1] = bv.y; bq[2] = bv.z; bq[3] = bv.w;
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float x = acc[mt][j][g];
                        if (!BWD) {
                            x = div_by(x + 2.0f * bq[g], den[mt], inv[mt]);   // gcn.py:270-271 (the bias enters twice), 390
                            x = x > 0.0f ? x : 0.0f;                         // gcn.py:392
                        }
                        v[g] = x;
                    }
                    if (!BWD && p.drop_p > 0.0f) {                            // gcn.py:393
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const unsigned dh = drop_hash(p.seed + seed_off, (unsigned)(r0 + row), (unsigned)(col0 >> 1) + h2);
                            v[2 * h2] = drop_keep(dh, 0u, p.drop_thresh16) ? v[2 * h2] * p.scale : 0.0f;
                            v[2 * h2 + 1] = drop_keep(dh, 1u, p.drop_thresh16) ? v[2 * h2 + 1] * p.scale : 0.0f;
                        }
                    }
                    OT* dst = O + (size_t)row * ostride + lcol0;
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
        GCNPT_STAMP(stamps, 2);
        lds_barrier();                                                  // O complete; everyone has read S[b] on the matrix cores
        GCNPT_STAMP(stamps, 3);

        // finish(k+1): its rows were requested one iteration ago, behind that iteration's stores
        Sw = S_of(b ^ 1); Zw = Z_of(b ^ 1);
        finish_rows(meta_of(mb1), tile_of(k + 1), R);
        GCNPT_STAMP(stamps, 4);

        // leave(k): whole rows leave in 16-byte pieces (8-byte ones when the width only allows those).  First, because the hand-over's
        // loads of the layer input should not queue behind the image's stores
        auto store_rows = [&](auto vtag) {
            using V = decltype(vtag);
            constexpr int PER = (int)sizeof(V) / (int)sizeof(OT);
            constexpr int NW = (int)sizeof(V) / 4;
            const int pieces = width / PER;
            const int row = tid >> 4, r = r0 + row;
            if (BWD && relu) {
                // hand-over to the layer below: its dZ instead of dh (gcn.py:390-393 differentiated where the rows are at hand)
                constexpr int RP = 4;
                const float f = p.next_scale / m.rden[row];
                const size_t rr = (size_t)min(r, p.N - 1) * p.NOUT + c_lo;
                for (int pc0 = tid & 15; pc0 < pieces; pc0 += 16 * RP) {
                    V hin[RP];
#pragma unroll
                    for (int u = 0; u < RP; ++u) hin[u] = *reinterpret_cast<const V*>(relu + rr + min(pc0 + 16 * u, pieces - 1) * PER);
#pragma unroll
                    for (int u = 0; u < RP; ++u) {
                        const int pc = pc0 + 16 * u;
                        if (pc >= pieces || r >= p.N) continue;
                        V o = *reinterpret_cast<const V*>(O + (size_t)row * ostride + pc * PER);
                        if constexpr (sizeof(OT) == 2) {
                            unsigned* ow = reinterpret_cast<unsigned*>(&o);
                            const unsigned* hw = reinterpret_cast<const unsigned*>(&hin[u]);
#pragma unroll
                            for (int q = 0; q < NW; ++q) {
                                const float lo = bf16_to_f32((bf16_t)(hw[q] & 0xffffu)) > 0.0f ? bf16_to_f32((bf16_t)(ow[q] & 0xffffu)) * f : 0.0f;
                                const float hi = bf16_to_f32((bf16_t)(hw[q] >> 16)) > 0.0f ? bf16_to_f32((bf16_t)(ow[q] >> 16)) * f : 0.0f;
                                ow[q] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                            }
                        } else {
                            float* ow = reinterpret_cast<float*>(&o);
                            const float* hw = reinterpret_cast<const float*>(&hin[u]);
#pragma unroll
                            for (int q = 0; q < NW; ++q) ow[q] = hw[q] > 0.0f ? ow[q] * f : 0.0f;
                        }
                        *reinterpret_cast<V*>(out + (size_t)r * p.NOUT + c_lo + pc * PER) = o;
                    }
                }
            } else if (r < p.N) {
                for (int pc = tid & 15; pc < pieces; pc += 16)
                    *reinterpret_cast<V*>(out + (size_t)r * p.NOUT + c_lo + pc * PER) = *reinterpret_cast<const V*>(O + (size_t)row * ostride + pc * PER);
            }
        };
        constexpr int PER16 = 16 / (int)sizeof(OT), PER8 = 8 / (int)sizeof(OT);
        if (p.vec_out == 16 && (width % PER16) == 0 && (c_lo % PER16) == 0) {
            store_rows(uint4{});
        } else if (p.vec_out >= 8 && (width % PER8) == 0 && (c_lo % PER8) == 0) {
            store_rows(uint2{});
        } else {
            for (int it = tid; it < ROWS * width; it += RTT) {
                const int row = it / width, c = it - row * width;
                const int r = r0 + row;
                if (r >= p.N) continue;
                OT v = O[(size_t)row * ostride + c];
                if (BWD && relu) {
                    const OT hv = relu[(size_t)r * p.NOUT + c_lo + c];
                    float x, hx;
                    if constexpr (sizeof(OT) == 2) { x = bf16_to_f32(v); hx = bf16_to_f32(hv); } else { x = v; hx = hv; }
                    x = hx > 0.0f ? x * (p.next_scale / m.rden[row]) : 0.0f;
                    if constexpr (sizeof(OT) == 2) v = f32_to_bf16(x); else v = x;
                }
                out[(size_t)r * p.NOUT + c_lo + c] = v;
            }
        }
        GCNPT_STAMP(stamps, 5);

        // ... and the tile in MFMA fragment order for the weight gradient (rows are its contraction index); the column tiles of the
        // image are dealt over the workgroups that share this row tile
        if (p.frag_out) {
            uint4* F = static_cast<uint4*>(p.frag_out);
            const CT* X = BWD ? Z_of(b) : Sr;
            const int i = lane & 15, g = lane >> 4, q4 = i >> 2, pp = i & 3;
            const int nt = ceil_div(p.K, 16);
            for (int t = wave + RTW * cpass; t < nt; t += RTW * C) {
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t*)(X + (size_t)(8 * g + q4) * stride + 16 * t + 4 * pp));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t*)(X + (size_t)(8 * g + 4 + q4) * stride + 16 * t + 4 * pp));
                uint4 u;
                u.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
                u.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
                u.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
                u.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
                F[((size_t)t * n_rt + tile_id) * 64 + lane] = u;
            }
        }
        GCNPT_STAMP(stamps, 6);

        // request(k+2): the youngest entries of the vector-memory queue when the next iteration waits for them
        park_heads(h_nxt, tile_of(k + 2), meta_of(mb2));
        wave_lds_fence();
        h_nxt = load_heads(tile_of(k + 3));
        issue_rows(meta_of(mb2), tile_of(k + 2), R);
        GCNPT_STAMP(stamps, 7);
        lds_barrier();                                                  // S[b^1] complete; O and meta[k % 3] are free again
        GCNPT_STAMP(stamps, 8);
        mb = mb1;
    }
}

#ifdef GCNPT_RT_PART
template <typename CT, typename IT, typename OT, bool BWD, int VEC, int NTW, int KS, bool DZIN, int PI>
static inline int launch_rowpersist_cfg(hipStream_t s, RowTileParams p, int col_split, int grid) {
    const int n_ctiles = ceil_div(p.NOUT, 16);
    p.col_split = col_split;
    p.tiles_pp = ceil_div(n_ctiles, col_split);
    const int stride = lds_stride_dw(p.Kpad * (int)sizeof(CT) / 4) * 4 / (int)sizeof(CT);
    const int ncols_pass = p.tiles_pp * 16;
    const int ostride = out_stride_dw(std::min(round_up(p.NOUT, 16), ncols_pass) * (int)sizeof(OT) / 4) * 4 / (int)sizeof(OT);
    const size_t s_bytes = (size_t)ROWS * stride * sizeof(CT), o_bytes = (size_t)ROWS * ostride * sizeof(OT);
    const size_t lds = (BWD ? 4 : 2) * s_bytes + o_bytes + (size_t)ROWS * 13 * 3 * sizeof(int) + (size_t)ncols_pass * sizeof(float);
    if (lds > 160 * 1024) return GCNPT_NOT_TAKEN;
    auto kern = rowpersist_kernel<CT, IT, OT, BWD, VEC, NTW, KS, DZIN, PI>;
    GCNPT_LDS_ATTR_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(RT_THREADS), lds, s, p);
    note_launch(grid, RT_THREADS, lds, sizeof(p));
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

// The persistent form's configurations: (column tiles per wave, resident k-steps, prefetched chunks of own rows per thread), each sized to
// stay inside 256 registers without spilling (a spilled register comes back through scratch memory behind a full drain of the prefetch).
// A layer takes the one that holds all its k-steps and its own rows with the fewest workgroups per row tile (col_split), then the fewest
// registers.  The backward from dY and Y (three loads per gathered row) only takes the two smallest.
//   A (2,12,3) K <= 384: the C2 input layer          B (2,7,2) K <= 224: the C2 hidden layers       C (3,7,2): their 360-column backward
//   D (1,20,5) K <= 640: the C5 input layer          E (2,10,3) K <= 320: the C5 hidden layers      F (1,10,3): their backward from dY
template <typename CT, typename IT, typename OT, bool BWD, int VEC, bool DZIN>
static inline int try_rowpersist(hipStream_t s, const RowTileParams& p) {
    if constexpr (sizeof(CT) != 2 || VEC == 0) {
        return GCNPT_NOT_TAKEN;
    } else {
        constexpr bool MASKED = BWD && !DZIN;
        const int forced = option(GCNPT_OPT_PERSIST);
        const int n_rt = ceil_div(p.N, ROWS);
        if (forced == 0 || !p.out || (forced < 0 && n_rt <= 256)) return GCNPT_NOT_TAKEN;
        const int ksteps = p.Kpad / 32, n_ctiles = ceil_div(p.NOUT, 16), chunks_pt = ceil_div(ROWS * (p.Kpad / 8), RT_THREADS);
        struct Cfg { int ntw, ks, pi; bool masked_ok; };
        static const Cfg cfgs[6] = {{2, 12, 3, false}, {2, 7, 2, true}, {3, 7, 2, false}, {1, 20, 5, false}, {2, 10, 3, false}, {1, 10, 3, true}};
        int best = -1, best_c = 1 << 30, best_regs = 1 << 30;
        for (int i = 0; i < 6; ++i) {
            if (ksteps > cfgs[i].ks || chunks_pt > cfgs[i].pi || (MASKED && !cfgs[i].masked_ok)) continue;
            const int c = ceil_div(n_ctiles, RT_WAVES * cfgs[i].ntw), regs = cfgs[i].ntw * cfgs[i].ks + cfgs[i].pi;
            if (c < best_c || (c == best_c && regs < best_regs)) { best = i; best_c = c; best_regs = regs; }
        }
        if (best < 0 || best_c > 4) return GCNPT_NOT_TAKEN;
        const int grid = forced > 0 ? round_up(std::max(forced, 8 * best_c), 8) : 256;
        switch (best) {
            case 0: if constexpr (!MASKED) return launch_rowpersist_cfg<CT, IT, OT, BWD, VEC, 2, 12, DZIN, 3>(s, p, best_c, grid); break;
            case 1: return launch_rowpersist_cfg<CT, IT, OT, BWD, VEC, 2, 7, DZIN, 2>(s, p, best_c, grid);
            case 2: if constexpr (!MASKED) return launch_rowpersist_cfg<CT, IT, OT, BWD, VEC, 3, 7, DZIN, 2>(s, p, best_c, grid); break;
            case 3: if constexpr (!MASKED) return launch_rowpersist_cfg<CT, IT, OT, BWD, VEC, 1, 20, DZIN, 5>(s, p, best_c, grid); break;
            case 4: if constexpr (!MASKED) return launch_rowpersist_cfg<CT, IT, OT, BWD, VEC, 2, 10, DZIN, 3>(s, p, best_c, grid); break;
            default: return launch_rowpersist_cfg<CT, IT, OT, BWD, VEC, 1, 10, DZIN, 3>(s, p, best_c, grid);
        }
        return GCNPT_NOT_TAKEN;
    }
}
#endif

}  // namespace gcnpt
