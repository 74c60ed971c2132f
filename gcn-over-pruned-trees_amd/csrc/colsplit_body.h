// The row-tile layer kernel for SMALL batches of WIDE layers: every 32-row tile is given to `col_split` workgroups that gather the same rows
// and each produce `tiles_pp` of the layer's output column tiles.
//
// Why: the one-tile-per-workgroup kernel of rowtile_body.h pulls ALL weight fragments of the layer through its CU's
// L1 path (~20 B per clock and CU) -- 360 KB at the C5 input layer, next to 50 KB of rows -- and when a batch has a few dozen row tiles
// (BASELINE configs[4] on 8 GPUs: 16 sentences per GPU, token-packed: 25 tiles) nine CUs in ten have no tile at all while the others
// wait for their weights.  Split 3 ... 8 ways by output COLUMNS a workgroup takes in 45 ... 120 KB of weights; the rows are gathered
// col_split times (from the same XCD's L2: the workgroups of a tile sit next to each other), which idle CUs do for free.  At the C2 widths
// (156 KB of weights) the split buys nothing and costs the backward launch its passenger (the weight gradient that rides on idle CUs):
// the dispatcher takes this form only from 170 KB of weight fragments on and for at most 128 row tiles.
//
// Structure: the one-shot kernel's, with two differences.  (1) A wave requests the fragments of its own column tiles FIRST, before the
// ELL heads -- they are few now and land while the heads make their round trip -- and a wave whose slot lies past the workgroup's share
// requests nothing and skips the matrix phase (the condition sits in front of every other load, so no later wait is counted across it).
// (2) All k-steps of a wave's tiles are resident (six register-budgeted configurations, no spills), the fragment image of S / dZ is dealt over the
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
    constexpr int NBU = WIDE ? 2 : 4;
    constexpr int META_INTS = 13 * ROWS;

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
    float* sbias = reinterpret_cast<float*>(meta0 + META_INTS);            // [tiles_pp * 16] fwd: the bias of this workgroup's columns
    struct Meta { int* rell; float* rinv; float* rden; int* glist; int* rsb; int* gcount; };
    const Meta m{meta0, reinterpret_cast<float*>(meta0 + 8 * ROWS), reinterpret_cast<float*>(meta0 + 9 * ROWS), meta0 + 10 * ROWS, meta0 + 11 * ROWS,
                 meta0 + 12 * ROWS};

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
        // rows with more than NBU entries: further round trips
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
        for (int base = RTT; base < n_g; base += RTT) {              // more than 512 (aggregating row, chunk) items: further rounds
            GItem g;
            g_issue(m, r0, n_g, base + tid, g);
            g_finish(m, r0, n_g, base + tid, g);
        }
        for (int first = PI * RTT; first < n_items; first += RTT) {   // K wider than PI covers: further batches
            raw8<IT> s, sy;
            issue_self(r0, first, s, sy);
            copy_item(m, r0, first + tid, s, sy);
        }
    };

    // ---- (0) this wave's weight fragments, FIRST: with the columns split they are few (19 ... 60 KB per workgroup) and land while the heads
    //      make their round trip.  Wave w owns column tiles cpass * tiles_pp + w, + 8, ...; a wave whose slot lies past the workgroup's share
    //      requests nothing (the condition sits in front of every other load of the kernel, so no later wait is counted across it).
    bool has_tile[NTW];
    bool any_tile = false;
    uint4 wreg[KS][NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int tloc = j * RTW + wave, tl = cpass * p.tiles_pp + tloc;
        has_tile[j] = tloc < p.tiles_pp && tl < n_ctiles;
        any_tile = any_tile || has_tile[j];
        if (has_tile[j]) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wreg[ks][j] = wfrag[((size_t)tl * ksteps + min(ks, ksteps - 1)) * 64 + lane];
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wreg[ks][j] = make_uint4(0, 0, 0, 0);
        }
    }
    GCNPT_STAMP_REAL(p.stamps);
    GCNPT_STAMP(p.stamps, 0);

    // ---- (1) the tile's adjacency (ELL heads, degrees) -> LDS; (2) its rows and its aggregating rows' neighbours -> S ---------------------
    const Heads heads = load_heads(tile_id);
    float bias_v = 0.0f;
    if constexpr (!BWD) bias_v = p.bias[min(cpass * ncols_pass + min(tid, ncols_pass - 1), p.NOUT - 1)];
    park_heads(heads, tile_id, m);
    if constexpr (!BWD) { if (tid < ncols_pass) sbias[tid] = bias_v; }
    wave_lds_fence();
    GCNPT_STAMP(p.stamps, 1);
    Rows R;
    issue_rows(m, tile_id, R);
    GCNPT_STAMP(p.stamps, 2);
    finish_rows(m, tile_id, R);
    GCNPT_STAMP(p.stamps, 3);
    lds_barrier();                                                      // S (and Z) complete
    GCNPT_STAMP(p.stamps, 4);

    OT* out = static_cast<OT*>(p.out);
    const int arow = lane & 15, kgrp = lane >> 4;
    const int c_lo = cpass * ncols_pass, c_hi = min(p.NOUT, c_lo + ncols_pass);
    const int width = c_hi - c_lo;
    const OT* relu = BWD ? static_cast<const OT*>(p.relu_src) : nullptr;

    // side output: the tile in MFMA fragment order for the weight gradient (rows are its contraction index); the column tiles of the
    // image are dealt over the workgroups that share this row tile
    if (p.frag_out) {
        uint4* F = static_cast<uint4*>(p.frag_out);
        const CT* X = BWD ? Zw : Sw;
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

    // (3) the tile meets the resident weights
    f32x4_t acc[2][NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) { acc[0][j] = (f32x4_t){0, 0, 0, 0}; acc[1][j] = (f32x4_t){0, 0, 0, 0}; }
    if (any_tile) {                                                 // (a wave without a column tile reads no operand either: LDS bandwidth)
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

    // epilogue on the accumulators -> O.  Lane (i = lane & 15, q = lane >> 4) holds row i and columns 4q..4q+3 of a 16x16 tile.
    {
        float den[2], inv[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) { den[mt] = m.rden[mt * 16 + (lane & 15)]; inv[mt] = m.rinv[mt * 16 + (lane & 15)]; }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int tl = cpass * p.tiles_pp + j * RTW + wave;
            if (!has_tile[j]) continue;
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
    GCNPT_STAMP(p.stamps, 6);
    lds_barrier();                                                      // O complete
    GCNPT_STAMP(p.stamps, 7);

    // whole rows leave in 16-byte pieces (8-byte ones when the width only allows those)
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
