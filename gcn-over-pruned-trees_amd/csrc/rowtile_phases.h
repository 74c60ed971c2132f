// The phases the two forms of the row-tile layer kernel share -- the one-shot form (rowtile_body.h) and the column-split form
// (colsplit_body.h) run the SAME gather, fragment-image emission, epilogue and row stores on a 32-row tile; they differ in how the
// weight fragments are scheduled and in which output columns a workgroup owns.  One copy here, so that the two forms stay
// bit-identical by construction (tests/test_gpu_parity.py::test_column_split_form_matches_one_shot checks it).
#pragma once
#include "layer_common.h"

// the kernel's big outputs (rows, fragment images): plain stores, or -DGCNPT_NT_STORES=1 non-temporal ones (experiment: does leaving
// less dirty data in the L2s shorten the launch boundary?  see EXPERIMENTS.md)
#if defined(GCNPT_NT_STORES) && GCNPT_NT_STORES
template <typename V> __device__ __forceinline__ void gcnpt_out_store(V* p, const V& v) {
    if constexpr (sizeof(V) == 16) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(p));
    } else {
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        __builtin_nontemporal_store(__builtin_bit_cast(u32x2, v), reinterpret_cast<u32x2*>(p));
    }
}
#define GCNPT_NT(ptr, val) gcnpt_out_store(ptr, val)
#else
#define GCNPT_NT_STORES 0
#define GCNPT_NT(ptr, val) (*(ptr) = (val))
#endif
#define GCNPT_PLAIN(ptr, val) (*(ptr) = (val))
// level 1: everything; 2: rows + the forward's S image (the dZ image is read by the very next launches); 3: rows only; 4: images only
#define GCNPT_ROW_STORE(ptr, val) do { if (GCNPT_NT_STORES >= 1 && GCNPT_NT_STORES <= 3) GCNPT_NT(ptr, val); else GCNPT_PLAIN(ptr, val); } while (0)
#define GCNPT_FRAG_STORE(bwd, ptr, val) do { if (GCNPT_NT_STORES == 1 || GCNPT_NT_STORES == 4 || (GCNPT_NT_STORES == 2 && !(bwd))) GCNPT_NT(ptr, val); else GCNPT_PLAIN(ptr, val); } while (0)

namespace gcnpt {

constexpr int ROWS = 32;             // token rows per workgroup (two 16-row MFMA tiles)
constexpr int NB_INLINE = 7;         // neighbours per row that the ELL head carries (include/gcnpt.h)

struct RowTileParams {
    const void* src;        // fwd: h [N,K]      bwd: dY [N,K]
    const void* yref;       // bwd: Y [N,K] (stored layer output)
    const void* wfrag;      // packed B operand, gcnpt_pack_weights
    const float* bias;      // fwd: [NOUT]
    const int32_t* g_row_ptr;   // pattern gathered over (fwd: A, bwd: A^T): CSR, only read for rows with > 7 entries
    const int32_t* g_col_idx;
    const int32_t* g_ell;       // its ELL head: [8r] = entries of row r, [8r+1..8r+7] = first columns
    const int32_t* d_ell;       // ELL head whose [8r] gives deg (always the forward pattern A)
    void* out;              // [N,NOUT]; NULL = only the side outputs below are wanted
    void* frag_out;         // NULL or fragment image (include/gcnpt.h) of the tile: fwd S = (A+I)h, bwd dZ
    float* zero_p[4];       // NULL or accumulators to clear for the weight gradients that follow (bwd: dW, db of this layer / the layer below)
    int zero_n[4];
    int N, T, K, NOUT, Kpad;
    unsigned chunk_magic;   // ceil(2^32 / (Kpad / 8)): division by the chunks per row as a multiply-high
    int vec_in, vec_out;    // vec_in: 8 / 4 / 0 elements per row load (selects the VEC instantiation); vec_out: 16 / 8 / 0 bytes per row store
    float scale;            // bwd: 1/(1-p) of the dropout applied to Y; fwd: 1/(1-drop_p)
    float drop_p;           // fwd
    unsigned drop_thresh16;
    uint64_t seed;
    const uint64_t* seed_dev;   // NULL, or a device word added to `seed` (a counter the caller advances between graph replays)
    const void* relu_src;       // bwd: NULL, or this layer's INPUT rows [N,NOUT] (the stored output of the layer below): the result then
    float next_scale;           //      leaves as that layer's dZ = dh * 1[input > 0] * next_scale / (deg + 1) instead of dh
    unsigned long long* stamps;   // diagnostic builds only
    int knob;
    int col_split, tiles_pp;    // column-split form only (colsplit_body.h): workgroups per row tile, column tiles each of them produces
};

// the tile's adjacency tables in LDS: [ROWS][8] ELL heads (count, 7 sentence-local columns), 1/(deg+1) (bwd: scale/(deg+1)), deg+1, the
// compacted list of rows that aggregate anything and its length, first row of each row's sentence (b * T)
struct TileMeta { int* rell; float* rinv; float* rden; int* glist; int* rsb; int* gcount; };
constexpr int TILE_META_INTS = 13 * ROWS;
__device__ __forceinline__ TileMeta tile_meta(int* meta) {
    return TileMeta{meta, reinterpret_cast<float*>(meta + 8 * ROWS), reinterpret_cast<float*>(meta + 9 * ROWS), meta + 10 * ROWS, meta + 11 * ROWS,
                    meta + 12 * ROWS};
}

// (1) the tile's adjacency: every wave loads all 32 ELL heads (64 lanes x 16 bytes) and the degrees for the denominators, and parks its
//     own copy of the derived tables: the load is unconditional and no wave waits for another one before it can start gathering.  All
//     waves write the same values to the same places; each reads back only after its own writes (wave_lds_fence)
struct TileHeads { int4 ell; int deg, sb; };
__device__ __forceinline__ TileHeads load_tile_heads(const RowTileParams& p, int r0, int lane) {
    const size_t er = (size_t)min(r0 + (lane >> 1), p.N - 1);
    TileHeads h;
    h.ell = reinterpret_cast<const int4*>(p.g_ell)[er * 2 + (lane & 1)];
    h.deg = p.d_ell[er * 8];                                                              // gcn.py:261
    h.sb = p.T ? (int)er / p.T * p.T : 0;   // the one division by T, done while the loads are on their way (T = 0: packed rows, columns are absolute)
    return h;
}
template <bool BWD>
__device__ __forceinline__ void park_tile_heads(const RowTileParams& p, const TileMeta& m, const TileHeads& h, int r0, int lane, bool want_rows) {
    const int erow = lane >> 1, ehalf = lane & 1;
    const bool first = ehalf == 0;
    const int e0 = (first && r0 + erow >= p.N) ? 0 : h.ell.x;                            // rows past the end aggregate nothing
    reinterpret_cast<int4*>(m.rell)[erow * 2 + ehalf] = make_int4(e0, h.ell.y, h.ell.z, h.ell.w);
    const float dn = (float)(h.deg + 1);
    m.rsb[erow] = h.sb;
    m.rinv[erow] = (BWD ? p.scale : 1.0f) / dn;      // both lanes of a row write it: a use under `first` only would let
    m.rden[erow] = dn;                               // hipcc sink the degree load into that branch, behind a full wait
    const bool agg = first && e0 > 0 && want_rows;
    const unsigned long long mk = __ballot(agg);
    if (agg) m.glist[__popcll(mk & ((1ull << lane) - 1ull))] = erow;
    if (lane == 0) *m.gcount = __popcll(mk);
}

template <typename IT, int NBU> struct GatherItem { raw8<IT> s, sy, nb[NBU], nby[NBU]; int dcnt[NBU]; };

// (2) gcn.py:269 as a gather, S[row,:] = x[row,:] + sum_{c in pattern row} x[c,:]:
//   (2a) rows that aggregate something, compacted: item gi = (list slot, 8-column chunk), so the ~4 such rows of a tile spread over ALL
//        waves and their neighbour loads (<= NBU per round; a pruned-tree row has 3-4 entries) leave in ONE round trip per item.  The
//        first 7 neighbours come from the ELL head in LDS, the (rare) rest from col_idx; lanes without an e-th neighbour load a dummy
//        line and drop it.  MASKED: the loader derives dZ = dY * 1[Y>0] * scale / (deg+1) for the row and its neighbours;
//   (2b) every other row is a plain copy (tile_copy_item).
// VEC: 8 = rows read 16 bytes at a time (K % 8 == 0), 4 = in 8-byte (bf16) / 16-byte (f32) halves (K % 4 == 0), 0 = element loads.
template <typename CT, typename IT, bool MASKED, int VEC, int NBU>
struct TileGather {
    const RowTileParams& p;
    const TileMeta& m;
    CT* S;                      // the tile in LDS, `stride` CT elements per row
    int stride, r0;

    __device__ __forceinline__ int nchunk() const { return p.Kpad / 8; }
    __device__ __forceinline__ int kmax8() const { return VEC == 8 ? p.K - 8 : (VEC == 4 ? p.K - 4 : p.K - 1); }
    __device__ __forceinline__ int div_chunk(int x) const { return (int)__umulhi((unsigned)x, p.chunk_magic); }    // x / nchunk, exact for x * nchunk < 2^32
    __device__ __forceinline__ const IT* src() const { return static_cast<const IT*>(p.src); }
    __device__ __forceinline__ const IT* yref() const { return static_cast<const IT*>(p.yref); }
    __device__ __forceinline__ void ld8(const IT* base, size_t row, int k0c, raw8<IT>& dst) const {
        if constexpr (VEC == 8) issue8<IT, true>(base, row, p.K, k0c, dst);
        else if constexpr (VEC == 4) issue8_half<IT>(base, row, p.K, k0c, dst);
        else issue8<IT, false>(base, row, p.K, k0c, dst);
    }
    // a chunk of one of the tile's own rows (item `it` of the row-major chunk order)
    __device__ __forceinline__ void issue_self(int it, raw8<IT>& s, raw8<IT>& sy) const {
        const int row = div_chunk(it), k0 = (it - row * nchunk()) * 8;
        const size_t r = (size_t)min(r0 + row, p.N - 1);
        ld8(src(), r, min(k0, kmax8()), s);
        if (MASKED) ld8(yref(), r, min(k0, kmax8()), sy);
    }
    __device__ __forceinline__ bool decode(int n_g, int gi, int& row, int& k0, int& n) const {
        const bool has = gi < n_g;
        const int li = has ? div_chunk(gi) : 0;
        row = has ? m.glist[li] : 0;
        k0 = has ? (gi - li * nchunk()) * 8 : 0;
        n = (has && k0 < p.K) ? m.rell[row * 8] : 0;
        return has;
    }
    __device__ __forceinline__ void issue(int n_g, int gi, GatherItem<IT, NBU>& g) const {
        int row, k0, n;
        decode(n_g, gi, row, k0, n);
        size_t r = (size_t)min(r0 + row, p.N - 1);
        const int sbase = m.rsb[row];                                  // first row of this row's sentence
        int k0c = min(k0, kmax8());
#ifdef GCNPT_STAMPS
        const bool in_lds = (p.knob & 64) != 0;                       // experiment (timing only, wrong values): what an in-tile neighbour and the
        if (in_lds) { r = (size_t)r0; k0c = 0; }                      // item's own piece would cost if they came from the parked rows in LDS
#endif
        ld8(src(), r, k0c, g.s);
        if (MASKED) ld8(yref(), r, k0c, g.sy);
#ifdef GCNPT_STAMPS
        k0c = min(k0, kmax8());
#endif
#pragma unroll
        for (int e = 0; e < NBU; ++e) {
            // no e-th neighbour: the first 16 bytes of the tile's first row, one cache line for all such lanes.  (NOT the
            // item's own row: hipcc would then reuse the load above, wait for it, and branch around the others.)
            bool on = e < min(n, NB_INLINE);
#ifdef GCNPT_STAMPS
            if (in_lds && on) { const int cc = sbase + m.rell[row * 8 + 1 + e]; if (cc >= r0 && cc < r0 + ROWS) on = false; }
#endif
            const size_t c = on ? (size_t)(sbase + m.rell[row * 8 + 1 + e]) : (size_t)r0;
            const int kc = on ? k0c : 0;
            ld8(src(), c, kc, g.nb[e]);
            if (MASKED) {
                ld8(yref(), c, kc, g.nby[e]);
                g.dcnt[e] = p.d_ell[c * 8];
            }
        }
    }
    __device__ __forceinline__ void finish(int n_g, int gi, const GatherItem<IT, NBU>& g) const {
        int row, k0, n;
        const bool has = decode(n_g, gi, row, k0, n);
        const bool live = has && k0 < p.K;
        const size_t rc = (size_t)min(r0 + row, p.N - 1);
        const int sbase = m.rsb[row];
        const int k0c = min(k0, kmax8());
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
        // rows with more than NBU entries (branching tree nodes, dense adjacency input): further round trips
        auto round = [&](int e0, int lim, auto from_lds) {
            raw8<IT> nb[NBU], nby[NBU];
            float ninv[NBU];
#pragma unroll
            for (int e = 0; e < NBU; ++e) {
                const bool on = e0 + e < lim;
                size_t c;
                if constexpr (decltype(from_lds)::value) {
                    c = (size_t)(sbase + m.rell[row * 8 + 1 + min(e0 + e, NB_INLINE - 1)]);
                } else {                                            // > 7 entries: continue in the CSR
                    const int beg = p.T ? p.g_row_ptr[(size_t)(sbase / p.T) * (p.T + 1) + (rc - sbase)] : p.g_row_ptr[rc];
                    c = (size_t)(sbase + p.g_col_idx[on ? beg + e0 + e : beg]);
                }
                c = on ? c : rc;
                ld8(src(), c, k0c, nb[e]);
                if (MASKED) {
                    ld8(yref(), c, k0c, nby[e]);
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
        if (has) tile<CT>::put8(S + (size_t)row * stride + k0, acc);
    }
    // (2b) a chunk of the tile's own rows: into S unless the row aggregates (finish() writes those), into Z for the backward's image
    template <bool BWD>
    __device__ __forceinline__ void copy_item(int it, const raw8<IT>& s, const raw8<IT>& sy, CT* Z) const {
        if (it >= ROWS * nchunk()) return;
        const int row = div_chunk(it), k0 = (it - row * nchunk()) * 8;
        const bool live = r0 + row < p.N && k0 < p.K;
        const bool to_s = !(p.out && m.rell[row * 8] > 0);
        if constexpr (!BWD && sizeof(IT) == 2 && sizeof(CT) == 2) {     // bf16 rows into a bf16 tile: the 16 bytes as they are
            if (to_s) *reinterpret_cast<uint4*>(S + (size_t)row * stride + k0) = live ? s.a : make_uint4(0, 0, 0, 0);
            return;
        }
        float acc[8];
        unpack8<IT>(s, live, acc);                                      // the explicit W(h) term, gcn.py:271
        if (MASKED) {
            float y[8];
            unpack8<IT>(sy, live, y);
            const float inv = m.rinv[row];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = (y[j] > 0.0f) ? acc[j] * inv : 0.0f;
        }
        if (BWD) {
            if (p.frag_out) tile<CT>::put8(Z + (size_t)row * stride + k0, acc);
        }
        if (to_s) tile<CT>::put8(S + (size_t)row * stride + k0, acc);
    }
};

// Side output: column tiles first, first + step, ... < nt of the LDS tile X (xstride elements per row) as the weight gradient's fragment
// image (include/gcnpt.h): rows are its contraction index, 8 consecutive rows per lane, read transposed with ds_read_b64_tr_b16
template <typename XT>
__device__ __forceinline__ void emit_tile_image(uint4* F, const XT* X, int xstride, int first, int step, int nt, int lane, size_t n_blocks, int tile_id,
                                                bool bwd_store) {
    if constexpr (sizeof(XT) == 2) {
        const int i = lane & 15, g = lane >> 4, q4 = i >> 2, pp = i & 3;
        for (int t = first; t < nt; t += step) {
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t*)(X + (size_t)(8 * g + q4) * xstride + 16 * t + 4 * pp));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t*)(X + (size_t)(8 * g + 4 + q4) * xstride + 16 * t + 4 * pp));
            uint4 u;
            u.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
            u.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
            u.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
            u.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
            GCNPT_FRAG_STORE(bwd_store, &F[((size_t)t * n_blocks + tile_id) * 64 + lane], u);
        }
    } else {
        const int i = lane & 15, g = lane >> 4;
        const size_t nks = n_blocks * 2;
        for (int tk = first; tk < nt * 2; tk += step) {
            const int t = tk >> 1, kk = tk & 1;
            uint4 u;
            u.x = __float_as_uint(X[(size_t)(16 * kk + 4 * g + 0) * xstride + 16 * t + i]);
            u.y = __float_as_uint(X[(size_t)(16 * kk + 4 * g + 1) * xstride + 16 * t + i]);
            u.z = __float_as_uint(X[(size_t)(16 * kk + 4 * g + 2) * xstride + 16 * t + i]);
            u.w = __float_as_uint(X[(size_t)(16 * kk + 4 * g + 3) * xstride + 16 * t + i]);
            GCNPT_FRAG_STORE(bwd_store, &F[((size_t)t * nks + 2 * (size_t)tile_id + kk) * 64 + lane], u);
        }
    }
}

// (4a) the epilogue on the accumulators of ONE 16-column tile (both 16-row halves of the row tile) -> the LDS out tile O.  Lane
//      (i = lane & 15, q = lane >> 4) holds row i and the 4 consecutive columns col0 .. col0+3 (col0 = 16 tile + 4 q); lcol0 = col0
//      relative to the first column O holds.  den / inv: deg+1 and its reciprocal for the lane's two rows.
template <typename OT, bool BWD>
__device__ __forceinline__ void epilogue_tile(const RowTileParams& p, const f32x4_t (&acc)[2], int col0, int lcol0, int r0, const float (&den)[2],
                                              const float (&inv)[2], const float* sbias, OT* O, int ostride, uint64_t seed_off, int lane) {
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
            float x = acc[mt][g];
            if (!BWD) {
                x = div_by(x + 2.0f * bq[g], den[mt], inv[mt]);   // gcn.py:270-271 (the bias enters twice), 390
                x = x > 0.0f ? x : 0.0f;                         // gcn.py:392
            }
            v[g] = x;
        }
        if (!BWD && p.drop_p > 0.0f) {                            // gcn.py:393: one hash per column pair
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

// (4b) columns [c_lo, c_lo + width) of the tile's rows leave O for `out` in 16-byte pieces (8-byte ones when the row width only allows
//      those: bf16 rows of 300 columns; elements otherwise).  bwd with relu_src: the hand-over to the layer below -- its dZ instead of dh
//      (gcn.py:390-393 differentiated where the rows are at hand): the input rows are fetched in one batch, then masked and scaled
//      while the tile leaves LDS.  RTT: threads of the workgroup (16 per row: one round with 512, two with 256).
template <typename OT, bool BWD, int RTT>
__device__ __forceinline__ void store_tile_rows(const RowTileParams& p, const OT* O, int ostride, const float* rden, int r0, int c_lo, int width, int tid) {
    OT* out = static_cast<OT*>(p.out);
    const OT* relu = BWD ? static_cast<const OT*>(p.relu_src) : nullptr;
    auto store_rows = [&](auto vtag) {
        using V = decltype(vtag);                                       // uint4 or uint2
        constexpr int PER = (int)sizeof(V) / (int)sizeof(OT);
        constexpr int NW = (int)sizeof(V) / 4;
        const int pieces = width / PER;                                  // 16 threads per row: no division, 256 contiguous bytes each round
#pragma unroll
        for (int rh = 0; rh < ROWS; rh += RTT / 16) {
            const int row = rh + (tid >> 4), r = r0 + row;
            if (BWD && relu) {
                constexpr int RP = 4;
                const float f = p.next_scale / rden[row];
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
                        GCNPT_ROW_STORE(reinterpret_cast<V*>(out + (size_t)r * p.NOUT + c_lo + pc * PER), o);
                    }
                }
            } else if (r < p.N) {
                for (int pc = tid & 15; pc < pieces; pc += 16)
                    GCNPT_ROW_STORE(reinterpret_cast<V*>(out + (size_t)r * p.NOUT + c_lo + pc * PER),
                                    *reinterpret_cast<const V*>(O + (size_t)row * ostride + pc * PER));
            }
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
                x = hx > 0.0f ? x * (p.next_scale / rden[row]) : 0.0f;
                if constexpr (sizeof(OT) == 2) v = f32_to_bf16(x); else v = x;
            }
            out[(size_t)r * p.NOUT + c_lo + c] = v;
        }
    }
}

}  // namespace gcnpt
