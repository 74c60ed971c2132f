// GCN layer forward / backward-data for BIG batches (more 32-row tiles than CUs): reference model/gcn.py:269-271, 390-393 and their
// autograd, the same arithmetic as rowtile_body.h in the same order (bit-identical rows), organised for throughput instead of for one
// workgroup's latency chain.
//
// Why a second form.  The row-tile kernel gives every 32 rows a workgroup that pulls the WHOLE weight image through its CU (361 KB at
// 600 -> 300) and gathers its tile in serial batches before the first MFMA: at the C5 shape (38 400 rows) 433 MB of L2 -> L1 weight
// traffic per launch, and stamps of 57 k cycles per tile of which 19 k wait for weight fragments and 24 k for the gather's round trips.
// Here a workgroup of 8 waves (2 x 4 over rows x columns) owns 128 (or 64) rows and walks the contraction one 32-column k-step at a time:
//   * the k-step's weight fragments and the tile's rows of those 32 columns (own rows; for the ~1 row in 8 that aggregates, the row and
//     its <= 7 ELL neighbours, summed in fp32 in the same order as the row-tile kernel) are requested RG_PF k-steps ahead into
//     registers, parked in a two-stage LDS ring, and read from there by all waves: a weight fragment crosses the CU's L1 path once per
//     128 rows instead of once per 32, and nothing waits for a whole tile's gather;
//   * every load is unconditional (clamped, a dead slot reads one fixed address): hipcc then counts the outstanding loads exactly
//     (`s_waitcnt vmcnt(N)`), a conditional load would make every wait a full drain;
//   * register rings are first-class vectors (ext_vector_type): arrays of uint4 STRUCTS were parked in scratch memory by hipcc;
//   * the fragment image of the tile for the weight gradient (S forward, own dZ rows backward) is emitted k-step by k-step from the ring
//     with ds_read_b64_tr_b16; epilogue and row stores as in the row-tile kernel, through an LDS out tile that aliases the ring.
// bf16 compute with bf16 activations only (what big batches run in); everything else keeps the row-tile kernel.
#include "rowtile_body.h"

namespace gcnpt {

constexpr int RG_WN = 4;                 // wave grid: WM (1 or 2) groups of rows x 4 groups of columns
constexpr int RG_PF = 2;                 // k-steps in flight in registers
constexpr int RG_NB = NB_INLINE;         // neighbours of an aggregating row held in registers (the ELL head's 7)
constexpr int RG_SSTR = 40;              // LDS row stride of a k-step's 32 columns, bf16 elements (64 B payload + 16: the 16 rows x 16 B of one
                                         // ds_read_b128 quarter-wave fall on distinct banks)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// Workgroup barrier for the LDS ring ONLY: this wave's LDS writes have landed (lgkmcnt(0)), then s_barrier.  __syncthreads() is a
// workgroup-scope fence over global memory too, i.e. it carries an s_waitcnt vmcnt(0): with it every k-step drained the register
// prefetch queue it is supposed to leave in flight (measured: 7.7 k cycles per k-step instead of ~1 k).
__device__ __forceinline__ void rg_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// MODE 0 forward, 1 backward-data on ready-made dZ rows (DZIN).  MTW / NTW: 16-row / 16-column MFMA tiles per wave.
// RG_WM = 1 (4 waves, 64 rows: two workgroups share a CU and overlap each other's prologue / epilogue -- up to 320 columns) or 2
// (8 waves, 64 rows x up to 640 columns).
template <int MODE, int RG_WM, int MTW, int NTW, int VEC>
__global__ __launch_bounds__(RG_WM * RG_WN * 64, 2) void rowgemm_kernel(const RowTileParams p) {
    constexpr int RG_WAVES = RG_WM * RG_WN, RG_THREADS = RG_WAVES * 64;
    constexpr bool BWD = MODE != 0;
    constexpr int R = RG_WM * MTW * 16;                       // rows per workgroup: 128 / 64
    constexpr int NCOL = RG_WN * NTW * 16;                    // columns a workgroup covers: 320 / 640 / 256
    constexpr int RB = R / 32;                                // 32-row blocks (k-steps of the fragment image) per tile
    constexpr int WPT = (RG_WN * NTW * 64 + RG_THREADS - 1) / RG_THREADS;      // weight fragment-lanes per thread and k-step
    constexpr size_t S_BYTES = (size_t)R * RG_SSTR * 2, W_BYTES = (size_t)RG_WN * NTW * 64 * 16;
    constexpr size_t STAGE = (BWD ? 2 : 1) * S_BYTES + W_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // meta | two stages (Sc | [Zc] | Wc); the out tile O aliases the stages after the k-loop
    int* rell = reinterpret_cast<int*>(smem);                 // [R][8]
    float* rinv = reinterpret_cast<float*>(rell + 8 * R);     // [R] fwd: 1/(deg+1)
    float* rden = rinv + R;                                   // [R] deg+1
    int* rsb = reinterpret_cast<int*>(rden + R);              // [R] first row of the row's sentence
    int* glist = rsb + R;                                     // [R] rows that aggregate at least one entry
    int* gcount = glist + R;                                  // [4]
    float* sbias = reinterpret_cast<float*>(gcount + 4);      // [NCOL]
    unsigned char* stages = smem + (size_t)(12 * R + 4 + NCOL) * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & (RG_WM - 1), wn = wave / RG_WM;
    const int n_blocks = gridDim.x, block_id = blockIdx.x;
    const int xg = block_id & 7, xq = n_blocks >> 3, xr = n_blocks & 7;
    const int tile_id = xg * xq + min(xg, xr) + (block_id >> 3);      // XCD x takes a contiguous run of tiles (rowtile_body.h)
    const int r0 = tile_id * R;
    const bf16_t* src = static_cast<const bf16_t*>(p.src);
    const u32x4_t* wfrag = static_cast<const u32x4_t*>(p.wfrag);
    const int n_tiles = ceil_div(p.NOUT, 16), ksteps = p.Kpad / 32, w_tiles = ceil_div(p.K, 16);
    uint64_t seed_off = 0;
    if (!BWD && p.seed_dev) seed_off = *p.seed_dev;
    GCNPT_STAMP_REAL(p.stamps);
    GCNPT_STAMP(p.stamps, 0);

    // ---- the tile's adjacency: ELL heads, degrees, sentence bases; the rows that aggregate anything, compacted --------------
    for (int q = tid; q < 2 * R; q += RG_THREADS) {
        const int row = q >> 1, half = q & 1;
        const size_t er = (size_t)min(r0 + row, p.N - 1);
        int4 e = reinterpret_cast<const int4*>(p.g_ell)[er * 2 + half];
        if (half == 0 && r0 + row >= p.N) e.x = 0;                                   // rows past the end aggregate nothing
        reinterpret_cast<int4*>(rell)[row * 2 + half] = e;
        if (half == 0) {
            const float dn = (float)(p.d_ell[er * 8] + 1);                           // gcn.py:261
            rden[row] = dn;
            rinv[row] = 1.0f / dn;
            rsb[row] = p.T ? (int)er / p.T * p.T : 0;
        }
    }
    if constexpr (!BWD)
        for (int c = tid; c < NCOL; c += RG_THREADS) sbias[c] = p.bias[min(c, p.NOUT - 1)];
    __syncthreads();
    if (wave == 0) {
        int ng = 0;
#pragma unroll
        for (int c = 0; c < R / 64; ++c) {
            const int row = c * 64 + lane;
            const bool agg = rell[row * 8] > 0 && p.out != nullptr;
            const unsigned long long m = __ballot(agg);
            if (agg) glist[ng + __popcll(m & ((1ull << lane) - 1ull))] = row;
            ng += __popcll(m);
        }
        if (lane == 0) *gcount = ng;
    }
#pragma unroll
    for (int z = 0; z < 4; ++z)
        if (p.zero_p[z])
            for (int i = block_id * RG_THREADS + tid; i < p.zero_n[z]; i += n_blocks * RG_THREADS) p.zero_p[z][i] = 0.0f;
    __syncthreads();
    const int ng = *gcount;

    // ---- this thread's share of a k-step, fixed for the whole tile ------------------------------------------------------------
    // own item: (row, 16-byte chunk of the k-step's 64 bytes); aggregating item: the same for list slot `tid >> 2`, with the row's
    // neighbours.  4 R <= threads, so one item of each kind per thread suffices.
    static_assert(4 * R <= RG_THREADS, "one own item per thread");
    const int o_row = tid >> 2, c16 = tid & 3;
    const bool o_on = o_row < R;
    const size_t o_r = (size_t)min(r0 + min(o_row, R - 1), p.N - 1);
    const bool o_live_row = o_on && r0 + o_row < p.N;
    const bool o_plain = o_on && !(p.out && rell[min(o_row, R - 1) * 8] > 0);          // an aggregating row is written by its aggregating item
    const bool a_on = (tid >> 2) < ng;
    const int a_row = a_on ? glist[tid >> 2] : 0;
    const int a_n = a_on ? rell[a_row * 8] : 0;
    const size_t a_r = (size_t)min(r0 + a_row, p.N - 1);
    size_t a_nb[RG_NB];
#pragma unroll
    for (int e = 0; e < RG_NB; ++e) a_nb[e] = (a_on && e < a_n) ? (size_t)(rsb[a_row] + rell[a_row * 8 + 1 + e]) : (size_t)0;
    const int kmax8 = VEC == 8 ? p.K - 8 : p.K - 4;

    struct StageRegs { u32x4_t own, aown, anb[RG_NB], w[WPT]; };
    auto ld16 = [&](size_t row, int k0c, bool on) -> u32x4_t {
        // 8 elements of `row` from column k0c (clamped by the caller); a dead slot reads the tile's first 16 bytes: one line, no traffic
        const bf16_t* q = src + (on ? row * (size_t)p.K + k0c : (size_t)r0 * p.K);
        if constexpr (VEC == 8) {
            return *reinterpret_cast<const u32x4_t*>(q);
        } else {                                             // rows only 8-byte aligned (K % 4 == 0): two halves, the upper one clamped into the row
            const int k1 = on ? min(k0c + 4, p.K - 4) - k0c : 0;
            const uint2 lo = *reinterpret_cast<const uint2*>(q);
            const uint2 hi = *reinterpret_cast<const uint2*>(q + k1);
            return (u32x4_t){lo.x, lo.y, hi.x, hi.y};
        }
    };
    // The aggregating items sit in the first ceil(4 ng / 64) waves; the other waves skip those 8 loads (wave-uniform branch: a dwordx4
    // wave-load costs ~16 cycles of address processing whatever its lanes do, and 8 dead ones per wave and k-step made the k-step
    // issue-bound: 5.6 k cycles).  They are issued LAST in a stage so that hipcc's count of the loads behind a stage stays exact on the
    // waves that skip them (it assumes the shorter path).
    const bool wave_agg = wave * 64 < 4 * ng;
    auto stage_load = [&](int ks, StageRegs& s) {
        const bool kon = ks < ksteps;
        const int k0 = ks * 32 + 8 * c16;
        const bool con = kon && k0 < p.K;                    // this chunk has columns inside the row
        const int k0c = min(k0, kmax8);
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int f = tid + u * RG_THREADS, t = min(f >> 6, n_tiles - 1);
            s.w[u] = wfrag[kon ? ((size_t)t * ksteps + ks) * 64 + (f & 63) : (size_t)0];
        }
        s.own = ld16(o_r, k0c, con && o_on);
        if (wave_agg) {
            s.aown = ld16(a_r, k0c, con && a_on);
#pragma unroll
            for (int e = 0; e < RG_NB; ++e) s.anb[e] = ld16(a_nb[e], k0c, con && e < a_n);
        }
    };
    auto bf16x8_to_f32 = [](const u32x4_t& u, float (&v)[8]) {
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
        v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
    };
    auto stage_store = [&](int ks, const StageRegs& s, unsigned char* stage) {
        bf16_t* Sc = reinterpret_cast<bf16_t*>(stage);
        bf16_t* Zc = reinterpret_cast<bf16_t*>(stage + S_BYTES);                       // BWD: the tile's own rows, before aggregation
        u32x4_t* Wc = reinterpret_cast<u32x4_t*>(stage + (BWD ? 2 : 1) * S_BYTES);
        const int k0 = ks * 32 + 8 * c16;
        const u32x4_t zero = {0u, 0u, 0u, 0u};
        // columns past K inside a straddling chunk (K % 8 == 4): the half loads returned real values of the row there; the row-tile
        // kernel leaves them (they meet zero weights and fall outside the weight gradient), so does this one -- same bits
        if (o_on) {
            const u32x4_t v = (o_live_row && k0 < p.K) ? s.own : zero;
            if constexpr (BWD) *reinterpret_cast<u32x4_t*>(Zc + (size_t)o_row * RG_SSTR + 8 * c16) = v;
            if (o_plain) *reinterpret_cast<u32x4_t*>(Sc + (size_t)o_row * RG_SSTR + 8 * c16) = v;
        }
        if (a_on) {
            const bool live = r0 + a_row < p.N && k0 < p.K;
            float acc[8], v[8];
            bf16x8_to_f32(live ? s.aown : zero, acc);                                     // the explicit W(h) term, gcn.py:271
#pragma unroll
            for (int e = 0; e < RG_NB; ++e) {
                bf16x8_to_f32((live && e < a_n) ? s.anb[e] : zero, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
            if (a_n > RG_NB) {                                                          // > 7 entries (dense adjacency input): continue in the CSR
                const int sb = rsb[a_row];
                const size_t rc = a_r;
                const int beg = p.T ? p.g_row_ptr[(size_t)(sb / p.T) * (p.T + 1) + (rc - sb)] : p.g_row_ptr[rc];
                const int k0c = min(k0, kmax8);
                for (int e = RG_NB; e < a_n; ++e) {
                    const size_t c = (size_t)(sb + p.g_col_idx[beg + e]);
                    bf16x8_to_f32(live ? ld16(c, k0c, true) : zero, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += v[j];
                }
            }
            u32x4_t o;
            o.x = (unsigned)f32_to_bf16(acc[0]) | ((unsigned)f32_to_bf16(acc[1]) << 16);
            o.y = (unsigned)f32_to_bf16(acc[2]) | ((unsigned)f32_to_bf16(acc[3]) << 16);
            o.z = (unsigned)f32_to_bf16(acc[4]) | ((unsigned)f32_to_bf16(acc[5]) << 16);
            o.w = (unsigned)f32_to_bf16(acc[6]) | ((unsigned)f32_to_bf16(acc[7]) << 16);
            *reinterpret_cast<u32x4_t*>(Sc + (size_t)a_row * RG_SSTR + 8 * c16) = o;
        }
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int f = tid + u * RG_THREADS;
            if (f < RG_WN * NTW * 64) Wc[f] = s.w[u];
        }
    };

    f32x4_t acc[MTW][NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4_t){0, 0, 0, 0};

    GCNPT_STAMP(p.stamps, 1);
    StageRegs sr0, sr1;
    static_assert(RG_PF == 2, "the ring below is written out");
    stage_load(0, sr0);
    stage_load(1, sr1);
    unsigned char* st0 = stages;
    unsigned char* st1 = stages + STAGE;
    const int i16 = lane & 15, g = lane >> 4, q4 = i16 >> 2, pp = i16 & 3;
    uint4* F = static_cast<uint4*>(p.frag_out);
    const size_t nks_img = (size_t)ceil_div(p.N, 32);

#define GCNPT_RG_KSTEP(KS, q, stage)                                                                                              \
    {                                                                                                                             \
        const int ks = (KS);                                                                                                      \
        if (ks == 4) GCNPT_STAMP(p.stamps, 5);                                                                                    \
        stage_store(ks, q, stage);                                                                                                \
        if (ks == 4) GCNPT_STAMP(p.stamps, 6);                                                                                    \
        stage_load(ks + RG_PF, q);                                                                                                \
        rg_lds_barrier();                                                                                                         \
        if (ks == 4) GCNPT_STAMP(p.stamps, 7);                                                                                    \
        if (ks < ksteps) {                                                                                                        \
            const bf16_t* Sc = reinterpret_cast<const bf16_t*>(stage);                                                            \
            const u32x4_t* Wc = reinterpret_cast<const u32x4_t*>(stage + (BWD ? 2 : 1) * S_BYTES);                                \
            u32x4_t sf[MTW];                                                                                                      \
            _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                                                       \
                sf[i] = *reinterpret_cast<const u32x4_t*>(Sc + (size_t)((wm * MTW + i) * 16 + i16) * RG_SSTR + 8 * g);             \
            if (p.out) {                                                                                                          \
                _Pragma("unroll") for (int j = 0; j < NTW; ++j) {                                                                 \
                    const u32x4_t wf = Wc[(wn * NTW + j) * 64 + lane];                                                            \
                    _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                                               \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf),                     \
                                                                             __builtin_bit_cast(bf16x8_t, sf[i]), acc[i][j], 0, 0, 0); \
                }                                                                                                                 \
            }                                                                                                                     \
            if (F) { /* the k-step's 2 column tiles x RB row blocks of the fragment image, one per wave */                       \
                const bf16_t* X = reinterpret_cast<const bf16_t*>(stage + (BWD ? S_BYTES : 0));                                   \
                for (int x = wave; x < 2 * RB; x += RG_WAVES) {                                                                   \
                    const int tl = x & 1, rk = x >> 1, t = 2 * ks + tl;                                                           \
                    const size_t blk = (size_t)tile_id * RB + rk;                                                                 \
                    if (t < w_tiles && blk < nks_img) {                                                                           \
                        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(  \
                            const_cast<bf16_t*>(X) + (size_t)(32 * rk + 8 * g + q4) * RG_SSTR + 16 * tl + 4 * pp));               \
                        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(  \
                            const_cast<bf16_t*>(X) + (size_t)(32 * rk + 8 * g + 4 + q4) * RG_SSTR + 16 * tl + 4 * pp));           \
                        uint4 u;                                                                                                  \
                        u.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);                          \
                        u.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);                          \
                        u.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);                          \
                        u.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);                          \
                        F[((size_t)t * nks_img + blk) * 64 + lane] = u;                                                           \
                    }                                                                                                             \
                }                                                                                                                 \
            }                                                                                                                     \
        }                                                                                                                         \
    }

    GCNPT_STAMP(p.stamps, 2);
    for (int base = 0; base < ksteps; base += RG_PF) {       // RG_PF is even: k-step parity == slot parity
        GCNPT_RG_KSTEP(base + 0, sr0, st0)
        if (base == 4) GCNPT_STAMP(p.stamps, 8);
        GCNPT_RG_KSTEP(base + 1, sr1, st1)
    }
#undef GCNPT_RG_KSTEP
    GCNPT_STAMP(p.stamps, 3);
    if (!p.out) return;
    __syncthreads();                                          // every wave has left the ring: it becomes the out tile

    // ---- epilogue on the accumulators -> LDS out tile -> whole rows in 16-byte (8-byte) pieces, as the row-tile kernel ------------
    bf16_t* O = reinterpret_cast<bf16_t*>(stages);
    const int ostride = out_stride_dw(min(round_up(p.NOUT, 16), NCOL) / 2) * 2;       // bf16 elements
    bf16_t* out = static_cast<bf16_t*>(p.out);
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int row = (wm * MTW + i) * 16 + i16;
        const float den = rden[row], inv = rinv[row];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int tl = wn * NTW + j;
            if (tl >= n_tiles) continue;
            const int col0 = tl * 16 + g * 4;
            float v[4];
            float bq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (!BWD) {
                const float4 bv = *reinterpret_cast<const float4*>(sbias + col0);
                bq[0] = bv.x; bq[1] = bv.y; bq[2] = bv.z; bq[3] = bv.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[i][j][e];
                if (!BWD) {
                    x = div_by(x + 2.0f * bq[e], den, inv);               // gcn.py:270-271 (the bias enters twice), 390
                    x = x > 0.0f ? x : 0.0f;                              // gcn.py:392
                }
                v[e] = x;
            }
            if (!BWD && p.drop_p > 0.0f) {                                // gcn.py:393: one hash per column pair
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const unsigned dh = drop_hash(p.seed + seed_off, (unsigned)(r0 + row), (unsigned)(col0 >> 1) + h2);
                    v[2 * h2] = drop_keep(dh, 0u, p.drop_thresh16) ? v[2 * h2] * p.scale : 0.0f;
                    v[2 * h2 + 1] = drop_keep(dh, 1u, p.drop_thresh16) ? v[2 * h2 + 1] * p.scale : 0.0f;
                }
            }
            uint2 pk;
            pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            *reinterpret_cast<uint2*>(O + (size_t)row * ostride + col0) = pk;
        }
    }
    __syncthreads();
    const bf16_t* relu = BWD ? static_cast<const bf16_t*>(p.relu_src) : nullptr;
    auto store_rows = [&](auto vtag) {
        using V = decltype(vtag);                                         // uint4 or uint2
        constexpr int PER = (int)sizeof(V) / 2;
        constexpr int NW = (int)sizeof(V) / 4;
        const int pieces = p.NOUT / PER;                                  // 16 threads per row, 32 rows per round
        for (int rh = 0; rh < R; rh += RG_THREADS / 16) {
            const int row = rh + (tid >> 4), r = r0 + row;
            if (BWD && relu) {
                // hand-over to the layer below: its dZ instead of dh (gcn.py:390-393 differentiated where the rows are at hand)
                constexpr int RP = 4;
                const float f = p.next_scale / rden[row];
                const size_t rr = (size_t)min(r, p.N - 1) * p.NOUT;
                for (int pc0 = tid & 15; pc0 < pieces; pc0 += 16 * RP) {
                    V hin[RP];
#pragma unroll
                    for (int u = 0; u < RP; ++u) hin[u] = *reinterpret_cast<const V*>(relu + rr + min(pc0 + 16 * u, pieces - 1) * PER);
#pragma unroll
                    for (int u = 0; u < RP; ++u) {
                        const int pc = pc0 + 16 * u;
                        if (pc >= pieces || r >= p.N) continue;
                        V o = *reinterpret_cast<const V*>(O + (size_t)row * ostride + pc * PER);
                        unsigned* ow = reinterpret_cast<unsigned*>(&o);
                        const unsigned* hw = reinterpret_cast<const unsigned*>(&hin[u]);
#pragma unroll
                        for (int q = 0; q < NW; ++q) {
                            const float lo = bf16_to_f32((bf16_t)(hw[q] & 0xffffu)) > 0.0f ? bf16_to_f32((bf16_t)(ow[q] & 0xffffu)) * f : 0.0f;
                            const float hi = bf16_to_f32((bf16_t)(hw[q] >> 16)) > 0.0f ? bf16_to_f32((bf16_t)(ow[q] >> 16)) * f : 0.0f;
                            ow[q] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                        }
                        *reinterpret_cast<V*>(out + (size_t)r * p.NOUT + pc * PER) = o;
                    }
                }
            } else if (r < p.N) {
                for (int pc = tid & 15; pc < pieces; pc += 16)
                    *reinterpret_cast<V*>(out + (size_t)r * p.NOUT + pc * PER) = *reinterpret_cast<const V*>(O + (size_t)row * ostride + pc * PER);
            }
        }
    };
    GCNPT_STAMP(p.stamps, 4);
    if (p.vec_out == 16) store_rows(uint4{});
    else store_rows(uint2{});
    GCNPT_STAMP(p.stamps, 9);
}

template <int MODE, int WM, int MTW, int NTW, int VEC>
static int rowgemm_launch(hipStream_t s, const RowTileParams& p) {
    constexpr int R = WM * MTW * 16, NCOL = RG_WN * NTW * 16, THREADS = WM * RG_WN * 64;
    constexpr size_t S_BYTES = (size_t)R * RG_SSTR * 2, W_BYTES = (size_t)RG_WN * NTW * 64 * 16;
    constexpr size_t STAGE = (MODE != 0 ? 2 : 1) * S_BYTES + W_BYTES;
    const int ostride = out_stride_dw(std::min(round_up(p.NOUT, 16), NCOL) / 2) * 2;
    const size_t meta = (size_t)(12 * R + 4 + NCOL) * 4;
    const size_t lds = meta + std::max(2 * STAGE, (size_t)R * ostride * 2);
    if (lds > 160 * 1024) return 0;
    auto kern = rowgemm_kernel<MODE, WM, MTW, NTW, VEC>;
    GCNPT_LDS_ATTR_ONCE(kern, 160 * 1024);
    const int grid = ceil_div(p.N, R);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, s, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    note_launch(grid, THREADS, lds, sizeof(p));
    return 1;
}

template <int MODE, int VEC>
static int rowgemm_shape(hipStream_t s, const RowTileParams& p) {
    const int n_tiles = ceil_div(p.NOUT, 16);
    if (n_tiles <= 16) return rowgemm_launch<MODE, 1, 4, 4, VEC>(s, p);       // <= 256 columns: 64 rows, 4 waves
    if (n_tiles <= 20) return rowgemm_launch<MODE, 1, 4, 5, VEC>(s, p);       // <= 320 columns: 64 rows, 4 waves
    if (n_tiles <= 40) return rowgemm_launch<MODE, 2, 2, 10, VEC>(s, p);      // <= 640 columns: 64 rows, 8 waves
    return 0;
}

// 1 = launched, 0 = this form does not apply (the caller takes the row-tile kernel), < 0 = error.
// Applies to: bf16 compute with bf16 rows in and out, forward or backward-data on ready-made dZ rows, row widths that are multiples of 4
// elements, <= 640 output columns, and batches of more 32-row tiles than CUs (GCNPT_OPT_FOUR_WAVES: 2 forces it, 0 / 1 exclude it).
int rowgemm_try(hipStream_t s, const RowTileParams& p, int mode, int in_dtype, int out_dtype, int compute_dtype) {
    const int forced = option(GCNPT_OPT_FOUR_WAVES);
    if (forced == 0 || forced == 1) return 0;
    if (compute_dtype != GCNPT_BF16 || in_dtype != GCNPT_BF16 || out_dtype != GCNPT_BF16) return 0;
    if (mode != 0 && mode != 2) return 0;                                  // (the top layer's dY / Y form keeps the row-tile kernel)
    if (forced != 2 && ceil_div(p.N, ROWS) <= 256) return 0;
    if (p.vec_in != 8 && p.vec_in != 4) return 0;
    if (p.out && p.vec_out != 16 && p.vec_out != 8) return 0;
    if (mode == 0) return p.vec_in == 8 ? rowgemm_shape<0, 8>(s, p) : rowgemm_shape<0, 4>(s, p);
    return p.vec_in == 8 ? rowgemm_shape<1, 8>(s, p) : rowgemm_shape<1, 4>(s, p);
}

}  // namespace gcnpt
