// Weight gradient dW += dZ^T S, db += 2 sum_r dZ[r,:] (reference model/gcn.py:270-271 differentiated) from ROW-MAJOR dZ rows and
// the forward's S fragment image: the device code shared by the stand-alone launch (gcnpt_layer_bwd_weight_rows[_multi]) and by the
// backward-data launches that carry weight gradients as a side job (rowtile_body.h).
//
// Why a second weight-gradient kernel (wgrad_common.h is the first): there every WAVE owned a whole (4 x 6)-tile block for a share of the
// k-steps, fetched its own fragments (0.42 KB through the CU's L1 path per MFMA) and the 8 waves then met in LDS and left through float
// atomics -- stamps at C2: 5.4 k cycles of streaming, 3.0 k of LDS hand-over, 6.3 k of reduction + atomics per unit.  Here a WORKGROUP owns
// a (<= 16 x NB)-tile block, every wave its own (<= 4 x NW) tiles of it for ALL k-steps of the unit (no reduction), and the operands of
// a k-step go through LDS once per workgroup: the dZ rows as they lie in memory (read transposed with ds_read_b64_tr_b16: no dZ fragment
// image is written by anybody), the S fragments from the forward's image.  <= 20 KB per k-step per workgroup feed <= 128 MFMAs.
#pragma once
#include "layer_common.h"

namespace gcnpt {

constexpr int WR_WAVES = 8, WR_THREADS = WR_WAVES * WAVE;
constexpr int WR_WM = 4, WR_WN = 2;          // wave grid over the block: wave (wm, wn) owns m-tiles wm, wm+4, ... and n-tiles wn*NW .. wn*NW+NW-1
constexpr int WR_MW = 4;                      // m-tiles per wave (<= 16 m-tiles per block)
constexpr int WR_NW = 4;                      // n-tiles per wave (<= 8 n-tiles per block)

struct WgradRowsParams {
    const void* dz;          // [N, H] row-major in the compute type's storage (bf16 / f32): dZ -- or, masked, dY
    const void* yref;        // masked: the layer's stored output Y [N, H]
    const int32_t* d_ell;    // masked: ELL head whose [8r] is deg(r)
    float scale;             // masked: 1/(1-p) of the dropout applied to Y:   dZ = dY * 1[Y > 0] * scale / (deg + 1)
    const uint4* sf;         // fragment image of S = (A+I) h  [n_tiles][nks][64]  (include/gcnpt.h)
    float* dW; float* db;
    int N, H, Din, m_tiles, n_tiles, nks;
    int n_mblocks, mb_tiles; // m-blocks of mb_tiles (<= 16) m-tiles
    int n_nblocks, nb_tiles; // n-blocks of nb_tiles (<= 8) n-tiles
    int slices, ks_per_unit; // contraction slices: units = n_mblocks * n_nblocks * slices, unit = (slice * n_nblocks + nblk) * n_mblocks + mblk
    int masked, vec;         // vec: 8 = rows read 16 bytes at a time, 4 = in 8-byte (bf16) / 16-byte (f32) halves
    unsigned long long* stamps;
};

// dynamic LDS of a workgroup: two stages of (dZ row tile + S fragments)
__host__ __device__ inline int wr_zstride(int es) { return lds_stride_dw(WR_MW * WR_WM * 16 * es / 4) * 4 / es; }       // elements
inline size_t wgrad_rows_lds(int compute_dtype) {
    const int es = compute_dtype == GCNPT_BF16 ? 2 : 4, rows = compute_dtype == GCNPT_BF16 ? 32 : 16;
    return 2 * ((size_t)rows * wr_zstride(es) * es + (size_t)WR_WN * WR_NW * 64 * 16);
}

// One unit: block (mblk, nblk) of dW over the k-steps of one slice.  All 512 threads of the workgroup.
template <typename CT, int VEC, bool MASKED>
__device__ __forceinline__ void wgrad_rows_unit(const WgradRowsParams& p, const int unit, unsigned char* smem) {
    constexpr int KROWS = sizeof(CT) == 2 ? 32 : 16;                    // rows per k-step
    constexpr int ZSTR = (sizeof(CT) == 2) ? 272 : 264;                 // == wr_zstride(sizeof(CT)): 256 payload elements, stride == 8 (mod 16) dwords
    constexpr size_t Z_BYTES = (size_t)KROWS * ZSTR * sizeof(CT), S_BYTES = (size_t)WR_WN * WR_NW * 64 * 16, STAGE = Z_BYTES + S_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & (WR_WM - 1), wn = wave >> 2;
    const int mblk = unit % p.n_mblocks, rest = unit / p.n_mblocks;
    const int nblk = rest % p.n_nblocks, slice = rest / p.n_nblocks;
    if (slice >= p.slices) return;
    const int m0 = mblk * p.mb_tiles, mt = min(p.mb_tiles, p.m_tiles - m0);          // m-tiles of this block
    const int n0 = nblk * p.nb_tiles, nt = min(p.nb_tiles, p.n_tiles - n0);
    const int ks_lo = slice * p.ks_per_unit, ks_hi = min(p.nks, ks_lo + p.ks_per_unit);
    const int c0 = m0 * 16;                                                             // first dZ column of the block
    const int ncols = min(p.H - c0, mt * 16);
    const bool want_db = nblk == 0 && p.db != nullptr;
    typedef typename std::conditional<sizeof(CT) == 2, bf16_t, float>::type IT;       // storage type of the rows == compute type
    const IT* dz = static_cast<const IT*>(p.dz);
    const IT* yref = static_cast<const IT*>(p.yref);
    GCNPT_STAMP(p.stamps, 11);

    f32x4_t acc[WR_MW][WR_NW];
    float dbp[WR_MW];
#pragma unroll
    for (int i = 0; i < WR_MW; ++i) {
        dbp[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < WR_NW; ++j) acc[i][j] = (f32x4_t){0, 0, 0, 0};
    }

    // ---- staging: the k-step's dZ rows (KROWS x <= 256 columns: 32 chunk slots of 8 elements per row) and its <= 8 S fragments.
    //      PF k-steps are in flight in registers (a k-step's MFMAs take ~500 cycles, a loaded round trip ~2000), two live in LDS.
    constexpr int ZI = KROWS * 32 / WR_THREADS;                          // chunk items per thread: 2 (bf16) / 1 (f32)
    constexpr int PF = 4;
    constexpr int ZW = sizeof(CT) == 2 ? 1 : 2;                          // 16-byte words per chunk
    struct StageRegs { uint4 z[ZI][ZW]; uint4 y[MASKED ? ZI : 1][ZW]; int deg[MASKED ? ZI : 1]; };
    uint4 ss0, ss1, ss2, ss3;                                            // the S fragment of each slot
    StageRegs sr0, sr1, sr2, sr3;                                        // (named, not an array: hipcc left an indexed ring in scratch memory)
    static_assert(PF == 4, "the ring below is written out");
    const int kmax8 = VEC == 8 ? p.H - 8 : p.H - 4;
    auto ld8 = [&](const IT* base, size_t row, int k0c, raw8<IT>& dst) {
        if constexpr (VEC == 8) issue8<IT, true>(base, row, p.H, k0c, dst);
        else issue8_half<IT>(base, row, p.H, k0c, dst);
    };
    // No load below sits behind a condition (hipcc counts outstanding loads only along straight-line code: one conditional load and
    // every later wait drains the whole queue, i.e. the prefetch distance collapses to one k-step).  A k-step past the unit's end
    // still issues its loads, all lanes at ONE address (16 bytes through the CU's L1 path instead of 20 KB).
    auto stage_load = [&](int ks, StageRegs& q, uint4& qs) {
        const bool on = ks < ks_hi;
#pragma unroll
        for (int u = 0; u < ZI; ++u) {
            const int it = u * WR_THREADS + tid, row = it >> 5, ch = it & 31;
            const size_t r = on ? (size_t)min(ks * KROWS + row, p.N - 1) : (size_t)0;
            const int k0c = on ? min(c0 + 8 * ch, kmax8) : 0;            // clamped: valid memory whatever the slot
            raw8<IT> t;
            ld8(dz, r, k0c, t);
            q.z[u][0] = t.a;
            if constexpr (ZW == 2) q.z[u][1] = t.b;
            if constexpr (MASKED) {
                raw8<IT> ty;
                ld8(yref, r, k0c, ty);
                q.y[u][0] = ty.a;
                if constexpr (ZW == 2) q.y[u][1] = ty.b;
                q.deg[u] = p.d_ell[r * 8];
            }
        }
        // S fragments: wave w fetches fragment w of the block (<= 8 per k-step)
        const int j = min(wave, nt - 1);
        qs = p.sf[on ? ((size_t)(n0 + j) * p.nks + min(ks, p.nks - 1)) * 64 + lane : (size_t)0];
    };
    auto stage_store = [&](int ks, const StageRegs& q, const uint4& qs, unsigned char* stage) {
        CT* Z = reinterpret_cast<CT*>(stage);
        uint4* SF = reinterpret_cast<uint4*>(stage + Z_BYTES);
#pragma unroll
        for (int u = 0; u < ZI; ++u) {
            const int it = u * WR_THREADS + tid, row = it >> 5, ch = it & 31;
            const int col = 8 * ch;                                       // column inside the block
            // a slot is live when its row exists and its first column does; a chunk that straddles the end of H (H % 8 == 4) keeps
            // its first 4 columns (the VEC == 4 loads clamp the upper half inside the row, see issue8_half; the rest is zeroed below)
            const bool live = ks < ks_hi && ks * KROWS + row < p.N && col < ncols;
            if constexpr (!MASKED && sizeof(CT) == 2 && VEC == 8) {       // bf16 rows into a bf16 tile: the 16 bytes as they are
                *reinterpret_cast<uint4*>(Z + (size_t)row * ZSTR + col) = live ? q.z[u][0] : make_uint4(0, 0, 0, 0);
                continue;
            }
            float v[8];
            raw8<IT> t;
            t.a = q.z[u][0];
            if constexpr (ZW == 2) t.b = q.z[u][1];
            unpack8<IT>(t, live, v);
            if constexpr (MASKED) {
                float y[8];
                raw8<IT> ty;
                ty.a = q.y[u][0];
                if constexpr (ZW == 2) ty.b = q.y[u][1];
                unpack8<IT>(ty, live, y);
                const float inv = p.scale / (float)(q.deg[u] + 1);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (y[e] > 0.0f) ? v[e] * inv : 0.0f;
            }
            if constexpr (VEC != 8) {                                     // zero what lies past H inside a straddling chunk
#pragma unroll
                for (int e = 4; e < 8; ++e) v[e] = (c0 + col + e < p.H) ? v[e] : 0.0f;
            }
            tile<CT>::put8(Z + (size_t)row * ZSTR + col, v);
        }
        if (wave < nt) SF[wave * 64 + lane] = qs;
    };

    unsigned char* st0 = smem;
    unsigned char* st1 = smem + STAGE;
    stage_load(ks_lo + 0, sr0, ss0);
    stage_load(ks_lo + 1, sr1, ss1);
    stage_load(ks_lo + 2, sr2, ss2);
    stage_load(ks_lo + 3, sr3, ss3);
    GCNPT_STAMP(p.stamps, 12);

    const int i16 = lane & 15, g = lane >> 4, q4 = i16 >> 2, pp = i16 & 3;
#define GCNPT_WR_KSTEP(KS, q, qs, stage)                                                                                        \
    { /* every k-step of a round runs: one past the unit's end stages zero rows (ks_per_unit % PF == 0, or ks_hi == nks) */  \
        const int ks = (KS);                                                                                                 \
        stage_store(ks, q, qs, stage); \
        stage_load(ks + PF, q, qs); \
        __syncthreads(); \
        const CT* Z = reinterpret_cast<const CT*>(stage); \
        const uint4* SF = reinterpret_cast<const uint4*>(stage + Z_BYTES); \
        uint4 b[WR_NW]; \
_Pragma("unroll") \
        for (int j = 0; j < WR_NW; ++j) b[j] = SF[min(wn * WR_NW + j, WR_WN * WR_NW - 1) * 64 + lane]; \
_Pragma("unroll") \
        for (int i = 0; i < WR_MW; ++i) { \
            const int t = wm + WR_WM * i; \
            if (t >= mt) continue; \
            uint4 a; \
            if constexpr (sizeof(CT) == 2) { \
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16( \
                    (__attribute__((address_space(3))) s16x4_t*)(const_cast<CT*>(Z) + (size_t)(8 * g + q4) * ZSTR + 16 * t + 4 * pp)); \
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16( \
                    (__attribute__((address_space(3))) s16x4_t*)(const_cast<CT*>(Z) + (size_t)(8 * g + 4 + q4) * ZSTR + 16 * t + 4 * pp)); \
                a.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16); \
                a.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16); \
                a.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16); \
                a.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16); \
            } else { \
                a.x = __float_as_uint(Z[(size_t)(4 * g + 0) * ZSTR + 16 * t + i16]); \
                a.y = __float_as_uint(Z[(size_t)(4 * g + 1) * ZSTR + 16 * t + i16]); \
                a.z = __float_as_uint(Z[(size_t)(4 * g + 2) * ZSTR + 16 * t + i16]); \
                a.w = __float_as_uint(Z[(size_t)(4 * g + 3) * ZSTR + 16 * t + i16]); \
            } \
            if (want_db && wn == 0) { \
                if constexpr (sizeof(CT) == 2) { \
                    dbp[i] += (__uint_as_float(a.x << 16) + __uint_as_float(a.x & 0xffff0000u)) + (__uint_as_float(a.y << 16) + __uint_as_float(a.y & 0xffff0000u)) + \
                              (__uint_as_float(a.z << 16) + __uint_as_float(a.z & 0xffff0000u)) + (__uint_as_float(a.w << 16) + __uint_as_float(a.w & 0xffff0000u)); \
                } else { \
                    dbp[i] += (__uint_as_float(a.x) + __uint_as_float(a.y)) + (__uint_as_float(a.z) + __uint_as_float(a.w)); \
                } \
            } \
_Pragma("unroll") \
            for (int j = 0; j < WR_NW; ++j) { \
                if (wn * WR_NW + j >= nt) continue; \
                if constexpr (sizeof(CT) == 2) { \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b[j]), acc[i][j], 0, 0, 0); \
                } else { \
                    const f32x4_t af = __builtin_bit_cast(f32x4_t, a), bf = __builtin_bit_cast(f32x4_t, b[j]); \
_Pragma("unroll") \
                    for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[e], acc[i][j], 0, 0, 0); \
                } \
            } \
        } \
    }
    for (int base = ks_lo; base < ks_hi; base += PF) {                   // PF is even: k-step parity == slot parity
        GCNPT_WR_KSTEP(base + 0, sr0, ss0, st0)
        GCNPT_WR_KSTEP(base + 1, sr1, ss1, st1)
        GCNPT_WR_KSTEP(base + 2, sr2, ss2, st0)
        GCNPT_WR_KSTEP(base + 3, sr3, ss3, st1)
    }
#undef GCNPT_WR_KSTEP
    GCNPT_STAMP(p.stamps, 13);

    // ---- every wave adds its own tiles: no cross-wave reduction (slices of different workgroups meet in the float atomics) ----
#pragma unroll
    for (int i = 0; i < WR_MW; ++i) {
        const int t = wm + WR_WM * i;
        if (t >= mt) continue;
#pragma unroll
        for (int j = 0; j < WR_NW; ++j) {
            const int jn = wn * WR_NW + j;
            if (jn >= nt) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = (m0 + t) * 16 + g * 4 + q;
                const int n = (n0 + jn) * 16 + i16;
                if (m < p.H && n < p.Din) atomicAdd(p.dW + (size_t)m * p.Din + n, acc[i][j][q]);
            }
        }
        if (want_db && wn == 0) {
            float v = dbp[i];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int m = (m0 + t) * 16 + lane;
            if (lane < 16 && m < p.H) atomicAdd(p.db + m, 2.0f * v);     // bias enters twice (gcn.py:270-271)
        }
    }
    GCNPT_STAMP(p.stamps, 14);
}

// block / slice plan of one layer for `budget` workgroup-units (host side, rowtile_kernels.hip)
int plan_wgrad_rows(WgradRowsParams& p, const void* dz, const void* yref, const int32_t* d_ell, float scale, int masked, int rows_dtype,
                    const void* s_frag, long long N, int Din, int H, float* dW, float* db, int compute_dtype, int budget, int min_ks_per_unit);
// whether rows of `H` elements of dtype at base a (and b) can be staged by this kernel: 8 (16-byte chunks), 4 (half chunks) or 0 (no)
int wgrad_rows_vec(int H, int dtype, const void* a, const void* b);

constexpr int WR_MAX_LAYERS = 8;
struct WgradRowsMulti {
    WgradRowsParams l[WR_MAX_LAYERS];
    int first[WR_MAX_LAYERS + 1];     // units [first[i], first[i+1]) belong to l[i]
    int n;
};

}  // namespace gcnpt
