// Weight gradient of a BIG batch (>= 512 k-steps, i.e. >= 16 k rows): dW += dZ^T S from the two fragment images, with the fragments of a
// k-step SHARED by the 8 waves of a workgroup through LDS.
//
// Why a second form (the first: weight_grad_body, wgrad_common.h): there every WAVE fetches the operands of its own (4 x 6)-tile block --
// 10 fragments (10 KB) per 24 MFMAs -- and at 38 400 / 102 400 rows the launch runs at what the L2s can deliver into the CUs' L1s
// (measured 16 TB/s chip-wide, 134 k of its 144 k cycles in the k loop at B = 1024).  Here a workgroup owns a (WM MBW) x (WN NBW)-tile block
// (64 ... 128 tiles); a k-step's WM MBW + WN NBW fragments cross the L1 once per WORKGROUP, each wave fetching an eighth of them, and meet
// in a two-slot LDS ring from which each wave reads the MBW + NBW it needs: 16 ... 24 KB per 64 ... 128 MFMAs, ~2x fewer bytes per MFMA.
// Bigger blocks would cut the traffic further but every workgroup adds its whole block to dW with float atomics at the end (~250 atomics
// per clock chip-wide): at 256 workgroups the sum of the two terms is flat between 64 and 96 tiles per block and rises beyond.
//
// Pipeline, per k-step i: request the fragments of k-step i+4 (registers, four sets) -- MFMAs on slot i % 2 -- park the registers of
// k-step i+1 (requested three k-steps ago: `s_waitcnt vmcnt(3 LPW)`, hipcc's own count) in slot (i+1) % 2 -- `s_waitcnt lgkmcnt(0)` + raw
// `s_barrier`.  No wait ever drains the queue (__syncthreads() would).  Register staging rather
// than LDS-DMA loads: with `global_load_lds` in flight hipcc puts `s_waitcnt vmcnt(0)` in front of every ds_read of the ring (it cannot
// tell the slots apart), which serialises request and use -- seen in the ISA of the first version of this file.
// No cross-wave reduction: a wave owns its output tiles, the contraction is split over WORKGROUPS (slices, XCD-aware like the first
// form) and meets in the float atomics.
#pragma once
#include "wgrad_common.h"

namespace gcnpt {

constexpr int WGB_NST = 2;                                     // LDS slots: one being read, one being filled
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));  // (first-class vectors stay in registers across the loop; uint4 structs may not)

template <typename CT, int WM, int MBW, int NBW>
__global__ __launch_bounds__(512, 2) void weight_grad_block_kernel(const WeightGradMulti mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wgb_smem[];
    constexpr int WN = 8 / WM, BM = WM * MBW, BN = WN * NBW, NF = BM + BN, LPW = (NF + 7) / 8;
    int layer = 0;
#pragma unroll
    for (int i = 1; i < WG_MAX_LAYERS; ++i) layer += (i < mp.n && (int)blockIdx.x >= mp.first[i]) ? 1 : 0;
    const WeightGradParams& p = mp.l[layer];
    const int id = (int)blockIdx.x - mp.first[layer];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (slice, output block): a contraction slice is pinned to one XCD group, the slices an XCD takes are the k-steps its row tiles
    // wrote (see weight_grad_body)
    const int xg = id & 7, rest = id >> 3;
    int slice, blk;
    if ((p.slices & 7) == 0) { const int sp = p.slices >> 3; slice = xg * sp + rest % sp; blk = rest / sp; }
    else                     { const int gp = 8 / p.slices;  slice = xg / gp;             blk = rest * gp + xg % gp; }
    if (blk >= p.mb * p.nb) return;
    const int bm = blk % p.mb, bn = blk / p.mb;
    const int m0 = bm * BM, n0 = bn * BN;
    const int ks_lo = slice * p.ks_per_wg, ks_hi = min(p.nks, ks_lo + p.ks_per_wg);
    const int n_ks = ks_hi - ks_lo;
    if (n_ks <= 0) return;
    const int wm = wave / WN, wn = wave % WN;
    // db = 2 sum of dZ over the rows: every block of a block row reads the same dZ fragments, so block column bn takes the k-steps
    // with ks % nb == bn, and among the WN waves that read a fragment wave wn takes the fragments x % WN == wn (all of it in the
    // bn = 0 blocks' wn = 0 waves made those workgroups' k loop 1.75x longer than everybody else's: 113 k against 66 k cycles)
    const bool want_db = p.db != nullptr;

    GCNPT_STAMP_REAL(p.stamps);
    GCNPT_STAMP(p.stamps, 11);
    u32x4_t* ring = reinterpret_cast<u32x4_t*>(wgb_smem);       // [2][NF][64]
    // this wave's share of a k-step's fragments: f = wave, wave + 8, ... (past NF: the last one again -- same bytes into the same place,
    // so that no load sits behind a condition)
    const u32x4_t* gsrc[LPW];
    int fslot[LPW];
#pragma unroll
    for (int u = 0; u < LPW; ++u) {
        const int f = min(wave + 8 * u, NF - 1);
        fslot[u] = f * 64 + lane;
        gsrc[u] = reinterpret_cast<const u32x4_t*>(f < BM ? p.zf + (size_t)min(m0 + f, p.m_tiles - 1) * p.nks * 64
                                                           : p.sf + (size_t)min(n0 + f - BM, p.n_tiles - 1) * p.nks * 64) + lane;
    }
    auto request = [&](int ks, u32x4_t (&r)[LPW]) {              // (a k-step past the end: the last one again, requested and dropped)
        const size_t koff = (size_t)min(ks, p.nks - 1) * 64;
#pragma unroll
        for (int u = 0; u < LPW; ++u) r[u] = gsrc[u][koff];
    };
    auto park = [&](const u32x4_t (&r)[LPW], int slot) {
#pragma unroll
        for (int u = 0; u < LPW; ++u) ring[slot * (NF * 64) + fslot[u]] = r[u];
    };
    auto lds_barrier = [&]() {
        asm volatile("" ::: "memory");                             // (s_barrier alone does not order LDS accesses for the compiler)
        __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): my LDS writes have landed; the loads stay in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    f32x4_t acc[MBW][NBW];
    float dbp[MBW];
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        dbp[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[i][j] = (f32x4_t){0, 0, 0, 0};
    }
    auto matrix = [&](int slot, bool db_step) {
        const u32x4_t* st = ring + slot * (NF * 64) + lane;
        u32x4_t a[MBW], b[NBW];
#pragma unroll
        for (int x = 0; x < MBW; ++x) a[x] = st[(wm * MBW + x) * 64];
#pragma unroll
        for (int j = 0; j < NBW; ++j) b[j] = st[(BM + wn * NBW + j) * 64];
#pragma unroll
        for (int x = 0; x < MBW; ++x) {
            if (db_step && x % WN == wn) dbp[x] += frag_sum<CT>(__builtin_bit_cast(uint4, a[x]));
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                if constexpr (sizeof(CT) == 2) {
                    acc[x][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[x]), __builtin_bit_cast(bf16x8_t, b[j]), acc[x][j], 0, 0, 0);
                } else {
                    const f32x4_t af = __builtin_bit_cast(f32x4_t, a[x]), bf = __builtin_bit_cast(f32x4_t, b[j]);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[x][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], acc[x][j], 0, 0, 0);
                }
            }
        }
    };

    // four register sets rotate (the loop is unrolled by four so that all are named statically): the fragments of three k-steps
    // (54 ... 72 KB per CU) are on their way while a fourth is on the matrix cores -- with two sets the launch ran at the loads' latency
    // (1.7 k cycles per k-step at the C5 shape), not at the rate the L1 path takes them in
    u32x4_t r0[LPW], r1[LPW], r2[LPW], r3[LPW];
    request(ks_lo, r0);
    request(ks_lo + 1, r1);
    request(ks_lo + 2, r2);
    request(ks_lo + 3, r3);
    park(r0, 0);
    lds_barrier();
    int i = 0, dbc = want_db ? (ks_lo + p.nb - bn) % p.nb : 1;     // (ks - bn) mod nb of the k-step on the matrix cores; 0 = a db step
    auto db_next = [&]() { const bool d = dbc == 0 && want_db; dbc = dbc + 1 == p.nb ? 0 : dbc + 1; return d; };
    for (; i + 4 <= n_ks; i += 4) {                                // (no exit inside the body: a path that skips a `park` would make hipcc
        request(ks_lo + i + 4, r0);                                //  protect the unparked registers with a full wait at the loop head)
        matrix(0, db_next());                                      // k-step i: slot 0 on the matrix cores, r1 (k-step i+1) -> slot 1
        park(r1, 1);
        lds_barrier();
        request(ks_lo + i + 5, r1);
        matrix(1, db_next());
        park(r2, 0);
        lds_barrier();
        request(ks_lo + i + 6, r2);
        matrix(0, db_next());
        park(r3, 1);
        lds_barrier();
        request(ks_lo + i + 7, r3);
        matrix(1, db_next());
        park(r0, 0);
        lds_barrier();
    }
    const int rem = n_ks - i;                                      // 0..3 k-steps left: k-step i sits in slot 0, i+1 / i+2 in r1 / r2
    if (rem >= 1) matrix(0, db_next());
    if (rem >= 2) {
        park(r1, 1);
        lds_barrier();
        matrix(1, db_next());
    }
    if (rem >= 3) {
        park(r2, 0);
        lds_barrier();
        matrix(0, db_next());
    }

    GCNPT_STAMP(p.stamps, 12);
    // every wave adds its own tiles (the slices of other workgroups meet them in the same accumulators)
#pragma unroll
    for (int x = 0; x < MBW; ++x) {
        const int mt = m0 + wm * MBW + x;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int nt = n0 + wn * NBW + j;
            if (mt >= p.m_tiles || nt >= p.n_tiles) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = mt * 16 + (lane >> 4) * 4 + g;
                const int n = nt * 16 + (lane & 15);
                if (m < p.H && n < p.Din) atomicAdd(p.dW + (size_t)m * p.Din + n, acc[x][j][g]);
            }
        }
        if (want_db && x % WN == wn) {
            float v = dbp[x];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int m = mt * 16 + lane;
            if (lane < 16 && mt < p.m_tiles && m < p.H) atomicAdd(p.db + m, 2.0f * v);      // bias enters twice
        }
    }
    GCNPT_STAMP(p.stamps, 14);
}

inline size_t weight_grad_block_lds(int wm, int mbw, int nbw) { return (size_t)WGB_NST * (wm * mbw + (8 / wm) * nbw) * 64 * 16; }

}  // namespace gcnpt
