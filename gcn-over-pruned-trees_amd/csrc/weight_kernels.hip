// Weight-side kernels (gfx950): nn.Linear weights -> MFMA fragment order, and the weight/bias gradient
// dW += dZ^T ((A+I) h), db += 2 sum dZ of reference model/gcn.py:270-271 (autograd).
#include "layer_common.h"
#include "pack_common.h"

namespace gcnpt {


template <typename CT>
__global__ void pack_weights_kernel(const PackParams p) {
    pack_fragments<CT>(p, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

// ---------------------------------------------------------------------------------------------------
// backward-weight: dW[H,Din] += dZ^T S,  db[H] += 2 sum_r dZ[r,:]      (S = (A+I) h)
//
// Both operands arrive as fragment images (include/gcnpt.h) written by the row-tile kernels, so every
// operand fetch is one fully coalesced 1-KiB wave load straight into registers: no LDS, no transposes,
// no CSR.  A workgroup owns a (4 x 3)-tile block of dW and one slice of the contraction (row) range;
// its 4 waves take every 4th k-step of the slice, issue up to WG_KB k-steps of loads at once, and meet
// in LDS at the end; the slices of different workgroups are combined with float atomics.
// ---------------------------------------------------------------------------------------------------
constexpr int WG_MT = 4, WG_NT = 3, WG_KB = 5;

struct WeightGradParams {
    const uint4* zf;     // fragment image of dZ  [m_tiles][nks][64]
    const uint4* sf;     // fragment image of S   [n_tiles][nks][64]
    float* dW; float* db;
    int H, Din, m_tiles, n_tiles, nks, ks_per_wg;
    int mb, nb, slices;  // block grid; the launch grid is 1-D so that the block -> (m, n, slice) map can follow the XCDs
    unsigned long long* stamps;   // diagnostic builds only
    int knob;
};

template <typename CT>
__device__ __forceinline__ float frag_sum(const uint4& u) {
    if constexpr (sizeof(CT) == 2) {
        return (__uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u)) + (__uint_as_float(u.y << 16) + __uint_as_float(u.y & 0xffff0000u)) +
               (__uint_as_float(u.z << 16) + __uint_as_float(u.z & 0xffff0000u)) + (__uint_as_float(u.w << 16) + __uint_as_float(u.w & 0xffff0000u));
    } else {
        return (__uint_as_float(u.x) + __uint_as_float(u.y)) + (__uint_as_float(u.z) + __uint_as_float(u.w));
    }
}

// One launch serves the weight gradients of several layers (they all become computable at the end of the backward
// sweep and each alone fills at most one workgroup per CU): layer l owns blocks [first[l], first[l+1]), first[l] % 8 == 0.
constexpr int WG_MAX_LAYERS = 8;
struct WeightGradMulti {
    WeightGradParams l[WG_MAX_LAYERS];
    int first[WG_MAX_LAYERS + 1];
    int n;
};

// WG_WAVES waves split a workgroup's k-steps: 4 when a layer has the launch to itself (its slices are short), 8 when several
// layers share the CUs (twice the k-steps per workgroup: 8 waves still take them in ONE batch of loads each)
// Block shape: (WG_MT x NT) output tiles per workgroup, KB k-steps of loads in flight per wave.  (4 x 3, 5) for small batches (one
// latency chain per CU: as many loads in flight as the registers hold); (4 x 6, 2) for big ones, where the fragment stream through
// the CU's L1 path is the bound: 10 fragment loads feed 24 MFMAs instead of 7 feeding 12.
template <typename CT, int WG_WAVES, int NT, int KB>
__global__ __launch_bounds__(WG_WAVES * WAVE, 2) void weight_grad_kernel(const WeightGradMulti mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
    constexpr int RT = WG_MT * NT, RH = RT > 12 ? RT / 2 : RT;                             // tiles reduced per LDS round (12 KiB per wave each)
    typedef f32x4_t RedTile[RH][WAVE];
    RedTile* red = reinterpret_cast<RedTile*>(wg_smem);                                   // [WG_WAVES] per-wave partial tiles
    typedef float DbTile[WG_MT][16];
    DbTile* dbred = reinterpret_cast<DbTile*>(wg_smem + sizeof(RedTile) * WG_WAVES);     // [WG_WAVES]

    int layer = 0;
#pragma unroll
    for (int i = 1; i < WG_MAX_LAYERS; ++i) layer += (i < mp.n && (int)blockIdx.x >= mp.first[i]) ? 1 : 0;
    const WeightGradParams& p = mp.l[layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups are dealt round-robin over the 8 XCDs (id % 8 labels the XCD group), each with its own L2.
    // All blocks of one contraction slice read the same rows of both images, so a slice is pinned to one
    // XCD group: its rows cross the fabric once and every other read hits that XCD's L2.  (Speed only.)
    const int id = (int)blockIdx.x - mp.first[layer], xg = id & 7, rest = id >> 3;
    int slice, blk;
    if ((p.slices & 7) == 0) { const int sp = p.slices >> 3; slice = xg + 8 * (rest % sp); blk = rest / sp; }
    else                     { const int gp = 8 / p.slices;  slice = xg % p.slices;       blk = rest * gp + xg / p.slices; }
    if (blk >= p.mb * p.nb) return;
    const int bm = blk % p.mb, bn = blk / p.mb;
    const int m0 = bm * WG_MT, n0 = bn * NT;
    const int ks_lo = slice * p.ks_per_wg, ks_hi = min(p.nks, ks_lo + p.ks_per_wg);
    const bool want_db = bn == 0 && p.db != nullptr;
    GCNPT_STAMP_REAL(p.stamps);
    GCNPT_STAMP(p.stamps, 0);

    f32x4_t acc[WG_MT][NT];
    float dbp[WG_MT];
#pragma unroll
    for (int i = 0; i < WG_MT; ++i) {
        dbp[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0, 0, 0, 0};
    }

    // Every load below is unconditional (clamped indices): a load behind a runtime condition gets its own basic
    // block and an s_waitcnt vmcnt(0) from hipcc, which would turn this batch into 35 serial round trips.
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    for (int base = ks_lo + wave; base < ks_hi; base += WG_WAVES * KB) {
        uint4 a[KB][WG_MT], b[KB][NT];
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            int ks = min(base + WG_WAVES * u, p.nks - 1);
#ifdef GCNPT_STAMPS
            if (p.knob & 2) ks = 0;                                       // experiment: every load hits the same lines
#endif
#pragma unroll
            for (int i = 0; i < WG_MT; ++i) a[u][i] = p.zf[((size_t)min(m0 + i, p.m_tiles - 1) * p.nks + ks) * 64 + lane];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[u][j] = p.sf[((size_t)min(n0 + j, p.n_tiles - 1) * p.nks + ks) * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const bool live = base + WG_WAVES * u < ks_hi;                       // past the slice: contributes zeros
#pragma unroll
            for (int i = 0; i < WG_MT; ++i) {
                const uint4 av = live ? a[u][i] : zero4;
                // db on the VALU, in every block.  Tried and measured slower: the column sums as one more MFMA against a fragment
                // of ones (+3.4 us: hipcc then schedules for occupancy and no longer keeps the batch of loads in flight), and
                // summing only in the blocks that store db (a branch here makes every wait in the batch a full drain, +1.8 us)
                dbp[i] += frag_sum<CT>(av);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (sizeof(CT) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av),
                                                                             __builtin_bit_cast(bf16x8_t, b[u][j]), acc[i][j], 0, 0, 0);
                    } else {
                        const f32x4_t af = __builtin_bit_cast(f32x4_t, av), bf = __builtin_bit_cast(f32x4_t, b[u][j]);
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    }

    GCNPT_STAMP(p.stamps, 1);
    // waves meet in LDS (RH tiles per round); wave w then owns tiles w, w + WG_WAVES, ... of the round
    if (want_db) {
#pragma unroll
        for (int i = 0; i < WG_MT; ++i) {
            float v = dbp[i];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) dbred[wave][i][lane] = v;
        }
    }
#pragma unroll
    for (int r0t = 0; r0t < RT; r0t += RH) {
        if (r0t > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < WG_MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int tt = i * NT + j;                                   // compile-time after unrolling
                if (tt >= r0t && tt < r0t + RH) red[wave][tt - r0t][lane] = acc[i][j];
            }
        __syncthreads();
        if (r0t == 0) GCNPT_STAMP(p.stamps, 2);
        for (int tl = wave; tl < RH; tl += WG_WAVES) {
            const int tt = r0t + tl;
            const int i = tt / NT, j = tt - i * NT;
            if (m0 + i >= p.m_tiles || n0 + j >= p.n_tiles) continue;
            f32x4_t v = red[0][tl][lane];
#pragma unroll
            for (int w = 1; w < WG_WAVES; ++w) v += red[w][tl][lane];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = (m0 + i) * 16 + (lane >> 4) * 4 + g;
                const int n = (n0 + j) * 16 + (lane & 15);
#ifdef GCNPT_STAMPS
                if (p.knob & 4) { if (m < p.H && n < p.Din) p.dW[(size_t)m * p.Din + n] = v[g]; continue; }     // experiment: stores for atomics
#endif
                if (m < p.H && n < p.Din) atomicAdd(p.dW + (size_t)m * p.Din + n, v[g]);
            }
        }
    }
    if (want_db && tid < WG_MT * 16) {
        const int i = tid >> 4, c = tid & 15;
        const int m = (m0 + i) * 16 + c;
        if (m0 + i < p.m_tiles && m < p.H)
        {
            float sdb = dbred[0][i][c];
#pragma unroll
            for (int w = 1; w < WG_WAVES; ++w) sdb += dbred[w][i][c];
            atomicAdd(p.db + m, 2.0f * sdb);                                 // bias enters twice
        }
    }
    GCNPT_STAMP(p.stamps, 3);
}

}  // namespace gcnpt

// =====================================================================================================
// C-ABI
// =====================================================================================================
using namespace gcnpt;


extern "C" size_t gcnpt_packed_bytes(int n_out, int k_in, int dtype) {
    if (n_out <= 0 || k_in <= 0 || (dtype != GCNPT_F32 && dtype != GCNPT_BF16)) return 0;
    const int ks = round_up(k_in, kstep_of(dtype)) / kstep_of(dtype);
    return (size_t)ceil_div(n_out, 16) * ks * 64 * 16;
}

namespace gcnpt {
int fill_pack_params(PackParams& p, int n_layers, const float* const* W, const int* H, const int* Din, int dtype, void* const* w_fwd,
                     void* const* w_bwd) {
    GCNPT_REQUIRE(n_layers >= 1 && n_layers <= PACK_MAX_LAYERS, "pack_weights: 1..%d layers per call", PACK_MAX_LAYERS);
    GCNPT_REQUIRE(W && H && Din && w_fwd && w_bwd, "pack_weights: null pointer");
    GCNPT_REQUIRE(dtype == GCNPT_F32 || dtype == GCNPT_BF16, "pack_weights: dtype %d", dtype);
    p = PackParams{};
    p.n_layers = n_layers;
    p.first[0] = 0;
    for (int l = 0; l < n_layers; ++l) {
        GCNPT_REQUIRE(W[l] && (w_fwd[l] || w_bwd[l]), "pack_weights: null pointer (layer %d)", l);
        GCNPT_REQUIRE(H[l] > 0 && Din[l] > 0, "pack_weights: H and Din must be positive (layer %d)", l);
        p.W[l] = W[l]; p.H[l] = H[l]; p.Din[l] = Din[l];
        p.wf[l] = static_cast<uint4*>(w_fwd[l]); p.wb[l] = static_cast<uint4*>(w_bwd[l]);
        const size_t frags = (w_fwd[l] ? gcnpt_packed_bytes(H[l], Din[l], dtype) : 0) / 16 +
                             (w_bwd[l] ? gcnpt_packed_bytes(Din[l], H[l], dtype) : 0) / 16;
        p.first[l + 1] = p.first[l] + (long long)frags;
    }
    return GCNPT_OK;
}
}  // namespace gcnpt

extern "C" int gcnpt_pack_weights_multi(void* stream, int n_layers, const float* const* W, const int* H, const int* Din,
                                        int dtype, void* const* w_fwd, void* const* w_bwd) {
    PackParams p;
    const int rc = fill_pack_params(p, n_layers, W, H, Din, dtype, w_fwd, w_bwd);
    if (rc != GCNPT_OK) return rc;
    const int grid = (int)((p.first[n_layers] + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GCNPT_BF16) hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(grid), dim3(256), 0, s, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_pack_weights(void* stream, const float* W, int H, int Din, int dtype, void* w_fwd, void* w_bwd) {
    return gcnpt_pack_weights_multi(stream, 1, &W, &H, &Din, dtype, &w_fwd, &w_bwd);
}

extern "C" size_t gcnpt_frag_bytes(int rows, int width, int dtype) {
    if (rows <= 0 || width <= 0 || !dtype_ok(dtype)) return 0;
    const size_t ksteps = (size_t)ceil_div(rows, 32) * (dtype == GCNPT_BF16 ? 1 : 2);
    return (size_t)ceil_div(width, 16) * ksteps * 64 * 16;
}

// fills p for one layer; returns the number of workgroups it takes (a multiple of 8)
static int plan_weight_grad(WeightGradParams& p, const void* z_frag, const void* s_frag, int nks, int Din, int H, float* dW, float* db,
                            int blocks_in_launch, int waves, int wg_budget, int nt) {
    p = WeightGradParams{};
    p.stamps = static_cast<unsigned long long*>(g_debug_stamps); p.knob = g_debug_knob;
    p.zf = static_cast<const uint4*>(z_frag); p.sf = static_cast<const uint4*>(s_frag);
    p.dW = dW; p.db = db; p.H = H; p.Din = Din;
    p.m_tiles = ceil_div(H, 16); p.n_tiles = ceil_div(Din, 16);
    p.nks = nks;
    const int mb = ceil_div(p.m_tiles, WG_MT), nb = ceil_div(p.n_tiles, nt);
    // split the contraction so that the launch as a whole has ~256 workgroups, one per CU (the layers of a launch share
    // them: measured 2 us faster per step than letting each layer bring 256 of its own), with at least one k-step per wave;
    // slices is 1, 2, 4 or a multiple of 8 so that each slice maps onto whole XCD groups (rounded DOWN: one more
    // workgroup than CUs costs a second round)
    // (big batches, wg_budget 768: 4-wave workgroups, three per CU, so that every CU's L1 path streams fragments -- with one slice a
    // 128 x 300-token batch ran on 135 workgroups for 121 us)
    int want = std::max(1, std::min(ceil_div(p.nks, waves), wg_budget / std::max(blocks_in_launch, mb * nb)));
    int slices = want >= 8 ? want / 8 * 8 : (want >= 4 ? 4 : (want >= 2 ? 2 : 1));
    p.ks_per_wg = ceil_div(p.nks, slices);
    p.mb = mb; p.nb = nb; p.slices = slices;
    const int per_group = (slices & 7) == 0 ? mb * nb * (slices / 8) : ceil_div(mb * nb, 8 / slices);
    return 8 * per_group;
}

template <typename CT, int NW, int NT, int KB>
static int launch_weight_grad_cfg(hipStream_t s, const WeightGradMulti& mp) {
    const size_t lds = (sizeof(f32x4_t) * 12 * WAVE + sizeof(float) * WG_MT * 16) * NW;      // 49 / 98 KiB
    GCNPT_LDS_ATTR_ONCE((weight_grad_kernel<CT, NW, NT, KB>), 160 * 1024);
    hipLaunchKernelGGL((weight_grad_kernel<CT, NW, NT, KB>), dim3(mp.first[mp.n]), dim3(NW * WAVE), lds, s, mp);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

// waves per workgroup / workgroups the launch may have: small batches are one latency chain per CU (8 waves when layers share the
// CUs); from 512 k-steps (16 k rows) on the fragment stream through the CUs' L1 paths is what counts: 4-wave workgroups, 2 per CU
static int wg_waves(int n_layers, int nks) { return (n_layers > 1 && nks < 512) ? 8 : 4; }
static int wg_budget(int nks) { return nks < 512 ? 256 : 512; }
// (4 x 3)-tile blocks of all layers of a launch: the budget is shared in proportion to them, so every workgroup gets the same k-steps
static int wg_nt(int nks) { return nks < 512 ? WG_NT : 6; }       // output tiles per block row: wide blocks for big batches (see the kernel)
static int wg_blocks(int n_layers, const int* Din, const int* H, int nt) {
    int b = 0;
    for (int l = 0; l < n_layers; ++l) b += ceil_div(ceil_div(H[l], 16), WG_MT) * ceil_div(ceil_div(Din[l], 16), nt);
    return b;
}

static int launch_weight_grad(hipStream_t s, const WeightGradMulti& mp, int compute_dtype, int waves, int nt) {
    if (compute_dtype == GCNPT_BF16) {
        if (nt == 6) return launch_weight_grad_cfg<bf16_t, 4, 6, 2>(s, mp);
        return waves == 8 ? launch_weight_grad_cfg<bf16_t, 8, WG_NT, WG_KB>(s, mp) : launch_weight_grad_cfg<bf16_t, 4, WG_NT, WG_KB>(s, mp);
    }
    if (nt == 6) return launch_weight_grad_cfg<float, 4, 6, 2>(s, mp);
    return waves == 8 ? launch_weight_grad_cfg<float, 8, WG_NT, WG_KB>(s, mp) : launch_weight_grad_cfg<float, 4, WG_NT, WG_KB>(s, mp);
}

extern "C" int gcnpt_layer_bwd_weight_multi(void* stream, int n_layers, const void* const* z_frag, const void* const* s_frag,
                                            int B, int T, const int* Din, const int* H, float* const* dW, float* const* db,
                                            int compute_dtype) {
    GCNPT_REQUIRE(n_layers >= 1 && n_layers <= WG_MAX_LAYERS, "layer_bwd_weight: 1..%d layers per call", WG_MAX_LAYERS);
    GCNPT_REQUIRE(z_frag && s_frag && Din && H && dW && db, "layer_bwd_weight: null pointer");
    GCNPT_REQUIRE(B > 0 && T >= 0, "layer_bwd_weight: sizes must be positive");
    GCNPT_REQUIRE(dtype_ok(compute_dtype), "layer_bwd_weight: bad dtype");
    const int nks = ceil_div((int)rows_of(B, T), 32) * (compute_dtype == GCNPT_BF16 ? 1 : 2);
    WeightGradMulti mp{};
    mp.n = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        GCNPT_REQUIRE(z_frag[l] && s_frag[l] && dW[l] && db[l], "layer_bwd_weight: null pointer (layer %d)", l);
        GCNPT_REQUIRE(Din[l] > 0 && H[l] > 0, "layer_bwd_weight: sizes must be positive (layer %d)", l);
        mp.first[l + 1] = mp.first[l] + plan_weight_grad(mp.l[l], z_frag[l], s_frag[l], nks, Din[l], H[l], dW[l], db[l],
                                                         wg_blocks(n_layers, Din, H, wg_nt(nks)), wg_waves(n_layers, nks), wg_budget(nks), wg_nt(nks));
    }
    return launch_weight_grad((hipStream_t)stream, mp, compute_dtype, wg_waves(n_layers, nks), wg_nt(nks));
}

extern "C" int gcnpt_layer_bwd_weight(void* stream, const void* z_frag, const void* s_frag, int B, int T, int Din, int H,
                                      float* dW, float* db, int compute_dtype) {
    return gcnpt_layer_bwd_weight_multi(stream, 1, &z_frag, &s_frag, B, T, &Din, &H, &dW, &db, compute_dtype);
}

// weight gradients of the sentence-resident stack: dW_l += G_l^T h_l from the two per-sentence fragment images
// (db_l is added by gcnpt_stack_bwd itself)
extern "C" int gcnpt_stack_bwd_weight(void* stream, int n_layers, const void* const* g_frag, const void* const* h_frag, int B, int T,
                                      int Din, int H, float* const* dW) {
    GCNPT_REQUIRE(g_frag && h_frag && dW && n_layers >= 1 && n_layers <= 8, "stack_bwd_weight: bad argument");
    GCNPT_REQUIRE(B > 0 && T > 0 && Din > 0 && H > 0, "stack_bwd_weight: sizes must be positive");
    const int nks = B * ceil_div(T, 32);
    WeightGradMulti mp{};
    mp.n = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        GCNPT_REQUIRE(g_frag[l] && h_frag[l] && dW[l], "stack_bwd_weight: null pointer (layer %d)", l);
        const int din_l[1] = {l == 0 ? Din : H}, h_l[1] = {H};
        mp.first[l + 1] = mp.first[l] + plan_weight_grad(mp.l[l], g_frag[l], h_frag[l], nks, l == 0 ? Din : H, H, dW[l], nullptr,
                                                         n_layers * wg_blocks(1, din_l, h_l, wg_nt(nks)), wg_waves(n_layers, nks), wg_budget(nks), wg_nt(nks));
    }
    return launch_weight_grad((hipStream_t)stream, mp, GCNPT_BF16, wg_waves(n_layers, nks), wg_nt(nks));
}
