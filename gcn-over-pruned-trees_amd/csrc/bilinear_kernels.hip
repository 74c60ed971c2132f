// The relation-conditioned traversal of adj_type == 'full_deprel' (SURVEY.md 8f row N3): reference model/gcn.py:400-415
//   y[m,:] += sum_d e[m,d] * (x[m,:] @ W3[d])          W3 = Linear.weight.reshape(D, Tin, H)   (gcn.py:301)
// for the M tokens that sit in a pruned tree (compacted by the caller).  The reference materialises the outer product
// e (x) x as [B,T,D,Tin] and contracts it with two einsums.  Here the per-relation products P_d = x @ W3[d] run on the
// matrix cores (bf16 operands, fp32 accumulate) and are folded into the result with one fp32 FMA per accumulator register:
//   * the MFMAs run with swapped operands (weights as A), so a lane holds ONE token row and 4 consecutive output columns:
//     the scale e[m,d] is a single scalar per lane and tile;
//   * a workgroup owns 64 tokens x 48 output columns and one slice of the relations; its x fragments stay in registers for
//     the whole kernel, its 4 waves take every 4th relation of the slice, weight fragments of the next relation are in
//     flight while the current one is multiplied; waves meet in LDS, slices with fp32 atomics (y is accumulated: the caller
//     initialises it, e.g. with the bias term e @ b3).
// (Measured and dropped: 256-token workgroups whose 4 waves share each relation's weight fragments through LDS, fetched with
// direct global-to-LDS loads one relation ahead -- a quarter of the L2 traffic, but 95 us instead of 69 at M=1200, D=200: with a
// single relation in flight per workgroup the MFMAs of a relation (0.6 us) are shorter than the fetch of the next one, while
// here the 4 waves keep 4 independent streams in flight.  A 3-4 deep LDS ring with counted waits is the next step.)
// Weights are packed once per step into MFMA fragment order: image[n_tile][d * TS + ts][lane] x 16 bytes, TS = ceil(Tin / 32),
// every relation padded to TS k-steps, lane l of a fragment = 8 values W3[d][32 ts + 8 (l >> 4) + j][16 n_tile + (l & 15)].
#include "layer_common.h"

namespace gcnpt {

constexpr int BL_THREADS = 256, BL_WAVES = BL_THREADS / WAVE;
constexpr int BL_MT = 4, BL_NT = 3;          // 16-row token tiles and 16-column output tiles per workgroup
constexpr int BL_TSMAX = 8;                  // k-steps per relation the registers are sized for: Tin <= 256

__global__ void bilinear_pack_kernel(const float* __restrict__ W, int D, int Tin, int H, int TS, long long n_frag_lanes,
                                     uint4* __restrict__ img) {
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < n_frag_lanes; gid += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(gid & 63);
        const long long f = gid >> 6;
        const int ks = (int)(f % ((long long)D * TS)), tl = (int)(f / ((long long)D * TS));
        const int d = ks / TS, ts = ks - d * TS;
        const int n = tl * 16 + (lane & 15), t0 = ts * 32 + 8 * (lane >> 4);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                       // unconditional clamped loads, dropped with a select
            const float x = W[((size_t)d * Tin + min(t0 + j, Tin - 1)) * H + min(n, H - 1)];
            v[j] = (t0 + j < Tin && n < H) ? x : 0.0f;
        }
        uint4 u;
        u.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        u.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        u.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
        u.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
        img[gid] = u;
    }
}

struct BilinearParams {
    const bf16_t* x;        // [M, TS*32] bf16, zero padded
    const float* e;         // [M, D]
    const uint4* img;       // packed W3
    float* y;               // [M, H], accumulated
    int M, D, H, TS, n_tiles, mb, nb, slices, d_per_slice;
};

__global__ __launch_bounds__(BL_THREADS, 1) void bilinear_fwd_kernel(const BilinearParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bl_smem[];
    typedef f32x4_t RedTile[BL_MT * BL_NT][WAVE];
    RedTile* red = reinterpret_cast<RedTile*>(bl_smem);                                  // [BL_WAVES], 12 KiB each
    float* es = reinterpret_cast<float*>(bl_smem + sizeof(RedTile) * BL_WAVES);         // [d_per_slice][65]: e of the tile, transposed
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = blockIdx.x;
    const int slice = id % p.slices, rest = id / p.slices;
    const int bm = rest % p.mb, bn = rest / p.mb;
    const int m0 = bm * 16 * BL_MT, nt0 = bn * BL_NT;
    const int d_lo = slice * p.d_per_slice, d_hi = min(p.D, d_lo + p.d_per_slice);
    const int nd = d_hi - d_lo;
    const int TS = p.TS, Tpad = TS * 32;
    const size_t d_stride = (size_t)p.D * TS;            // fragments (of 64 lanes) per output tile

    // the tile's x fragments: lane (l & 15) = token row, 8 consecutive k per lane -- registers for the whole kernel
    uint4 xf[BL_MT][BL_TSMAX];
#pragma unroll
    for (int mt = 0; mt < BL_MT; ++mt) {
        const size_t row = (size_t)min(m0 + 16 * mt + (lane & 15), p.M - 1);
#pragma unroll
        for (int ts = 0; ts < BL_TSMAX; ++ts)
            xf[mt][ts] = *reinterpret_cast<const uint4*>(p.x + row * Tpad + min(ts, TS - 1) * 32 + 8 * (lane >> 4));
    }
    // this wave's first relation: its weight fragments
    uint4 wa[BL_TSMAX][BL_NT], wb[BL_TSMAX][BL_NT];
    auto load_w = [&](int dd, uint4 (&w)[BL_TSMAX][BL_NT]) {
        const int d = d_lo + min(dd, nd - 1);
#pragma unroll
        for (int ts = 0; ts < BL_TSMAX; ++ts)
#pragma unroll
            for (int j = 0; j < BL_NT; ++j)
                w[ts][j] = p.img[((size_t)min(nt0 + j, p.n_tiles - 1) * d_stride + (size_t)d * TS + min(ts, TS - 1)) * 64 + lane];
    };
    load_w(wave, wa);
    // e of the tile -> LDS, relation-major (a lane then reads its row's scalar without bank conflicts)
    for (int q = tid; q < nd * 64; q += BL_THREADS) {
        const int row = q / nd, dl = q - row * nd;
        es[dl * 65 + row] = p.e[(size_t)min(m0 + row, p.M - 1) * p.D + d_lo + dl];
    }
    __syncthreads();

    f32x4_t acc[BL_MT][BL_NT];
#pragma unroll
    for (int mt = 0; mt < BL_MT; ++mt)
#pragma unroll
        for (int j = 0; j < BL_NT; ++j) acc[mt][j] = (f32x4_t){0, 0, 0, 0};

    auto relation = [&](int dd, const uint4 (&w)[BL_TSMAX][BL_NT]) {
        f32x4_t P[BL_MT][BL_NT];
#pragma unroll
        for (int mt = 0; mt < BL_MT; ++mt)
#pragma unroll
            for (int j = 0; j < BL_NT; ++j) P[mt][j] = (f32x4_t){0, 0, 0, 0};
#pragma unroll
        for (int ts = 0; ts < BL_TSMAX; ++ts) {
            if (ts < TS) {                                      // workgroup-uniform, no load inside
#pragma unroll
                for (int j = 0; j < BL_NT; ++j)
#pragma unroll
                    for (int mt = 0; mt < BL_MT; ++mt)
                        P[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w[ts][j]),
                                                                           __builtin_bit_cast(bf16x8_t, xf[mt][ts]), P[mt][j], 0, 0, 0);
            }
        }
        const bool live = dd < nd;                              // past the slice: contributes zeros
#pragma unroll
        for (int mt = 0; mt < BL_MT; ++mt) {
            const float ev = live ? es[min(dd, nd - 1) * 65 + 16 * mt + (lane & 15)] : 0.0f;
#pragma unroll
            for (int j = 0; j < BL_NT; ++j) acc[mt][j] += ev * P[mt][j];
        }
    };
    // two relations per round: the fragments of the next one are requested before the current one is multiplied
    for (int dd = wave; dd < nd; dd += 2 * BL_WAVES) {
        load_w(dd + BL_WAVES, wb);
        relation(dd, wa);
        load_w(dd + 2 * BL_WAVES, wa);
        relation(dd + BL_WAVES, wb);
    }

    // waves meet in LDS; slices of the relation range are combined with float atomics
#pragma unroll
    for (int mt = 0; mt < BL_MT; ++mt)
#pragma unroll
        for (int j = 0; j < BL_NT; ++j) red[wave][mt * BL_NT + j][lane] = acc[mt][j];
    __syncthreads();
    for (int tt = wave; tt < BL_MT * BL_NT; tt += BL_WAVES) {
        const int mt = tt / BL_NT, j = tt - mt * BL_NT;
        f32x4_t v = red[0][tt][lane];
#pragma unroll
        for (int w = 1; w < BL_WAVES; ++w) v += red[w][tt][lane];
        const int m = m0 + 16 * mt + (lane & 15);
        const int n = (nt0 + j) * 16 + 4 * (lane >> 4);
        if (m < p.M && nt0 + j < p.n_tiles) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (n + g < p.H) atomicAdd(p.y + (size_t)m * p.H + n + g, v[g]);
        }
    }
}

}  // namespace gcnpt

using namespace gcnpt;

extern "C" size_t gcnpt_bilinear_packed_bytes(int D, int Tin, int H) {
    if (D <= 0 || Tin <= 0 || H <= 0) return 0;
    return (size_t)ceil_div(H, 16) * D * ceil_div(Tin, 32) * 64 * 16;
}

extern "C" int gcnpt_bilinear_supported(int D, int Tin, int H) {
    return D > 0 && H > 0 && Tin > 0 && ceil_div(Tin, 32) <= BL_TSMAX;
}

extern "C" int gcnpt_bilinear_pack(void* stream, const float* W, int D, int Tin, int H, void* w_img) {
    GCNPT_REQUIRE(W && w_img, "bilinear_pack: null pointer");
    GCNPT_REQUIRE(D > 0 && Tin > 0 && H > 0, "bilinear_pack: sizes must be positive");
    const int TS = ceil_div(Tin, 32);
    const long long n = (long long)ceil_div(H, 16) * D * TS * 64;
    const int grid = (int)std::min<long long>((n + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(bilinear_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, D, Tin, H, TS, n, static_cast<uint4*>(w_img));
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_bilinear_fwd(void* stream, const void* x, const float* e, const void* w_img, int M, int D, int Tin, int H,
                                  float* y) {
    GCNPT_REQUIRE(x && e && w_img && y, "bilinear_fwd: null pointer");
    GCNPT_REQUIRE(M > 0 && D > 0 && Tin > 0 && H > 0, "bilinear_fwd: sizes must be positive");
    GCNPT_REQUIRE(aligned16(x) && aligned16(w_img), "bilinear_fwd: x and w_img must be 16-byte aligned");
    if (!gcnpt_bilinear_supported(D, Tin, H))
        return fail(GCNPT_E_UNSUPPORTED, "bilinear_fwd: Tin=%d needs more than %d k-steps per relation", Tin, BL_TSMAX);
    BilinearParams p{};
    p.x = static_cast<const bf16_t*>(x); p.e = e; p.img = static_cast<const uint4*>(w_img); p.y = y;
    p.M = M; p.D = D; p.H = H; p.TS = ceil_div(Tin, 32); p.n_tiles = ceil_div(H, 16);
    p.mb = ceil_div(M, 16 * BL_MT); p.nb = ceil_div(p.n_tiles, BL_NT);
    // relation slices: about one workgroup per CU, each wave at least 2 relations
#ifndef GCNPT_BL_TARGET_WGS
#define GCNPT_BL_TARGET_WGS 256     // measured: 256 beats 512 (56 vs 70 us at M=1200, D=200), 1024 and 2048 are far worse (float atomics + per-workgroup setup)
#endif
    int slices = std::max(1, std::min(GCNPT_BL_TARGET_WGS / std::max(1, p.mb * p.nb), D / (2 * BL_WAVES)));
    slices = std::max(1, std::min(slices, D));
    p.d_per_slice = ceil_div(D, slices);
    p.slices = ceil_div(D, p.d_per_slice);
    const size_t lds = sizeof(f32x4_t) * BL_MT * BL_NT * WAVE * BL_WAVES + sizeof(float) * 65 * (size_t)p.d_per_slice;
    if (lds > 160 * 1024) return fail(GCNPT_E_UNSUPPORTED, "bilinear_fwd: %zu B of LDS", lds);
    static bool big_lds = false;
    if (lds > 64 * 1024 && !big_lds) {
        GCNPT_HIP_CHECK(hipFuncSetAttribute((const void*)bilinear_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        big_lds = true;
    }
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(p.mb * p.nb * p.slices), dim3(BL_THREADS), lds, (hipStream_t)stream, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}
