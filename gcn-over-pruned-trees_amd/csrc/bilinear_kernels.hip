// The relation-conditioned traversal of adj_type == 'full_deprel' (SURVEY.md 8f row N3): reference model/gcn.py:400-415
//   y[m,:] = sum_d e[m,d] * (x[m,:] @ W3[d])           W3 = Linear.weight.reshape(D, Tin, H)   (gcn.py:301)
// for the M tokens that sit in a pruned tree (compacted by the caller).  The reference materialises the outer product
// e (x) x as [B,T,D,Tin] and contracts it with two einsums.  Here the per-relation products P_d = x @ W3[d] run on the
// matrix cores (bf16 operands, fp32 accumulate) and are folded into the result with one fp32 FMA per accumulator register:
//   * the MFMAs run with swapped operands (weights as A), so a lane holds ONE token row and 4 consecutive output columns:
//     the scale e[m,d] is a single scalar per lane and tile;
//   * a workgroup owns 256 tokens x 48 output columns and one slice of the relations; each wave keeps the x fragments of its 64
//     tokens in registers for the whole kernel; the weight fragments of a relation come through a 4-deep LDS ring once per
//     workgroup; every relation slice writes its own plane of the result, the caller sums the planes (and adds e @ b3).
// Weights are packed once per step into MFMA fragment order: image[n_tile][d * TS + ts][lane] x 16 bytes, TS = ceil(Tin / 32),
// every relation padded to TS k-steps, lane l of a fragment = 8 values W3[d][32 ts + 8 (l >> 4) + j][16 n_tile + (l & 15)].
#include "layer_common.h"

namespace gcnpt {

constexpr int BL_THREADS = 256, BL_WAVES = BL_THREADS / WAVE;
constexpr int BL_MT = 4, BL_NT = 3;          // 16-row token tiles and 16-column output tiles per workgroup
constexpr int BL_TSMAX = 8;                  // k-steps per relation the registers are sized for: Tin <= 256

// transposed = 0: contraction over Tin, output columns H (forward);  1: contraction over H, output columns Tin (the same kernel
// then computes dx = sum_d e_d (gy @ W3[d]^T))
__global__ void bilinear_pack_kernel(const float* __restrict__ W, int D, int Tin, int H, int TS, long long n_frag_lanes,
                                     uint4* __restrict__ img, int transposed) {
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < n_frag_lanes; gid += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(gid & 63);
        const long long f = gid >> 6;
        const int ks = (int)(f % ((long long)D * TS)), tl = (int)(f / ((long long)D * TS));
        const int d = ks / TS, ts = ks - d * TS;
        const int n = tl * 16 + (lane & 15), k0 = ts * 32 + 8 * (lane >> 4);
        const int Kd = transposed ? H : Tin, Nd = transposed ? Tin : H;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                       // unconditional clamped loads, dropped with a select
            const int kc = min(k0 + j, Kd - 1), nc = min(n, Nd - 1);
            const float x = transposed ? W[((size_t)d * Tin + nc) * H + kc] : W[((size_t)d * Tin + kc) * H + nc];
            v[j] = (k0 + j < Kd && n < Nd) ? x : 0.0f;
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
    float* y;               // MODE 0: [slices][M, H], one plane per relation slice; MODE 1: [nb][M, D], one plane per column block
    const float* gy;        // MODE 1: [M, H] upstream gradient
    int M, D, H, TS, n_tiles, mb, nb, slices, d_per_slice;
    int TSA, chunks;        // k-steps per relation in all (x rows and the image are that wide); the contraction is cut into `chunks` runs
                            // of <= TS k-steps (Tin > 256: what the registers hold), each with planes of its own
};

// A workgroup owns 256 tokens (its 4 waves take 64 each: 4 token tiles whose x fragments stay in registers), 48 output columns
// and one slice of the relations.  All waves walk the SAME relations: the 3 x TS weight fragments of a relation come from L2 ONCE
// per workgroup, straight into LDS (global_load_lds, one 1-KiB fragment per wave-instruction, lane-linear on both sides), and feed
// 16 MFMAs each.  The LDS ring holds 4 relations; three are in flight while one is multiplied: the wait at the top of a round is a
// COUNTED s_waitcnt (two relations' loads of this wave may stay outstanding) and the barrier a raw s_barrier -- __syncthreads()
// would drain the direct-to-LDS loads (they count as pending LDS writes) and with it the pipeline.
constexpr int BL_ROWS = 16 * BL_MT * BL_WAVES;       // 256 token rows per workgroup
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
constexpr int BL_RING = 4;                           // relations in the LDS ring
constexpr int BL_GL = (BL_NT * BL_TSMAX + BL_WAVES - 1) / BL_WAVES;      // direct-to-LDS loads per wave and relation: always 6
constexpr int BL_FRAGS = BL_GL * BL_WAVES;           // fragment slots per ring entry (24 KiB), TS < 8 leaves some unused

// MODE 0: y planes = sum_d e_d P_d.   MODE 1: de[m,d] = P_d[m,:] . gy[m,:] (the gradient of the relation vectors: the same per-relation
// products, dotted with the upstream gradient instead of scaled and summed), one plane per block of 48 columns.
template <int MODE>
__global__ __launch_bounds__(BL_THREADS, 1) void bilinear_fwd_kernel(const BilinearParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bl_smem[];              // ONE LDS object: ring | dummy | e tile
    const int per_chunk = p.mb * p.nb * p.slices;
    const int chunk = (int)blockIdx.x / per_chunk, ts0 = chunk * p.TS;                   // this workgroup's run of the contraction
    const int TS = min(p.TS, p.TSA - ts0), Tpad = p.TSA * 32;
    const int n_frag = BL_NT * TS;
    uint4* wl = reinterpret_cast<uint4*>(bl_smem);                                       // [BL_RING][BL_FRAGS][64]
    uint4* dummy = wl + (size_t)BL_RING * BL_FRAGS * 64;                                 // [BL_WAVES][64]: where unused slots' loads land
    float* es = reinterpret_cast<float*>(dummy + BL_WAVES * 64);                         // [d_per_slice][BL_ROWS + 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = (int)blockIdx.x - chunk * per_chunk;
    const int slice = id % p.slices, rest = id / p.slices;
    const int bm = rest % p.mb, bn = rest / p.mb;
    const int m0 = bm * BL_ROWS + wave * 16 * BL_MT, nt0 = bn * BL_NT;
    const int d_lo = slice * p.d_per_slice, d_hi = min(p.D, d_lo + p.d_per_slice);
    const int nd = d_hi - d_lo;
    const size_t d_stride = (size_t)p.D * p.TSA;         // fragments per output tile in the image

    // this wave's x fragments: lane (l & 15) = token row, 8 consecutive k per lane -- registers for the whole kernel
    uint4 xf[BL_MT][BL_TSMAX];
#pragma unroll
    for (int mt = 0; mt < BL_MT; ++mt) {
        const size_t row = (size_t)min(m0 + 16 * mt + (lane & 15), p.M - 1);
#pragma unroll
        for (int ts = 0; ts < BL_TSMAX; ++ts)
            xf[mt][ts] = *reinterpret_cast<const uint4*>(p.x + row * Tpad + (ts0 + min(ts, TS - 1)) * 32 + 8 * (lane >> 4));
    }
    // relation dd -> ring entry: exactly BL_GL loads per wave (fragments wave, wave + 4, ...; slots past 3 TS go to the dummy)
    auto fetch_w = [&](int dd) {
        const int d = d_lo + min(dd, nd - 1);
        uint4* ring = wl + (size_t)(dd & (BL_RING - 1)) * BL_FRAGS * 64;
#pragma unroll
        for (int u = 0; u < BL_GL; ++u) {
            const int f = wave + u * BL_WAVES;
            const int fc = min(f, n_frag - 1);
            const int j = fc / TS, ts = fc - j * TS;
            const uint4* src = p.img + ((size_t)min(nt0 + j, p.n_tiles - 1) * d_stride + (size_t)d * p.TSA + ts0 + ts) * 64 + lane;
            uint4* dst = f < n_frag ? ring + (size_t)f * 64 : dummy + wave * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    fetch_w(0);
    fetch_w(1);
    fetch_w(2);
    // e of the workgroup's rows -> LDS, relation-major
    const int rows0 = bm * BL_ROWS;
    constexpr int BL_EU = 8;                                 // loads in flight per thread and round (a load per round would be ~20 round trips)
    for (int q0 = tid; MODE == 0 && q0 < nd * BL_ROWS; q0 += BL_THREADS * BL_EU) {
        float v[BL_EU];
#pragma unroll
        for (int u = 0; u < BL_EU; ++u) {
            const int q = min(q0 + u * BL_THREADS, nd * BL_ROWS - 1);
            const int row = q / nd, dl = q - row * nd;
            v[u] = p.e[(size_t)min(rows0 + row, p.M - 1) * p.D + d_lo + dl];
        }
#pragma unroll
        for (int u = 0; u < BL_EU; ++u) {
            const int q = q0 + u * BL_THREADS;
            const int row = q / nd, dl = q - row * nd;
            if (q < nd * BL_ROWS) es[dl * (BL_ROWS + 1) + row] = v[u];
        }
    }
    __syncthreads();                                         // (one full drain before the loop: e tile and the first three relations)

    f32x4_t acc[BL_MT][BL_NT];             // MODE 0: the result tile; MODE 1: the upstream gradient's values at the tile's positions
#pragma unroll
    for (int mt = 0; mt < BL_MT; ++mt)
#pragma unroll
        for (int j = 0; j < BL_NT; ++j) {
            acc[mt][j] = (f32x4_t){0, 0, 0, 0};
            if constexpr (MODE == 1) {
                const size_t row = (size_t)min(m0 + 16 * mt + (lane & 15), p.M - 1);
                const int n = (nt0 + j) * 16 + 4 * (lane >> 4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float v = p.gy[row * p.H + min(n + g, p.H - 1)];
                    acc[mt][j][g] = (n + g < p.H && nt0 + j < p.n_tiles) ? v : 0.0f;
                }
            }
        }
    if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // those loads are not part of the loop's counted queue

    for (int dd = 0; dd < nd; ++dd) {
        // relation dd has landed when at most the loads of dd+1 and dd+2 (2 x BL_GL = 12 of this wave) are outstanding
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // ... in every wave; and every wave is done reading relation dd-1
        fetch_w(dd + 3);                                     // into the ring entry relation dd-1 has just left
        // Every LDS read of the loop is inline asm: hipcc cannot tell a ds_read from the ring apart from the direct-to-LDS loads in
        // flight and would put an s_waitcnt vmcnt(0) in front of each (seen in the ISA), which is the drain the counted wait avoids.
        const unsigned ring = lds_addr(wl) + (unsigned)((dd & (BL_RING - 1)) * BL_FRAGS * 1024) + (unsigned)lane * 16u;
        const unsigned eaddr = lds_addr(es) + (unsigned)((dd * (BL_ROWS + 1) + wave * 16 * BL_MT + (lane & 15)) * 4);
        float ev[BL_MT] = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (MODE == 0) {
#pragma unroll
            for (int mt = 0; mt < BL_MT; ++mt) asm volatile("ds_read_b32 %0, %1" : "=v"(ev[mt]) : "v"(eaddr + (unsigned)(mt * 64)));
        }
        uint4 wc[2][BL_NT];
        auto read_w = [&](int ts, uint4 (&w)[BL_NT]) {
#pragma unroll
            for (int j = 0; j < BL_NT; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(w[j]) : "v"(ring + (unsigned)((j * TS + ts) * 1024)));
        };
        read_w(0, wc[0]);
        f32x4_t P[BL_MT][BL_NT];
#pragma unroll
        for (int mt = 0; mt < BL_MT; ++mt)
#pragma unroll
            for (int j = 0; j < BL_NT; ++j) P[mt][j] = (f32x4_t){0, 0, 0, 0};
#pragma unroll
        for (int ts = 0; ts < BL_TSMAX; ++ts) {
            if (ts < TS) {                                   // workgroup-uniform
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // this k-step's fragments (requested one k-step ago)
                if (ts + 1 < TS) read_w(ts + 1, wc[(ts + 1) & 1]);                  // the next one's, behind this k-step's MFMAs
#pragma unroll
                for (int j = 0; j < BL_NT; ++j)
#pragma unroll
                    for (int mt = 0; mt < BL_MT; ++mt)
                        P[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wc[ts & 1][j]),
                                                                           __builtin_bit_cast(bf16x8_t, xf[mt][ts]), P[mt][j], 0, 0, 0);
            }
        }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int mt = 0; mt < BL_MT; ++mt)
#pragma unroll
                for (int j = 0; j < BL_NT; ++j) acc[mt][j] += ev[mt] * P[mt][j];
        } else {
            float* plane = p.y + (size_t)(chunk * p.nb + bn) * p.M * p.D;
#pragma unroll
            for (int mt = 0; mt < BL_MT; ++mt) {
                float sdot = 0.0f;
#pragma unroll
                for (int j = 0; j < BL_NT; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) sdot += P[mt][j][g] * acc[mt][j][g];
                sdot += __shfl_xor(sdot, 16);
                sdot += __shfl_xor(sdot, 32);
                const int m = m0 + 16 * mt + (lane & 15);
                if (lane < 16 && m < p.M) plane[(size_t)m * p.D + d_lo + dd] = sdot;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped re-fetches past the slice's end

    if constexpr (MODE == 1) return;
    // every wave owns its rows: plain 16-byte stores into this relation slice's own plane of y (the caller sums the planes: ten
    // float atomics per element cost more than the whole contraction -- ~50 G atomics/s device-wide -- a plane costs one store)
    float* plane = p.y + (size_t)(chunk * p.slices + slice) * p.M * p.H;
    const bool vec = (p.H & 3) == 0;
#pragma unroll
    for (int mt = 0; mt < BL_MT; ++mt) {
        const int m = m0 + 16 * mt + (lane & 15);
#pragma unroll
        for (int j = 0; j < BL_NT; ++j) {
            const int n = (nt0 + j) * 16 + 4 * (lane >> 4);
            if (m < p.M && nt0 + j < p.n_tiles) {
                float* dst = plane + (size_t)m * p.H + n;
                if (vec && n + 4 <= p.H) {
                    *reinterpret_cast<float4*>(dst) = make_float4(acc[mt][j][0], acc[mt][j][1], acc[mt][j][2], acc[mt][j][3]);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (n + g < p.H) dst[g] = acc[mt][j][g];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// dW3[d][t][h] = sum_m e[m,d] * x[m,t] * gy[m,h]      (the weight gradient of the traversal: the outer product e (x) x contracted
// with the upstream gradient over the M tokens; the reference's autograd materialises [M, D*Tin] for it)
// Both token-side operands come as "row-contraction" fragment images (lane = column, 8 consecutive TOKENS per lane, as the
// saved operands of the layer kernels): xI[t_tile][m_step][lane], gI[h_tile][m_step][lane].  A workgroup owns a (4 x 3)-tile block
// of dW3 for BW_ND relations; per m-step it loads the 4 + 3 fragments once and, for each of its relations, scales the three gy
// fragments by e[m, d] along the tokens (fp32 multiply, repacked to bf16) and issues the 12 MFMAs.  Its 4 waves take every 4th
// m-step and meet in LDS; every element of dW3 is written once (no atomics).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int BW_MT = 8, BW_NT = 3, BW_ND = 2;       // 8 x 3 tiles x 2 relations: a scaled gy fragment feeds 8 MFMAs (4 x 3 x 4 was VALU-bound)
constexpr int BW_RH = 4;                             // tile rows reduced per LDS round (12 tiles x 4 waves = 48 KiB)

__global__ void rows_pack_kernel(const float* __restrict__ src, int M, int W, int m_steps, long long n_lanes, uint4* __restrict__ img) {
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < n_lanes; gid += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(gid & 63);
        const long long f = gid >> 6;
        const int ms = (int)(f % m_steps), wt = (int)(f / m_steps);
        const int c = wt * 16 + (lane & 15), m0 = ms * 32 + 8 * (lane >> 4);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = src[(size_t)min(m0 + j, M - 1) * W + min(c, W - 1)];
            v[j] = (m0 + j < M && c < W) ? x : 0.0f;
        }
        uint4 u;
        u.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        u.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        u.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
        u.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
        img[gid] = u;
    }
}

struct BilinearDwParams {
    const uint4* xI;        // [t_tiles][m_steps][64]
    const uint4* gI;        // [h_tiles][m_steps][64]
    const float* eT;        // [D][m_steps * 32]: e transposed, zero padded along the tokens
    float* dW;              // [D][Tin][H] == the Linear weight's [D*H, Tin] memory
    int D, Tin, H, t_tiles, h_tiles, m_steps, tb, hb, dgroups;
};

__global__ __launch_bounds__(BL_THREADS, 1) void bilinear_dw_kernel(const BilinearDwParams p) {
    __shared__ f32x4_t red[BL_WAVES][BW_RH * BW_NT][WAVE];                  // 48 KiB: half of one relation's tiles at a time
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = blockIdx.x;
    const int bt = id % p.tb, rest = id / p.tb;
    const int bh = rest % p.hb, dg = rest / p.hb;
    const int t0 = bt * BW_MT, h0 = bh * BW_NT, d0 = dg * BW_ND;
    f32x4_t acc[BW_ND][BW_MT][BW_NT];
#pragma unroll
    for (int q = 0; q < BW_ND; ++q)
#pragma unroll
        for (int i = 0; i < BW_MT; ++i)
#pragma unroll
            for (int j = 0; j < BW_NT; ++j) acc[q][i][j] = (f32x4_t){0, 0, 0, 0};
    const size_t Mpad = (size_t)p.m_steps * 32;
    struct Operands { uint4 xa[BW_MT], gb[BW_NT]; float4 ea[BW_ND][2]; };
    auto fetch = [&](int ms_raw, Operands& o) {              // clamped: the step past the end re-reads the last one and is not used
        const int ms = min(ms_raw, p.m_steps - 1);
#pragma unroll
        for (int i = 0; i < BW_MT; ++i) o.xa[i] = p.xI[((size_t)min(t0 + i, p.t_tiles - 1) * p.m_steps + ms) * 64 + lane];
#pragma unroll
        for (int j = 0; j < BW_NT; ++j) o.gb[j] = p.gI[((size_t)min(h0 + j, p.h_tiles - 1) * p.m_steps + ms) * 64 + lane];
#pragma unroll
        for (int q = 0; q < BW_ND; ++q) {
            const float* ep = p.eT + (size_t)min(d0 + q, p.D - 1) * Mpad + (size_t)ms * 32 + 8 * (lane >> 4);
            o.ea[q][0] = *reinterpret_cast<const float4*>(ep);
            o.ea[q][1] = *reinterpret_cast<const float4*>(ep + 4);
        }
    };
    auto consume = [&](const Operands& o) {
#pragma unroll
        for (int q = 0; q < BW_ND; ++q) {
#pragma unroll
            for (int j = 0; j < BW_NT; ++j) {
                // gy fragment scaled along the tokens by e[., d]: 8 bf16 -> fp32, multiply, back to bf16
                const uint4 g = o.gb[j];
                uint4 sg;
                sg.x = (unsigned)f32_to_bf16(__uint_as_float(g.x << 16) * o.ea[q][0].x) | ((unsigned)f32_to_bf16(__uint_as_float(g.x & 0xffff0000u) * o.ea[q][0].y) << 16);
                sg.y = (unsigned)f32_to_bf16(__uint_as_float(g.y << 16) * o.ea[q][0].z) | ((unsigned)f32_to_bf16(__uint_as_float(g.y & 0xffff0000u) * o.ea[q][0].w) << 16);
                sg.z = (unsigned)f32_to_bf16(__uint_as_float(g.z << 16) * o.ea[q][1].x) | ((unsigned)f32_to_bf16(__uint_as_float(g.z & 0xffff0000u) * o.ea[q][1].y) << 16);
                sg.w = (unsigned)f32_to_bf16(__uint_as_float(g.w << 16) * o.ea[q][1].z) | ((unsigned)f32_to_bf16(__uint_as_float(g.w & 0xffff0000u) * o.ea[q][1].w) << 16);
#pragma unroll
                for (int i = 0; i < BW_MT; ++i)       // swapped operands: rows = h (scaled gy), columns = t (x): a lane ends with 4 consecutive h of one t
                    acc[q][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, sg), __builtin_bit_cast(bf16x8_t, o.xa[i]),
                                                                          acc[q][i][j], 0, 0, 0);
            }
        }
    };
    // two m-steps per round: the operands of the next one are in flight while the current one is multiplied
    Operands oa, ob;
    fetch(wave, oa);
    for (int ms = wave; ms < p.m_steps; ms += 2 * BL_WAVES) {
        fetch(ms + BL_WAVES, ob);
        consume(oa);
        fetch(ms + 2 * BL_WAVES, oa);
        if (ms + BL_WAVES < p.m_steps) consume(ob);          // wave-uniform, no load inside
    }
    // waves meet in LDS, BW_RH tile rows of one relation at a time
    for (int q = 0; q < BW_ND; ++q) {
        for (int ih = 0; ih < BW_MT; ih += BW_RH) {
            __syncthreads();
#pragma unroll
            for (int q2 = 0; q2 < BW_ND; ++q2)
#pragma unroll
                for (int i = 0; i < BW_MT; ++i)
#pragma unroll
                    for (int j = 0; j < BW_NT; ++j)
                        if (q2 == q && i >= ih && i < ih + BW_RH) red[wave][(i - ih) * BW_NT + j][lane] = acc[q2][i][j];      // static register indices
            __syncthreads();
            const int d = d0 + q;
            for (int tt = wave; tt < BW_RH * BW_NT; tt += BL_WAVES) {
                const int i = ih + tt / BW_NT, j = tt % BW_NT;
                f32x4_t v = red[0][tt][lane];
#pragma unroll
                for (int w = 1; w < BL_WAVES; ++w) v += red[w][tt][lane];
                const int t = (t0 + i) * 16 + (lane & 15);
                const int h = (h0 + j) * 16 + 4 * (lane >> 4);
                if (d < p.D && t < p.Tin && t0 + i < p.t_tiles && h0 + j < p.h_tiles) {
                    float* dst = p.dW + ((size_t)d * p.Tin + t) * p.H + h;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (h + g < p.H) dst[g] = v[g];
                }
            }
        }
    }
}


// ===========================================================================================================================
// The same traversal in EXACT fp32 (v_mfma_f32_16x16x4_f32): the module's default precision (opt['gcn_dtype'] = 'fp32'), the mode the
// reference-recorded gradient goldens are checked in.  Reference model/gcn.py:400-415 (+ 417-434 through the caller).
// fp32 MFMA runs at 1/16 of the bf16 rate (32 cycles per 16x16x4 MFMA), so these kernels are MFMA-bound by a wide margin and need none
// of the bf16 kernel's machinery: a relation's weight fragments are prefetched through registers one relation ahead into a two-stage
// LDS buffer (a relation's MFMAs take ~10 k cycles per wave), the token fragments stay in registers.
// Fragment = 16 bytes per lane = 4 floats X[k = 16 kstep + 4 (lane >> 4) + s][n = 16 tile + (lane & 15)], s = 0..3 (the layer kernels'
// fp32 image order, include/gcnpt.h); MFMA s of a k-step multiplies the s-th values.
// ===========================================================================================================================
constexpr int BF_MT = 2, BF_NT = 3;          // token tiles per wave, output tiles per workgroup (one workgroup per CU: ~250 registers)
constexpr int BF_TSMAX = 16;                 // 16-wide k-steps per relation the registers are sized for: Tin <= 256
constexpr int BF_ROWS = 16 * BF_MT * BL_WAVES;       // 128 token rows per workgroup

__global__ void bilinear_pack_f32_kernel(const float* __restrict__ W, int D, int Tin, int H, int TS, long long n_frag_lanes,
                                         uint4* __restrict__ img, int transposed) {
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < n_frag_lanes; gid += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(gid & 63);
        const long long f = gid >> 6;
        const int ks = (int)(f % ((long long)D * TS)), tl = (int)(f / ((long long)D * TS));
        const int d = ks / TS, ts = ks - d * TS;
        const int n = tl * 16 + (lane & 15), k0 = ts * 16 + 4 * (lane >> 4);
        const int Kd = transposed ? H : Tin, Nd = transposed ? Tin : H;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                       // unconditional clamped loads, dropped with a select
            const int kc = min(k0 + j, Kd - 1), nc = min(n, Nd - 1);
            const float x = transposed ? W[((size_t)d * Tin + nc) * H + kc] : W[((size_t)d * Tin + kc) * H + nc];
            v[j] = (k0 + j < Kd && n < Nd) ? x : 0.0f;
        }
        img[gid] = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    }
}

struct BilinearF32Params {
    const float* x;         // [M, TS*16] fp32, zero padded
    const float* e;         // [M, D]
    const uint4* img;       // packed W3 (fp32 fragments)
    float* y;               // MODE 0: [slices][M, H]; MODE 1: [nb][M, D]
    const float* gy;        // MODE 1: [M, H]
    int M, D, H, TS, n_tiles, mb, nb, slices, d_per_slice;
    int TSA, chunks;        // as BilinearParams: all k-steps of a relation / runs of <= TS k-steps the contraction is cut into
};

template <int MODE>
__global__ __launch_bounds__(BL_THREADS, 1) void bilinear_f32_kernel(const BilinearF32Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bf_smem[];              // [2][BF_NT * BF_TSMAX][64] uint4 | e tile
    const int per_chunk = p.mb * p.nb * p.slices;
    const int chunk = (int)blockIdx.x / per_chunk, ts0 = chunk * p.TS;
    const int TS = min(p.TS, p.TSA - ts0), Tpad = p.TSA * 16;
    const int n_frag = BF_NT * TS;
    constexpr int STAGE = BF_NT * BF_TSMAX * 64;                                         // uint4 per stage (all slots: the stores are unconditional)
    uint4* wl = reinterpret_cast<uint4*>(bf_smem);
    float* es = reinterpret_cast<float*>(wl + (size_t)2 * STAGE);                        // [d_per_slice][BF_ROWS + 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = (int)blockIdx.x - chunk * per_chunk;
    const int slice = id % p.slices, rest = id / p.slices;
    const int bm = rest % p.mb, bn = rest / p.mb;
    const int m0 = bm * BF_ROWS + wave * 16 * BF_MT, nt0 = bn * BF_NT;
    const int d_lo = slice * p.d_per_slice, d_hi = min(p.D, d_lo + p.d_per_slice);
    const int nd = d_hi - d_lo;
    const size_t d_stride = (size_t)p.D * p.TSA;

    // this wave's x fragments: lane (l & 15) = token row, 4 consecutive k per lane -- registers for the whole kernel
    uint4 xf[BF_MT][BF_TSMAX];
#pragma unroll
    for (int mt = 0; mt < BF_MT; ++mt) {
        const size_t row = (size_t)min(m0 + 16 * mt + (lane & 15), p.M - 1);
#pragma unroll
        for (int ts = 0; ts < BF_TSMAX; ++ts)
            xf[mt][ts] = *reinterpret_cast<const uint4*>(p.x + row * Tpad + (ts0 + min(ts, TS - 1)) * 16 + 4 * (lane >> 4));
    }
    // a relation's fragments: thread t fetches fragment-lanes t, t + 256, ... (<= BF_NT * BF_TSMAX * 64 / 256 = 12 per thread)
    constexpr int WPT = BF_NT * BF_TSMAX * 64 / BL_THREADS;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));      // a first-class vector: hipcc kept an array of uint4 STRUCTS in scratch memory
    u32x4_t wr[WPT];
    auto fetch_w = [&](int dd) {
        const int d = d_lo + min(dd, nd - 1);
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int q = min(tid + u * BL_THREADS, n_frag * 64 - 1);
            const int f = q >> 6, l = q & 63;
            const int j = f / TS, ts = f - j * TS;
            wr[u] = *reinterpret_cast<const u32x4_t*>(p.img + ((size_t)min(nt0 + j, p.n_tiles - 1) * d_stride + (size_t)d * p.TSA + ts0 + ts) * 64 + l);
        }
    };
    // (unconditional stores into a stage sized for BF_TSMAX k-steps: a prefetched register whose only use is a CONDITIONAL store gets
    // parked in scratch memory by hipcc right behind its load, i.e. the prefetch becomes a blocking load -- seen in the ISA)
    auto store_w = [&](int dd) {
        uint4* dst = wl + (size_t)(dd & 1) * STAGE;
#pragma unroll
        for (int u = 0; u < WPT; ++u) *reinterpret_cast<u32x4_t*>(dst + tid + u * BL_THREADS) = wr[u];
    };
    fetch_w(0);
    // e of the workgroup's rows -> LDS, relation-major
    const int rows0 = bm * BF_ROWS;
    constexpr int EU = 8;
    for (int q0 = tid; MODE == 0 && q0 < nd * BF_ROWS; q0 += BL_THREADS * EU) {
        float v[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int q = min(q0 + u * BL_THREADS, nd * BF_ROWS - 1);
            const int row = q / nd, dl = q - row * nd;
            v[u] = p.e[(size_t)min(rows0 + row, p.M - 1) * p.D + d_lo + dl];
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int q = q0 + u * BL_THREADS;
            const int row = q / nd, dl = q - row * nd;
            if (q < nd * BF_ROWS) es[dl * (BF_ROWS + 1) + row] = v[u];
        }
    }
    store_w(0);
    __syncthreads();

    f32x4_t acc[BF_MT][BF_NT];             // MODE 0: the result tile; MODE 1: the upstream gradient's values at the tile's positions
#pragma unroll
    for (int mt = 0; mt < BF_MT; ++mt)
#pragma unroll
        for (int j = 0; j < BF_NT; ++j) {
            acc[mt][j] = (f32x4_t){0, 0, 0, 0};
            if constexpr (MODE == 1) {
                const size_t row = (size_t)min(m0 + 16 * mt + (lane & 15), p.M - 1);
                const int n = (nt0 + j) * 16 + 4 * (lane >> 4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float v = p.gy[row * p.H + min(n + g, p.H - 1)];
                    acc[mt][j][g] = (n + g < p.H && nt0 + j < p.n_tiles) ? v : 0.0f;
                }
            }
        }

    for (int dd = 0; dd < nd; ++dd) {
        fetch_w(dd + 1);                                     // the next relation's fragments, in flight during this one's MFMAs
        const uint4* ring = wl + (size_t)(dd & 1) * STAGE;
        float ev[BF_MT];
#pragma unroll
        for (int mt = 0; mt < BF_MT; ++mt) ev[mt] = MODE == 0 ? es[dd * (BF_ROWS + 1) + wave * 16 * BF_MT + 16 * mt + (lane & 15)] : 0.0f;
        f32x4_t P[BF_MT][BF_NT];
#pragma unroll
        for (int mt = 0; mt < BF_MT; ++mt)
#pragma unroll
            for (int j = 0; j < BF_NT; ++j) P[mt][j] = (f32x4_t){0, 0, 0, 0};
#pragma unroll
        for (int ts = 0; ts < BF_TSMAX; ++ts) {
            if (ts < TS) {                                   // workgroup-uniform, no global load inside
                // the 4 MFMAs of a k-step that share an accumulator are kept BF_NT * BF_MT instructions apart (a dependent MFMA right
                // behind its producer waits for the 8 passes to drain)
                f32x4_t wq[BF_NT];
#pragma unroll
                for (int j = 0; j < BF_NT; ++j) wq[j] = __builtin_bit_cast(f32x4_t, ring[(size_t)(j * TS + ts) * 64 + lane]);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int j = 0; j < BF_NT; ++j)
#pragma unroll
                        for (int mt = 0; mt < BF_MT; ++mt) {
                            const f32x4_t xq = __builtin_bit_cast(f32x4_t, xf[mt][ts]);
                            P[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[j][s4], xq[s4], P[mt][j], 0, 0, 0);
                        }
            }
        }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int mt = 0; mt < BF_MT; ++mt)
#pragma unroll
                for (int j = 0; j < BF_NT; ++j) acc[mt][j] += ev[mt] * P[mt][j];
        } else {
            float* plane = p.y + (size_t)(chunk * p.nb + bn) * p.M * p.D;
#pragma unroll
            for (int mt = 0; mt < BF_MT; ++mt) {
                float sdot = 0.0f;
#pragma unroll
                for (int j = 0; j < BF_NT; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) sdot += P[mt][j][g] * acc[mt][j][g];
                sdot += __shfl_xor(sdot, 16);
                sdot += __shfl_xor(sdot, 32);
                const int m = m0 + 16 * mt + (lane & 15);
                if (lane < 16 && m < p.M) plane[(size_t)m * p.D + d_lo + dd] = sdot;
            }
        }
        store_w(dd + 1);                                     // into the stage relation dd-1 left (everybody passed the barrier since)
        __syncthreads();
    }

    if constexpr (MODE == 1) return;
    float* plane = p.y + (size_t)(chunk * p.slices + slice) * p.M * p.H;
    const bool vec = (p.H & 3) == 0;
#pragma unroll
    for (int mt = 0; mt < BF_MT; ++mt) {
        const int m = m0 + 16 * mt + (lane & 15);
#pragma unroll
        for (int j = 0; j < BF_NT; ++j) {
            const int n = (nt0 + j) * 16 + 4 * (lane >> 4);
            if (m < p.M && nt0 + j < p.n_tiles) {
                float* dst = plane + (size_t)m * p.H + n;
                if (vec && n + 4 <= p.H) {
                    *reinterpret_cast<float4*>(dst) = make_float4(acc[mt][j][0], acc[mt][j][1], acc[mt][j][2], acc[mt][j][3]);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (n + g < p.H) dst[g] = acc[mt][j][g];
                }
            }
        }
    }
}

// rows -> fp32 row-contraction fragment image: img[w_tile][m16 step][lane] = 4 floats src[16 ms + 4 (lane >> 4) + s][16 wt + (lane & 15)]
__global__ void rows_pack_f32_kernel(const float* __restrict__ src, int M, int W, int m_steps, long long n_lanes, uint4* __restrict__ img) {
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < n_lanes; gid += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(gid & 63);
        const long long f = gid >> 6;
        const int ms = (int)(f % m_steps), wt = (int)(f / m_steps);
        const int c = wt * 16 + (lane & 15), m0 = ms * 16 + 4 * (lane >> 4);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = src[(size_t)min(m0 + j, M - 1) * W + min(c, W - 1)];
            v[j] = (m0 + j < M && c < W) ? x : 0.0f;
        }
        img[gid] = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    }
}

// dW3[d][t][h] = sum_m e[m,d] x[m,t] gy[m,h] in exact fp32: (4 x 3)-tile blocks x 2 relations per workgroup; per 16-token step the gy
// fragments are scaled by e[., d] (one fp32 multiply per value) and meet the x fragments on the matrix cores; the 4 waves take every
// 4th step and meet in LDS; every element of dW3 is written once (no atomics).
constexpr int BWF_MT = 4, BWF_NT = 3, BWF_ND = 2;

__global__ __launch_bounds__(BL_THREADS, 1) void bilinear_dw_f32_kernel(const BilinearDwParams p) {
    __shared__ f32x4_t red[BL_WAVES][BWF_MT * BWF_NT][WAVE];                // 48 KiB: one relation's tiles at a time
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = blockIdx.x;
    const int bt = id % p.tb, rest = id / p.tb;
    const int bh = rest % p.hb, dg = rest / p.hb;
    const int t0 = bt * BWF_MT, h0 = bh * BWF_NT, d0 = dg * BWF_ND;
    f32x4_t acc[BWF_ND][BWF_MT][BWF_NT];
#pragma unroll
    for (int q = 0; q < BWF_ND; ++q)
#pragma unroll
        for (int i = 0; i < BWF_MT; ++i)
#pragma unroll
            for (int j = 0; j < BWF_NT; ++j) acc[q][i][j] = (f32x4_t){0, 0, 0, 0};
    const size_t Mpad = (size_t)p.m_steps * 16;
    struct Operands { uint4 xa[BWF_MT], gb[BWF_NT]; float4 ea[BWF_ND]; };
    auto fetch = [&](int ms_raw, Operands& o) {              // clamped: the step past the end re-reads the last one and is not used
        const int ms = min(ms_raw, p.m_steps - 1);
#pragma unroll
        for (int i = 0; i < BWF_MT; ++i) o.xa[i] = p.xI[((size_t)min(t0 + i, p.t_tiles - 1) * p.m_steps + ms) * 64 + lane];
#pragma unroll
        for (int j = 0; j < BWF_NT; ++j) o.gb[j] = p.gI[((size_t)min(h0 + j, p.h_tiles - 1) * p.m_steps + ms) * 64 + lane];
#pragma unroll
        for (int q = 0; q < BWF_ND; ++q)
            o.ea[q] = *reinterpret_cast<const float4*>(p.eT + (size_t)min(d0 + q, p.D - 1) * Mpad + (size_t)ms * 16 + 4 * (lane >> 4));
    };
    auto consume = [&](const Operands& o) {
#pragma unroll
        for (int q = 0; q < BWF_ND; ++q) {
            const float ev[4] = {o.ea[q].x, o.ea[q].y, o.ea[q].z, o.ea[q].w};
            // swapped operands: rows = h (scaled gy), columns = t (x): a lane ends with 4 consecutive h of one t.  MFMAs that share an
            // accumulator are 12 instructions apart
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int j = 0; j < BWF_NT; ++j) {
                    const float sg = __builtin_bit_cast(f32x4_t, o.gb[j])[s4] * ev[s4];
#pragma unroll
                    for (int i = 0; i < BWF_MT; ++i)
                        acc[q][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(sg, __builtin_bit_cast(f32x4_t, o.xa[i])[s4], acc[q][i][j], 0, 0, 0);
                }
        }
    };
    Operands oa, ob;
    fetch(wave, oa);
    for (int ms = wave; ms < p.m_steps; ms += 2 * BL_WAVES) {
        fetch(ms + BL_WAVES, ob);
        consume(oa);
        fetch(ms + 2 * BL_WAVES, oa);
        if (ms + BL_WAVES < p.m_steps) consume(ob);          // wave-uniform, no load inside
    }
    for (int q = 0; q < BWF_ND; ++q) {
        __syncthreads();
#pragma unroll
        for (int q2 = 0; q2 < BWF_ND; ++q2)
#pragma unroll
            for (int i = 0; i < BWF_MT; ++i)
#pragma unroll
                for (int j = 0; j < BWF_NT; ++j)
                    if (q2 == q) red[wave][i * BWF_NT + j][lane] = acc[q2][i][j];      // static register indices
        __syncthreads();
        const int d = d0 + q;
        for (int tt = wave; tt < BWF_MT * BWF_NT; tt += BL_WAVES) {
            const int i = tt / BWF_NT, j = tt % BWF_NT;
            f32x4_t v = red[0][tt][lane];
#pragma unroll
            for (int w = 1; w < BL_WAVES; ++w) v += red[w][tt][lane];
            const int t = (t0 + i) * 16 + (lane & 15);
            const int h = (h0 + j) * 16 + 4 * (lane >> 4);
            if (d < p.D && t < p.Tin && t0 + i < p.t_tiles && h0 + j < p.h_tiles) {
                float* dst = p.dW + ((size_t)d * p.Tin + t) * p.H + h;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (h + g < p.H) dst[g] = v[g];
            }
        }
    }
}

}  // namespace gcnpt

using namespace gcnpt;

// k-step width of the traversal's fragments: 32 values per lane group of bf16, 16 of fp32
static int bl_kstep(int dtype) { return dtype == GCNPT_BF16 ? 32 : 16; }

extern "C" size_t gcnpt_bilinear_packed_bytes(int D, int Tin, int H, int dtype) {
    if (D <= 0 || Tin <= 0 || H <= 0 || !dtype_ok(dtype)) return 0;
    return (size_t)ceil_div(H, 16) * D * ceil_div(Tin, bl_kstep(dtype)) * 64 * 16;
}

// (Tin beyond what a wave's registers hold -- 256 in bf16 -- is cut into runs of k-steps, each run with planes of its own: see the
// `chunks` of the parameter structs.  The bound below only keeps the plane count sane.)
extern "C" int gcnpt_bilinear_supported(int D, int Tin, int H, int dtype) {
    if (!dtype_ok(dtype)) return 0;
    return D > 0 && H > 0 && Tin > 0 && ceil_div(Tin, bl_kstep(dtype)) <= 8 * (dtype == GCNPT_BF16 ? BL_TSMAX : BF_TSMAX);
}

extern "C" int gcnpt_bilinear_pack(void* stream, const float* W, int D, int Tin, int H, void* w_img, int transposed, int dtype) {
    GCNPT_REQUIRE(W && w_img, "bilinear_pack: null pointer");
    GCNPT_REQUIRE(D > 0 && Tin > 0 && H > 0 && dtype_ok(dtype), "bilinear_pack: sizes must be positive, dtype f32 or bf16");
    const int TS = ceil_div(transposed ? H : Tin, bl_kstep(dtype));
    const long long n = (long long)ceil_div(transposed ? Tin : H, 16) * D * TS * 64;
    const int grid = (int)std::min<long long>((n + 255) / 256, 1 << 16);
    if (dtype == GCNPT_BF16)
        hipLaunchKernelGGL(bilinear_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, D, Tin, H, TS, n, static_cast<uint4*>(w_img), transposed);
    else
        hipLaunchKernelGGL(bilinear_pack_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, D, Tin, H, TS, n, static_cast<uint4*>(w_img), transposed);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

static void bilinear_f32_plan(BilinearF32Params& p, int M, int D, int Tin, int H) {
    p.M = M; p.D = D; p.H = H; p.TSA = ceil_div(Tin, 16); p.n_tiles = ceil_div(H, 16);
    p.chunks = ceil_div(p.TSA, BF_TSMAX); p.TS = ceil_div(p.TSA, p.chunks);
    p.mb = ceil_div(M, BF_ROWS); p.nb = ceil_div(p.n_tiles, BF_NT);
    int slices = std::max(1, std::min(256 / std::max(1, p.mb * p.nb * p.chunks), D / 4));       // one workgroup per CU, one round
    slices = std::max(slices, ceil_div(D, 64));                 // the e tile of a slice must fit LDS beside the two weight stages
    p.d_per_slice = ceil_div(D, slices);
    p.slices = ceil_div(D, p.d_per_slice);
}

template <int MODE>
static int bilinear_f32_launch(hipStream_t s, BilinearF32Params& p) {
    const size_t lds = (size_t)2 * BF_NT * BF_TSMAX * 64 * sizeof(uint4) + sizeof(float) * (BF_ROWS + 1) * (size_t)p.d_per_slice;
    if (lds > 160 * 1024) return fail(GCNPT_E_UNSUPPORTED, "bilinear (fp32): %zu B of LDS", lds);
    GCNPT_LDS_ATTR_ONCE(bilinear_f32_kernel<MODE>, 160 * 1024);
    hipLaunchKernelGGL(bilinear_f32_kernel<MODE>, dim3(p.mb * p.nb * p.slices * p.chunks), dim3(BL_THREADS), lds, s, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

static void bilinear_plan(BilinearParams& p, int M, int D, int Tin, int H) {
    p.M = M; p.D = D; p.H = H; p.TSA = ceil_div(Tin, 32); p.n_tiles = ceil_div(H, 16);
    p.chunks = ceil_div(p.TSA, BL_TSMAX); p.TS = ceil_div(p.TSA, p.chunks);
    p.mb = ceil_div(M, BL_ROWS); p.nb = ceil_div(p.n_tiles, BL_NT);
    // relation slices: about one workgroup per CU (each holds ~120 KB of LDS and all registers), at least 4 relations each
    int slices = std::max(1, std::min(256 / std::max(1, p.mb * p.nb * p.chunks), D / 4));
    slices = std::max(slices, ceil_div(D, 48));                 // the e tile of a slice must fit LDS beside the ring (48 x 257 words)
    p.d_per_slice = ceil_div(D, slices);
    p.slices = ceil_div(D, p.d_per_slice);
}

extern "C" int gcnpt_bilinear_planes(int M, int D, int Tin, int H, int dtype) {
    if (M <= 0 || !gcnpt_bilinear_supported(D, Tin, H, dtype)) return 0;
    if (dtype == GCNPT_F32) {
        BilinearF32Params q{};
        bilinear_f32_plan(q, M, D, Tin, H);
        return q.slices * q.chunks;
    }
    BilinearParams p{};
    bilinear_plan(p, M, D, Tin, H);
    return p.slices * p.chunks;
}

template <int MODE>
static int bilinear_launch(hipStream_t s, BilinearParams& p) {
    const size_t lds = ((size_t)BL_RING * BL_FRAGS + BL_WAVES) * 64 * sizeof(uint4) + sizeof(float) * (BL_ROWS + 1) * (size_t)p.d_per_slice;
    if (lds > 160 * 1024) return fail(GCNPT_E_UNSUPPORTED, "bilinear: %zu B of LDS", lds);
    GCNPT_LDS_ATTR_ONCE(bilinear_fwd_kernel<MODE>, 160 * 1024);
    hipLaunchKernelGGL(bilinear_fwd_kernel<MODE>, dim3(p.mb * p.nb * p.slices * p.chunks), dim3(BL_THREADS), lds, s, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_bilinear_fwd(void* stream, const void* x, const float* e, const void* w_img, int M, int D, int Tin, int H,
                                  float* y_planes, int dtype) {
    GCNPT_REQUIRE(x && e && w_img && y_planes, "bilinear_fwd: null pointer");
    GCNPT_REQUIRE(M > 0 && D > 0 && Tin > 0 && H > 0, "bilinear_fwd: sizes must be positive");
    GCNPT_REQUIRE(aligned16(x) && aligned16(w_img) && aligned16(y_planes), "bilinear_fwd: x, w_img and y_planes must be 16-byte aligned");
    if (!gcnpt_bilinear_supported(D, Tin, H, dtype))
        return fail(GCNPT_E_UNSUPPORTED, "bilinear_fwd: Tin=%d is beyond what the traversal kernels are planned for", Tin);
    if (dtype == GCNPT_F32) {
        BilinearF32Params q{};
        q.x = static_cast<const float*>(x); q.e = e; q.img = static_cast<const uint4*>(w_img); q.y = y_planes;
        bilinear_f32_plan(q, M, D, Tin, H);
        return bilinear_f32_launch<0>((hipStream_t)stream, q);
    }
    BilinearParams p{};
    p.x = static_cast<const bf16_t*>(x); p.e = e; p.img = static_cast<const uint4*>(w_img); p.y = y_planes;
    bilinear_plan(p, M, D, Tin, H);
    return bilinear_launch<0>((hipStream_t)stream, p);
}

extern "C" int gcnpt_bilinear_de_planes(int M, int D, int Tin, int H, int dtype) {
    if (M <= 0 || !gcnpt_bilinear_supported(D, Tin, H, dtype)) return 0;
    const int tsa = ceil_div(Tin, bl_kstep(dtype));
    return ceil_div(ceil_div(H, 16), dtype == GCNPT_BF16 ? BL_NT : BF_NT) * ceil_div(tsa, dtype == GCNPT_BF16 ? BL_TSMAX : BF_TSMAX);
}

extern "C" int gcnpt_bilinear_bwd_e(void* stream, const void* x, const float* gy, const void* w_img, int M, int D, int Tin, int H,
                                    float* de_planes, int dtype) {
    GCNPT_REQUIRE(x && gy && w_img && de_planes, "bilinear_bwd_e: null pointer");
    GCNPT_REQUIRE(M > 0 && D > 0 && Tin > 0 && H > 0, "bilinear_bwd_e: sizes must be positive");
    GCNPT_REQUIRE(aligned16(x) && aligned16(w_img), "bilinear_bwd_e: x and w_img must be 16-byte aligned");
    if (!gcnpt_bilinear_supported(D, Tin, H, dtype))
        return fail(GCNPT_E_UNSUPPORTED, "bilinear_bwd_e: Tin=%d is beyond what the traversal kernels are planned for", Tin);
    if (dtype == GCNPT_F32) {
        BilinearF32Params q{};
        q.x = static_cast<const float*>(x); q.e = nullptr; q.gy = gy; q.img = static_cast<const uint4*>(w_img); q.y = de_planes;
        bilinear_f32_plan(q, M, D, Tin, H);
        return bilinear_f32_launch<1>((hipStream_t)stream, q);
    }
    BilinearParams p{};
    p.x = static_cast<const bf16_t*>(x); p.e = nullptr; p.gy = gy;
    p.img = static_cast<const uint4*>(w_img); p.y = de_planes;
    bilinear_plan(p, M, D, Tin, H);
    return bilinear_launch<1>((hipStream_t)stream, p);
}

extern "C" size_t gcnpt_rows_image_bytes(int M, int W, int dtype) {
    if (M <= 0 || W <= 0 || !dtype_ok(dtype)) return 0;
    return (size_t)ceil_div(W, 16) * ceil_div(M, bl_kstep(dtype)) * 64 * 16;
}

extern "C" int gcnpt_rows_pack(void* stream, const float* src, int M, int W, void* img, int dtype) {
    GCNPT_REQUIRE(src && img, "rows_pack: null pointer");
    GCNPT_REQUIRE(M > 0 && W > 0 && dtype_ok(dtype), "rows_pack: sizes must be positive, dtype f32 or bf16");
    const int m_steps = ceil_div(M, bl_kstep(dtype));
    const long long n = (long long)ceil_div(W, 16) * m_steps * 64;
    const int grid = (int)std::min<long long>((n + 255) / 256, 1 << 16);
    if (dtype == GCNPT_BF16) hipLaunchKernelGGL(rows_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, M, W, m_steps, n, static_cast<uint4*>(img));
    else hipLaunchKernelGGL(rows_pack_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, M, W, m_steps, n, static_cast<uint4*>(img));
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_bilinear_bwd_w(void* stream, const void* x_img, const void* gy_img, const float* eT, int M, int D, int Tin, int H,
                                    float* dW, int dtype) {
    GCNPT_REQUIRE(x_img && gy_img && eT && dW, "bilinear_bwd_w: null pointer");
    GCNPT_REQUIRE(M > 0 && D > 0 && Tin > 0 && H > 0 && dtype_ok(dtype), "bilinear_bwd_w: sizes must be positive, dtype f32 or bf16");
    GCNPT_REQUIRE(aligned16(x_img) && aligned16(gy_img) && aligned16(eT), "bilinear_bwd_w: images and eT must be 16-byte aligned");
    BilinearDwParams p{};
    p.xI = static_cast<const uint4*>(x_img); p.gI = static_cast<const uint4*>(gy_img); p.eT = eT; p.dW = dW;
    p.D = D; p.Tin = Tin; p.H = H; p.t_tiles = ceil_div(Tin, 16); p.h_tiles = ceil_div(H, 16); p.m_steps = ceil_div(M, bl_kstep(dtype));
    const bool bf = dtype == GCNPT_BF16;
    p.tb = ceil_div(p.t_tiles, bf ? BW_MT : BWF_MT); p.hb = ceil_div(p.h_tiles, bf ? BW_NT : BWF_NT); p.dgroups = ceil_div(D, bf ? BW_ND : BWF_ND);
    const long long blocks = (long long)p.tb * p.hb * p.dgroups;
    if (blocks > 0x7fffffffLL) return fail(GCNPT_E_UNSUPPORTED, "bilinear_bwd_w: too many blocks");
    if (bf) hipLaunchKernelGGL(bilinear_dw_kernel, dim3((unsigned)blocks), dim3(BL_THREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(bilinear_dw_f32_kernel, dim3((unsigned)blocks), dim3(BL_THREADS), 0, (hipStream_t)stream, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}
