// adj_type == 'diagonal_deprel' layer: reference model/gcn.py:272-294 + 390-393.
// There is no weight matrix in this variant: every neighbour row is scaled ELEMENT-WISE by the embedding of a
// dependency relation and summed, so the layer is pure gather + FMA, HBM/latency bound, no MFMA:
//   z[r] = sum_{c: 0 < adj[r,c] < 42} E[deprel[c]] * h[c] + sum_{c: 42 < adj[r,c] < 84} E[deprel[c]+42] * h[c] + E[84] * h[r]
//   out[r] = dropout(relu(z[r] / (deg[r] + 1)))
// (the COLUMN token's relation picks the embedding row, as the reference does; adj values 42 and 84 fall in neither
// range, so the adjacency's own diagonal contributes nothing and E[84] * h[r] is applied to EVERY row, in-tree or not.)
//
// Forward : one wave per row, 4 (or 1) columns per lane, CSR row with labels.
// Backward: GATHER form over the transposed CSR, so dh is written exactly once per element (no atomics, no memset):
//   dz[r]  = dY[r] * (Y[r] > 0) * scale / (deg[r] + 1)
//   uf[c]  = sum_{r: 0 < adj[r,c] < 42} dz[r]        ur[c] = sum_{r: 42 < adj[r,c] < 84} dz[r]
//   dh[c]  = uf[c] * E[deprel[c]] + ur[c] * E[deprel[c]+42] + dz[c] * E[84]
//   dE[deprel[c]] += uf[c] * h[c]   dE[deprel[c]+42] += ur[c] * h[c]   (fp32 atomics, only tokens that have such edges)
//   dE[84] += sum_c dz[c] * h[c]    (its own column-reduction launch: 64 atomics per element instead of one per workgroup)
// adj[r,c] for r in column c's transposed list is looked up in row r's (short) CSR segment.
#include "layer_common.h"

namespace gcnpt {

constexpr int DG_THREADS = 256;
constexpr int DG_WAVES = DG_THREADS / 64;
constexpr int DG_ROWS = DG_WAVES;           // rows per block: one per wave, so the dependent index chains of different rows overlap
constexpr int DG_SELF_BLOCKS = 32;          // workgroups of the dE[84] column reduction (= float atomics per element of that row:
                                            // same-address device atomics cost ~0.1 us each, 128 of them were slower than 64)
constexpr int DG_SELF_WAVES = 8;
constexpr int DG_FWD = 42, DG_REV = 84, DG_SELF = 84, DG_NE = 85;   // utils/constant.py:14-17 and len(DEPREL_TO_ID)*2+1

struct DiagParams {
    const void* h;            // layer input [N,H]
    const void* y;            // bwd: layer output (post dropout) [N,H]
    const void* dy;           // bwd: its gradient
    const float* E;           // [85,H]
    const int64_t* deprel;    // [N]
    const int32_t *row_ptr, *col_idx, *label, *rowT_ptr, *colT_idx;
    void* out;                // fwd: [N,H]      bwd: dh [N,H]
    float* dE;                // bwd: [85,H], accumulated
    int N, T, H;
    float scale;
    unsigned drop_thresh16;
    uint64_t seed;
    const uint64_t* seed_dev;  // NULL or a device word added to seed
};

// embedding row picked by adj value `lab` for column token relation `rel`: -1 = the entry contributes nothing
__device__ __forceinline__ int diag_emb_row(int lab, int rel) {
    int id = -1;
    if (lab > 0 && lab < DG_FWD) id = rel;                       // gcn.py:276-280
    else if (lab > DG_FWD && lab < DG_REV) id = rel + DG_FWD;    // gcn.py:281-288
    return id < 0 ? -1 : min(id, DG_NE - 1);                      // memory safety for relation ids the vocabulary lacks
}
__device__ __forceinline__ int diag_rel(const int64_t* deprel, int i) { return (int)max((int64_t)0, min(deprel[i], (int64_t)(DG_NE - 1))); }

template <typename T, int CPL>
__global__ __launch_bounds__(DG_THREADS) void diag_fwd_kernel(const DiagParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const T* h = static_cast<const T*>(p.h);
    T* out = static_cast<T*>(p.out);
    const int H = p.H;
    const uint64_t seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
    for (int rr = wave; rr < DG_ROWS; rr += DG_WAVES) {
        const int r = blockIdx.x * DG_ROWS + rr;
        if (r >= p.N) break;
        const int b = r / p.T, t = r - b * p.T;
        const int beg = p.row_ptr[b * (p.T + 1) + t], end = p.row_ptr[b * (p.T + 1) + t + 1];
        const float den = (float)(end - beg + 1);                                       // gcn.py:138 + 390
        for (int c0 = lane * CPL; c0 < round_up(H, 64 * CPL); c0 += 64 * CPL) {
            const int cc = min(c0, H - CPL), live = c0 < H;
            float acc[CPL], hv[CPL], ev[CPL];
            dgio<T, CPL>::ld(h + (size_t)r * H + cc, live, hv);
            dgio<float, CPL>::ld(p.E + (size_t)DG_SELF * H + cc, live, ev);
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[j] = ev[j] * hv[j];                      // gcn.py:289-294
            for (int e = beg; e < end; ++e) {
                const int c = p.col_idx[e];
                const int id = diag_emb_row(p.label[e], diag_rel(p.deprel, b * p.T + c));
                if (id < 0) continue;                                                   // wave-uniform
                dgio<T, CPL>::ld(h + ((size_t)b * p.T + c) * H + cc, live, hv);
                dgio<float, CPL>::ld(p.E + (size_t)id * H + cc, live, ev);
#pragma unroll
                for (int j = 0; j < CPL; ++j) acc[j] = __builtin_fmaf(ev[j], hv[j], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                float x = acc[j] / den;
                x = x > 0.0f ? x : 0.0f;
                if (p.scale != 1.0f) {                                                  // gcn.py:393
                    const unsigned col = (unsigned)(cc + j);
                    x = drop_keep(drop_hash(seed, (unsigned)r, col >> 1), col & 1u, p.drop_thresh16) ? x * p.scale : 0.0f;
                }
                acc[j] = x;
            }
            if (live) dgio<T, CPL>::st(out + (size_t)r * H + cc, live, acc);
        }
    }
}

template <typename T, int CPL>
__global__ __launch_bounds__(DG_THREADS) void diag_bwd_kernel(const DiagParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const T* h = static_cast<const T*>(p.h);
    const T* Y = static_cast<const T*>(p.y);
    const T* dY = static_cast<const T*>(p.dy);
    T* dh = static_cast<T*>(p.out);
    const int H = p.H;
    for (int c0 = lane * CPL; c0 < round_up(H, 64 * CPL); c0 += 64 * CPL) {              // uniform trip count over the block
        const int cc = min(c0, H - CPL), live = c0 < H;
        float eself[CPL];
        dgio<float, CPL>::ld(p.E + (size_t)DG_SELF * H + cc, live, eself);
        for (int rr = wave; rr < DG_ROWS; rr += DG_WAVES) {
            const int c = blockIdx.x * DG_ROWS + rr;                                      // this wave's token (a COLUMN of adj)
            if (c >= p.N) break;
            const int b = c / p.T, t = c - b * p.T;
            const int rp = b * (p.T + 1);
            const int rel = diag_rel(p.deprel, c);
            float uf[CPL], ur[CPL], dz[CPL], yv[CPL], gv[CPL], hv[CPL], out[CPL];
            bool any_f = false, any_r = false;
#pragma unroll
            for (int j = 0; j < CPL; ++j) uf[j] = ur[j] = 0.0f;
            {                                                                             // own dz
                const float s = p.scale / (float)(p.row_ptr[rp + t + 1] - p.row_ptr[rp + t] + 1);
                dgio<T, CPL>::ld(Y + (size_t)c * H + cc, live, yv);
                dgio<T, CPL>::ld(dY + (size_t)c * H + cc, live, gv);
#pragma unroll
                for (int j = 0; j < CPL; ++j) dz[j] = yv[j] > 0.0f ? gv[j] * s : 0.0f;
            }
            for (int e = p.rowT_ptr[rp + t]; e < p.rowT_ptr[rp + t + 1]; ++e) {
                const int tr = p.colT_idx[e];                                             // adj[b, tr, t] != 0
                const int rb = p.row_ptr[rp + tr], re = p.row_ptr[rp + tr + 1];
                int lab = 0;
                for (int q = rb; q < re; ++q) lab = p.col_idx[q] == t ? p.label[q] : lab;
                const int id = diag_emb_row(lab, rel);
                if (id < 0) continue;                                                     // wave-uniform
                const float s = p.scale / (float)(re - rb + 1);
                const size_t r = (size_t)b * p.T + tr;
                dgio<T, CPL>::ld(Y + r * H + cc, live, yv);
                dgio<T, CPL>::ld(dY + r * H + cc, live, gv);
                if (lab < DG_FWD) {
                    any_f = true;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) uf[j] += yv[j] > 0.0f ? gv[j] * s : 0.0f;
                } else {
                    any_r = true;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) ur[j] += yv[j] > 0.0f ? gv[j] * s : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) { out[j] = dz[j] * eself[j]; hv[j] = 0.0f; }
            if (any_f || any_r) dgio<T, CPL>::ld(h + (size_t)c * H + cc, live, hv);      // wave-uniform: only tokens with edges
            if (any_f) {
                const int id = min(rel, DG_NE - 1);
                float ev[CPL];
                dgio<float, CPL>::ld(p.E + (size_t)id * H + cc, live, ev);
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    out[j] = __builtin_fmaf(uf[j], ev[j], out[j]);
                    if (live) atomicAdd(p.dE + (size_t)id * H + cc + j, uf[j] * hv[j]);
                }
            }
            if (any_r) {
                const int id = min(rel + DG_FWD, DG_NE - 1);
                float ev[CPL];
                dgio<float, CPL>::ld(p.E + (size_t)id * H + cc, live, ev);
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    out[j] = __builtin_fmaf(ur[j], ev[j], out[j]);
                    if (live) atomicAdd(p.dE + (size_t)id * H + cc + j, ur[j] * hv[j]);
                }
            }
            if (live) dgio<T, CPL>::st(dh + (size_t)c * H + cc, live, out);
        }
    }
}

// dE[84] += sum_c dz[c] * h[c]: a column reduction over all N rows on its own, so that one element of that row receives
// DG_SELF_BLOCKS float atomics instead of one per workgroup of the kernel above (same-address atomics serialise)
template <typename T, int CPL>
__global__ __launch_bounds__(DG_SELF_WAVES * 64) void diag_selfgrad_kernel(const DiagParams p) {
    __shared__ float red[DG_SELF_WAVES][64 * CPL];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const T* h = static_cast<const T*>(p.h);
    const T* Y = static_cast<const T*>(p.y);
    const T* dY = static_cast<const T*>(p.dy);
    const int H = p.H;
    const int per = ceil_div(p.N, (int)gridDim.x);
    const int r_lo = blockIdx.x * per, r_hi = min(p.N, r_lo + per);
    const int c0 = (blockIdx.y * 64 + lane) * CPL;
    const int cc = min(c0, H - CPL), live = c0 < H;
    float acc[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] = 0.0f;
    constexpr int RU = 4;                                                  // rows per round: twelve loads in flight
    for (int r0 = r_lo + wave; r0 < r_hi; r0 += RU * DG_SELF_WAVES) {
        float yv[RU][CPL], gv[RU][CPL], hv[RU][CPL], s[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int r = min(r0 + u * DG_SELF_WAVES, r_hi - 1);
            const int b = r / p.T, t = r - b * p.T;
            dgio<T, CPL>::ld(Y + (size_t)r * H + cc, live, yv[u]);
            dgio<T, CPL>::ld(dY + (size_t)r * H + cc, live, gv[u]);
            dgio<T, CPL>::ld(h + (size_t)r * H + cc, live, hv[u]);
            s[u] = p.scale / (float)(p.row_ptr[b * (p.T + 1) + t + 1] - p.row_ptr[b * (p.T + 1) + t] + 1);
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const bool on = r0 + u * DG_SELF_WAVES < r_hi;
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[j] += (on && yv[u][j] > 0.0f) ? gv[u][j] * s[u] * hv[u][j] : 0.0f;
        }
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) red[wave][lane * CPL + j] = acc[j];
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < DG_SELF_WAVES; ++w) v += red[w][lane * CPL + j];
            atomicAdd(p.dE + (size_t)DG_SELF * H + cc + j, v);
        }
    }
}

template <bool BWD>
static int launch_diag(hipStream_t s, const DiagParams& p, int dtype, bool vec) {
    const dim3 grid(ceil_div(p.N, DG_ROWS)), block(DG_THREADS);
#define GCNPT_DIAG_LAUNCH(T, CPL)                                                             \
    do {                                                                                      \
        if (BWD) {                                                                            \
            hipLaunchKernelGGL((diag_bwd_kernel<T, CPL>), grid, block, 0, s, p);              \
            hipLaunchKernelGGL((diag_selfgrad_kernel<T, CPL>), dim3(std::min(DG_SELF_BLOCKS, p.N), ceil_div(p.H, 64 * CPL)), dim3(DG_SELF_WAVES * 64), 0, s, p); \
        } else hipLaunchKernelGGL((diag_fwd_kernel<T, CPL>), grid, block, 0, s, p);           \
    } while (0)
    if (dtype == GCNPT_F32) {
        if (vec) GCNPT_DIAG_LAUNCH(float, 4); else GCNPT_DIAG_LAUNCH(float, 1);
    } else {
        if (vec) GCNPT_DIAG_LAUNCH(bf16_t, 4); else GCNPT_DIAG_LAUNCH(bf16_t, 1);
    }
#undef GCNPT_DIAG_LAUNCH
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

}  // namespace gcnpt

using namespace gcnpt;

extern "C" int gcnpt_diag_layer_fwd(void* stream, const void* h, int dtype, const float* E, const int64_t* deprel,
                                    const int32_t* row_ptr, const int32_t* col_idx, const int32_t* label, int B, int T, int H,
                                    void* out, float drop_p, uint64_t seed, const uint64_t* seed_dev) {
    GCNPT_REQUIRE(h && E && deprel && row_ptr && col_idx && label && out, "diag_layer_fwd: null pointer");
    GCNPT_REQUIRE(B > 0 && T > 0 && H > 0 && dtype_ok(dtype), "diag_layer_fwd: bad argument");
    GCNPT_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "diag_layer_fwd: drop_p must be in [0,1)");
    if ((long long)B * T > 0x7fffffffLL / 2) return fail(GCNPT_E_UNSUPPORTED, "diag_layer_fwd: B*T too large");
    DiagParams p{};
    p.h = h; p.E = E; p.deprel = deprel; p.row_ptr = row_ptr; p.col_idx = col_idx; p.label = label; p.out = out;
    p.N = B * T; p.T = T; p.H = H;
    p.scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    p.drop_thresh16 = (unsigned)((double)drop_p * 65536.0);
    p.seed = seed; p.seed_dev = seed_dev;
    const bool vec = H % 4 == 0 && aligned16(h) && aligned16(out) && aligned16(E);
    return launch_diag<false>((hipStream_t)stream, p, dtype, vec);
}

extern "C" int gcnpt_diag_layer_bwd(void* stream, const void* dY, const void* Y, const void* h, int dtype, const float* E,
                                    const int64_t* deprel, const int32_t* row_ptr, const int32_t* col_idx, const int32_t* label,
                                    const int32_t* rowT_ptr, const int32_t* colT_idx, int B, int T, int H, void* dh, float* dE,
                                    float scale) {
    GCNPT_REQUIRE(dY && Y && h && E && deprel && row_ptr && col_idx && label && rowT_ptr && colT_idx && dh && dE, "diag_layer_bwd: null pointer");
    GCNPT_REQUIRE(B > 0 && T > 0 && H > 0 && dtype_ok(dtype), "diag_layer_bwd: bad argument");
    if ((long long)B * T > 0x7fffffffLL / 2) return fail(GCNPT_E_UNSUPPORTED, "diag_layer_bwd: B*T too large");
    DiagParams p{};
    p.h = h; p.y = Y; p.dy = dY; p.E = E; p.deprel = deprel; p.row_ptr = row_ptr; p.col_idx = col_idx; p.label = label;
    p.rowT_ptr = rowT_ptr; p.colT_idx = colT_idx; p.out = dh; p.dE = dE;
    p.N = B * T; p.T = T; p.H = H; p.scale = scale;
    const bool vec = H % 4 == 0 && aligned16(h) && aligned16(Y) && aligned16(dY) && aligned16(dh) && aligned16(E);
    return launch_diag<true>((hipStream_t)stream, p, dtype, vec);
}
