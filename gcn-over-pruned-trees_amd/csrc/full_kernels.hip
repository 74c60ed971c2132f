// adj_type == 'full_deprel' (SURVEY.md 8f row N3), the part AROUND the relation-conditioned contraction: reference
// model/gcn.py:308-311 + 331 (forward edges), 340-344 + 362 (reverse edges), 366-385 (self loop), 390-393 (normalise, ReLU, dropout).
//
// The reference builds two dense [B,T,T] 0/1 matrices from value ranges of the labelled adjacency and multiplies them with the
// traversed encodings of ALL tokens.  Here the traversal (csrc/bilinear_kernels.hip, or one library GEMM in fp32 mode) has been
// evaluated for the M tokens that sit in a pruned tree only, Yf / Yr [M,H], and ONE kernel does the rest of the layer:
//   agg[r] = sum_{k in row r, 0 < label_k < 42} kf_k * Yf[pos[col_k]] + sum_{k, 42 < label_k < 84} kr_k * Yr[pos[col_k]] + self[r]
//   out[r] = dropout(relu(agg[r] / (deg[r] + 1)))
// (kf / kr: the training-time edge dropout of gcn.py:436-449, one keep flag per CSR slot and direction; pos: token -> row of Yf / Yr;
// self: the self-loop traversal, a plain [N,Tin] x [Tin,H] product computed by the host's BLAS, or NULL.)
// Backward: dagg[r] = dY[r] * 1[out[r] > 0] * scale / (deg[r] + 1) is written out (it IS the gradient of `self`) and scattered to
// dYf / dYr along the same entries with float atomics (a token's traversed row is read by its parent or its children only: 1-3 adds).
// One wave per row, 4 (or 1) columns per lane; HBM / latency bound, no MFMA.
#include "layer_common.h"

namespace gcnpt {

constexpr int FA_THREADS = 256;
constexpr int FA_ROWS = FA_THREADS / 64;
constexpr int FA_FWD = 42, FA_REV = 84;        // utils/constant.py:14,16

struct FullAggParams {
    const float *yf, *yr, *self_term;          // [M,H], [M,H] or NULL (directed), [N,H] or NULL
    const int32_t* pos;                        // [N] token -> row of yf / yr (entries only ever name tokens of a tree)
    const int32_t *row_ptr, *col_idx, *label;
    const uint8_t *keep_f, *keep_r;            // NULL or one flag per CSR slot (edge dropout)
    const float *dy, *y;                       // bwd: gradient of out, out
    float *out;                                // fwd: [N,H]   bwd: dagg [N,H]
    float *dyf, *dyr;                          // bwd: [M,H] accumulated (cleared by the caller); dyr NULL when directed
    int N, T, H, M;
    float scale;                               // 1/(1-p) of the dropout applied to out
    unsigned drop_thresh16;
    uint64_t seed;
    const uint64_t* seed_dev;
};

template <int CPL>
__global__ __launch_bounds__(FA_THREADS) void full_agg_fwd_kernel(const FullAggParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * FA_ROWS + wave;
    if (r >= p.N) return;
    const int H = p.H;
    const uint64_t seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
    const int b = r / p.T, t = r - b * p.T;
    const int beg = p.row_ptr[b * (p.T + 1) + t], end = p.row_ptr[b * (p.T + 1) + t + 1];
    const float den = (float)(end - beg + 1);                                              // gcn.py:261 + 390
    for (int c0 = lane * CPL; c0 < round_up(H, 64 * CPL); c0 += 64 * CPL) {
        const int cc = min(c0, H - CPL), live = c0 < H;
        float acc[CPL], v[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] = 0.0f;
        if (p.self_term) dgio<float, CPL>::ld(p.self_term + (size_t)r * H + cc, live, acc);   // gcn.py:385
        for (int e = beg; e < end; ++e) {
            const int lab = p.label[e];
            const bool fw = lab > 0 && lab < FA_FWD, rv = lab > FA_FWD && lab < FA_REV;   // gcn.py:308-311, 340-344 (42 and 84 fall in neither)
            if (!(fw || (rv && p.yr))) continue;                                           // wave-uniform
            const uint8_t* keep = fw ? p.keep_f : p.keep_r;
            if (keep && !keep[e]) continue;                                                // gcn.py:436-449
            const int m = p.pos[(size_t)b * p.T + p.col_idx[e]];
            if (m < 0 || m >= p.M) continue;
            dgio<float, CPL>::ld((fw ? p.yf : p.yr) + (size_t)m * H + cc, live, v);
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[j] += v[j];                                  // gcn.py:331, 362
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            float x = acc[j] / den;                                                        // gcn.py:390
            x = x > 0.0f ? x : 0.0f;                                                       // gcn.py:392
            if (p.scale != 1.0f) {                                                         // gcn.py:393
                const unsigned col = (unsigned)(cc + j);
                x = drop_keep(drop_hash(seed, (unsigned)r, col >> 1), col & 1u, p.drop_thresh16) ? x * p.scale : 0.0f;
            }
            acc[j] = x;
        }
        if (live) dgio<float, CPL>::st(p.out + (size_t)r * H + cc, live, acc);
    }
}

template <int CPL>
__global__ __launch_bounds__(FA_THREADS) void full_agg_bwd_kernel(const FullAggParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * FA_ROWS + wave;
    if (r >= p.N) return;
    const int H = p.H;
    const int b = r / p.T, t = r - b * p.T;
    const int beg = p.row_ptr[b * (p.T + 1) + t], end = p.row_ptr[b * (p.T + 1) + t + 1];
    const float f = p.scale / (float)(end - beg + 1);
    for (int c0 = lane * CPL; c0 < round_up(H, 64 * CPL); c0 += 64 * CPL) {
        const int cc = min(c0, H - CPL), live = c0 < H;
        float g[CPL], yv[CPL];
        dgio<float, CPL>::ld(p.dy + (size_t)r * H + cc, live, g);
        dgio<float, CPL>::ld(p.y + (size_t)r * H + cc, live, yv);
#pragma unroll
        for (int j = 0; j < CPL; ++j) g[j] = yv[j] > 0.0f ? g[j] * f : 0.0f;
        if (live) dgio<float, CPL>::st(p.out + (size_t)r * H + cc, live, g);               // = d self_term
        for (int e = beg; e < end; ++e) {
            const int lab = p.label[e];
            const bool fw = lab > 0 && lab < FA_FWD, rv = lab > FA_FWD && lab < FA_REV;
            if (!(fw || (rv && p.dyr))) continue;
            const uint8_t* keep = fw ? p.keep_f : p.keep_r;
            if (keep && !keep[e]) continue;
            const int m = p.pos[(size_t)b * p.T + p.col_idx[e]];
            if (m < 0 || m >= p.M) continue;
            float* dst = (fw ? p.dyf : p.dyr) + (size_t)m * H + cc;
            if (live) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) atomicAdd(dst + j, g[j]);
            }
        }
    }
}

}  // namespace gcnpt

using namespace gcnpt;

static int full_agg_check(const char* what, const void* a, const void* pos, const void* rp, const void* ci, const void* lab, const void* out,
                          int B, int T, int H, int M) {
    GCNPT_REQUIRE(a && pos && rp && ci && lab && out, "%s: null pointer", what);
    GCNPT_REQUIRE(B > 0 && T > 0 && H > 0 && M >= 0, "%s: sizes must be positive", what);
    return GCNPT_OK;
}

extern "C" int gcnpt_full_agg_fwd(void* stream, const float* yf, const float* yr, const float* self_term, const int32_t* pos,
                                  const int32_t* row_ptr, const int32_t* col_idx, const int32_t* label, const uint8_t* keep_f,
                                  const uint8_t* keep_r, int B, int T, int H, int M, float* out, float drop_p, uint64_t seed,
                                  const uint64_t* seed_dev) {
    const int rc = full_agg_check("full_agg_fwd", yf, pos, row_ptr, col_idx, label, out, B, T, H, M);
    if (rc != GCNPT_OK) return rc;
    GCNPT_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "full_agg_fwd: drop_p=%f outside [0,1)", (double)drop_p);
    FullAggParams p{};
    p.yf = yf; p.yr = yr; p.self_term = self_term; p.pos = pos; p.row_ptr = row_ptr; p.col_idx = col_idx; p.label = label;
    p.keep_f = keep_f; p.keep_r = keep_r; p.out = out; p.N = B * T; p.T = T; p.H = H; p.M = M;
    p.scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    p.drop_thresh16 = (unsigned)((double)drop_p * 65536.0);
    p.seed = seed; p.seed_dev = seed_dev;
    const bool vec = H % 4 == 0 && aligned16(yf) && (!yr || aligned16(yr)) && (!self_term || aligned16(self_term)) && aligned16(out);
    const dim3 grid(ceil_div(p.N, FA_ROWS));
    if (vec) hipLaunchKernelGGL(full_agg_fwd_kernel<4>, grid, dim3(FA_THREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(full_agg_fwd_kernel<1>, grid, dim3(FA_THREADS), 0, (hipStream_t)stream, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_full_agg_bwd(void* stream, const float* dy, const float* y, const int32_t* pos, const int32_t* row_ptr,
                                  const int32_t* col_idx, const int32_t* label, const uint8_t* keep_f, const uint8_t* keep_r, int B, int T,
                                  int H, int M, float scale, float* dagg, float* dyf, float* dyr) {
    const int rc = full_agg_check("full_agg_bwd", dy, pos, row_ptr, col_idx, label, dagg, B, T, H, M);
    if (rc != GCNPT_OK) return rc;
    GCNPT_REQUIRE(y && dyf, "full_agg_bwd: null pointer");
    FullAggParams p{};
    p.dy = dy; p.y = y; p.pos = pos; p.row_ptr = row_ptr; p.col_idx = col_idx; p.label = label; p.keep_f = keep_f; p.keep_r = keep_r;
    p.out = dagg; p.dyf = dyf; p.dyr = dyr; p.N = B * T; p.T = T; p.H = H; p.M = M; p.scale = scale;
    const bool vec = H % 4 == 0 && aligned16(dy) && aligned16(y) && aligned16(dagg);
    const dim3 grid(ceil_div(p.N, FA_ROWS));
    if (vec) hipLaunchKernelGGL(full_agg_bwd_kernel<4>, grid, dim3(FA_THREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(full_agg_bwd_kernel<1>, grid, dim3(FA_THREADS), 0, (hipStream_t)stream, p);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}
