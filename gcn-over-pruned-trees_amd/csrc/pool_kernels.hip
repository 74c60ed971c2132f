// Fused masked pooling of the GCN output: reference model/gcn.py:116-121 runs pool() three times
// (all in-tree tokens, subject tokens, object tokens; model/gcn.py:473-483), i.e. three masked_fill + reduce passes over
// h [B,T,H] plus the mask tensors.  Here ONE pass reads h once and writes the concatenated [B, 3H] row the output MLP
// consumes; the subject / object masks are taken straight from the position tensors (pos != 0 <=> masked, gcn.py:116).
//   max : out = max over unmasked t (ties: first t, as torch.max(dim) returns); all masked -> -1e12 (INFINITY_NUMBER)
//   avg : sum over unmasked / (T - #masked)          sum : sum over unmasked
// Backward routes the gradient to the recorded argmax (max) or to every unmasked token (avg / sum).
#include "layer_common.h"

namespace gcnpt {

constexpr int POOL_THREADS = 512;
constexpr int POOL_WAVES = POOL_THREADS / WAVE;
constexpr int POOL_LANES = 16;                           // lanes across the columns of a workgroup (x CPL columns each)
constexpr int POOL_STREAMS = POOL_THREADS / POOL_LANES;  // 32 token streams: stream s takes tokens s, s + 32, ...
constexpr float POOL_NEG = -1e12f;     // utils/constant.py:35 INFINITY_NUMBER

// bit k of the result = token t is MASKED for pooling k (0: pool_mask, 1: not a subject token, 2: not an object token)
__device__ __forceinline__ int pool_mask_bits(const uint8_t* pm, const int64_t* sp, const int64_t* op, size_t i) {
    return (pm[i] ? 1 : 0) | (sp[i] != 0 ? 2 : 0) | (op[i] != 0 ? 4 : 0);
}

// A workgroup owns 16 x CPL columns of one sentence (grid B x ceil(H / (16 CPL)): 200 workgroups at B=50, H=200, so that the
// read of h is spread over the CUs -- one CU sustains only ~30 GB/s from HBM).  Its 512 threads form 32 token streams; all
// loads of a sentence of up to 128 tokens are issued before the masks are read.  The streams of a wave meet by shuffles, the
// waves in LDS.  max keeps the FIRST maximum (torch.max(dim) on the CPU reference): strict > along a stream (tokens
// ascending), ties between streams go to the smaller token.
template <typename T, int CPL>
__global__ __launch_bounds__(POOL_THREADS) void pool3_fwd_kernel(const T* __restrict__ h, const uint8_t* __restrict__ pool_mask,
                                                                const int64_t* __restrict__ subj_pos, const int64_t* __restrict__ obj_pos,
                                                                int Tn, int H, int type, float* __restrict__ out, int32_t* __restrict__ argmax) {
    extern __shared__ int mbits[];                       // [Tn]
    __shared__ float red[POOL_WAVES][3][POOL_LANES * CPL];
    __shared__ int redarg[POOL_WAVES][3][POOL_LANES * CPL];
    __shared__ int s_cnt[3];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = lane & (POOL_LANES - 1), stream = tid / POOL_LANES;
    const int c0 = (blockIdx.y * POOL_LANES + cl) * CPL;
    const int cc = min(c0, H - CPL), live = c0 < H;
    constexpr int PU = 4;                                // tokens per stream per round
    float v[PU][CPL];
#pragma unroll
    for (int u = 0; u < PU; ++u)
        dgio<T, CPL>::ld(h + ((size_t)b * Tn + min(stream + u * POOL_STREAMS, Tn - 1)) * H + cc, live, v[u]);
    if (tid < 3) s_cnt[tid] = 0;
    __syncthreads();
    for (int t = tid; t < Tn; t += POOL_THREADS) {
        const int m = pool_mask_bits(pool_mask, subj_pos, obj_pos, (size_t)b * Tn + t);
        mbits[t] = m;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (!(m >> k & 1)) atomicAdd(&s_cnt[k], 1);
    }
    __syncthreads();
    float acc[3][CPL];
    int arg[3][CPL];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < CPL; ++j) { acc[k][j] = type == 0 ? -INFINITY : 0.0f; arg[k][j] = 0x7fffffff; }
    for (int t0 = stream; t0 < Tn; t0 += POOL_STREAMS * PU) {
        if (t0 != stream) {
#pragma unroll
            for (int u = 0; u < PU; ++u)
                dgio<T, CPL>::ld(h + ((size_t)b * Tn + min(t0 + u * POOL_STREAMS, Tn - 1)) * H + cc, live, v[u]);
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int t = t0 + u * POOL_STREAMS;
            const bool have = t < Tn;
            const int m = mbits[min(t, Tn - 1)];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool masked = m >> k & 1;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    if (type == 0) {
                        const float x = masked ? POOL_NEG : v[u][j];         // masked_fill(mask, -1e12), gcn.py:476
                        if (have && x > acc[k][j]) { acc[k][j] = x; arg[k][j] = t; }
                    } else {
                        acc[k][j] += (have && !masked) ? v[u][j] : 0.0f;
                    }
                }
            }
        }
    }
    // the 4 streams of a wave (lanes 16 apart), then the 8 waves
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
#pragma unroll
            for (int d = POOL_LANES; d < WAVE; d <<= 1) {
                const float x = __shfl_xor(acc[k][j], d);
                const int ax = __shfl_xor(arg[k][j], d);
                if (type == 0) {
                    if (x > acc[k][j] || (x == acc[k][j] && ax < arg[k][j])) { acc[k][j] = x; arg[k][j] = ax; }
                } else {
                    acc[k][j] += x;
                }
            }
            if (lane < POOL_LANES) { red[wave][k][cl * CPL + j] = acc[k][j]; redarg[wave][k][cl * CPL + j] = arg[k][j]; }
        }
    __syncthreads();
    for (int q = tid; q < 3 * POOL_LANES * CPL; q += POOL_THREADS) {
        const int k = q / (POOL_LANES * CPL), cq = q - k * (POOL_LANES * CPL);
        const int c = blockIdx.y * POOL_LANES * CPL + cq;
        if (c >= H) continue;
        float r = red[0][k][cq];
        int a = redarg[0][k][cq];
#pragma unroll
        for (int w = 1; w < POOL_WAVES; ++w) {
            const float x = red[w][k][cq];
            const int ax = redarg[w][k][cq];
            if (type == 0) {
                if (x > r || (x == r && ax < a)) { r = x; a = ax; }
            } else {
                r += x;
            }
        }
        if (type == 1) r = r / (float)s_cnt[k];                       // gcn.py:480 (0/0 = nan when everything is masked, as there)
        out[(size_t)b * 3 * H + (size_t)k * H + c] = r;
        if (argmax) argmax[((size_t)b * 3 + k) * H + c] = a == 0x7fffffff ? 0 : a;
    }
}

// Backward: dh is written completely (masked tokens get 0, as masked_fill blocks their gradient); same workgroup shape.
template <typename T, int CPL>
__global__ __launch_bounds__(POOL_THREADS) void pool3_bwd_kernel(const float* __restrict__ g, const int32_t* __restrict__ argmax,
                                                                const uint8_t* __restrict__ pool_mask, const int64_t* __restrict__ subj_pos,
                                                                const int64_t* __restrict__ obj_pos, int Tn, int H, int type,
                                                                T* __restrict__ dh, const T* __restrict__ y, const int32_t* __restrict__ d_ell,
                                                                float scale) {
    extern __shared__ int mbits[];
    __shared__ int s_cnt[3];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int cl = lane & (POOL_LANES - 1), stream = tid / POOL_LANES;
    const int c0 = (blockIdx.y * POOL_LANES + cl) * CPL;
    const int cc = min(c0, H - CPL), live = c0 < H;
    float gk[3][CPL];
    int ak[3][CPL];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        dgio<float, CPL>::ld(g + (size_t)b * 3 * H + (size_t)k * H + cc, live, gk[k]);
#pragma unroll
        for (int j = 0; j < CPL; ++j) ak[k][j] = type == 0 ? argmax[((size_t)b * 3 + k) * H + min(cc + j, H - 1)] : 0;
    }
    if (tid < 3) s_cnt[tid] = 0;
    __syncthreads();
    for (int t = tid; t < Tn; t += POOL_THREADS) {
        const int m = pool_mask_bits(pool_mask, subj_pos, obj_pos, (size_t)b * Tn + t);
        mbits[t] = m;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (!(m >> k & 1)) atomicAdd(&s_cnt[k], 1);
    }
    __syncthreads();
    if (type == 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float inv = 1.0f / (float)s_cnt[k];
#pragma unroll
            for (int j = 0; j < CPL; ++j) gk[k][j] *= inv;
        }
    }
    for (int t = stream; t < Tn; t += POOL_STREAMS) {
        const int m = mbits[t];
        float s[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            s[j] = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool on = !(m >> k & 1) && (type != 0 || ak[k][j] == t);     // masked positions never receive gradient
                s[j] += on ? gk[k][j] : 0.0f;
            }
        }
        if (y) {
            // hand-over to the layer stack (gcnpt_pool3_bwd_dz): the pooled rows are the top GCN layer's output, so its
            // dZ = dh * 1[Y > 0] * scale / (deg + 1) (gcn.py:390-393 differentiated) leaves from here, where dh is at hand
            float yv[CPL];
            dgio<T, CPL>::ld(y + ((size_t)b * Tn + t) * H + cc, live, yv);
            const float f = scale / (float)(d_ell[((size_t)b * Tn + t) * 8] + 1);
#pragma unroll
            for (int j = 0; j < CPL; ++j) s[j] = yv[j] > 0.0f ? s[j] * f : 0.0f;
        }
        if (live) dgio<T, CPL>::st(dh + ((size_t)b * Tn + t) * H + cc, live, s);
    }
}

}  // namespace gcnpt

using namespace gcnpt;

extern "C" int gcnpt_pool3_fwd(void* stream, const void* h, int h_dtype, const uint8_t* pool_mask, const int64_t* subj_pos,
                               const int64_t* obj_pos, int B, int T, int H, int type, float* out, int32_t* argmax) {
    GCNPT_REQUIRE(h && pool_mask && subj_pos && obj_pos && out, "pool3_fwd: null pointer");
    GCNPT_REQUIRE(B > 0 && T > 0 && H > 0 && dtype_ok(h_dtype) && type >= 0 && type <= 2, "pool3_fwd: bad argument");
    GCNPT_REQUIRE(type != 0 || argmax, "pool3_fwd: max pooling needs the argmax buffer");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = H % 4 == 0 && aligned16(h);
    const dim3 grid(B, ceil_div(H, POOL_LANES * (vec ? 4 : 1)));
    const size_t lds = sizeof(int) * T;
    if (h_dtype == GCNPT_F32) {
        if (vec) hipLaunchKernelGGL((pool3_fwd_kernel<float, 4>), grid, dim3(POOL_THREADS), lds, s, (const float*)h, pool_mask, subj_pos, obj_pos, T, H, type, out, argmax);
        else hipLaunchKernelGGL((pool3_fwd_kernel<float, 1>), grid, dim3(POOL_THREADS), lds, s, (const float*)h, pool_mask, subj_pos, obj_pos, T, H, type, out, argmax);
    } else {
        if (vec) hipLaunchKernelGGL((pool3_fwd_kernel<bf16_t, 4>), grid, dim3(POOL_THREADS), lds, s, (const bf16_t*)h, pool_mask, subj_pos, obj_pos, T, H, type, out, argmax);
        else hipLaunchKernelGGL((pool3_fwd_kernel<bf16_t, 1>), grid, dim3(POOL_THREADS), lds, s, (const bf16_t*)h, pool_mask, subj_pos, obj_pos, T, H, type, out, argmax);
    }
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

static int pool3_bwd_launch(void* stream, const float* g, const int32_t* argmax, const uint8_t* pool_mask, const int64_t* subj_pos,
                            const int64_t* obj_pos, int B, int T, int H, int type, void* dh, int dh_dtype, const void* y, const int32_t* ell,
                            float scale, const char* what) {
    GCNPT_REQUIRE(g && pool_mask && subj_pos && obj_pos && dh, "%s: null pointer", what);
    GCNPT_REQUIRE(B > 0 && T > 0 && H > 0 && dtype_ok(dh_dtype) && type >= 0 && type <= 2, "%s: bad argument", what);
    GCNPT_REQUIRE(type != 0 || argmax, "%s: max pooling needs the argmax buffer", what);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = H % 4 == 0 && aligned16(dh) && aligned16(g) && (!y || aligned16(y));
    const dim3 grid(B, ceil_div(H, POOL_LANES * (vec ? 4 : 1)));
    const size_t lds = sizeof(int) * T;
    if (dh_dtype == GCNPT_F32) {
        if (vec) hipLaunchKernelGGL((pool3_bwd_kernel<float, 4>), grid, dim3(POOL_THREADS), lds, s, g, argmax, pool_mask, subj_pos, obj_pos, T, H, type, (float*)dh, (const float*)y, ell, scale);
        else hipLaunchKernelGGL((pool3_bwd_kernel<float, 1>), grid, dim3(POOL_THREADS), lds, s, g, argmax, pool_mask, subj_pos, obj_pos, T, H, type, (float*)dh, (const float*)y, ell, scale);
    } else {
        if (vec) hipLaunchKernelGGL((pool3_bwd_kernel<bf16_t, 4>), grid, dim3(POOL_THREADS), lds, s, g, argmax, pool_mask, subj_pos, obj_pos, T, H, type, (bf16_t*)dh, (const bf16_t*)y, ell, scale);
        else hipLaunchKernelGGL((pool3_bwd_kernel<bf16_t, 1>), grid, dim3(POOL_THREADS), lds, s, g, argmax, pool_mask, subj_pos, obj_pos, T, H, type, (bf16_t*)dh, (const bf16_t*)y, ell, scale);
    }
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_pool3_bwd(void* stream, const float* g, const int32_t* argmax, const uint8_t* pool_mask, const int64_t* subj_pos,
                               const int64_t* obj_pos, int B, int T, int H, int type, void* dh, int dh_dtype) {
    return pool3_bwd_launch(stream, g, argmax, pool_mask, subj_pos, obj_pos, B, T, H, type, dh, dh_dtype, nullptr, nullptr, 1.0f, "pool3_bwd");
}

extern "C" int gcnpt_pool3_bwd_dz(void* stream, const float* g, const int32_t* argmax, const uint8_t* pool_mask, const int64_t* subj_pos,
                                  const int64_t* obj_pos, int B, int T, int H, int type, const void* y, const int32_t* ell, float scale, void* dz,
                                  int dtype) {
    GCNPT_REQUIRE(y && ell, "pool3_bwd_dz: null pointer");
    return pool3_bwd_launch(stream, g, argmax, pool_mask, subj_pos, obj_pos, B, T, H, type, dz, dtype, y, ell, scale, "pool3_bwd_dz");
}
