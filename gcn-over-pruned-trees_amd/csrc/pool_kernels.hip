// Fused masked pooling of the GCN output: reference model/gcn.py:116-121 runs pool() three times
// (all in-tree tokens, subject tokens, object tokens; model/gcn.py:473-483), i.e. three masked_fill + reduce passes over
// h [B,T,H] plus the mask tensors.  Here ONE pass reads h once and writes the concatenated [B, 3H] row the output MLP
// consumes; the subject / object masks are taken straight from the position tensors (pos != 0 <=> masked, gcn.py:116).
//   max : out = max over unmasked t (ties: first t, as torch.max(dim) returns); all masked -> -1e12 (INFINITY_NUMBER)
//   avg : sum over unmasked / (T - #masked)          sum : sum over unmasked
// Backward routes the gradient to the recorded argmax (max) or to every unmasked token (avg / sum).
#include "layer_common.h"

namespace gcnpt {

constexpr int POOL_THREADS = 256;
constexpr float POOL_NEG = -1e12f;     // utils/constant.py:35 INFINITY_NUMBER

// bit k of the result = token t is MASKED for pooling k (0: pool_mask, 1: not a subject token, 2: not an object token)
__device__ __forceinline__ int pool_mask_bits(const uint8_t* pm, const int64_t* sp, const int64_t* op, size_t i) {
    return (pm[i] ? 1 : 0) | (sp[i] != 0 ? 2 : 0) | (op[i] != 0 ? 4 : 0);
}

template <typename T>
__global__ __launch_bounds__(POOL_THREADS) void pool3_fwd_kernel(const T* __restrict__ h, const uint8_t* __restrict__ pool_mask,
                                                                const int64_t* __restrict__ subj_pos, const int64_t* __restrict__ obj_pos,
                                                                int Tn, int H, int type, float* __restrict__ out, int32_t* __restrict__ argmax) {
    extern __shared__ int mbits[];                       // [Tn]
    const int b = blockIdx.x;
    for (int t = threadIdx.x; t < Tn; t += POOL_THREADS) mbits[t] = pool_mask_bits(pool_mask, subj_pos, obj_pos, (size_t)b * Tn + t);
    __syncthreads();
    int unmasked[3] = {0, 0, 0};
    if (type == 1)
        for (int t = 0; t < Tn; ++t) {
#pragma unroll
            for (int k = 0; k < 3; ++k) unmasked[k] += (mbits[t] >> k & 1) ? 0 : 1;
        }
    for (int c = blockIdx.y * POOL_THREADS + threadIdx.x; c < H; c += gridDim.y * POOL_THREADS) {
        float acc[3];
        int arg[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] = type == 0 ? -INFINITY : 0.0f;
        for (int t = 0; t < Tn; ++t) {
            const float v = io<T>::load1(h + ((size_t)b * Tn + t) * H + c);
            const int m = mbits[t];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (type == 0) {
                    const float x = (m >> k & 1) ? POOL_NEG : v;         // masked_fill(mask, -1e12), gcn.py:476
                    if (x > acc[k]) { acc[k] = x; arg[k] = t; }           // strict: the FIRST maximum wins
                } else {
                    acc[k] += (m >> k & 1) ? 0.0f : v;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float r = acc[k];
            if (type == 1) r = r / (float)unmasked[k];                     // gcn.py:480 (0/0 = nan when everything is masked, as there)
            out[(size_t)b * 3 * H + (size_t)k * H + c] = r;
            if (argmax) argmax[((size_t)b * 3 + k) * H + c] = arg[k];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(POOL_THREADS) void pool3_bwd_kernel(const float* __restrict__ g, const int32_t* __restrict__ argmax,
                                                                const uint8_t* __restrict__ pool_mask, const int64_t* __restrict__ subj_pos,
                                                                const int64_t* __restrict__ obj_pos, int Tn, int H, int type,
                                                                T* __restrict__ dh) {
    extern __shared__ int mbits[];
    const int b = blockIdx.x;
    for (int t = threadIdx.x; t < Tn; t += POOL_THREADS) mbits[t] = pool_mask_bits(pool_mask, subj_pos, obj_pos, (size_t)b * Tn + t);
    __syncthreads();
    float inv[3] = {1.0f, 1.0f, 1.0f};
    if (type == 1) {
        int unmasked[3] = {0, 0, 0};
        for (int t = 0; t < Tn; ++t) {
#pragma unroll
            for (int k = 0; k < 3; ++k) unmasked[k] += (mbits[t] >> k & 1) ? 0 : 1;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) inv[k] = 1.0f / (float)unmasked[k];
    }
    for (int c = blockIdx.y * POOL_THREADS + threadIdx.x; c < H; c += gridDim.y * POOL_THREADS) {
        float gk[3];
        int ak[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gk[k] = g[(size_t)b * 3 * H + (size_t)k * H + c] * inv[k];
            if (type == 0) ak[k] = argmax[((size_t)b * 3 + k) * H + c];
        }
        for (int t = 0; t < Tn; ++t) {
            const int m = mbits[t];
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool on = !(m >> k & 1) && (type != 0 || ak[k] == t);     // masked positions never receive gradient
                s += on ? gk[k] : 0.0f;
            }
            io<T>::store1(dh + ((size_t)b * Tn + t) * H + c, s);
        }
    }
}

}  // namespace gcnpt

using namespace gcnpt;

extern "C" int gcnpt_pool3_fwd(void* stream, const void* h, int h_dtype, const uint8_t* pool_mask, const int64_t* subj_pos,
                               const int64_t* obj_pos, int B, int T, int H, int type, float* out, int32_t* argmax) {
    GCNPT_REQUIRE(h && pool_mask && subj_pos && obj_pos && out, "pool3_fwd: null pointer");
    GCNPT_REQUIRE(B > 0 && T > 0 && H > 0 && dtype_ok(h_dtype) && type >= 0 && type <= 2, "pool3_fwd: bad argument");
    GCNPT_REQUIRE(type != 0 || argmax, "pool3_fwd: max pooling needs the argmax buffer");
    const dim3 grid(B, std::max(1, std::min(ceil_div(H, POOL_THREADS), 8)));
    hipStream_t s = (hipStream_t)stream;
    if (h_dtype == GCNPT_F32)
        hipLaunchKernelGGL(pool3_fwd_kernel<float>, grid, dim3(POOL_THREADS), sizeof(int) * T, s, (const float*)h, pool_mask, subj_pos, obj_pos, T, H, type, out, argmax);
    else
        hipLaunchKernelGGL(pool3_fwd_kernel<bf16_t>, grid, dim3(POOL_THREADS), sizeof(int) * T, s, (const bf16_t*)h, pool_mask, subj_pos, obj_pos, T, H, type, out, argmax);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_pool3_bwd(void* stream, const float* g, const int32_t* argmax, const uint8_t* pool_mask, const int64_t* subj_pos,
                               const int64_t* obj_pos, int B, int T, int H, int type, void* dh, int dh_dtype) {
    GCNPT_REQUIRE(g && pool_mask && subj_pos && obj_pos && dh, "pool3_bwd: null pointer");
    GCNPT_REQUIRE(B > 0 && T > 0 && H > 0 && dtype_ok(dh_dtype) && type >= 0 && type <= 2, "pool3_bwd: bad argument");
    GCNPT_REQUIRE(type != 0 || argmax, "pool3_bwd: max pooling needs the argmax buffer");
    const dim3 grid(B, std::max(1, std::min(ceil_div(H, POOL_THREADS), 8)));
    hipStream_t s = (hipStream_t)stream;
    if (dh_dtype == GCNPT_F32)
        hipLaunchKernelGGL(pool3_bwd_kernel<float>, grid, dim3(POOL_THREADS), sizeof(int) * T, s, g, argmax, pool_mask, subj_pos, obj_pos, T, H, type, (float*)dh);
    else
        hipLaunchKernelGGL(pool3_bwd_kernel<bf16_t>, grid, dim3(POOL_THREADS), sizeof(int) * T, s, g, argmax, pool_mask, subj_pos, obj_pos, T, H, type, (bf16_t*)dh);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}
