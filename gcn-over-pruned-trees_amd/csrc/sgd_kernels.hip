// The parameter update of synchronous data-parallel SGD as the reference trains (train.py:224-227: clip_grad_norm_(max_grad_norm), then
// plain SGD, utils/torch_utils.py), on the flat fp32 parameter / gradient buffers of shard.FlatGradBucket, in TWO small launches instead of
// the six element-wise library kernels the same arithmetic takes through torch (vector_norm, div, add, clamp, mul, addcmul: ~20 us of
// device time and ~40 us of host time per step, which made the N > 1 loop host-bound):
//   1. every workgroup writes the sum of squares of its slice of g * g_scale to partials[wg] (plain stores: no atomics, nothing to zero);
//   2. every workgroup adds the SGD_PARTIALS partials (+ *extra_sq, the row-sparse parameters' share of the norm, if given), takes
//      coef = min(1, max_norm / (norm + 1e-6)) -- clip_grad_norm_'s coefficient -- and updates its slice: w -= lr * coef * g_scale * g.
//      Workgroup 0 leaves coef in partials[SGD_PARTIALS] for the caller's row-sparse updates.
// Sums in double: the result does not depend on how the slices are cut.
#include "gcnpt_common.h"

namespace gcnpt {

constexpr int SGD_THREADS = 256, SGD_PARTIALS = 64;

__global__ __launch_bounds__(SGD_THREADS) void sgd_sumsq_kernel(const float* __restrict__ g, long long n, float g_scale, float* __restrict__ partials) {
    __shared__ double red[SGD_THREADS / WAVE];
    const long long n4 = n / 4;
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * SGD_THREADS + threadIdx.x; i < n4; i += (long long)SGD_PARTIALS * SGD_THREADS) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        const double a = (double)(v.x * g_scale), b = (double)(v.y * g_scale), c = (double)(v.z * g_scale), d = (double)(v.w * g_scale);
        s += a * a + b * b + c * c + d * d;
    }
    if (blockIdx.x == 0)
        for (long long i = n4 * 4 + threadIdx.x; i < n; i += SGD_THREADS) { const double a = (double)(g[i] * g_scale); s += a * a; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < SGD_THREADS / WAVE; ++w) t += red[w];
        partials[blockIdx.x] = (float)t;
    }
}

__global__ __launch_bounds__(SGD_THREADS) void sgd_update_kernel(float* __restrict__ w, const float* __restrict__ g, long long n, float g_scale,
                                                                float max_norm, float lr, float* __restrict__ partials,
                                                                const float* __restrict__ extra_sq) {
    float coef = 1.0f;
    if (max_norm > 0.0f) {
        double sq = threadIdx.x < SGD_PARTIALS ? (double)partials[threadIdx.x] : 0.0;       // (one wave holds all 64)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        sq = __shfl(sq, 0);
        __shared__ float s_coef;
        if (threadIdx.x == 0) {
            if (extra_sq) sq += (double)*extra_sq;
            const float norm = (float)sqrt(sq);
            s_coef = fminf(1.0f, max_norm / (norm + 1e-6f));
        }
        __syncthreads();
        coef = s_coef;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) partials[SGD_PARTIALS] = coef;
    const float step = lr * coef * g_scale;
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * SGD_THREADS + threadIdx.x; i < n4; i += (long long)gridDim.x * SGD_THREADS) {
        float4 p = reinterpret_cast<float4*>(w)[i];
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        p.x -= step * v.x; p.y -= step * v.y; p.z -= step * v.z; p.w -= step * v.w;
        reinterpret_cast<float4*>(w)[i] = p;
    }
    if (blockIdx.x == 0)
        for (long long i = n4 * 4 + threadIdx.x; i < n; i += SGD_THREADS) w[i] -= step * g[i];
}

}  // namespace gcnpt

using namespace gcnpt;

extern "C" int gcnpt_sgd_clip_update(void* stream, float* w, const float* g, long long n, float g_scale, float max_norm, float lr,
                                     float* partials, const float* extra_sq) {
    GCNPT_REQUIRE(w && g && partials, "sgd_clip_update: null pointer");
    GCNPT_REQUIRE(n > 0, "sgd_clip_update: n must be positive");
    GCNPT_REQUIRE(((uintptr_t)w & 15) == 0 && ((uintptr_t)g & 15) == 0, "sgd_clip_update: w and g must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (max_norm > 0.0f) {
        hipLaunchKernelGGL(sgd_sumsq_kernel, dim3(SGD_PARTIALS), dim3(SGD_THREADS), 0, s, g, n, g_scale, partials);
        GCNPT_HIP_CHECK(hipGetLastError());
    }
    const int grid = (int)std::min<long long>(256, (n / 4 + SGD_THREADS - 1) / SGD_THREADS + 1);
    hipLaunchKernelGGL(sgd_update_kernel, dim3(grid), dim3(SGD_THREADS), 0, s, w, g, n, g_scale, max_norm, lr, partials, extra_sq);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}
