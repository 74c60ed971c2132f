// nn.Linear weights -> MFMA fragment order: the device code of gcnpt_pack_weights, shared by the launches that carry it as a side job
// (the tree build and the tree gather have idle CUs while they run: gcnpt_prune_to_csr_pack, gcnpt_gather_trees_pack).
#pragma once
#include "layer_common.h"

namespace gcnpt {

// ---------------------------------------------------------------------------------------------------
// nn.Linear weight [H,Din] fp32 -> MFMA B-operand fragments
//   fragment (tile, kstep, lane) = 16 bytes:
//     bf16: 8 values  B[k = 32 kstep + 8 (lane>>4) + j][n = 16 tile + (lane&15)],  j = 0..7
//     f32 : 4 values  B[k = 16 kstep + 4 (lane>>4) + s][n = 16 tile + (lane&15)],  s = 0..3
//   forward image : B[k][n] = W[n][k]  (n over H,   k over Din)
//   backward image: B[k][n] = W[k][n]  (n over Din, k over H)
// ---------------------------------------------------------------------------------------------------
constexpr int PACK_MAX_LAYERS = 8;
struct PackParams {
    const float* W[PACK_MAX_LAYERS];
    uint4* wf[PACK_MAX_LAYERS];
    uint4* wb[PACK_MAX_LAYERS];
    int H[PACK_MAX_LAYERS], Din[PACK_MAX_LAYERS];
    long long first[PACK_MAX_LAYERS + 1];      // fragment index range of each layer in the launch
    int n_layers;
};

// fragments gid0, gid0 + stride, ... of every layer of the stack (weights change once per optimizer step)
template <typename CT>
__device__ __forceinline__ void pack_fragments(const PackParams& p, long long gid0, long long stride) {
    constexpr int KSTEP = sizeof(CT) == 2 ? 32 : 16;
    constexpr int PER = sizeof(CT) == 2 ? 8 : 4;
    for (long long gid = gid0; gid < p.first[p.n_layers]; gid += stride) {
        int l = 0;
#pragma unroll
        for (int i = 1; i < PACK_MAX_LAYERS; ++i) l += (i < p.n_layers && gid >= p.first[i]) ? 1 : 0;
        const float* W = p.W[l];
        const int H = p.H[l], Din = p.Din[l];
        const int ksf = round_up(Din, KSTEP) / KSTEP, ntf = ceil_div(H, 16);
        const int ksb = round_up(H, KSTEP) / KSTEP;
        const long long nf = p.wf[l] ? (long long)ntf * ksf * 64 : 0;
        const long long id = gid - p.first[l];
        const bool bwd = id >= nf;
        const long long f = bwd ? id - nf : id;
        const int ks_n = bwd ? ksb : ksf;
        const int lane = (int)(f & 63);
        const int ks = (int)((f >> 6) % ks_n), tl = (int)((f >> 6) / ks_n);
        const int n = tl * 16 + (lane & 15);
        const int kb = ks * KSTEP + (lane >> 4) * PER;
        float v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int k = kb + j;
            // unconditional clamped load + select (a conditional load would serialise the 8 of them)
            const int rr = bwd ? k : n, cc = bwd ? n : k;
            const float x = W[(size_t)min(rr, H - 1) * Din + min(cc, Din - 1)];
            v[j] = (rr < H && cc < Din) ? x : 0.0f;
        }
        uint4 u;
        if constexpr (sizeof(CT) == 2) {
            u.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            u.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            u.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
            u.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
        } else {
            u.x = __float_as_uint(v[0]); u.y = __float_as_uint(v[1]); u.z = __float_as_uint(v[2]); u.w = __float_as_uint(v[3]);
        }
        (bwd ? p.wb[l] : p.wf[l])[f] = u;
    }
}

// host side: fills p from the argument lists of gcnpt_pack_weights_multi (validated by the caller); returns the number of fragments
int fill_pack_params(PackParams& p, int n_layers, const float* const* W, const int* H, const int* Din, int dtype, void* const* w_fwd,
                     void* const* w_bwd);

// the side job itself: workgroups [first_block, gridDim.x) of a launch pack while the others do the launch's own work
__device__ __forceinline__ void pack_side_job(const PackParams& p, int dtype, int first_block) {
    const long long gid0 = (long long)(blockIdx.x - first_block) * blockDim.x + threadIdx.x;
    const long long stride = (long long)(gridDim.x - first_block) * blockDim.x;
    if (dtype == GCNPT_BF16) pack_fragments<bf16_t>(p, gid0, stride);
    else pack_fragments<float>(p, gid0, stride);
}

}  // namespace gcnpt
