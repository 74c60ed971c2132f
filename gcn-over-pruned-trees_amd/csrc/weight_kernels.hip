// Weight-side kernels (gfx950): nn.Linear weights -> MFMA fragment order, and the weight/bias gradient
// dW += dZ^T ((A+I) h), db += 2 sum dZ of reference model/gcn.py:270-271 (autograd).
#include "layer_common.h"
#include "pack_common.h"
#include "wgrad_common.h"

namespace gcnpt {


template <typename CT>
__global__ void pack_weights_kernel(const PackParams p) {
    pack_fragments<CT>(p, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
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
    note_launch(grid, 256, 0, sizeof(p));
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
namespace gcnpt {
int plan_weight_grad(WeightGradParams& p, const void* z_frag, const void* s_frag, int nks, int Din, int H, float* dW, float* db,
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
    // GCNPT_OPT_DETERMINISTIC: one slice, i.e. every element of dW / db is summed by exactly ONE workgroup in a fixed order and meets
    // its zero-initialised accumulator in a single atomic add: run-to-run bit-identical gradients, as the reference's CPU / single-GPU
    // path gives, at the price of the split contraction's parallelism
    if (option(GCNPT_OPT_DETERMINISTIC) == 1) want = 1;
    int slices = want >= 8 ? want / 8 * 8 : (want >= 4 ? 4 : (want >= 2 ? 2 : 1));
    p.ks_per_wg = ceil_div(p.nks, slices);
    p.mb = mb; p.nb = nb; p.slices = slices;
    const int per_group = (slices & 7) == 0 ? mb * nb * (slices / 8) : ceil_div(mb * nb, 8 / slices);
    return 8 * per_group;
}
}  // namespace gcnpt

template <typename CT, int NW, int NT, int KB>
static int launch_weight_grad_cfg(hipStream_t s, const WeightGradMulti& mp) {
    const size_t lds = weight_grad_lds(NW);      // 49 / 98 KiB
    GCNPT_LDS_ATTR_ONCE((weight_grad_kernel<CT, NW, NT, KB>), 160 * 1024);
    hipLaunchKernelGGL((weight_grad_kernel<CT, NW, NT, KB>), dim3(mp.first[mp.n]), dim3(NW * WAVE), lds, s, mp);
    note_launch(mp.first[mp.n], NW * WAVE, lds, sizeof(mp));
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

// waves per workgroup / workgroups the launch may have: small batches are one latency chain per CU (8 waves when layers share the
// CUs); from 512 k-steps (16 k rows) on the fragment stream through the CUs' L1 paths is what counts: 4-wave workgroups, 2 per CU
#ifndef GCNPT_WG_SINGLE_WAVES
#define GCNPT_WG_SINGLE_WAVES 4      // waves per workgroup when a launch holds ONE layer's gradient of a small batch (A/B: 4 / 8)
#endif
#ifndef GCNPT_WG_SMALL_BUDGET
#define GCNPT_WG_SMALL_BUDGET 256    // workgroups a small batch's launch may have (A/B: 128 / 256 / 512)
#endif
static int wg_waves(int n_layers, int nks) { return nks < 512 ? (n_layers > 1 ? 8 : GCNPT_WG_SINGLE_WAVES) : 4; }
static int wg_budget(int nks) { return nks < 512 ? GCNPT_WG_SMALL_BUDGET : 512; }
// (4 x 3)-tile blocks of all layers of a launch: the budget is shared in proportion to them, so every workgroup gets the same k-steps
static int wg_nt(int nks) { return nks < 512 ? WG_NT : 6; }       // output tiles per block row: wide blocks for big batches (see the kernel)
static int wg_blocks(int n_layers, const int* Din, const int* H, int nt) {
    int b = 0;
    for (int l = 0; l < n_layers; ++l) b += ceil_div(ceil_div(H[l], 16), WG_MT) * ceil_div(ceil_div(Din[l], 16), nt);
    return b;
}

static int launch_weight_grad(hipStream_t s, const WeightGradMulti& mp, int compute_dtype, int waves, int nt) {
    // a wave issues KB k-steps of loads per batch whether they are live or not (no load behind a condition): a few dozen row tiles leave
    // a wave one or two k-steps, and the other three of a 5-deep batch would be 21 dead wave-loads of address processing each
    int per_wave = 0;
    for (int l = 0; l < mp.n; ++l) per_wave = std::max(per_wave, ceil_div(mp.l[l].ks_per_wg, waves));
    const bool shallow = per_wave <= 2;
    if (compute_dtype == GCNPT_BF16) {
        if (nt == 6) return launch_weight_grad_cfg<bf16_t, 4, 6, 3>(s, mp);
        if (shallow) return waves == 8 ? launch_weight_grad_cfg<bf16_t, 8, WG_NT, 2>(s, mp) : launch_weight_grad_cfg<bf16_t, 4, WG_NT, 2>(s, mp);
        return waves == 8 ? launch_weight_grad_cfg<bf16_t, 8, WG_NT, WG_KB>(s, mp) : launch_weight_grad_cfg<bf16_t, 4, WG_NT, WG_KB>(s, mp);
    }
    if (nt == 6) return launch_weight_grad_cfg<float, 4, 6, 2>(s, mp);
    if (shallow) return waves == 8 ? launch_weight_grad_cfg<float, 8, WG_NT, 2>(s, mp) : launch_weight_grad_cfg<float, 4, WG_NT, 2>(s, mp);
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

