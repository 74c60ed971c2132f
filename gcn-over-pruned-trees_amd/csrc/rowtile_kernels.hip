// GCN layer forward and backward-data for gfx950 (CDNA4): reference model/gcn.py:269-271, 390-393.
//
//   forward        out = dropout(relu((((A+I) h) W^T + 2 b) / (deg + 1)))
//   backward-data  dh  = ((A+I)^T dZ) W            with dZ = dY * 1[Y>0] * scale / (deg + 1)
//
// Both are ONE row-tile kernel.  A workgroup (8 waves) owns ROWS = 32 consecutive token rows and is the
// only workgroup on its CU (157 workgroups at B=50, T=100), so it is built to have every byte it
// needs in flight at once rather than for occupancy:
//   (0) each wave issues the loads of ALL weight fragments it will use (registers, up to KSMAX k-steps);
//   (1) 32 lanes fetch the rows' CSR extents, degrees and first neighbours -> LDS;
//       meanwhile every thread issues the loads of its share of the tile's own rows;
//   (2) neighbour rows (<= 3 per kept token of a pruned tree) are added in fp32 and the tile is parked
//       in LDS in the MFMA operand type;
//   (3) the tile meets the register-resident weight fragments on the matrix cores;
//   (4) the epilogue runs on the accumulators, the result is staged through LDS and leaves as
//       whole rows in 16-byte stores.
// The dense [B,T,T] bmm of the reference (gcn.py:269) never exists: aggregation is a gather.
#include "rowtile_body.h"

// =====================================================================================================
// C-ABI
// =====================================================================================================
using namespace gcnpt;

namespace gcnpt {
thread_local SideWgrad t_side;
#define GCNPT_RT_DECL(n) int rowtile_launch_part##n(hipStream_t s, const RowTileParams& p);
GCNPT_RT_DECL(0) GCNPT_RT_DECL(1) GCNPT_RT_DECL(2) GCNPT_RT_DECL(3) GCNPT_RT_DECL(4) GCNPT_RT_DECL(5) GCNPT_RT_DECL(6) GCNPT_RT_DECL(7)
GCNPT_RT_DECL(8) GCNPT_RT_DECL(9) GCNPT_RT_DECL(10) GCNPT_RT_DECL(11) GCNPT_RT_DECL(12) GCNPT_RT_DECL(13) GCNPT_RT_DECL(14)
int rowtile_launch(int combo, int mode, hipStream_t s, const RowTileParams& p) {
    typedef int (*fn_t)(hipStream_t, const RowTileParams&);
    static const fn_t table[15] = {rowtile_launch_part0, rowtile_launch_part1, rowtile_launch_part2, rowtile_launch_part3, rowtile_launch_part4,
                                   rowtile_launch_part5, rowtile_launch_part6, rowtile_launch_part7, rowtile_launch_part8, rowtile_launch_part9,
                                   rowtile_launch_part10, rowtile_launch_part11, rowtile_launch_part12, rowtile_launch_part13, rowtile_launch_part14};
    return table[combo * 3 + mode](s, p);
}
}  // namespace gcnpt

// how rows of `width` elements of `es` bytes at base a (and b, if given) may be read: 8 elements per load when rows are 16-byte
// aligned and whole chunks, 4 when they are aligned to half chunks (bf16: 8 bytes, f32: 16 bytes; width % 4 == 0), else element-wise
static int vec_elems(int width, size_t es, const void* a, const void* b) {
    auto al = [&](size_t n) { return (reinterpret_cast<uintptr_t>(a) % n) == 0 && (!b || (reinterpret_cast<uintptr_t>(b) % n) == 0); };
    if (width % 8 == 0 && al(16)) return 8;
    if (width % 4 == 0 && width >= 4 && al(4 * es)) return 4;
    return 0;
}
// bytes per row-store piece: 16, 8 or 0 (element stores)
static int vec_bytes(int width, size_t es, const void* a, const void* b) {
    auto al = [&](size_t n) { return (reinterpret_cast<uintptr_t>(a) % n) == 0 && (!b || (reinterpret_cast<uintptr_t>(b) % n) == 0); };
    if ((width * es) % 16 == 0 && al(16)) return 16;
    if ((width * es) % 8 == 0 && al(8)) return 8;
    return 0;
}


template <bool BWD, bool DZIN = false>
static int dispatch_rowtile(hipStream_t s, const RowTileParams& p, int in_dtype, int out_dtype, int compute) {
    const int mode = BWD ? (DZIN ? 2 : 1) : 0;
    if (compute == GCNPT_F32) {
        if (in_dtype != GCNPT_F32 || out_dtype != GCNPT_F32)
            return fail(GCNPT_E_UNSUPPORTED, "compute_dtype f32 needs f32 activations");
        return rowtile_launch(0, mode, s, p);
    }
    const int combo = in_dtype == GCNPT_F32 ? (out_dtype == GCNPT_F32 ? 1 : 2) : (out_dtype == GCNPT_F32 ? 3 : 4);
    return rowtile_launch(combo, mode, s, p);
}

static int layer_fwd_impl(void* stream, const void* h, int h_dtype, const void* w_fwd, const float* bias,
                          const int32_t* row_ptr, const int32_t* col_idx, const int32_t* ell, const int32_t* deg_ell,
                          int B, int T, int Din, int H, void* out, int out_dtype, int compute_dtype, float drop_p,
                          uint64_t seed, void* s_frag, const uint64_t* seed_dev) {
    GCNPT_REQUIRE(h && w_fwd && bias && row_ptr && col_idx && ell && out, "layer_fwd: null pointer");
    GCNPT_REQUIRE(B > 0 && T >= 0 && Din > 0 && H > 0, "layer_fwd: sizes must be positive");
    GCNPT_REQUIRE(dtype_ok(h_dtype) && dtype_ok(out_dtype) && dtype_ok(compute_dtype), "layer_fwd: bad dtype");
    GCNPT_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "layer_fwd: drop_p=%f outside [0,1)", (double)drop_p);
    if (rows_of(B, T) > 0x7fffffffLL / 2) return fail(GCNPT_E_UNSUPPORTED, "layer_fwd: B*T too large");
    RowTileParams p{};
    p.stamps = static_cast<unsigned long long*>(g_debug_stamps); p.knob = g_debug_knob;
    p.src = h; p.yref = nullptr; p.wfrag = w_fwd; p.bias = bias;
    p.g_row_ptr = row_ptr; p.g_col_idx = col_idx; p.g_ell = ell; p.d_ell = deg_ell ? deg_ell : ell; p.out = out;
    p.frag_out = s_frag;
    p.N = (int)rows_of(B, T); p.T = T; p.K = Din; p.NOUT = H; p.Kpad = round_up(Din, kstep_of(compute_dtype)); p.chunk_magic = 0xffffffffu / (unsigned)(p.Kpad / 8) + 1u;
    p.vec_in = vec_elems(Din, esize(h_dtype), h, nullptr);
    p.vec_out = vec_bytes(H, esize(out_dtype), out, nullptr);
    p.drop_p = drop_p; p.scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    p.drop_thresh16 = (unsigned)((double)drop_p * 65536.0);
    p.seed = seed; p.seed_dev = seed_dev;
    return dispatch_rowtile<false>((hipStream_t)stream, p, h_dtype, out_dtype, compute_dtype);
}

extern "C" int gcnpt_layer_fwd(void* stream, const void* h, int h_dtype, const void* w_fwd, const float* bias,
                               const int32_t* row_ptr, const int32_t* col_idx, const int32_t* ell, const int32_t* deg_ell,
                               int B, int T, int Din, int H, void* out, int out_dtype, int compute_dtype, float drop_p,
                               uint64_t seed, void* s_frag, const uint64_t* seed_dev) {
    return layer_fwd_impl(stream, h, h_dtype, w_fwd, bias, row_ptr, col_idx, ell, deg_ell, B, T, Din, H, out, out_dtype, compute_dtype, drop_p, seed,
                          s_frag, seed_dev);
}

// One weight gradient some launch should compute: the two fragment images, the layer's widths, its accumulators
struct WgradReq { const void* z; const void* s; int Din, H; float* dW; float* db; };

static int layer_bwd_data_impl(void* stream, const void* dY, const void* Y, int g_dtype, const void* w_bwd,
                               const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                               const int32_t* ellT, int B, int T, int Din, int H, void* dh, int dh_dtype,
                               int compute_dtype, float scale, void* z_frag, float* zero_dW, float* zero_db,
                               const void* relu_src, float next_scale, int src_is_dz) {
    GCNPT_REQUIRE(dY && (Y || src_is_dz) && w_bwd && ell && rowT_ptr && colT_idx && ellT, "layer_bwd_data: null pointer");
    GCNPT_REQUIRE(!relu_src || dh, "layer_bwd_data: relu_src without dh");
    GCNPT_REQUIRE(dh || z_frag, "layer_bwd_data: nothing to produce (dh and z_frag both NULL)");
    GCNPT_REQUIRE(B > 0 && T >= 0 && Din > 0 && H > 0, "layer_bwd_data: sizes must be positive");
    GCNPT_REQUIRE(dtype_ok(g_dtype) && dtype_ok(dh_dtype) && dtype_ok(compute_dtype), "layer_bwd_data: bad dtype");
    if (rows_of(B, T) > 0x7fffffffLL / 2) return fail(GCNPT_E_UNSUPPORTED, "layer_bwd_data: B*T too large");
    RowTileParams p{};
    p.stamps = static_cast<unsigned long long*>(g_debug_stamps); p.knob = g_debug_knob;
    p.src = dY; p.yref = Y; p.wfrag = w_bwd; p.bias = nullptr;
    p.g_row_ptr = rowT_ptr; p.g_col_idx = colT_idx; p.g_ell = ellT; p.d_ell = ell; p.out = dh;
    p.frag_out = z_frag;
    p.zero_p[0] = zero_dW; p.zero_n[0] = H * Din; p.zero_p[1] = zero_db; p.zero_n[1] = H;
    p.N = (int)rows_of(B, T); p.T = T; p.K = H; p.NOUT = Din; p.Kpad = round_up(H, kstep_of(compute_dtype)); p.chunk_magic = 0xffffffffu / (unsigned)(p.Kpad / 8) + 1u;
    p.vec_in = vec_elems(H, esize(g_dtype), dY, src_is_dz ? nullptr : Y);
    p.vec_out = dh ? vec_bytes(Din, esize(dh_dtype), dh, relu_src) : 0;
    p.scale = scale; p.drop_p = 0.0f;
    p.relu_src = relu_src; p.next_scale = next_scale;
    if (src_is_dz) return dispatch_rowtile<true, true>((hipStream_t)stream, p, g_dtype, dh_dtype, compute_dtype);
    return dispatch_rowtile<true>((hipStream_t)stream, p, g_dtype, dh_dtype, compute_dtype);
}

extern "C" int gcnpt_layer_bwd_data(void* stream, const void* dY, const void* Y, int g_dtype, const void* w_bwd,
                                    const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                                    const int32_t* ellT, int B, int T, int Din, int H, void* dh, int dh_dtype,
                                    int compute_dtype, float scale, void* z_frag, float* zero_dW, float* zero_db,
                                    const void* relu_src, float next_scale, int src_is_dz) {
    return layer_bwd_data_impl(stream, dY, Y, g_dtype, w_bwd, ell, rowT_ptr, colT_idx, ellT, B, T, Din, H, dh, dh_dtype, compute_dtype, scale, z_frag,
                               zero_dW, zero_db, relu_src, next_scale, src_is_dz);
}

// Plans the weight gradient of the layer ABOVE (its two fragment images, left by earlier launches) for the passenger workgroups of a
// backward-data launch over B x T rows: taken by the row-tile launch of a small batch (<= GCNPT_OPT_SIDE_TILES row tiles), see
// rowtile_wgrad_kernel.  false: it cannot ride (the caller launches it on its own).
static bool plan_side_wgrad(SideWgrads& sw, const WgradReq& r, int B, int T, int compute_dtype) {
    sw = SideWgrads{};
    const int n_tiles = ceil_div((int)rows_of(B, T), ROWS);
    if (!r.z || !r.s || !r.dW || !r.db || n_tiles > option(GCNPT_OPT_SIDE_TILES)) return false;
    const int nks = n_tiles * (compute_dtype == GCNPT_BF16 ? 1 : 2);
    const int blocks_l = ceil_div(ceil_div(r.H, 16), WG_MT) * ceil_div(ceil_div(r.Din, 16), WG_NT);
    sw.blocks = plan_weight_grad(sw.l, r.z, r.s, nks, r.Din, r.H, r.dW, r.db, blocks_l, RT_WAVES, std::max(64, 256 - n_tiles), WG_NT);
    return sw.blocks > 0;
}

// Whether the backward-data launch of a layer carries `ride` (the layer above's weight gradient): what launch_rowtile_cfg / try_colsplit
// decide when the launch is issued, as ONE predicate for the sweep's bookkeeping of launches it does not issue (gcnpt_layers_bwd_range):
// the 8-wave one-shot form on ready-made dZ rows in uniform precision carries it, the 4-wave and the column-split forms never do.
static bool would_carry(const void* g, const void* Y, int g_dtype, const void* dh, int dh_dtype, int compute_dtype, bool handed, int B, int T,
                        int H, int Din, const WgradReq& ride) {
    if (!handed || g_dtype != dh_dtype || esize(g_dtype) != esize(compute_dtype) || option(GCNPT_OPT_FOUR_WAVES) == 1) return false;
    const int N = (int)rows_of(B, T), Kpad = round_up(H, kstep_of(compute_dtype));
    int split = 0;
    if (colsplit_plan(N, Kpad, Din, (int)esize(compute_dtype), vec_elems(H, esize(g_dtype), g, Y), false, dh != nullptr, &split) >= 0) return false;
    SideWgrads sw;
    return plan_side_wgrad(sw, ride, B, T, compute_dtype);
}

extern "C" int gcnpt_layer_bwd_weight_multi(void* stream, int n_layers, const void* const* z_frag, const void* const* s_frag,
                                            int B, int T, const int* Din, const int* H, float* const* dW, float* const* db,
                                            int compute_dtype);

// the weight gradients of `req` in a launch of their own
static int launch_wgrads(void* stream, const WgradReq* req, int n, int B, int T, int compute_dtype) {
    constexpr int MAXR = 8;
    const void* zf[MAXR]; const void* sf[MAXR]; float* dW[MAXR]; float* db[MAXR]; int Din[MAXR], H[MAXR];
    for (int i = 0; i < n && i < MAXR; ++i) { zf[i] = req[i].z; sf[i] = req[i].s; dW[i] = req[i].dW; db[i] = req[i].db; Din[i] = req[i].Din; H[i] = req[i].H; }
    return n ? gcnpt_layer_bwd_weight_multi(stream, n, zf, sf, B, T, Din, H, dW, db, compute_dtype) : GCNPT_OK;
}

// backward-data of one layer; the weight gradient `ride` (of the layer above) rides in the launch when it can (*carried says whether it
// did); launch_behind: a gradient that could not ride is launched right after, else it is left to the caller
static int bwd_data_with_rider(void* stream, const void* dY, const void* Y, int g_dtype, const void* w_bwd, const int32_t* ell,
                               const int32_t* rowT_ptr, const int32_t* colT_idx, const int32_t* ellT, int B, int T, int Din, int H, void* dh,
                               int dh_dtype, int compute_dtype, float scale, void* z_frag, float* zero_dW, float* zero_db, const void* relu_src,
                               float next_scale, int src_is_dz, const WgradReq* ride, bool launch_behind, bool* carried_out) {
    SideWgrads sw;
    t_side = SideWgrad{};
    if (ride && plan_side_wgrad(sw, *ride, B, T, compute_dtype)) t_side.sw = &sw;
    const int rc = layer_bwd_data_impl(stream, dY, Y, g_dtype, w_bwd, ell, rowT_ptr, colT_idx, ellT, B, T, Din, H, dh, dh_dtype, compute_dtype, scale,
                                       z_frag, zero_dW, zero_db, relu_src, next_scale, src_is_dz);
    const bool carried = t_side.carried;
    t_side = SideWgrad{};
    if (carried_out) *carried_out = carried;
    if (rc != GCNPT_OK) return rc;
    return (carried || !ride || !launch_behind) ? GCNPT_OK : launch_wgrads(stream, ride, 1, B, T, compute_dtype);
}

extern "C" int gcnpt_layer_bwd_data_wgrad(void* stream, const void* dY, const void* Y, int g_dtype, const void* w_bwd,
                                          const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                                          const int32_t* ellT, int B, int T, int Din, int H, void* dh, int dh_dtype,
                                          int compute_dtype, float scale, void* z_frag, float* zero_dW, float* zero_db,
                                          const void* relu_src, float next_scale, int src_is_dz, const void* up_z_frag,
                                          const void* up_s_frag, int up_Din, int up_H, float* up_dW, float* up_db) {
    GCNPT_REQUIRE(up_z_frag && up_s_frag && up_dW && up_db && up_Din > 0 && up_H > 0, "layer_bwd_data_wgrad: the layer above's weight gradient needs "
                  "its two fragment images, dW and db");
    const WgradReq ride{up_z_frag, up_s_frag, up_Din, up_H, up_dW, up_db};
    return bwd_data_with_rider(stream, dY, Y, g_dtype, w_bwd, ell, rowT_ptr, colT_idx, ellT, B, T, Din, H, dh, dh_dtype, compute_dtype, scale, z_frag,
                               zero_dW, zero_db, relu_src, next_scale, src_is_dz, &ride, true, nullptr);
}

// ---- the whole layer loop / its autograd in one host call: the launches above, back to back (no kernel of their own) ----
constexpr int LAYERS_MAX = 8;

extern "C" int gcnpt_layers_fwd(void* stream, int n_layers, const void* x, int x_dtype, const void* const* w_fwd,
                                const float* const* bias, const int32_t* row_ptr, const int32_t* col_idx, const int32_t* ell,
                                const int32_t* deg_ell, int B, int T, const int* Din, const int* H, void* const* out,
                                const int* out_dtype, int compute_dtype, const float* drop_p, const uint64_t* seed,
                                void* const* s_frag, const uint64_t* seed_dev) {
    GCNPT_REQUIRE(n_layers >= 1 && n_layers <= LAYERS_MAX, "layers_fwd: 1..%d layers per call", LAYERS_MAX);
    GCNPT_REQUIRE(x && w_fwd && bias && Din && H && out && out_dtype && drop_p && seed, "layers_fwd: null pointer");
    for (int l = 1; l < n_layers; ++l)
        GCNPT_REQUIRE(Din[l] == H[l - 1], "layers_fwd: layer %d reads %d columns but layer %d writes %d", l, Din[l], l - 1, H[l - 1]);
    const void* h = x;
    int h_dtype = x_dtype;
    for (int l = 0; l < n_layers; ++l) {
        const int rc = layer_fwd_impl(stream, h, h_dtype, w_fwd[l], bias[l], row_ptr, col_idx, ell, deg_ell, B, T, Din[l], H[l], out[l],
                                      out_dtype[l], compute_dtype, drop_p[l], seed[l], s_frag ? s_frag[l] : nullptr, seed_dev);
        if (rc != GCNPT_OK) return rc;
        h = out[l];
        h_dtype = out_dtype[l];
    }
    return GCNPT_OK;
}

// The backward sweep, top layer first.  Every layer but the bottom one hands the layer below its dZ ready-made (dh[l] then holds dZ of
// layer l-1) and every layer but the top one receives it: one load per neighbour in the gather instead of three.  Small batches
// (<= GCNPT_OPT_SIDE_TILES row tiles): the weight gradient of layer l+1 rides in the backward-data launch of layer l, on the CUs
// without a row tile (rowtile_wgrad_kernel); what is left for the launch at the end of the sweep is the bottom layer (and any layer
// whose launch could not carry one).
static int layers_bwd_impl(void* stream, int n_layers, const void* gy, const void* const* Y, const int* y_dtype,
                           const void* const* w_bwd, const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                           const int32_t* ellT, int B, int T, const int* Din, const int* H, void* const* dh,
                           const int* dh_dtype, int compute_dtype, const float* scale, void* const* z_frag,
                           const void* const* s_frag, float* const* dW, float* const* db, bool gy_is_dz, int first_launch, int n_launches) {
    GCNPT_REQUIRE(n_layers >= 1 && n_layers <= LAYERS_MAX, "layers_bwd: 1..%d layers per call", LAYERS_MAX);
    GCNPT_REQUIRE(gy && Y && y_dtype && w_bwd && Din && H && dh && dh_dtype && scale, "layers_bwd: null pointer");
    GCNPT_REQUIRE(!z_frag || (s_frag && dW && db), "layers_bwd: weight gradients need z_frag, s_frag, dW and db");
    for (int l = 1; l < n_layers; ++l) {
        GCNPT_REQUIRE(Din[l] == H[l - 1], "layers_bwd: layer %d reads %d columns but layer %d writes %d", l, Din[l], l - 1, H[l - 1]);
        GCNPT_REQUIRE(dh[l] && dh_dtype[l] == y_dtype[l - 1], "layers_bwd: dh[%d] must exist and have the dtype of Y[%d]", l, l - 1);
    }
    const void* g = gy;
    bool wg_done[LAYERS_MAX] = {};
    int launch = 0;
    auto wanted = [&](void) { const bool w = launch >= first_launch && launch < first_launch + n_launches; ++launch; return w; };
    for (int l = n_layers - 1; l >= 0; --l) {
        if (dh[l] || z_frag) {
            const bool hand_down = l > 0, handed = l < n_layers - 1 || gy_is_dz;
            WgradReq ride{};
            const bool offer = z_frag && l + 1 < n_layers;           // carry layer l+1's gradient?
            if (offer) ride = WgradReq{z_frag[l + 1], s_frag[l + 1], Din[l + 1], H[l + 1], dW[l + 1], db[l + 1]};
            bool carried = false;
            if (wanted()) {
                const int rc = bwd_data_with_rider(stream, g, Y[l], y_dtype[l], w_bwd[l], ell, rowT_ptr, colT_idx, ellT, B, T, Din[l], H[l], dh[l],
                                                   dh_dtype[l], compute_dtype, scale[l], z_frag ? z_frag[l] : nullptr, z_frag ? dW[l] : nullptr,
                                                   z_frag ? db[l] : nullptr, hand_down ? Y[l - 1] : nullptr, hand_down ? scale[l - 1] : 1.0f,
                                                   handed ? 1 : 0, offer ? &ride : nullptr, false, &carried);
                if (rc != GCNPT_OK) return rc;
            } else if (offer) {                                      // (a launch outside the requested range: would it have carried?)
                carried = would_carry(g, handed ? nullptr : Y[l], y_dtype[l], dh[l], dh_dtype[l], compute_dtype, handed, B, T, H[l], Din[l], ride);
            }
            if (carried) wg_done[l + 1] = true;                      // what no launch carried goes into the launch at the end of the sweep
        }
        g = dh[l];
    }
    if (!z_frag) return GCNPT_OK;
    // the weight gradients no launch has carried, in one launch
    WgradReq rest[LAYERS_MAX];
    int n_r = 0;
    for (int l = 0; l < n_layers; ++l)
        if (!wg_done[l]) rest[n_r++] = WgradReq{z_frag[l], s_frag[l], Din[l], H[l], dW[l], db[l]};
    if (n_r == 0 || !wanted()) return GCNPT_OK;
    return launch_wgrads(stream, rest, n_r, B, T, compute_dtype);
}

extern "C" int gcnpt_layers_bwd(void* stream, int n_layers, const void* gy, const void* const* Y, const int* y_dtype,
                                const void* const* w_bwd, const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                                const int32_t* ellT, int B, int T, const int* Din, const int* H, void* const* dh,
                                const int* dh_dtype, int compute_dtype, const float* scale, void* const* z_frag,
                                const void* const* s_frag, float* const* dW, float* const* db) {
    return layers_bwd_impl(stream, n_layers, gy, Y, y_dtype, w_bwd, ell, rowT_ptr, colT_idx, ellT, B, T, Din, H, dh, dh_dtype, compute_dtype, scale,
                           z_frag, s_frag, dW, db, false, 0, 1 << 30);
}

// the same sweep when the caller already holds dZ of the TOP layer (gcnpt_pool3_bwd_dz leaves it): the top layer then gathers one
// row per neighbour like the layers below it, instead of dY, Y and a degree
extern "C" int gcnpt_layers_bwd_dz(void* stream, int n_layers, const void* dz_top, const void* const* Y, const int* y_dtype,
                                   const void* const* w_bwd, const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                                   const int32_t* ellT, int B, int T, const int* Din, const int* H, void* const* dh,
                                   const int* dh_dtype, int compute_dtype, const float* scale, void* const* z_frag,
                                   const void* const* s_frag, float* const* dW, float* const* db) {
    return layers_bwd_impl(stream, n_layers, dz_top, Y, y_dtype, w_bwd, ell, rowT_ptr, colT_idx, ellT, B, T, Din, H, dh, dh_dtype, compute_dtype,
                           scale, z_frag, s_frag, dW, db, true, 0, 1 << 30);
}

// launches [first_launch, first_launch + n_launches) of the sweep gcnpt_layers_bwd would enqueue (measurement: bench.py charges a launch
// t(k) - t(k-1) from truncated steps; a partial sweep leaves partial results)
extern "C" int gcnpt_layers_bwd_range(void* stream, int n_layers, const void* gy, const void* const* Y, const int* y_dtype,
                                      const void* const* w_bwd, const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                                      const int32_t* ellT, int B, int T, const int* Din, const int* H, void* const* dh,
                                      const int* dh_dtype, int compute_dtype, const float* scale, void* const* z_frag,
                                      const void* const* s_frag, float* const* dW, float* const* db, int gy_is_dz, int first_launch,
                                      int n_launches) {
    GCNPT_REQUIRE(first_launch >= 0 && n_launches >= 0, "layers_bwd_range: negative range");
    return layers_bwd_impl(stream, n_layers, gy, Y, y_dtype, w_bwd, ell, rowT_ptr, colT_idx, ellT, B, T, Din, H, dh, dh_dtype, compute_dtype, scale,
                           z_frag, s_frag, dW, db, gy_is_dz != 0, first_launch, n_launches);
}

// pack + forward sweep + backward sweep from one call, arguments marshalled once (include/gcnpt.h, gcnpt_step_t)
extern "C" int gcnpt_layers_step(void* stream, const gcnpt_step_t* st) {
    GCNPT_REQUIRE(st, "layers_step: null pointer");
    const int L = st->n_layers;
    GCNPT_REQUIRE(L >= 1 && L <= LAYERS_MAX, "layers_step: 1..%d layers per call", LAYERS_MAX);
    if (st->parts & 1) {
        const int rc = gcnpt_pack_weights_multi(stream, L, st->W, st->H, st->Din, st->compute_dtype, st->w_fwd, st->w_bwd);
        if (rc != GCNPT_OK) return rc;
    }
    if (st->parts & 2) {
        const int rc = gcnpt_layers_fwd(stream, L, st->x, st->x_dtype, st->w_fwd, st->bias, st->row_ptr, st->col_idx, st->ell, st->deg_ell, st->B, st->T,
                                        st->Din, st->H, st->out, st->out_dtype, st->compute_dtype, st->drop_p, st->seed, st->s_frag, st->seed_dev);
        if (rc != GCNPT_OK) return rc;
    }
    if (st->parts & 4) {
        const bool wg = st->z_frag[0] != nullptr;
        return layers_bwd_impl(stream, L, st->gy, st->out, st->out_dtype, st->w_bwd, st->ell_bwd ? st->ell_bwd : st->ell, st->rowT_ptr, st->colT_idx, st->ellT,
                               st->B, st->T, st->Din, st->H, st->dh, st->dh_dtype, st->compute_dtype, st->scale, wg ? st->z_frag : nullptr,
                               wg ? st->s_frag : nullptr, wg ? st->dW : nullptr, wg ? st->db : nullptr, st->gy_is_dz != 0, 0, 1 << 30);
    }
    return GCNPT_OK;
}
