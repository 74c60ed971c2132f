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

// One weight gradient some launch should compute.  rows form (wgrad_rows.h): dZ rows as they lie in memory -- `dz` [N,H] in the compute
// type's storage, or, masked, dY with Y / the degrees / the dropout scale -- plus the forward's S image; image form (wgrad_common.h): both
// fragment images.
struct WgradReq {
    const void* dz = nullptr; const void* yref = nullptr; const int32_t* d_ell = nullptr; float scale = 1.0f; int masked = 0;
    const void* z_img = nullptr;
    const void* s = nullptr; int Din = 0, H = 0; float* dW = nullptr; float* db = nullptr;
};
// What a backward-data launch clears beside its own accumulators: those of the layer below ([Din x down_Din], [Din])
struct BwdExtras { float* down_zero_dW = nullptr; float* down_zero_db = nullptr; int down_Din = 0; };

namespace gcnpt {
int wgrad_rows_vec(int H, int dtype, const void* a, const void* b) {
    const size_t es = esize(dtype);
    auto al = [&](size_t n) { return (reinterpret_cast<uintptr_t>(a) % n) == 0 && (!b || (reinterpret_cast<uintptr_t>(b) % n) == 0); };
    if (H % 8 == 0 && al(16)) return 8;
    if (H % 4 == 0 && H >= 4 && al(4 * es)) return 4;
    return 0;
}

int plan_wgrad_rows(WgradRowsParams& p, const void* dz, const void* yref, const int32_t* d_ell, float scale, int masked, int rows_dtype,
                    const void* s_frag, long long N, int Din, int H, float* dW, float* db, int compute_dtype, int budget, int min_ks_per_unit) {
    p = WgradRowsParams{};
    p.stamps = static_cast<unsigned long long*>(g_debug_stamps);
    p.dz = dz; p.yref = yref; p.d_ell = d_ell; p.scale = scale; p.masked = masked;
    p.vec = wgrad_rows_vec(H, rows_dtype, dz, masked ? yref : nullptr);
    p.sf = static_cast<const uint4*>(s_frag); p.dW = dW; p.db = db;
    p.N = (int)N; p.H = H; p.Din = Din;
    p.m_tiles = ceil_div(H, 16); p.n_tiles = ceil_div(Din, 16);
    p.nks = ceil_div((int)N, 32) * (compute_dtype == GCNPT_BF16 ? 1 : 2);        // the S image's k-steps (include/gcnpt.h, gcnpt_frag_bytes)
    p.n_mblocks = ceil_div(p.m_tiles, WR_WM * WR_MW); p.mb_tiles = ceil_div(p.m_tiles, p.n_mblocks);
    p.n_nblocks = ceil_div(p.n_tiles, WR_WN * WR_NW); p.nb_tiles = ceil_div(p.n_tiles, p.n_nblocks);
    const int blocks = p.n_mblocks * p.n_nblocks;
    int slices = std::max(1, std::min(budget / blocks, ceil_div(p.nks, std::max(1, min_ks_per_unit))));
    if (option(GCNPT_OPT_DETERMINISTIC) == 1) slices = 1;      // every element of dW / db summed by one workgroup in a fixed order
    p.ks_per_unit = std::min(p.nks, round_up(ceil_div(p.nks, slices), 4));      // whole prefetch rounds (wgrad_rows.h, PF)
    p.slices = ceil_div(p.nks, p.ks_per_unit);
    return blocks * p.slices;
}

template <typename CT>
__global__ __launch_bounds__(WR_THREADS, 2) void wgrad_rows_kernel(const WgradRowsMulti mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wr_smem[];
    int layer = 0;
#pragma unroll
    for (int i = 1; i < WR_MAX_LAYERS; ++i) layer += (i < mp.n && (int)blockIdx.x >= mp.first[i]) ? 1 : 0;
    const WgradRowsParams& w = mp.l[layer];
    const int u = (int)blockIdx.x - mp.first[layer];
    if (w.masked) { if (w.vec == 8) wgrad_rows_unit<CT, 8, true>(w, u, wr_smem); else wgrad_rows_unit<CT, 4, true>(w, u, wr_smem); }
    else          { if (w.vec == 8) wgrad_rows_unit<CT, 8, false>(w, u, wr_smem); else wgrad_rows_unit<CT, 4, false>(w, u, wr_smem); }
}
}  // namespace gcnpt

// can this gradient take the rows form?  (activations in the compute type's storage, rows readable in 8- or 16-byte pieces)
static bool rows_form_ok(const WgradReq& r, int rows_dtype, int compute_dtype) {
    return r.dz && rows_dtype == compute_dtype && wgrad_rows_vec(r.H, rows_dtype, r.dz, r.masked ? r.yref : nullptr) != 0;
}

static int layer_bwd_data_impl(void* stream, const void* dY, const void* Y, int g_dtype, const void* w_bwd,
                               const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                               const int32_t* ellT, int B, int T, int Din, int H, void* dh, int dh_dtype,
                               int compute_dtype, float scale, void* z_frag, float* zero_dW, float* zero_db,
                               const void* relu_src, float next_scale, int src_is_dz, const BwdExtras& ex = BwdExtras{}) {
    GCNPT_REQUIRE(dY && (Y || src_is_dz) && w_bwd && ell && rowT_ptr && colT_idx && ellT, "layer_bwd_data: null pointer");
    GCNPT_REQUIRE(!relu_src || dh, "layer_bwd_data: relu_src without dh");
    GCNPT_REQUIRE(dh || z_frag, "layer_bwd_data: nothing to produce (dh and z_frag both NULL)");
    GCNPT_REQUIRE(B > 0 && T >= 0 && Din > 0 && H > 0, "layer_bwd_data: sizes must be positive");
    GCNPT_REQUIRE(dtype_ok(g_dtype) && dtype_ok(dh_dtype) && dtype_ok(compute_dtype), "layer_bwd_data: bad dtype");
    GCNPT_REQUIRE(!(ex.down_zero_dW || ex.down_zero_db) || ex.down_Din > 0, "layer_bwd_data: the layer below's accumulators need its input width");
    if (rows_of(B, T) > 0x7fffffffLL / 2) return fail(GCNPT_E_UNSUPPORTED, "layer_bwd_data: B*T too large");
    RowTileParams p{};
    p.stamps = static_cast<unsigned long long*>(g_debug_stamps); p.knob = g_debug_knob;
    p.src = dY; p.yref = Y; p.wfrag = w_bwd; p.bias = nullptr;
    p.g_row_ptr = rowT_ptr; p.g_col_idx = colT_idx; p.g_ell = ellT; p.d_ell = ell; p.out = dh;
    p.frag_out = z_frag;
    p.zero_p[0] = zero_dW; p.zero_n[0] = H * Din; p.zero_p[1] = zero_db; p.zero_n[1] = H;
    p.zero_p[2] = ex.down_zero_dW; p.zero_n[2] = Din * ex.down_Din; p.zero_p[3] = ex.down_zero_db; p.zero_n[3] = Din;
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

// Plans up to SIDE_MAX rows-form weight gradients for the passenger workgroups of a backward-data launch over B x T rows: taken by the
// row-tile launch of a small batch (<= GCNPT_OPT_SIDE_TILES row tiles), see rowtile_wgrad_kernel.  false: nothing can ride (the caller
// launches the gradients on their own).
static bool plan_side_wgrads(SideWgrads& sw, int& passengers, const WgradReq* req, int n, int B, int T, int rows_dtype, int compute_dtype) {
    sw = SideWgrads{};
    const long long N = rows_of(B, T);
    const int n_tiles = ceil_div((int)N, ROWS);
    if (n <= 0 || n > SIDE_MAX || n_tiles > option(GCNPT_OPT_SIDE_TILES)) return false;
    for (int i = 0; i < n; ++i)
        if (!rows_form_ok(req[i], rows_dtype, compute_dtype)) return false;
    passengers = std::max(64, 256 - round_up(n_tiles, 8));          // one workgroup per CU in all
    // the passengers' budget is shared by the riding layers in proportion to their blocks; a unit is at least 4 k-steps long
    int blocks[SIDE_MAX] = {}, total = 0;
    for (int i = 0; i < n; ++i) {
        blocks[i] = ceil_div(ceil_div(req[i].H, 16), WR_WM * WR_MW) * ceil_div(ceil_div(req[i].Din, 16), WR_WN * WR_NW);
        total += blocks[i];
    }
    for (int i = 0; i < SIDE_MAX; ++i) {
        sw.first[i + 1] = sw.first[i];
        if (i >= n) continue;
        const int budget = std::max(blocks[i], passengers * blocks[i] / total);
        sw.first[i + 1] += plan_wgrad_rows(sw.l[i], req[i].dz, req[i].yref, req[i].d_ell, req[i].scale, req[i].masked, rows_dtype, req[i].s, N,
                                           req[i].Din, req[i].H, req[i].dW, req[i].db, compute_dtype, budget, 4);
        sw.vec[i] = sw.l[i].vec; sw.masked[i] = sw.l[i].masked;
    }
    // every unit on a passenger (round-robin) unless there are more than two per passenger: the rest then goes to the row-tile
    // workgroups, one unit each after their tile (the units need nothing this launch computes)
    const int units = sw.first[SIDE_MAX];
    sw.tile_unit0 = std::min(units, std::max(2 * passengers, units - n_tiles));
    return true;
}

extern "C" int gcnpt_layer_bwd_weight_multi(void* stream, int n_layers, const void* const* z_frag, const void* const* s_frag,
                                            int B, int T, const int* Din, const int* H, float* const* dW, float* const* db,
                                            int compute_dtype);

// the weight gradients of `req` in launches of their own: the rows-form ones in ONE launch, the image-form ones in another
static int launch_wgrads(void* stream, const WgradReq* req, int n, int B, int T, int rows_dtype, int compute_dtype) {
    constexpr int MAXR = 8;
    const void* zf[MAXR]; const void* sf[MAXR]; float* dW[MAXR]; float* db[MAXR]; int Din[MAXR], H[MAXR];
    int n_img = 0, n_rows = 0, blocks[MAXR], total = 0;
    const WgradReq* rows[MAXR];
    const long long N = rows_of(B, T);
    for (int i = 0; i < n && i < MAXR; ++i) {
        if (rows_form_ok(req[i], rows_dtype, compute_dtype)) {
            blocks[n_rows] = ceil_div(ceil_div(req[i].H, 16), WR_WM * WR_MW) * ceil_div(ceil_div(req[i].Din, 16), WR_WN * WR_NW);
            total += blocks[n_rows];
            rows[n_rows++] = &req[i];
        } else {
            GCNPT_REQUIRE(req[i].z_img, "layer_bwd_weight: this layer's dZ rows cannot be read in 8-byte pieces (width %d) and no fragment image "
                          "was provided", req[i].H);
            zf[n_img] = req[i].z_img; sf[n_img] = req[i].s; dW[n_img] = req[i].dW; db[n_img] = req[i].db; Din[n_img] = req[i].Din; H[n_img] = req[i].H;
            ++n_img;
        }
    }
    if (n_rows) {
        WgradRowsMulti mp{};
        mp.n = n_rows;
        // one workgroup per CU; from 16 k rows on two rounds' worth of units so that the tail is short
        const int budget_all = N >= 16384 ? 512 : 256;
        for (int i = 0; i < n_rows; ++i) {
            const WgradReq& r = *rows[i];
            mp.first[i + 1] = mp.first[i] + plan_wgrad_rows(mp.l[i], r.dz, r.yref, r.d_ell, r.scale, r.masked, rows_dtype, r.s, N, r.Din, r.H, r.dW, r.db,
                                                            compute_dtype, std::max(blocks[i], budget_all * blocks[i] / total), 4);
        }
        const size_t lds = wgrad_rows_lds(compute_dtype);
        hipStream_t s = (hipStream_t)stream;
        if (compute_dtype == GCNPT_BF16) {
            GCNPT_LDS_ATTR_ONCE(wgrad_rows_kernel<bf16_t>, 160 * 1024);
            hipLaunchKernelGGL(wgrad_rows_kernel<bf16_t>, dim3(mp.first[n_rows]), dim3(WR_THREADS), lds, s, mp);
        } else {
            GCNPT_LDS_ATTR_ONCE(wgrad_rows_kernel<float>, 160 * 1024);
            hipLaunchKernelGGL(wgrad_rows_kernel<float>, dim3(mp.first[n_rows]), dim3(WR_THREADS), lds, s, mp);
        }
        GCNPT_HIP_CHECK(hipGetLastError());
        note_launch(mp.first[n_rows], WR_THREADS, lds, sizeof(mp));
    }
    return n_img ? gcnpt_layer_bwd_weight_multi(stream, n_img, zf, sf, B, T, Din, H, dW, db, compute_dtype) : GCNPT_OK;
}

// backward-data of one layer + the extras; the weight gradients of `req` ride in the launch when they can, else follow it
static int bwd_data_with_riders(void* stream, const void* dY, const void* Y, int g_dtype, const void* w_bwd, const int32_t* ell,
                                const int32_t* rowT_ptr, const int32_t* colT_idx, const int32_t* ellT, int B, int T, int Din, int H, void* dh,
                                int dh_dtype, int compute_dtype, float scale, void* z_frag, float* zero_dW, float* zero_db, const void* relu_src,
                                float next_scale, int src_is_dz, const BwdExtras& ex, const WgradReq* req, int n_req, int rows_dtype) {
    SideWgrads sw;
    t_side = SideWgrad{};
    int passengers = 0;
    if (n_req > 0 && plan_side_wgrads(sw, passengers, req, n_req, B, T, rows_dtype, compute_dtype)) { t_side.sw = &sw; t_side.passengers = passengers; }
    const int rc = layer_bwd_data_impl(stream, dY, Y, g_dtype, w_bwd, ell, rowT_ptr, colT_idx, ellT, B, T, Din, H, dh, dh_dtype, compute_dtype, scale,
                                       z_frag, zero_dW, zero_db, relu_src, next_scale, src_is_dz, ex);
    const bool carried = t_side.carried;
    t_side = SideWgrad{};
    if (rc != GCNPT_OK) return rc;
    return carried ? GCNPT_OK : launch_wgrads(stream, req, n_req, B, T, rows_dtype, compute_dtype);
}

// weight gradient of one layer from its dZ ROWS (or dY, Y, degrees, scale: masked) and the forward's S image, in a launch of its own
extern "C" int gcnpt_layer_bwd_weight_rows(void* stream, const void* dz, const void* Y, const int32_t* ell, float scale, int rows_dtype,
                                           const void* s_frag, int B, int T, int Din, int H, float* dW, float* db, int compute_dtype) {
    GCNPT_REQUIRE(dz && s_frag && dW && db, "layer_bwd_weight_rows: null pointer");
    GCNPT_REQUIRE(!Y || ell, "layer_bwd_weight_rows: Y (dz is dY) needs the ELL head for the degrees");
    GCNPT_REQUIRE(B > 0 && T >= 0 && Din > 0 && H > 0, "layer_bwd_weight_rows: sizes must be positive");
    GCNPT_REQUIRE(dtype_ok(rows_dtype) && dtype_ok(compute_dtype), "layer_bwd_weight_rows: bad dtype");
    WgradReq r;
    r.dz = dz; r.yref = Y; r.d_ell = ell; r.scale = scale; r.masked = Y ? 1 : 0; r.s = s_frag; r.Din = Din; r.H = H; r.dW = dW; r.db = db;
    if (!rows_form_ok(r, rows_dtype, compute_dtype))
        return fail(GCNPT_E_UNSUPPORTED, "layer_bwd_weight_rows: rows must be in the compute dtype's storage, %d-byte aligned, width %% 4 == 0 "
                    "(width %d): use gcnpt_layer_bwd_weight with the dZ image", (int)(4 * esize(rows_dtype)), H);
    return launch_wgrads(stream, &r, 1, B, T, rows_dtype, compute_dtype);
}

extern "C" int gcnpt_layer_bwd_data_ex(void* stream, const void* dY, const void* Y, int g_dtype, const void* w_bwd,
                                       const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                                       const int32_t* ellT, int B, int T, int Din, int H, void* dh, int dh_dtype,
                                       int compute_dtype, float scale, void* z_frag, float* zero_dW, float* zero_db,
                                       const void* relu_src, float next_scale, int src_is_dz, float* down_zero_dW,
                                       float* down_zero_db, int down_Din, int n_riders, const void* const* r_dz, const void* const* r_Y,
                                       const float* r_scale, const void* const* r_s_frag, const int* r_Din, const int* r_H, float* const* r_dW,
                                       float* const* r_db) {
    GCNPT_REQUIRE(n_riders >= 0 && n_riders <= SIDE_MAX, "layer_bwd_data_ex: 0..%d weight gradients can ride", SIDE_MAX);
    GCNPT_REQUIRE(n_riders == 0 || (r_dz && r_Y && r_scale && r_s_frag && r_Din && r_H && r_dW && r_db), "layer_bwd_data_ex: null rider arrays");
    WgradReq req[SIDE_MAX];
    for (int i = 0; i < n_riders; ++i) {
        GCNPT_REQUIRE(r_dz[i] && r_s_frag[i] && r_dW[i] && r_db[i] && r_Din[i] > 0 && r_H[i] > 0,
                      "layer_bwd_data_ex: rider %d needs its dZ rows, its S image, dW, db and positive widths", i);
        req[i].dz = r_dz[i]; req[i].yref = r_Y[i]; req[i].d_ell = ell; req[i].scale = r_scale[i]; req[i].masked = r_Y[i] ? 1 : 0;
        req[i].s = r_s_frag[i]; req[i].Din = r_Din[i]; req[i].H = r_H[i]; req[i].dW = r_dW[i]; req[i].db = r_db[i];
        GCNPT_REQUIRE(rows_form_ok(req[i], g_dtype, compute_dtype), "layer_bwd_data_ex: rider %d: rows must be in the compute dtype's storage, "
                      "aligned, width %% 4 == 0", i);
    }
    BwdExtras ex;
    ex.down_zero_dW = down_zero_dW; ex.down_zero_db = down_zero_db; ex.down_Din = down_Din;
    return bwd_data_with_riders(stream, dY, Y, g_dtype, w_bwd, ell, rowT_ptr, colT_idx, ellT, B, T, Din, H, dh, dh_dtype, compute_dtype, scale, z_frag,
                                zero_dW, zero_db, relu_src, next_scale, src_is_dz, ex, req, n_riders, g_dtype);
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
// layer l-1) and every layer but the top one receives it: one load per neighbour in the gather instead of three.
// Weight gradients.  A layer whose dZ rows are in the compute type's storage and readable in 8-byte pieces takes the ROWS form
// (wgrad_rows.h: no dZ fragment image at all): its gradient only needs rows that exist once the launch ABOVE its own backward-data launch
// has ended (the top layer's: dY and Y, from the start).  Small batches (<= GCNPT_OPT_SIDE_TILES row tiles): the gradient of layer l then
// rides in layer l's backward-data launch, on the CUs without a row tile, the top layer's in the launch below it, and the launch above
// clears the accumulators -- an L >= 2 sweep is L launches.  Big batches: one launch at the end for all of them.  Other layers (odd
// widths, mixed precisions) write their dZ image as before (z_frag[l] required) and share that last launch's image-form kernel.
static int layers_bwd_impl(void* stream, int n_layers, const void* gy, const void* const* Y, const int* y_dtype,
                           const void* const* w_bwd, const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                           const int32_t* ellT, int B, int T, const int* Din, const int* H, void* const* dh,
                           const int* dh_dtype, int compute_dtype, const float* scale, void* const* z_frag,
                           const void* const* s_frag, float* const* dW, float* const* db, bool gy_is_dz, int first_launch, int n_launches) {
    GCNPT_REQUIRE(n_layers >= 1 && n_layers <= LAYERS_MAX, "layers_bwd: 1..%d layers per call", LAYERS_MAX);
    GCNPT_REQUIRE(gy && Y && y_dtype && w_bwd && Din && H && dh && dh_dtype && scale, "layers_bwd: null pointer");
    const bool want_w = s_frag != nullptr;
    GCNPT_REQUIRE(!want_w || (dW && db), "layers_bwd: weight gradients need s_frag, dW and db");
    for (int l = 1; l < n_layers; ++l) {
        GCNPT_REQUIRE(Din[l] == H[l - 1], "layers_bwd: layer %d reads %d columns but layer %d writes %d", l, Din[l], l - 1, H[l - 1]);
        GCNPT_REQUIRE(dh[l] && dh_dtype[l] == y_dtype[l - 1], "layers_bwd: dh[%d] must exist and have the dtype of Y[%d]", l, l - 1);
    }
    const int top = n_layers - 1;
    const int n_tiles = ceil_div((int)rows_of(B, T), ROWS);
    // the gradient requests, and which form each takes
    WgradReq req[LAYERS_MAX];
    bool rows_form[LAYERS_MAX] = {}, wg_done[LAYERS_MAX] = {};
    if (want_w) {
        for (int l = 0; l < n_layers; ++l) {
            GCNPT_REQUIRE(s_frag[l] && dW[l] && db[l], "layers_bwd: null pointer (weight gradient of layer %d)", l);
            WgradReq& r = req[l];
            r.s = s_frag[l]; r.Din = Din[l]; r.H = H[l]; r.dW = dW[l]; r.db = db[l]; r.d_ell = ell;
            if (l == top) { r.dz = gy; r.yref = gy_is_dz ? nullptr : Y[top]; r.masked = gy_is_dz ? 0 : 1; r.scale = scale[top]; }
            else          { r.dz = dh[l + 1]; }
            rows_form[l] = rows_form_ok(r, y_dtype[l], compute_dtype);
            r.z_img = z_frag ? z_frag[l] : nullptr;
            GCNPT_REQUIRE(rows_form[l] || r.z_img, "layers_bwd: layer %d needs z_frag[%d] (its dZ rows are not in the compute dtype's storage or "
                          "their width %d is not a multiple of 4)", l, l, H[l]);
        }
    }
    const bool riders = want_w && n_tiles <= option(GCNPT_OPT_SIDE_TILES);
    const void* g = gy;
    int launch = 0;
    auto wanted = [&](void) { const bool w = launch >= first_launch && launch < first_launch + n_launches; ++launch; return w; };
    for (int l = top; l >= 0; --l) {
        if (dh[l] || want_w) {
            const bool hand_down = l > 0, handed = l < top || gy_is_dz;
            const bool own_image = want_w && !rows_form[l];
            // accumulators this launch clears: its own unless an earlier launch did (the rows-form gradient of this layer rides HERE, so
            // the launch above cleared them -- or, for the top layer / big batches, this launch does and the gradient comes later), and
            // the layer below's when that layer's gradient will ride in its launch
            const bool cleared_above = riders && rows_form[l] && l < top;
            BwdExtras ex;
            if (riders && hand_down && rows_form[l - 1]) { ex.down_zero_dW = dW[l - 1]; ex.down_zero_db = db[l - 1]; ex.down_Din = Din[l - 1]; }
            WgradReq ride[SIDE_MAX];
            int ride_l[SIDE_MAX], n_ride = 0;
            if (riders && handed) {
                if (l == top - 1 && rows_form[top] && !wg_done[top]) { ride[n_ride] = req[top]; ride_l[n_ride++] = top; }
                if (l < top && rows_form[l] && !wg_done[l]) { ride[n_ride] = req[l]; ride_l[n_ride++] = l; }
                if (l == top && gy_is_dz && n_layers == 1) { /* a single layer: its accumulators are cleared by this very launch */ }
            }
            const bool clear_own = want_w && !cleared_above;
            if (!dh[l] && !own_image && !clear_own) {
                // no input gradient wanted, no image to write, nothing to clear: the launch would only carry the riders
                if (n_ride > 0 && wanted()) {
                    const int rc = launch_wgrads(stream, ride, n_ride, B, T, y_dtype[l], compute_dtype);
                    if (rc != GCNPT_OK) return rc;
                }
            } else if (wanted()) {
                const int rc = bwd_data_with_riders(stream, g, Y[l], y_dtype[l], w_bwd[l], ell, rowT_ptr, colT_idx, ellT, B, T, Din[l], H[l], dh[l],
                                                    dh_dtype[l], compute_dtype, scale[l], own_image ? z_frag[l] : nullptr,
                                                    clear_own ? dW[l] : nullptr, clear_own ? db[l] : nullptr, hand_down ? Y[l - 1] : nullptr,
                                                    hand_down ? scale[l - 1] : 1.0f, handed ? 1 : 0, ex, ride, n_ride, y_dtype[l]);
                if (rc != GCNPT_OK) return rc;
            }
            for (int i = 0; i < n_ride; ++i) wg_done[ride_l[i]] = true;          // carried, or launched right behind
        }
        g = dh[l];
    }
    if (!want_w) return GCNPT_OK;
    // the weight gradients no launch has carried, in one launch per form
    WgradReq rest[LAYERS_MAX];
    int n_r = 0;
    for (int l = 0; l < n_layers; ++l)
        if (!wg_done[l]) rest[n_r++] = req[l];
    if (n_r == 0 || !wanted()) return GCNPT_OK;
    return launch_wgrads(stream, rest, n_r, B, T, y_dtype[top], compute_dtype);
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
