// Host side of the sentence-slice layer kernels (sent_body.h): the plan (groups of whole sentences x column slices, chunking, LDS),
// the launch table, and the layer entry points the C-ABI dispatches to (rowtile_kernels.hip: gcnpt_layer_fwd / _bwd_data, gcnpt_layers_*).
#include "sent_common.h"
#include "sent_wgrad.h"

using namespace gcnpt;

namespace gcnpt {

#define GCNPT_SS_DECL(n) int sent_launch_part##n(hipStream_t s, const SentParams& p, int cfg, int vec, size_t lds, int grid);
GCNPT_SS_DECL(0) GCNPT_SS_DECL(1) GCNPT_SS_DECL(2) GCNPT_SS_DECL(3) GCNPT_SS_DECL(4) GCNPT_SS_DECL(5) GCNPT_SS_DECL(6) GCNPT_SS_DECL(7)
GCNPT_SS_DECL(8) GCNPT_SS_DECL(9) GCNPT_SS_DECL(10) GCNPT_SS_DECL(11) GCNPT_SS_DECL(12) GCNPT_SS_DECL(13) GCNPT_SS_DECL(14)
int sent_launch(int combo, int mode, hipStream_t s, const SentParams& p, const SentPlan& plan) {
    typedef int (*fn_t)(hipStream_t, const SentParams&, int, int, size_t, int);
    static const fn_t table[15] = {sent_launch_part0, sent_launch_part1, sent_launch_part2, sent_launch_part3, sent_launch_part4,
                                   sent_launch_part5, sent_launch_part6, sent_launch_part7, sent_launch_part8, sent_launch_part9,
                                   sent_launch_part10, sent_launch_part11, sent_launch_part12, sent_launch_part13, sent_launch_part14};
    return table[combo * 3 + mode](s, p, plan.cfg, plan.vec, plan.lds, plan.grid);
}

static unsigned magic_of(int d) { return (unsigned)(0xffffffffu / (unsigned)d + 1u); }     // ceil(2^32 / d) for d >= 2 that is no power of two; exact use: x * d < 2^32

// A workgroup = (spg whole sentences) x (one of n_slices column slices of <= 4 tiles).  All shapes that fit are scored by what one
// workgroup pulls through its CU (its rows, its weight slice, its output) times the rounds the grid needs: per-CU delivery, not HBM,
// bounds a launch of this size (~100 KB per workgroup at the C2 shape against ~25 KB if the chip's HBM rate were the limit).
// bwd: the side outputs for the weight gradient are wanted (a slice then also stages its share of the K columns of all its rows).
bool plan_sent(SentParams& p, SentPlan& plan, int B, int T, int K, int NOUT, int ct_size, int it_size, int ot_size, bool masked, bool bwd,
               int vec_in, int vec_out) {
    if (T < 2 || B < 1 || vec_in < 4 || vec_out < 8 || (ot_size == 4 && vec_out != 16) || K < 8 || NOUT < 8) return false;
    const int kstep = ct_size == 2 ? 32 : 16;
    const int ksteps = ceil_div(K, kstep), n_ct = ceil_div(NOUT, 16);
    const int forced_slices = option(GCNPT_OPT_SENT_SLICES);
    // registers: a wave keeps 2 row tiles x ksh k-steps of fragments (fp32 rows feeding bf16 fragments take twice the registers, the
    // dZ-deriving loader two streams): the widest configuration that still fits
    const int heavy = (it_size * (ct_size == 2 ? 8 : 4) == 32 ? 2 : 1) * (masked ? 2 : 1);
    const int ksh_cap = heavy >= 4 ? 2 : (heavy == 2 ? ss_ksh(0) : ss_ksh(SS_CFGS - 1));
    long long best = -1;
    const int max_spg = std::max(1, 128 / T);
    for (int spg = 1; spg <= std::min(max_spg, B); ++spg) {
        const int R = spg * T, rtn = ceil_div(R, 16);
        if (rtn > SS_RTMAX) continue;
        const int n_groups = ceil_div(B, spg);
        for (int n_slices = ceil_div(n_ct, SS_CTW); n_slices <= n_ct; ++n_slices) {
            if (forced_slices > 0 && n_slices != std::max(std::min(forced_slices, n_ct), ceil_div(n_ct, SS_CTW))) continue;
            const int ctw = ceil_div(n_ct, n_slices);
            const int zq = bwd ? ceil_div(ceil_div(K, 8), n_slices) : 0;
            if (zq > 8) continue;
            const int RP = rtn * 16;
            const int ksh = std::min(ceil_div(ksteps, 2), ksh_cap), kc = std::min(ksteps, 2 * ksh), n_chunks = ceil_div(ksteps, kc);
            const int cfg = ksh <= ss_ksh(0) ? 0 : (ksh <= ss_ksh(1) ? 1 : 2);
            const size_t w_bytes = (size_t)ctw * kc * 1024, p_bytes = 2 * (size_t)RP * SS_PSTRIDE * 4;
            const size_t z_bytes = bwd ? ((size_t)RP * (zq * 8 + 8) * ct_size + 15) / 16 * 16 : 0, meta = (size_t)RP * 40 + 256;
            const size_t lds = w_bytes + p_bytes + z_bytes + meta;
            if (lds > 160 * 1024) continue;
            // every slice of a group runs on the group's XCD (32 CUs, one workgroup each): what does not fit waits for a second round
            const long long rounds = ceil_div(ceil_div(n_groups, 8) * n_slices, 32);
            const long long bytes = (long long)R * K * it_size * (masked ? 2 : 1) + (long long)ctw * 16 * ksteps * kstep * ct_size +
                                    (long long)R * ctw * 16 * ot_size + 24576 + (long long)(n_chunks - 1) * 16384;
            const long long cost = rounds * bytes;
            if (best >= 0 && cost >= best) continue;
            best = cost;
            p.R = R; p.rtn = rtn; p.n_groups = n_groups; p.n_slices = n_slices; p.n_ct = n_ct;
            p.ksteps = ksteps; p.kc = kc; p.ksh = ceil_div(kc, 2); p.n_chunks = n_chunks;
            p.p_off = (int)w_bytes; p.z_off = (int)(w_bytes + p_bytes); p.meta_off = (int)(w_bytes + p_bytes + z_bytes);
            p.zq = zq;
            p.t_magic = magic_of(T);
            plan.cfg = cfg; plan.vec = vec_in >= 8 ? 8 : 4; plan.lds = lds;
            plan.grid = 8 * ceil_div(n_groups, 8) * n_slices;
        }
    }
    return best >= 0;
}

static int combo_of(int in_dtype, int out_dtype, int compute) {
    if (compute == GCNPT_F32) return (in_dtype == GCNPT_F32 && out_dtype == GCNPT_F32) ? 0 : -1;
    return in_dtype == GCNPT_F32 ? (out_dtype == GCNPT_F32 ? 1 : 2) : (out_dtype == GCNPT_F32 ? 3 : 4);
}

// how rows of `width` elements of `es` bytes at base a (and b) may be read: 8 elements per load, 4 (half loads), else 0
static int ss_vec_elems(int width, size_t es, const void* a, const void* b) {
    auto al = [&](size_t n) { return (reinterpret_cast<uintptr_t>(a) % n) == 0 && (!b || (reinterpret_cast<uintptr_t>(b) % n) == 0); };
    if (width % 8 == 0 && al(16)) return 8;
    if (width % 4 == 0 && width >= 4 && al(4 * es)) return 4;
    return 0;
}
static int ss_vec_bytes(int width, size_t es, const void* a, const void* b) {
    auto al = [&](size_t n) { return (reinterpret_cast<uintptr_t>(a) % n) == 0 && (!b || (reinterpret_cast<uintptr_t>(b) % n) == 0); };
    if ((width * es) % 16 == 0 && al(16)) return 16;
    if ((width * es) % 8 == 0 && al(8)) return 8;
    return 0;
}

// ---- which form a stack runs in (include/gcnpt.h, GCNPT_OPT_DATAFLOW) ----------------------------------------------------------
// A pure function of the option, the shapes and the dtypes, so that gcnpt_layers_fwd and gcnpt_layers_bwd* of one step agree without
// telling each other: the sentence-slice form is taken when EVERY layer's forward and backward launch can be planned for it (rows are
// assumed aligned as torch allocates them; a launch whose pointers turn out not to be fails loudly instead of changing form).
static bool plan_ok(int B, int T, int K, int NOUT, int compute, int in_dtype, int out_dtype, bool masked, bool bwd) {
    if (combo_of(in_dtype, out_dtype, compute) < 0) return false;
    SentParams p{};
    SentPlan plan{};
    const int vin = K % 8 == 0 ? 8 : (K % 4 == 0 ? 4 : 0);
    const size_t ob = (size_t)NOUT * esize(out_dtype);
    const int vout = ob % 16 == 0 ? 16 : (ob % 8 == 0 ? 8 : 0);
    return plan_sent(p, plan, B, T, K, NOUT, (int)esize(compute), (int)esize(in_dtype), (int)esize(out_dtype), masked, bwd, vin, vout);
}

bool sent_stack_form(int n_layers, int B, int T, const int* Din, const int* H, int x_dtype, const int* out_dtype, int compute) {
    if (option(GCNPT_OPT_DATAFLOW) == 0 || T <= 0) return false;
    for (int l = 0; l < n_layers; ++l) {
        const int in_dtype = l == 0 ? x_dtype : out_dtype[l - 1];
        if (!plan_ok(B, T, Din[l], H[l], compute, in_dtype, out_dtype[l], false, false)) return false;
        // backward: gradients arrive in the layer's output dtype and leave in its input dtype; the top layer may have to derive dZ itself
        if (!plan_ok(B, T, H[l], Din[l], compute, out_dtype[l], in_dtype, false, true)) return false;
        if (l == n_layers - 1 && !plan_ok(B, T, H[l], Din[l], compute, out_dtype[l], in_dtype, true, true)) return false;
        // the weight gradient reads rows of G (compute type) and of the layer's input: 8-byte aligned pieces at least
        if ((H[l] * esize(compute)) % 8 != 0 || (Din[l] * esize(in_dtype)) % 8 != 0) return false;
        if (compute == GCNPT_F32 && in_dtype != GCNPT_F32) return false;
    }
    return true;
}

// forward of one layer in the sentence-slice form; GCNPT_SS_NOT_TAKEN when the shape keeps it on the row tiles
int sent_layer_fwd(hipStream_t s, const void* h, int h_dtype, const void* w_fwd, const float* bias, const int32_t* row_ptr,
                   const int32_t* col_idx, const int32_t* ell, const int32_t* deg_ell, int B, int T, int Din, int H, void* out, int out_dtype,
                   int compute_dtype, float drop_p, uint64_t seed, const uint64_t* seed_dev) {
    if (option(GCNPT_OPT_DATAFLOW) == 0 || T <= 0) return GCNPT_SS_NOT_TAKEN;
    const int combo = combo_of(h_dtype, out_dtype, compute_dtype);
    if (combo < 0) return GCNPT_SS_NOT_TAKEN;
    SentParams p{};
    SentPlan plan{};
    if (!plan_sent(p, plan, B, T, Din, H, (int)esize(compute_dtype), (int)esize(h_dtype), (int)esize(out_dtype), false, false,
                   ss_vec_elems(Din, esize(h_dtype), h, nullptr), ss_vec_bytes(H, esize(out_dtype), out, nullptr)))
        return GCNPT_SS_NOT_TAKEN;
    p.stamps = static_cast<unsigned long long*>(g_debug_stamps); p.knob = g_debug_knob;
    p.src = h; p.wfrag = w_fwd; p.bias = bias;
    p.g_row_ptr = row_ptr; p.g_col_idx = col_idx; p.g_ell = ell; p.d_ell = deg_ell ? deg_ell : ell; p.out = out;
    p.N = B * T; p.T = T; p.K = Din; p.NOUT = H;
    p.vec_out = ss_vec_bytes(H, esize(out_dtype), out, nullptr);
    p.drop_p = drop_p; p.scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    p.drop_thresh16 = (unsigned)((double)drop_p * 65536.0);
    p.seed = seed; p.seed_dev = seed_dev;
    return sent_launch(combo, 0, s, p, plan);
}

// What the backward-data launch of a layer leaves for its weight gradient, in one caller-owned buffer: the rows of G (compute type),
// then the per-group column sums of dZ (at most one group per sentence)
static size_t g_rows_bytes(int B, int T, int H, int compute_dtype) { return ((size_t)B * T * H * esize(compute_dtype) + 255) / 256 * 256; }
size_t sent_wgrad_scratch_bytes(int B, int T, int H, int compute_dtype) {
    if (T <= 0) return 0;
    return g_rows_bytes(B, T, H, compute_dtype) + (size_t)B * H * sizeof(float);
}

int sent_layer_bwd(hipStream_t s, const void* dY, const void* Y, int g_dtype, const void* w_bwd, const int32_t* ell, const int32_t* rowT_ptr,
                   const int32_t* colT_idx, const int32_t* ellT, int B, int T, int Din, int H, void* dh, int dh_dtype, int compute_dtype,
                   float scale, void* wg_scratch, float* zero_dW, float* zero_db, const void* relu_src, float next_scale, int src_is_dz,
                   int* n_groups_out) {
    if (option(GCNPT_OPT_DATAFLOW) == 0 || T <= 0) return GCNPT_SS_NOT_TAKEN;
    const int combo = combo_of(g_dtype, dh_dtype, compute_dtype);
    if (combo < 0) return GCNPT_SS_NOT_TAKEN;
    SentParams p{};
    SentPlan plan{};
    const int vin = ss_vec_elems(H, esize(g_dtype), dY, src_is_dz ? nullptr : Y);
    // (no dh wanted: the launch only leaves G and the column sums; planned like the real one, its matrix phase is skipped)
    const int vout = dh ? ss_vec_bytes(Din, esize(dh_dtype), dh, relu_src) : 16;
    if (!plan_sent(p, plan, B, T, H, Din, (int)esize(compute_dtype), (int)esize(g_dtype), (int)esize(dh_dtype), !src_is_dz, true, vin, vout))
        return GCNPT_SS_NOT_TAKEN;
    p.stamps = static_cast<unsigned long long*>(g_debug_stamps); p.knob = g_debug_knob;
    p.src = dY; p.yref = Y; p.wfrag = w_bwd;
    p.g_row_ptr = rowT_ptr; p.g_col_idx = colT_idx; p.g_ell = ellT; p.d_ell = ell; p.out = dh;
    if (wg_scratch) {
        p.g_out = wg_scratch;
        p.dbpart = reinterpret_cast<float*>(static_cast<unsigned char*>(wg_scratch) + g_rows_bytes(B, T, H, compute_dtype));
        p.vec_k = ss_vec_bytes(H, esize(compute_dtype), wg_scratch, nullptr);
        if (p.vec_k < 8) return fail(GCNPT_E_INVALID, "layers_bwd: the weight-gradient scratch buffer must be 16-byte aligned");
    }
    p.zero_p[0] = zero_dW; p.zero_n[0] = H * Din; p.zero_p[1] = zero_db; p.zero_n[1] = H;
    p.N = B * T; p.T = T; p.K = H; p.NOUT = Din;
    p.vec_out = vout;
    p.scale = scale;
    p.relu_src = relu_src; p.next_scale = next_scale;
    if (n_groups_out) *n_groups_out = p.n_groups;
    return sent_launch(combo, src_is_dz ? 2 : 1, s, p, plan);
}

// ---- the weight gradients of a stack from rows, one launch (sent_wgrad.h) ----
template <typename CT, typename HT, int VB>
static int launch_sent_wgrad(hipStream_t s, const SentWgradMulti& mp) {
    const size_t lds = sent_wgrad_lds(sizeof(CT));
    GCNPT_LDS_ATTR_ONCE((sent_wgrad_kernel<CT, HT, VB>), 160 * 1024);
    hipLaunchKernelGGL((sent_wgrad_kernel<CT, HT, VB>), dim3(mp.first[mp.n]), dim3(SW_WAVES * WAVE), lds, s, mp);
    note_launch(mp.first[mp.n], SW_WAVES * WAVE, lds, sizeof(mp));
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

// wg_scratch[l] as sent_layer_bwd left it; h_rows[l] / h_dtype[l]: layer l's input rows; n_groups[l]: the groups of its backward launch.
// All layers of a launch must share the input dtype class (the kernel is instantiated per (compute, input) type pair).
int sent_wgrads(hipStream_t s, int n_layers, const void* const* wg_scratch, const void* const* h_rows, const int* h_dtype, const int* n_groups,
                int B, int T, const int* Din, const int* H, float* const* dW, float* const* db, int compute_dtype) {
    SentWgradMulti mp{};
    mp.n = n_layers;
    const int rk = compute_dtype == GCNPT_BF16 ? 32 : 16;
    const int nks = ceil_div(B * T, rk);
    int blocks = 0, vb = 16;
    for (int l = 0; l < n_layers; ++l) blocks += ceil_div(ceil_div(H[l], 16), WG_MT) * ceil_div(ceil_div(Din[l], 16), WG_NT);
    for (int l = 0; l < n_layers; ++l) {
        GCNPT_REQUIRE(wg_scratch[l] && h_rows[l] && dW[l] && db[l], "layers_bwd: weight gradient of layer %d: null pointer", l);
        GCNPT_REQUIRE(h_dtype[l] == h_dtype[0], "layers_bwd: the layers' inputs must share one dtype for the row-operand weight gradient");
        WeightGradParams wp;
        mp.first[l + 1] = mp.first[l] + plan_weight_grad(wp, nullptr, nullptr, nks, Din[l], H[l], dW[l], db[l], blocks, SW_WAVES, 256, WG_NT);
        SentWgradParams& q = mp.l[l];
        q.stamps = static_cast<unsigned long long*>(g_debug_stamps); q.knob = g_debug_knob;
        q.g = wg_scratch[l]; q.h = h_rows[l];
        q.dbpart = reinterpret_cast<const float*>(static_cast<const unsigned char*>(wg_scratch[l]) + g_rows_bytes(B, T, H[l], compute_dtype));
        q.dW = dW[l]; q.db = db[l]; q.N = B * T; q.H = H[l]; q.Din = Din[l]; q.n_groups = n_groups[l];
        q.m_tiles = wp.m_tiles; q.n_tiles = wp.n_tiles; q.nks = wp.nks; q.ks_per_wg = wp.ks_per_wg; q.mb = wp.mb; q.nb = wp.nb; q.slices = wp.slices;
        if (ss_vec_bytes(H[l], esize(compute_dtype), wg_scratch[l], nullptr) < 16 || ss_vec_bytes(Din[l], esize(h_dtype[l]), h_rows[l], nullptr) < 16) vb = 8;
        if (ss_vec_bytes(H[l], esize(compute_dtype), wg_scratch[l], nullptr) < 8 || ss_vec_bytes(Din[l], esize(h_dtype[l]), h_rows[l], nullptr) < 8)
            return fail(GCNPT_E_INVALID, "layers_bwd: rows of layer %d are not 8-byte aligned", l);
    }
    if (compute_dtype == GCNPT_F32) {
        GCNPT_REQUIRE(h_dtype[0] == GCNPT_F32, "layers_bwd: compute_dtype f32 needs f32 activations");
        return vb == 16 ? launch_sent_wgrad<float, float, 16>(s, mp) : launch_sent_wgrad<float, float, 8>(s, mp);
    }
    if (h_dtype[0] == GCNPT_BF16) return vb == 16 ? launch_sent_wgrad<bf16_t, bf16_t, 16>(s, mp) : launch_sent_wgrad<bf16_t, bf16_t, 8>(s, mp);
    return vb == 16 ? launch_sent_wgrad<bf16_t, float, 16>(s, mp) : launch_sent_wgrad<bf16_t, float, 8>(s, mp);
}

}  // namespace gcnpt

extern "C" size_t gcnpt_wgrad_scratch_bytes(int B, int T, int H, int dtype) {
    if (B <= 0 || T < 0 || H <= 0 || !dtype_ok(dtype)) return 0;
    const size_t a = gcnpt_frag_bytes((int)rows_of(B, T), H, dtype), b = sent_wgrad_scratch_bytes(B, T, H, dtype);
    return a > b ? a : b;
}

extern "C" int gcnpt_layers_form(int n_layers, int B, int T, const int* Din, const int* H, int x_dtype, const int* out_dtype, int compute_dtype) {
    if (n_layers < 1 || n_layers > 8 || !Din || !H || !out_dtype) return fail(GCNPT_E_INVALID, "layers_form: bad argument");
    return sent_stack_form(n_layers, B, T, Din, H, x_dtype, out_dtype, compute_dtype) ? 1 : 0;
}
