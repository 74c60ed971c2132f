// Launch configurations and the host-side plan of the sentence-slice layer kernel (sent_body.h): shared by sent_part.hip (the
// instantiations) and sent_kernels.hip (planning, the weight gradient, the C-ABI glue).
#pragma once
#include "sent_body.h"

namespace gcnpt {

// Kernel configurations (CFG): the k-steps a wave keeps in registers for its half of a chunk -- 4 (K <= 256 in one chunk: the hidden
// width), 6 (K <= 384: the C2 input layer), 7 (K <= 448: the C-GCN input layer); wider rows take several chunks
constexpr int SS_CFGS = 3;
constexpr int ss_ksh(int cfg) { return cfg == 0 ? 4 : (cfg == 1 ? 6 : 7); }

struct SentPlan {
    int cfg, vec, grid;
    size_t lds;
};

// Fills the geometry fields of p (R, rtn, n_groups, n_slices, n_ct, ctw, ksteps, kc, n_chunks, wc_shift, LDS offsets, npr, magics) for a
// layer of `rows` = B x T rows; false: the shape is outside what the kernel is built for (the caller takes the row-tile form)
bool plan_sent(SentParams& p, SentPlan& plan, int B, int T, int K, int NOUT, int ct_size, int it_size, int ot_size, bool masked, bool bwd,
               int vec_in, int vec_out);

// one precision combination x one direction (sent_part.hip): combo 0 = exact f32; 1..4 = bf16 MFMA operands with (in, out) activations
// (f32, f32), (f32, bf16), (bf16, f32), (bf16, bf16); mode as MODE of sent_kernel
int sent_launch(int combo, int mode, hipStream_t s, const SentParams& p, const SentPlan& plan);

// the layer entry points (sent_kernels.hip): GCNPT_OK / an error when the launch was taken, GCNPT_SS_NOT_TAKEN when the shape or
// GCNPT_OPT_DATAFLOW keeps the layer on the row tiles
constexpr int GCNPT_SS_NOT_TAKEN = 1;
int sent_layer_fwd(hipStream_t s, const void* h, int h_dtype, const void* w_fwd, const float* bias, const int32_t* row_ptr,
                   const int32_t* col_idx, const int32_t* ell, const int32_t* deg_ell, int B, int T, int Din, int H, void* out, int out_dtype,
                   int compute_dtype, float drop_p, uint64_t seed, const uint64_t* seed_dev);
int sent_layer_bwd(hipStream_t s, const void* dY, const void* Y, int g_dtype, const void* w_bwd, const int32_t* ell, const int32_t* rowT_ptr,
                   const int32_t* colT_idx, const int32_t* ellT, int B, int T, int Din, int H, void* dh, int dh_dtype, int compute_dtype,
                   float scale, void* wg_scratch, float* zero_dW, float* zero_db, const void* relu_src, float next_scale, int src_is_dz,
                   int* n_groups_out);
size_t sent_wgrad_scratch_bytes(int B, int T, int H, int compute_dtype);
bool sent_stack_form(int n_layers, int B, int T, const int* Din, const int* H, int x_dtype, const int* out_dtype, int compute);
int sent_wgrads(hipStream_t s, int n_layers, const void* const* wg_scratch, const void* const* h_rows, const int* h_dtype, const int* n_groups,
                int B, int T, const int* Din, const int* H, float* const* dW, float* const* db, int compute_dtype);

}  // namespace gcnpt
