/*
 * gcnpt.h -- C-ABI of libgcnpt.so: the MI355X (gfx950) implementation of the pruned-tree GCN hot path
 * of gstoica27/gcn-over-pruned-trees.
 *
 * The reference has NO foreign-function interface for this path: its boundary is the Python nn.Module
 * surface (model/gcn.py:15-126 GCNClassifier / GCNRelationModel, model/gcn.py:128-395 GCN.forward,
 * model/tree.py:58 head_to_tree, model/tree.py:167 tree_to_adj).  The entry points below are what a
 * ctypes binding of that surface needs; each cites the reference lines it replaces.  The host-side
 * mirror that binds them lives in gcn-over-pruned-trees_amd/model/{tree,gcn}.py; INTEGRATION.md shows
 * the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; every pointer marked [dev] is a DEVICE pointer owned by the caller
 *     (PyTorch-ROCm allocations in practice); the library allocates nothing and keeps no global state except the option
 *     table below (gcnpt_set_option), which no call reads from the environment.
 *   - `stream` is a hipStream_t passed as void*.  Every call only ENQUEUES work on that stream and never
 *     synchronises, so calls can be captured into a hipGraph.
 *   - return value: 0 = enqueued, <0 = error (GCNPT_E_*); the text is in gcnpt_last_error() (thread-local).
 *   - a "row" is one token slot: row r = b*T + i of the padded [B,T,*] tensors the reference uses (or, with T = 0, a row of the
 *     token-packed layout, see gcnpt_pack_trees).
 *
 * Pruned-tree adjacency in HBM ("CSR", shared by every kernel):
 *   row_ptr  int32 [B*(T+1)]   entries of row (b,i) are col_idx[row_ptr[b*(T+1)+i] .. row_ptr[b*(T+1)+i+1])
 *                              (offsets are absolute; sentence b owns the slice [b*cap, (b+1)*cap))
 *   col_idx  int32 [B*cap]     sentence-local column (token) index, ascending inside a row
 *   label    int32 [B*cap]     the value tree_to_adj writes (deprel id / +42 / 84), optional
 *   rowT_ptr/colT_idx          the same for the transposed pattern (needed by backward)
 *   cap >= 3*T for pruned trees (nnz = 3n-2), cap = T*T for an arbitrary dense adjacency.
 *   ell      int32 [B*T*8]     "ELL head" of every row: ell[8r] = number of entries of row r (its degree incl. the
 *                              diagonal), ell[8r+1 .. 8r+7] = its first 7 columns (unused slots 0).  A kept token of a
 *                              pruned tree has <= 3-4 entries, so a 32-row tile's whole adjacency is one coalesced
 *                              1-KiB load; rows with more entries continue in row_ptr/col_idx.   ellT: transposed.
 */
#ifndef GCNPT_H
#define GCNPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCNPT_ABI_VERSION 7

/* element types of activation / gradient buffers and of the MFMA operands */
#define GCNPT_F32 0
#define GCNPT_BF16 1

/* return codes.  -2..-7 are the reference's per-sentence failures and are ALSO what the pruner writes
 * into status[b] (they mirror oracle/prune_ref.c). */
#define GCNPT_OK 0
#define GCNPT_E_INVALID -1        /* bad argument (null pointer, non-positive size, unsupported dtype) */
#define GCNPT_E_PRUNE_NEGATIVE -2 /* model/tree.py:67-79 + 194: prune < 0 crashes the fork (AttributeError) */
#define GCNPT_E_NO_SUBJECT -3     /* model/tree.py:109,113: no token with subj_pos == 0 */
#define GCNPT_E_NO_LCA -4         /* model/tree.py:112-124: entities under different roots (UnboundLocalError) */
#define GCNPT_E_CYCLE -5          /* model/tree.py:91-94: a head cycle never terminates in the reference */
#define GCNPT_E_BAD_HEAD -6       /* model/tree.py:94: head points past the sentence (IndexError) */
#define GCNPT_E_ASSERT -7         /* model/tree.py:159 */
#define GCNPT_E_CAPACITY -8       /* a sentence needs more than `cap` adjacency entries */
#define GCNPT_E_HIP -9            /* a HIP runtime call failed */
#define GCNPT_E_UNSUPPORTED -10   /* shape outside what the kernels are built for */
#define GCNPT_E_LENGTH -11        /* gcnpt_gather_trees / gcnpt_compact_trees: a sentence does not fit the width asked for */

int gcnpt_abi_version(void);
const char* gcnpt_last_error(void);

/* ---- process-wide options (the only mutable state of the library) ---------------------------------------------------------
 * Kernel-selection and numerics switches that are not per-call arguments.  The defaults come from the environment variables of
 * the same names ONCE, when the library is loaded; nothing reads the environment afterwards.  Setting an option is a relaxed
 * atomic store: it affects calls that start after it.
 *   GCNPT_OPT_DETERMINISTIC (env GCNPT_DETERMINISTIC, default 0)  1: every element of dW / db is summed by exactly ONE workgroup in a
 *       fixed order (one contraction slice): run-to-run bit-identical weight gradients, as the reference's single-device autograd
 *       gives (model/gcn.py:270-271), at the price of the split contraction's parallelism
 *   GCNPT_OPT_FOUR_WAVES    (env GCNPT_WAVES4, default -1 = by batch size)  0 / 1 forces the 8- / 4-wave form of the layer kernel
 *   GCNPT_OPT_SIDE_TILES    (env GCNPT_SIDE_TILES, default 192)  batches of up to this many 32-row tiles carry the weight gradient of
 *       layer l+1 as a passenger of layer l's backward-data launch (gcnpt_layers_bwd)
 *   GCNPT_OPT_COL_SPLIT     (env GCNPT_COL_SPLIT, default -1 = by shape)  the column-split form of the layer kernel (bf16 MFMA operands): every
 *       32-row tile is given to 2 ... 8 workgroups that gather the same rows and each produce a share of the output columns, so that a
 *       small batch of a wide layer spreads the layer's weight fragments over the CUs that have no tile.  By itself: <= 128 row tiles and
 *       >= 170 KB of weight fragments.  0: never;  n >= 1: always, with at least n workgroups per tile (tests).  Same values bit for bit */
#define GCNPT_OPT_DETERMINISTIC 0
#define GCNPT_OPT_FOUR_WAVES 1
#define GCNPT_OPT_SIDE_TILES 2
#define GCNPT_OPT_COL_SPLIT 3
#define GCNPT_OPT_COUNT 4
int gcnpt_set_option(int option, int value);
int gcnpt_get_option(int option);

/* ---- the data-parallel update (SURVEY.md 8 row e; reference train.py:224-227: clip_grad_norm_(max_grad_norm) then SGD) ---------------------
 * On the flat fp32 buffers the all-reduced gradient bucket and the parameters live in (shard.FlatGradBucket):
 *     g_eff = g * g_scale   (g_scale = 1 / world after a SUM all-reduce)
 *     coef  = max_norm > 0 ? min(1, max_norm / (sqrt(sum(g_eff^2) + *extra_sq) + 1e-6)) : 1      (torch.nn.utils.clip_grad_norm_'s coefficient)
 *     w    -= lr * coef * g_eff
 * Two launches, no host sync.  partials: [dev] float[65] scratch, no initialisation needed; partials[64] = coef afterwards (for the caller's
 * row-sparse parameters, whose squared norm -- already scaled -- comes in through extra_sq [dev] float[1] or NULL).  w, g 16-byte aligned. */
int gcnpt_sgd_clip_update(void* stream, float* w, const float* g, long long n, float g_scale, float max_norm, float lr, float* partials,
                          const float* extra_sq);

/* ---- measurement aids (no reference counterpart) --------------------------------------------------------------
 * gcnpt_last_launch: grid, workgroup size, dynamic LDS bytes and kernel-argument bytes of the calling thread's most recent launch of
 * the layer path (pack, layer forward / backward-data, weight gradient).  gcnpt_launch_empty: enqueues a kernel of that shape whose
 * body returns at entry.  bench.py's launch-floor leg replays a step's launches with empty bodies to show how much of a step is
 * dispatch / ramp / drain rather than kernel work. */
int gcnpt_last_launch(int* grid, int* block, int* lds_bytes, int* kernarg_bytes);
int gcnpt_launch_empty(void* stream, int grid, int block, int lds_bytes, int kernarg_bytes);
/* n of them back to back from ONE native call (host arrays), as gcnpt_layers_fwd / _bwd enqueue their launches */
int gcnpt_launch_empty_seq(void* stream, int n, const int* grid, const int* block, const int* lds_bytes, const int* kernarg_bytes);

/* ---- A1-A4: model/gcn.py:96-110 (lengths, head_to_tree x B, tree_to_adj x B, upload) ------------------
 * One workgroup per sentence prunes the dependency tree to the tokens within `prune_k` of the
 * subject<->object path (model/tree.py:80-162) and emits the CSR of the labelled adjacency that
 * tree_to_adj(directed=False, self_loop=True) would have produced (model/tree.py:167-204).
 *   head, subj_pos, obj_pos, deprel  [dev] int64 [B,T]   exactly the loader tensors (data/loader.py:111-121)
 *   pad_mask  [dev] uint8/bool [B,T], non-zero = pad (model/gcn.py:96); may be NULL if `len` is given
 *   len       [dev] int32 [B], used when pad_mask is NULL
 *   label, rowT_ptr + colT_idx + ellT may be NULL (not produced); ell is required
 *   pool_mask [dev] uint8 [B*T]: 1 where (row sum + column sum == 0), the mask GCN.forward returns (gcn.py:262)
 *   status    [dev] int32 [B+1]: status[b] = 0 or GCNPT_E_* for sentence b (its rows are then empty);
 *             status[B] = max sentence length seen (the reference needs it to equal T, gcn.py:97,269)
 */
int gcnpt_prune_to_csr(void* stream, const int64_t* head, const int64_t* subj_pos, const int64_t* obj_pos,
                       const int64_t* deprel, const uint8_t* pad_mask, const int32_t* len, int B, int T,
                       int prune_k, int cap, int32_t* row_ptr, int32_t* col_idx, int32_t* label,
                       int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell, int32_t* ellT, uint8_t* pool_mask,
                       int32_t* status);

/* ---- A5: model/gcn.py:260-262 on an explicit dense adjacency (GCN.forward(adj, inputs)) ----------------
 * adj [dev] float32 [B,T,T] -> CSR of (adj != 0) and of its transpose, pool_mask as above.  cap >= max nnz
 * per sentence (T*T always suffices); status[b] = GCNPT_E_CAPACITY when exceeded. */
int gcnpt_adj_to_csr(void* stream, const float* adj, int B, int T, int cap, int32_t* row_ptr, int32_t* col_idx,
                     int32_t* label, int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell, int32_t* ellT,
                     uint8_t* pool_mask, int32_t* status);

/* inverse of the above for callers that want the reference's dense float32 [B,T,T] adjacency
 * (model/gcn.py:106-108); adj is fully overwritten. */
int gcnpt_csr_to_adj(void* stream, const int32_t* row_ptr, const int32_t* col_idx, const int32_t* label, int B,
                     int T, float* adj);

/* ---- A8: nn.Linear parameters of GCN.W[l] (model/gcn.py:170-176) -> MFMA fragment order ---------------
 * W [dev] float32 [H,Din] row-major.  Produces the B-operand images the layer kernels stream:
 *   w_fwd  (for (A+I)h . W^T)   gcnpt_packed_bytes(H, Din, dtype) bytes
 *   w_bwd  (for ((A+I)^T dZ) . W) gcnpt_packed_bytes(Din, H, dtype) bytes       (either may be NULL)
 * Must be re-run whenever the optimizer changed W. */
size_t gcnpt_packed_bytes(int n_out, int k_in, int dtype);
int gcnpt_pack_weights(void* stream, const float* W, int H, int Din, int dtype, void* w_fwd, void* w_bwd);
/* the same for up to 8 layers in ONE launch (host arrays of n_layers device pointers / sizes) */
int gcnpt_pack_weights_multi(void* stream, int n_layers, const float* const* W, const int* H, const int* Din, int dtype,
                             void* const* w_fwd, void* const* w_bwd);

/* ---- saved operands in MFMA fragment order ------------------------------------------------------------
 * The weight gradient contracts over the ROW index, so its MFMA operands need 8 consecutive rows per lane.
 * Instead of transposing later, the forward kernel saves its gathered tile S = (A+I) h, and backward-data its
 * dZ tile, already in that order ("fragment image"):
 *   image[tile t][k-step ks][lane l] = 16 bytes
 *     bf16: 8 values X[32 ks + 8 (l>>4) + j][16 t + (l&15)], j = 0..7        (rows >= B*T and columns >= width are 0)
 *     f32 : 4 values X[16 ks + 4 (l>>4) + s][16 t + (l&15)], s = 0..3
 * gcnpt_frag_bytes(rows, width, dtype) is the size of one image. */
size_t gcnpt_frag_bytes(int rows, int width, int dtype);

/* ---- A6: one GCN layer, model/gcn.py:269-271 + 390-393 --------------------------------------------------
 *   out[r,:] = dropout( relu( ( (sum_{c in row r} h[c,:] + h[r,:]) . W^T + 2 b ) / (deg[r] + 1) ) )
 * h [dev] [B*T, Din] of h_dtype; out [dev] [B*T, H] of out_dtype; bias [dev] float32 [H].
 * compute_dtype = GCNPT_BF16 (bf16 MFMA operands, fp32 accumulate; h/out may be f32 or bf16) or
 *                 GCNPT_F32 (exact fp32 MFMA; h/out must be f32).
 * drop_p in [0,1): 0 disables dropout (eval mode / last layer, gcn.py:393); otherwise element e of the
 * output is kept iff hash(seed, e) >= drop_p and scaled by 1/(1-drop_p).
 * row_ptr/col_idx/ell describe the aggregated pattern.  deg_ell: NULL, or the ELL head whose degrees are used when
 * they differ from the aggregated pattern's (the `no_adj` ablation, gcn.py:264-265: denominators from the real
 * adjacency, aggregation over an empty one = an all-zero ell).
 * s_frag: NULL (inference), or gcnpt_frag_bytes(B*T, Din, compute_dtype) bytes that receive the fragment image
 * of S = (A+I) h for gcnpt_layer_bwd_weight.
 * seed_dev: NULL, or [dev] one uint64 that the kernel adds to `seed` when it runs: a counter the caller advances
 * between replays of a captured hipGraph, so that a replayed training step draws a new dropout mask (a by-value
 * seed is frozen into the graph). */
int gcnpt_layer_fwd(void* stream, const void* h, int h_dtype, const void* w_fwd, const float* bias,
                    const int32_t* row_ptr, const int32_t* col_idx, const int32_t* ell, const int32_t* deg_ell, int B,
                    int T, int Din, int H, void* out, int out_dtype, int compute_dtype, float drop_p, uint64_t seed,
                    void* s_frag, const uint64_t* seed_dev);

/* ---- A7: autograd of A6 -----------------------------------------------------------------------------------
 * With dZ[r,:] = dY[r,:] * 1[Y[r,:] > 0] * scale / (deg[r] + 1)   (Y = the layer's stored output, which
 * already carries the dropout zeros; scale = 1/(1-drop_p)):
 *   data   : dh[r,:]  = (sum_{c in rowT r} dZ[c,:] + dZ[r,:]) . W                  -> [B*T, Din] of dh_dtype
 *   weight : dW       = dZ^T ((A+I) h)   [H,Din] float32,   db = 2 * sum_r dZ[r,:]  [H] float32
 * gcnpt_layer_bwd_data: dY and Y share g_dtype; ell is the forward pattern's ELL head (degrees), rowT_ptr /
 * colT_idx / ellT the transposed pattern that is aggregated over.  dh may be NULL (input needs no gradient).  z_frag: NULL or gcnpt_frag_bytes(B*T, H, compute_dtype) bytes
 * receiving the fragment image of dZ; zero_dW [H*Din] / zero_db [H]: NULL or the accumulators the FOLLOWING
 * gcnpt_layer_bwd_weight adds into, cleared here so that no separate memset is needed.
 * Hand-over between stacked layers (optional; gcnpt_layers_bwd uses it): the gather of the layer below needs, per neighbour,
 * dY, Y and the neighbour's degree -- unless the layer above, which has its rows at hand, leaves dZ instead of dh:
 *   relu_src / next_scale  NULL, or this layer's INPUT rows [B*T, Din] in dh_dtype (= the stored output of the layer below)
 *                          and that layer's dropout scale: dh then receives dh * 1[relu_src > 0] * next_scale / (deg + 1),
 *                          i.e. dZ of the layer below;
 *   src_is_dz              non-zero: dY already IS this layer's dZ (written that way by the layer above); Y and scale are
 *                          not used.  Same values either way, fp32 bit for bit.
 * gcnpt_layer_bwd_weight: streams the two fragment images (z_frag from bwd_data, s_frag from fwd) and adds the
 * K-slices into dW/db with float atomics; dW/db must be zero on entry (see zero_dW/zero_db above). */
int gcnpt_layer_bwd_data(void* stream, const void* dY, const void* Y, int g_dtype, const void* w_bwd,
                         const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx, const int32_t* ellT,
                         int B, int T, int Din, int H, void* dh, int dh_dtype, int compute_dtype, float scale,
                         void* z_frag, float* zero_dW, float* zero_db, const void* relu_src, float next_scale, int src_is_dz);
int gcnpt_layer_bwd_weight(void* stream, const void* z_frag, const void* s_frag, int B, int T, int Din, int H,
                           float* dW, float* db, int compute_dtype);
/* The weight gradients of n_layers (<= 8) layers in ONE launch: they all become computable at the end of the backward
 * sweep, and each alone fills at most one workgroup per CU.  Arrays of n_layers entries (host memory), same meaning
 * per layer as above; every layer sees the same B x T rows. */
int gcnpt_layer_bwd_weight_multi(void* stream, int n_layers, const void* const* z_frag, const void* const* s_frag, int B,
                                 int T, const int* Din, const int* H, float* const* dW, float* const* db,
                                 int compute_dtype);

/* ---- the reference's whole layer loop (model/gcn.py:266-393) and its autograd in ONE host call each -------------------
 * The per-layer entry points above enqueue one launch per host call; a step of an L-layer stack is 2L+1 launches, and what
 * the host spends between them (interpreter, ctypes, hipGraph replay set-up: measured 5.3 us per replay) shows as idle
 * gaps on the device.  These two enqueue the same launches back to back from native code.  Arrays have n_layers (<= 8)
 * entries in host memory; every layer sees the same B x T rows and the same pattern.
 * gcnpt_layers_fwd:  out[l] = layer l applied to out[l-1] (x for l = 0), Din[l] must equal H[l-1]; out_dtype[l] is also the
 *   dtype layer l+1 reads; s_frag[l] may be NULL (or s_frag itself NULL) as in gcnpt_layer_fwd.
 * gcnpt_layers_bwd:  top layer first, gy = gradient of out[L-1] in y_dtype[L-1]; dh[l] = gradient of layer l's input in
 *   dh_dtype[l] (= y_dtype[l-1] for l > 0; dh[0] may be NULL when the input needs no gradient; dh[l] for l > 0 is scratch for
 *   the caller: it holds dZ of layer l-1, see the hand-over above); scale[l] = 1/(1-p_l) of the
 *   dropout layer l's forward applied.  z_frag == NULL: no weight gradients (dW, db, s_frag unused); otherwise z_frag[l],
 *   s_frag[l], dW[l], db[l] for every layer; the weight gradient of layer l+1 rides in layer l's backward-data launch (small batches,
 *   see gcnpt_layer_bwd_data_wgrad) and what is left follows in ONE launch (gcnpt_layer_bwd_weight_multi). */
int gcnpt_layers_fwd(void* stream, int n_layers, const void* x, int x_dtype, const void* const* w_fwd, const float* const* bias,
                     const int32_t* row_ptr, const int32_t* col_idx, const int32_t* ell, const int32_t* deg_ell, int B, int T,
                     const int* Din, const int* H, void* const* out, const int* out_dtype, int compute_dtype,
                     const float* drop_p, const uint64_t* seed, void* const* s_frag, const uint64_t* seed_dev);
int gcnpt_layers_bwd(void* stream, int n_layers, const void* gy, const void* const* Y, const int* y_dtype,
                     const void* const* w_bwd, const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                     const int32_t* ellT, int B, int T, const int* Din, const int* H, void* const* dh, const int* dh_dtype,
                     int compute_dtype, const float* scale, void* const* z_frag, const void* const* s_frag, float* const* dW,
                     float* const* db);

/* ---- a whole step of the stack from ONE host call ----------------------------------------------------------------------------------
 * gcnpt_pack_weights_multi + gcnpt_layers_fwd + gcnpt_layers_bwd (or _bwd_dz) enqueue a step's launches from three calls whose ~50
 * arguments a binding has to marshal every time; on a slow host that is what decides whether the device queue stays fed (bench.py:
 * launch_floor).  gcnpt_layers_step takes the same arguments ONCE, in a struct the caller fills at set-up and keeps: the per-step cost
 * is one call and one pointer.  Fields have exactly the meaning of the arguments of the three entry points above (arrays: entry l =
 * layer l, n_layers <= 8).  parts: bit 0 = pack the weights, bit 1 = forward sweep, bit 2 = backward sweep + weight gradients; the
 * parts run in that order.  gy_is_dz != 0: `gy` already is dZ of the top layer (gcnpt_layers_bwd_dz).  z_frag[0] == NULL: no weight
 * gradients.  The struct is read during the call only. */
typedef struct gcnpt_step {
    int n_layers, B, T, compute_dtype, parts, gy_is_dz;
    /* weights (gcnpt_pack_weights_multi) */
    const float* W[8]; const float* bias[8]; int Din[8], H[8];
    void* w_fwd[8]; void* w_bwd[8];
    /* pattern (gcnpt_prune_to_csr / gcnpt_pack_trees) */
    const int32_t* row_ptr; const int32_t* col_idx; const int32_t* ell; const int32_t* deg_ell;
    const int32_t* rowT_ptr; const int32_t* colT_idx; const int32_t* ellT; const int32_t* ell_bwd;   /* ell_bwd: the degrees' ELL head the backward uses (= the forward pattern's) */
    /* forward (gcnpt_layers_fwd) */
    const void* x; int x_dtype;
    void* out[8]; int out_dtype[8]; float drop_p[8]; uint64_t seed[8]; const uint64_t* seed_dev;
    void* s_frag[8];
    /* backward (gcnpt_layers_bwd): gradients of the layers' inputs, scales 1/(1-p), saved-operand images, accumulators */
    const void* gy; void* dh[8]; int dh_dtype[8]; float scale[8];
    void* z_frag[8]; float* dW[8]; float* db[8];
} gcnpt_step_t;
int gcnpt_layers_step(void* stream, const gcnpt_step_t* step);

/* ---- token-packed variable-length batches (north_star "packed") ----------------------------------------
 * The reference pads every batch to its longest sentence (data/loader.py:109-121, model/gcn.py:96-97,106): B*T token rows of
 * which only sum(len) are real.  The batch adjacency is block diagonal (model/tree.py:167-204), i.e. ONE sparse matrix over the
 * packed rows r = cu_seqlens[b] + i.  gcnpt_pack_trees rewrites the [B,T] arrays of gcnpt_prune_to_csr / gcnpt_gather_trees /
 * gcnpt_adj_to_csr into that form: same entries in the same order, columns = packed row numbers, offsets contiguous:
 *   cu_seqlens int32 [B+1]   first packed row of each sentence (cu_seqlens[B] = sum(len) = N)
 *   row_ptr / rowT_ptr int32 [N+1], col_idx / colT_idx / label int32 [nnz_cap], ell / ellT int32 [N*8], pool_mask uint8 [N],
 *   row_sent int32 [N] (sentence of each packed row).   len [dev] int32 [B]: tokens per sentence (clamped to T).
 *   n_rows = rows allocated (>= sum(len)), nnz_cap = entries allocated; status [dev] int32 [2]: [0] = 0 or GCNPT_E_CAPACITY when
 *   either is too small (nothing usable is written then), [1] = sum(len) seen.
 * EVERY layer entry point above takes the packed form with T = 0: then `B` is the number of packed rows N, the pattern's
 * columns are absolute row numbers and row_ptr has N+1 entries (gcnpt_layer_fwd, gcnpt_layer_bwd_data, gcnpt_layer_bwd_weight[_multi],
 * gcnpt_layers_fwd, gcnpt_layers_bwd; fragment images: gcnpt_frag_bytes(N, width, dtype)).  Rows keep their values: a packed row
 * is bit-identical to the same token's row of the padded batch.
 * gcnpt_pack_rows: src [B*T, W] -> dst [N, W] (dst[cu[b]+i] = src[b*T+i], i < len[b]); gcnpt_unpack_rows: the inverse, slots past a
 * sentence's end are zero-filled (the reference leaves relu(pad-row . W + 2b) there; every consumer masks those slots,
 * model/gcn.py:116-121).  dtype = element type of both (GCNPT_F32 / GCNPT_BF16). */
int gcnpt_pack_trees(void* stream, const int32_t* src_row_ptr, const int32_t* src_col_idx, const int32_t* src_label,
                     const int32_t* src_rowT_ptr, const int32_t* src_colT_idx, const int32_t* src_ell, const int32_t* src_ellT,
                     const uint8_t* src_pool_mask, const int32_t* len, int B, int T, int cap, int32_t* cu_seqlens, int32_t* row_ptr,
                     int32_t* col_idx, int32_t* label, int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell, int32_t* ellT,
                     uint8_t* pool_mask, int32_t* row_sent, int n_rows, int nnz_cap, int32_t* status);
/* gcnpt_prune_to_csr writing that layout ITSELF, in the same launch (and, with n_layers > 0, packing the weights as gcnpt_prune_to_csr_pack
 * does): exactly the arrays gcnpt_pack_trees(gcnpt_prune_to_csr(...)) gives, bit for bit -- one launch instead of two or three per batch
 * (model/tree.py:167-204 + data/loader.py:109-121).  A sentence's offsets are prefix sums over the sentences before it, computed inside
 * the launch: workgroups publish their counts and wait for those of the sentences before them, which they take in the order of an atomic
 * ticket (no assumption about dispatch order; see tree_kernels.hip, PackedOut).
 *   status      [dev] int32 [2]    as gcnpt_pack_trees' (0 / GCNPT_E_CAPACITY, sum(len)); no memset needed
 *   sent_status [dev] int32 [B+1]  per-sentence codes and the longest sentence, as gcnpt_prune_to_csr's `status`
 *   pool_mask_padded [dev] uint8 [B*T] or NULL: the mask in the padded layout as well (what GCN.forward returns, model/gcn.py:262,395)
 *   sync_ws     [dev] uint64 [B+2] workspace of the launch; must be ZERO before the first call and is left zero by every call (also under
 *               hipGraph replay): allocate it once per stream
 *   n_layers = 0: no weight pack (the trailing arguments are then unused). */
int gcnpt_prune_to_csr_packed(void* stream, const int64_t* head, const int64_t* subj_pos, const int64_t* obj_pos, const int64_t* deprel,
                              const uint8_t* pad_mask, const int32_t* len, int B, int T, int prune_k, int32_t* cu_seqlens, int32_t* row_ptr,
                              int32_t* col_idx, int32_t* label, int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell, int32_t* ellT,
                              uint8_t* pool_mask, int32_t* row_sent, int n_rows, int nnz_cap, int32_t* status, int32_t* sent_status,
                              uint8_t* pool_mask_padded, uint64_t* sync_ws, int n_layers, const float* const* W, const int* H, const int* Din,
                              int dtype, void* const* w_fwd, void* const* w_bwd);
/* gcnpt_gather_trees (N4: a batch from a dataset pruned once) writing the packed layout itself, likewise: what gcnpt_pack_trees makes of
 * gcnpt_gather_trees' arrays (cap = 3 T), bit for bit, in one launch; the offsets come from the cache's own lengths and row offsets. */
int gcnpt_gather_trees_packed(void* stream, const int32_t* src_row_ptr, const int32_t* src_col_idx, const int32_t* src_label,
                              const int32_t* src_rowT_ptr, const int32_t* src_colT_idx, const int32_t* src_ell, const int32_t* src_ellT,
                              const uint8_t* src_pool_mask, const int32_t* src_status, const int32_t* src_len, int S, int Ts, int cap_s,
                              const int64_t* idx, int B, int T, int32_t* cu_seqlens, int32_t* row_ptr, int32_t* col_idx, int32_t* label,
                              int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell, int32_t* ellT, uint8_t* pool_mask, int32_t* row_sent,
                              int n_rows, int nnz_cap, int32_t* status, int32_t* sent_status, uint8_t* pool_mask_padded, int n_layers,
                              const float* const* W, const int* H, const int* Din, int dtype, void* const* w_fwd, void* const* w_bwd);
int gcnpt_pack_rows(void* stream, const void* src, int dtype, const int32_t* cu_seqlens, int B, int T, int W, void* dst);
int gcnpt_unpack_rows(void* stream, const void* src, int dtype, const int32_t* cu_seqlens, int B, int T, int W, void* dst);

/* ---- the weight pack as a side job of the tree launch ---------------------------------------------------------------------
 * A training step needs the weights re-packed (gcnpt_pack_weights_multi) after every optimizer step and a batch's trees built
 * (gcnpt_prune_to_csr) or assembled (gcnpt_gather_trees); both launches leave most of the 256 CUs idle and neither depends on the
 * other.  These forms do both in ONE launch: extra workgroups pack while the others build the trees (same results bit for bit; the
 * trailing arguments are gcnpt_pack_weights_multi's).  Saves one launch boundary (~4 us) per step. */
int gcnpt_prune_to_csr_pack(void* stream, const int64_t* head, const int64_t* subj_pos, const int64_t* obj_pos, const int64_t* deprel,
                            const uint8_t* pad_mask, const int32_t* len, int B, int T, int prune_k, int cap, int32_t* row_ptr,
                            int32_t* col_idx, int32_t* label, int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell, int32_t* ellT,
                            uint8_t* pool_mask, int32_t* status, int n_layers, const float* const* W, const int* H, const int* Din,
                            int dtype, void* const* w_fwd, void* const* w_bwd);
int gcnpt_gather_trees_pack(void* stream, const int32_t* src_row_ptr, const int32_t* src_col_idx, const int32_t* src_label,
                            const int32_t* src_rowT_ptr, const int32_t* src_colT_idx, const int32_t* src_ell, const int32_t* src_ellT,
                            const uint8_t* src_pool_mask, const int32_t* src_status, const int32_t* src_len, int S, int Ts, int cap_s,
                            const int64_t* idx, int B, int T, int cap, int32_t* row_ptr, int32_t* col_idx, int32_t* label,
                            int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell, int32_t* ellT, uint8_t* pool_mask, int32_t* status,
                            int n_layers, const float* const* W, const int* H, const int* Din, int dtype, void* const* w_fwd,
                            void* const* w_bwd);

/* ---- N1: the consumer right after the path, model/gcn.py:116-121 + pool() 473-483 ---------------------------------
 * One pass over h [B*T,H] (h_dtype) produces out [B, 3H] float32 = [pool(h, pool_mask) | pool(h, subj_pos != 0) |
 * pool(h, obj_pos != 0)], the row the output MLP reads.  type: 0 = max (masked tokens count as -1e12; ties go to the
 * first token, as torch.max(dim) does; argmax int32 [B,3,H] is recorded for backward), 1 = avg, 2 = sum.
 * gcnpt_pool3_bwd writes dh [B*T,H] completely (masked tokens get 0, as masked_fill blocks their gradient). */
int gcnpt_pool3_fwd(void* stream, const void* h, int h_dtype, const uint8_t* pool_mask, const int64_t* subj_pos,
                    const int64_t* obj_pos, int B, int T, int H, int type, float* out, int32_t* argmax);
int gcnpt_pool3_bwd(void* stream, const float* g, const int32_t* argmax, const uint8_t* pool_mask, const int64_t* subj_pos,
                    const int64_t* obj_pos, int B, int T, int H, int type, void* dh, int dh_dtype);
/* Hand-over from the pooling to the layer stack (model/gcn.py:114-121: the pooled tensor IS the top GCN layer's output): instead of dh
 * the pooling's backward leaves dz = dh * 1[y > 0] * scale / (deg + 1), i.e. dZ of that layer (gcn.py:390-393 differentiated), with y
 * [dev] [B*T,H] = the layer's stored output (same dtype as dz), ell = the forward pattern's ELL head (degrees), scale = 1/(1-p) of the
 * dropout applied to y.  gcnpt_layers_bwd_dz then runs the backward sweep from it: same arguments and results as gcnpt_layers_bwd, but
 * its first tensor is that dZ, so the top layer gathers ONE row per neighbour (as the layers below it do) instead of dY, Y and a degree. */
int gcnpt_pool3_bwd_dz(void* stream, const float* g, const int32_t* argmax, const uint8_t* pool_mask, const int64_t* subj_pos,
                       const int64_t* obj_pos, int B, int T, int H, int type, const void* y, const int32_t* ell, float scale, void* dz,
                       int dtype);
int gcnpt_layers_bwd_dz(void* stream, int n_layers, const void* dz_top, const void* const* Y, const int* y_dtype,
                        const void* const* w_bwd, const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                        const int32_t* ellT, int B, int T, const int* Din, const int* H, void* const* dh, const int* dh_dtype,
                        int compute_dtype, const float* scale, void* const* z_frag, const void* const* s_frag, float* const* dW,
                        float* const* db);

/* gcnpt_layer_bwd_data for layer l that ALSO computes the weight gradient of the layer above it (up_*: that layer's two fragment images,
 * widths and accumulators, exactly gcnpt_layer_bwd_weight's arguments): for batches of up to GCNPT_OPT_SIDE_TILES row tiles (6 144 token
 * rows) the gradient rides in the same launch, on the CUs that have no row tile (one launch boundary less on the step's critical path: the
 * last launch of a backward sweep is then the bottom layer's weight gradient alone); otherwise it is launched right after.
 * gcnpt_layers_bwd / _bwd_dz do this for every layer but the top one.  (Round 3 measured putting BOTH gradients of a two-layer sweep into
 * the bottom layer's launch, four ways: slower every time, EXPERIMENTS.md.) */
int gcnpt_layer_bwd_data_wgrad(void* stream, const void* dY, const void* Y, int g_dtype, const void* w_bwd, const int32_t* ell,
                               const int32_t* rowT_ptr, const int32_t* colT_idx, const int32_t* ellT, int B, int T, int Din, int H,
                               void* dh, int dh_dtype, int compute_dtype, float scale, void* z_frag, float* zero_dW, float* zero_db,
                               const void* relu_src, float next_scale, int src_is_dz, const void* up_z_frag, const void* up_s_frag,
                               int up_Din, int up_H, float* up_dW, float* up_db);
/* Launches [first_launch, first_launch + n_launches) of the sweep gcnpt_layers_bwd (gy_is_dz = 0) / gcnpt_layers_bwd_dz (1) would
 * enqueue, in its order (measurement aid: bench.py times truncated steps to charge each launch its in-step duration). */
int gcnpt_layers_bwd_range(void* stream, int n_layers, const void* gy, const void* const* Y, const int* y_dtype,
                           const void* const* w_bwd, const int32_t* ell, const int32_t* rowT_ptr, const int32_t* colT_idx,
                           const int32_t* ellT, int B, int T, const int* Din, const int* H, void* const* dh, const int* dh_dtype,
                           int compute_dtype, const float* scale, void* const* z_frag, const void* const* s_frag, float* const* dW,
                           float* const* db, int gy_is_dz, int first_launch, int n_launches);

/* ---- N2: adj_type == 'diagonal_deprel', model/gcn.py:272-294 (+ 390-393) -------------------------------------------
 * No weight matrix: out[r] = dropout(relu((sum_{c: 0<adj[r,c]<42} E[deprel[c]]*h[c] + sum_{c: 42<adj[r,c]<84}
 * E[deprel[c]+42]*h[c] + E[84]*h[r]) / (deg[r]+1))), all products element-wise over the H columns.  h, out, Y, dY, dh
 * are [B*T,H] of `dtype`; E is the float32 [85,H] relation embedding table (model/gcn.py:56-57); deprel int64 [B*T];
 * row_ptr/col_idx/label and rowT_ptr/colT_idx as gcnpt_prune_to_csr / gcnpt_adj_to_csr write them (label = the adj value).
 * The preprocessor Linear in front of the layers (gcn.py:257) is a plain GEMM and stays with the host's BLAS.
 * gcnpt_diag_layer_bwd writes dh completely and ACCUMULATES into dE [85,H] float32 (the same table feeds every layer:
 * the caller clears it once per backward pass); scale = 1/(1-drop_p) of the forward call. */
int gcnpt_diag_layer_fwd(void* stream, const void* h, int dtype, const float* E, const int64_t* deprel,
                         const int32_t* row_ptr, const int32_t* col_idx, const int32_t* label, int B, int T, int H,
                         void* out, float drop_p, uint64_t seed, const uint64_t* seed_dev);
int gcnpt_diag_layer_bwd(void* stream, const void* dY, const void* Y, const void* h, int dtype, const float* E,
                         const int64_t* deprel, const int32_t* row_ptr, const int32_t* col_idx, const int32_t* label,
                         const int32_t* rowT_ptr, const int32_t* colT_idx, int B, int T, int H, void* dh, float* dE,
                         float scale);

/* ---- N4: loader-side pre-pruning (data/loader.py:81-141 builds batches from per-sentence features; trainer.py:52-73) ----
 * Pruning depends on the parse alone, so a dataset is pruned ONCE -- gcnpt_prune_to_csr over all S sentences padded to
 * the dataset's longest sentence Ts, capacity cap_s -- and the result stays in HBM.  gcnpt_gather_trees assembles the
 * arrays of a batch from it: cached sentence idx[b] becomes sentence b of a [B, T] batch (src_* = the cached arrays,
 * src_len int32 [S] = sentence lengths; the outputs have exactly the layout gcnpt_prune_to_csr writes for B, T, cap and
 * are bit-identical to pruning that batch directly).  label / rowT_ptr+colT_idx+ellT / pool_mask may be NULL.
 * status[b] = the cached sentence's own code, GCNPT_E_INVALID for an index outside [0,S), GCNPT_E_LENGTH when the
 * sentence has more than T tokens, GCNPT_E_CAPACITY when it has more than cap entries; status[B] = longest sentence. */
int gcnpt_gather_trees(void* stream, const int32_t* src_row_ptr, const int32_t* src_col_idx, const int32_t* src_label,
                       const int32_t* src_rowT_ptr, const int32_t* src_colT_idx, const int32_t* src_ell,
                       const int32_t* src_ellT, const uint8_t* src_pool_mask, const int32_t* src_status,
                       const int32_t* src_len, int S, int Ts, int cap_s, const int64_t* idx, int B, int T, int cap,
                       int32_t* row_ptr, int32_t* col_idx, int32_t* label, int32_t* rowT_ptr, int32_t* colT_idx,
                       int32_t* ell, int32_t* ellT, uint8_t* pool_mask, int32_t* status);

/* ---- N1, second half: "pooled-only" rows (model/gcn.py:116-121 pools over the tokens of the pruned tree only; a tree token's
 * row of every layer depends on tree tokens only, the adjacency has no entry outside the tree, model/tree.py:167-204) ----
 * Rewrites the arrays of a [B, T] batch (src_*, as gcnpt_prune_to_csr / gcnpt_gather_trees / gcnpt_adj_to_csr wrote them) for a
 * [B, Tc] batch that holds only the tokens with pool_mask == 0, renumbered 0..kept-1 in token order: same entries, same
 * order, columns in slot numbers; slots kept..Tc-1 are empty and carry pool_mask 1.  The layer kernels run unchanged on the
 * result and give, in slot j of sentence b, exactly the row they give for token tok[b*Tc+j] of the full batch.
 *   tok  [dev] int64 [B*Tc]  token position of each slot, -1 for an empty slot (the caller gathers its layer inputs with it)
 *   kept [dev] int32 [B]     tokens kept per sentence
 *   status[b] = the source's code, GCNPT_E_LENGTH when a sentence keeps more than Tc tokens, GCNPT_E_CAPACITY when it has
 *   more than cap_c entries (the sentence is then empty); status[B] = most tokens kept by a sentence of the batch.
 * label / rowT_ptr+colT_idx+ellT may be NULL. */
int gcnpt_compact_trees(void* stream, const int32_t* src_row_ptr, const int32_t* src_col_idx, const int32_t* src_label,
                        const int32_t* src_rowT_ptr, const int32_t* src_colT_idx, const int32_t* src_ell,
                        const int32_t* src_ellT, const uint8_t* src_pool_mask, const int32_t* src_status, int B, int T,
                        int cap, int Tc, int cap_c, int32_t* row_ptr, int32_t* col_idx, int32_t* label, int32_t* rowT_ptr,
                        int32_t* colT_idx, int32_t* ell, int32_t* ellT, uint8_t* pool_mask, int32_t* status, int64_t* tok,
                        int32_t* kept);

/* ---- N3: the relation-conditioned traversal of adj_type == 'full_deprel', model/gcn.py:400-415 (traverse_deprel) ------------
 *   y[m,:] = sum_d e[m,d] * (x[m,:] @ W3[d]),   W3 = Linear.weight.reshape(D, Tin, H) (gcn.py:301: a reinterpretation of the
 *   [D*H, Tin] weight's memory, no transpose), for M token rows (the caller compacts the tokens that sit in a pruned tree).
 * Every entry point takes the MFMA operand type `dtype`: GCNPT_BF16 (bf16 operands, fp32 accumulation) or GCNPT_F32 (exact fp32 MFMA,
 * v_mfma_f32_16x16x4_f32: the module's default precision, the mode the reference-recorded goldens are checked in).  k = 32 (bf16) or
 * 16 (f32) below is the fragments' k-step.
 * gcnpt_bilinear_pack: W [dev] float32, the Linear weight as it lies in memory -> w_img, gcnpt_bilinear_packed_bytes(D, Tin, H, dtype)
 * bytes of MFMA fragment order (once per optimizer step).
 * gcnpt_bilinear_fwd: x [dev] [M, k*ceil(Tin/k)] of dtype, zero padded, 16-byte aligned; e [dev] float32 [M, D] (relation vectors:
 * embeddings, or ones past deprel_max_depth); y_planes [dev] float32 [gcnpt_bilinear_planes(M,D,Tin,H,dtype)][M, H], written
 * completely: the relations are split into that many slices (so that ~256 workgroups exist) and each slice leaves its partial
 * sums in its own plane -- the result is the sum of the planes (plus the bias term e @ b3, gcn.py:413).  No float atomics.
 * A wave keeps the fragments of its tokens for <= 8 k-steps (Tin <= 256 in bf16) in registers; wider inputs are cut into runs of k-steps and
 * every run writes planes of its own (gcnpt_bilinear_planes counts them).  gcnpt_bilinear_supported: 0 only for absurd widths (> 64 k-steps). */
size_t gcnpt_bilinear_packed_bytes(int D, int Tin, int H, int dtype);
int gcnpt_bilinear_supported(int D, int Tin, int H, int dtype);
int gcnpt_bilinear_pack(void* stream, const float* W, int D, int Tin, int H, void* w_img, int transposed, int dtype);
int gcnpt_bilinear_planes(int M, int D, int Tin, int H, int dtype);
int gcnpt_bilinear_fwd(void* stream, const void* x, const float* e, const void* w_img, int M, int D, int Tin, int H,
                       float* y_planes, int dtype);
/* Gradients of the traversal that reuse the same kernel (the op is linear in each argument):
 *   dx = sum_d e_d * (gy @ W3[d]^T)        -> gcnpt_bilinear_fwd on (gy as x, the image packed with transposed = 1, widths swapped:
 *                                             gcnpt_bilinear_pack(.., transposed=1, dtype) fills gcnpt_bilinear_packed_bytes(D, H, Tin, dtype) bytes;
 *                                             gcnpt_bilinear_fwd(stream, gy [M, k*ceil(H/k)] of dtype, e, imgT, M, D, H, Tin, dx_planes, dtype))
 *   de[m,d] = (x[m] @ W3[d]) . gy[m]       -> gcnpt_bilinear_bwd_e: gy [dev] float32 [M,H]; de_planes [dev] float32
 *                                             [gcnpt_bilinear_de_planes(M,D,Tin,H,dtype)][M, D] written completely, summed by the caller. */
/*   dW3[d][t][h] = sum_m e[m,d] x[m,t] gy[m,h] -> gcnpt_bilinear_bwd_w: x_img / gy_img = gcnpt_rows_pack of x [M,Tin] / gy [M,H]
 *                                             (float32 in, fragment images of dtype out, gcnpt_rows_image_bytes(M, width, dtype) bytes:
 *                                             lane = column, k/4 consecutive rows per lane); eT [dev] float32 [D, k*ceil(M/k)] = e
 *                                             transposed, zero padded; dW [dev] float32 in the Linear weight's own layout
 *                                             ([D*H, Tin] memory read as [D,Tin,H]), written completely, no atomics. */
size_t gcnpt_rows_image_bytes(int M, int W, int dtype);
int gcnpt_rows_pack(void* stream, const float* src, int M, int W, void* img, int dtype);
int gcnpt_bilinear_bwd_w(void* stream, const void* x_img, const void* gy_img, const float* eT, int M, int D, int Tin, int H,
                         float* dW, int dtype);
int gcnpt_bilinear_de_planes(int M, int D, int Tin, int H, int dtype);
int gcnpt_bilinear_bwd_e(void* stream, const void* x, const float* gy, const void* w_img, int M, int D, int Tin, int H,
                         float* de_planes, int dtype);

/* ---- N3, the layer AROUND the traversal: model/gcn.py:308-311 + 331, 340-344 + 362 (aggregation of the traversed encodings over the
 * forward / reverse edges, picked by value ranges of the labelled adjacency), 366-385 (self loop), 390-393 (normalise, ReLU, dropout) ----
 *   out[r] = dropout(relu((sum_{k in row r, 0 < label_k < 42} keep_f[k] * yf[pos[col_k]] + sum_{k, 42 < label_k < 84} keep_r[k] * yr[pos[col_k]]
 *                          + self_term[r]) / (deg[r] + 1)))
 * yf / yr [dev] float32 [M,H]: the traversal of the M tokens that sit in a pruned tree (gcnpt_bilinear_fwd + bias term, or the host's GEMM);
 * yr NULL = deprel_directed; pos [dev] int32 [B*T]: token -> its row of yf / yr; self_term [dev] float32 [B*T,H] or NULL; keep_f / keep_r
 * [dev] uint8, one flag per CSR slot, or NULL (training-time edge dropout, gcn.py:436-449); row_ptr / col_idx / label as the pruner writes
 * them.  gcnpt_full_agg_bwd: dagg [B*T,H] = dY * 1[y > 0] * scale / (deg + 1) (= the gradient of self_term) is written completely and
 * ACCUMULATED into dyf / dyr [M,H] along the same entries (float atomics; the caller clears them). */
int gcnpt_full_agg_fwd(void* stream, const float* yf, const float* yr, const float* self_term, const int32_t* pos, const int32_t* row_ptr,
                       const int32_t* col_idx, const int32_t* label, const uint8_t* keep_f, const uint8_t* keep_r, int B, int T, int H, int M,
                       float* out, float drop_p, uint64_t seed, const uint64_t* seed_dev);
int gcnpt_full_agg_bwd(void* stream, const float* dy, const float* y, const int32_t* pos, const int32_t* row_ptr, const int32_t* col_idx,
                       const int32_t* label, const uint8_t* keep_f, const uint8_t* keep_r, int B, int T, int H, int M, float scale,
                       float* dagg, float* dyf, float* dyr);

#ifdef __cplusplus
}
#endif
#endif /* GCNPT_H */
