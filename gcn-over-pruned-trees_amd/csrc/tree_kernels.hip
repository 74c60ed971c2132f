// Pruned-dependency-tree adjacency on the device (gfx950).
//
// Replaces the host round trip of reference model/gcn.py:96-110 (6 device->host copies, a Python loop
// over model/tree.py:58 head_to_tree and model/tree.py:167 tree_to_adj, one host->device copy of a
// dense [B,T,T] float tensor) with one launch: a workgroup per sentence stages the head array in LDS,
// prunes the tree there and writes the CSR pattern (3n-2 entries for an n-node tree) the layer kernels
// consume.  Integer work only; results are exact.
//
// The pruning uses the closed form of SURVEY.md 3c (fuzz-verified against the reference):
//   chain(t) = t and its ancestors;  CA = intersection of chain(t) over entity tokens;
//   lca = the member of CA with no child in CA;  P = (union of chains - CA) + {lca};
//   dist[i] = edges from i up to its first ancestor-or-self in P (infinite if none);  keep = dist <= K;
//   edges = {(head[i]-1 -> i) : keep[i], i != lca, head[i] > 0}.
#include "gcnpt_common.h"
#include "pack_common.h"

namespace gcnpt {

constexpr int PRUNE_THREADS = 1024;  // wave 0 prunes the sentence (wave-local phases); all 16 waves then emit the rows (one each, typically)
constexpr int PRUNE_SCAN_MAX = 1 << 16;   // B*T up to which workgroup 0 finds the longest sentence itself (no memset, no atomics)
constexpr int ADJ_THREADS = 256;

enum : int {
    F_SUBJ = 1, F_OBJ = 2, F_FWD_NZ = 4, F_REV_NZ = 8, F_CA = 16, F_CA_HASCHILD = 32, F_PATH = 64, F_HASEDGE = 128
};
enum : int { K_KEEP = 1, K_CHILD = 2 };
enum : int { ERR_CHAIN_BADHEAD = 1, ERR_CHAIN_CYCLE = 2, ERR_BADHEAD = 4, ERR_CYCLE = 8, ERR_ASSERT = 16 };

// Token-packed output of the pruner (gcnpt_prune_to_csr_packed; layout of gcnpt_pack_trees in include/gcnpt.h): row cu[b] + i, columns
// shifted by cu[b], entry offsets contiguous over the batch.  A sentence's offsets are prefix sums of (len, nnz, nnzT) over the sentences
// before it, which other workgroups of the SAME launch compute: every workgroup publishes its three counts in ONE 8-byte word (valid bit |
// len << 40 | nnz << 20 | nnzT, an agent-scope atomic store: the word is its own flag, nothing to order), then sums the words of the
// sentences before it (agent-scope atomic loads, spinning on the valid bit).  Sentences are taken in TICKET order (an atomic counter), not
// in blockIdx order, so a workgroup only ever waits for workgroups that already run: no assumption about dispatch order.  The last
// workgroup to finish polling clears the words and both counters: the workspace is all-zero between launches (graph replay included).
struct PackedOut {
    int32_t* cu;                 // NULL = the padded layout (everything below unused)
    int32_t* row_sent;
    uint8_t* pool_mask_padded;   // NULL or [B*T]: the mask in the padded layout as well (what GCN.forward returns, gcn.py:262,395)
    unsigned long long* sync;    // [B + 2], zero between launches: the sentences' words, ticket counter, done counter
    int32_t* pk_status;          // [2]: 0 / GCNPT_E_CAPACITY, sum(len)
    int n_rows, nnz_cap;
};
constexpr unsigned long long PK_VALID = 1ull << 63;

// exclusive scan of a[0..n) in place, a[n] = total; `scratch` holds blockDim.x ints (any block size)
__device__ void block_exclusive_scan(int* a, int n, int* scratch) {
    const int t = threadIdx.x, nt = blockDim.x;
    const int seg = (n + nt - 1) / nt;
    const int lo = min(n, t * seg), hi = min(n, lo + seg);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    scratch[t] = s;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int k = 0; k < nt; ++k) { int v = scratch[k]; scratch[k] = run; run += v; }
        a[n] = run;
    }
    __syncthreads();
    int run = scratch[t];
    for (int i = lo; i < hi; ++i) { int v = a[i]; a[i] = run; run += v; }
    __syncthreads();
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
// exclusive scan of a[0..n) in place by ONE wave (lane l owns the contiguous segment l); returns the total
__device__ __forceinline__ int wave_exclusive_scan(int* a, int n, int lane) {
    const int seg = (n + WAVE - 1) / WAVE;
    const int lo = min(n, lane * seg), hi = min(n, lo + seg);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    int inc = s;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    const int total = __shfl(inc, WAVE - 1);
    int run = inc - s;
    for (int i = lo; i < hi; ++i) { const int v = a[i]; a[i] = run; run += v; }
    return total;
}

// the same for three arrays at once: three independent shuffle chains share the latency of one
__device__ __forceinline__ void wave_exclusive_scan3(int* a, int* b, int* c, int n, int lane, int& ta, int& tb, int& tc) {
    const int seg = (n + WAVE - 1) / WAVE;
    const int lo = min(n, lane * seg), hi = min(n, lo + seg);
    int sa = 0, sb = 0, sc = 0;
    for (int i = lo; i < hi; ++i) { sa += a[i]; sb += b[i]; sc += c[i]; }
    int ia = sa, ib = sb, ic = sc;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const int ua = __shfl_up(ia, o), ub = __shfl_up(ib, o), uc = __shfl_up(ic, o);
        if (lane >= o) { ia += ua; ib += ub; ic += uc; }
    }
    ta = __shfl(ia, WAVE - 1); tb = __shfl(ib, WAVE - 1); tc = __shfl(ic, WAVE - 1);
    int ra = ia - sa, rb = ib - sb, rc = ic - sc;
    for (int i = lo; i < hi; ++i) {
        const int va = a[i], vb = b[i], vc = c[i];
        a[i] = ra; b[i] = rb; c[i] = rc;
        ra += va; rb += vb; rc += vc;
    }
}

// One 32-bit word per token keeps its parent and its flags together, so that every step of a walk up the tree is
// ONE LDS read:  bits 0..11 = parent + 2 (0 = head points past the sentence, 1 = root / none), bits 12.. = F_* flags.
constexpr int PW_SHIFT = 12, PW_MASK = (1 << PW_SHIFT) - 1, PRUNE_MAX_T = PW_MASK - 3;
__device__ __forceinline__ int pw_par(int w) { return (w & PW_MASK) - 2; }

// Sentences of up to PRUNE_WAVE0_T tokens: wave 0 runs the pruning phases alone, separated by wave_lds_fence() (gcnpt_common.h): no
// workgroup barrier.  Longer ones (ALLW): every phase is a loop over the tokens by ALL 1024 threads with a
// workgroup barrier behind it, so a 300-token sentence is one round per phase instead of five (round 2: 30 us at T = 300).
#ifndef GCNPT_PRUNE_WAVE0_T
#define GCNPT_PRUNE_WAVE0_T 64       // measured (tools/ab_libs.sh): T = 100 takes 8.4 us with wave 0 alone, 7.3 us with all waves; T = 50: 4.1 against 5.4
#endif
constexpr int PRUNE_WAVE0_T = GCNPT_PRUNE_WAVE0_T;

// Phases: lengths, LCA, path, distances, kept tokens, degrees, row offsets, compacted edge rows.
// s_red: [0] pad slots, [1] subject tokens, [2] entity tokens, [3] lca (workgroup-wide sums / minimum of the ALLW form)
template <bool ALLW>
__device__ void prune_sentence(const int64_t* __restrict__ head, const int64_t* __restrict__ subj_pos,
                               const int64_t* __restrict__ obj_pos, const int64_t* __restrict__ deprel,
                               const uint8_t* __restrict__ pad_mask, const int32_t* __restrict__ len_in, int b, int B, int T,
                               int prune_k, int cap, int* smem, int* s_err, int* s_status, int* s_nrows, int* s_red,
                               int32_t* __restrict__ row_ptr, int32_t* __restrict__ rowT_ptr, int32_t* __restrict__ ell,
                               int32_t* __restrict__ ellT, uint8_t* __restrict__ pool_mask, int32_t* __restrict__ status,
                               unsigned long long* stamps, const PackedOut& po, int* s_out) {
    int* pw = smem;                // [T]   parent + flags (see above)
    int* cnt = pw + T;             // [T]   #entity chains through the token; later K_KEEP/K_CHILD bits
    int* deg = cnt + T;            // [T+1] row degree -> row offsets
    int* degT = deg + T + 1;       // [T+1] column degree -> transposed row offsets
    int* rank = degT + T + 1;      // [T+1] rank of the token among the rows that carry an edge
    int* lab = rank + T + 1;       // [T]   deprel id of the token
    int* einfo = lab + T;          // [T]   compacted edge rows: token | (parent+2) << 12 | child/label bits << 24
    const int lane = threadIdx.x & 63;
    const int t0 = ALLW ? (int)threadIdx.x : lane;            // this thread's first token
    constexpr int NT = ALLW ? PRUNE_THREADS : WAVE;           // token stride
    const size_t base = (size_t)b * T;
    auto phase_sync = [&]() { if constexpr (ALLW) __syncthreads(); else wave_lds_fence(); };
    // sum over the participating threads (ALLW: through s_red[slot], which the kernel cleared; the caller syncs before reading)
    auto team_sum = [&](int v, int slot) {
        v = wave_sum(v);
        if constexpr (ALLW) { if (lane == 0 && v) atomicAdd(&s_red[slot], v); }
        return v;
    };

    // ---- stage the parse (tree.py:60-63, 82-83).  Wave-0 form: two tokens per lane per round, every load of the round issued before
    // the first use (clamped addresses, no load behind a condition): a sentence of up to 128 tokens costs ONE memory round trip.
    // ALLW: one token per thread, one round for up to 1024 tokens.
    int npad = 0;
    constexpr int PER = ALLW ? 1 : 2;
    for (int i0 = 0; i0 < T; i0 += PER * NT) {
        int64_t h[PER], sp[PER], op[PER], d[PER];
        bool pad[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const size_t j = base + min(i0 + u * NT + t0, T - 1);
            h[u] = head[j]; sp[u] = subj_pos[j]; op[u] = obj_pos[j]; d[u] = deprel[j];
            pad[u] = pad_mask ? pad_mask[j] != 0 : false;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = i0 + u * NT + t0;
            if (i >= T) continue;
            npad += pad[u] ? 1 : 0;
            int f = 0;
            if (sp[u] == 0) f |= F_SUBJ;
            if (op[u] == 0) f |= F_OBJ;
            if (d[u] != 0) f |= F_FWD_NZ;                  // adj[p,c] = deprel[c]        survives `adj != 0`
            if (d[u] + FWD_BOUND != 0) f |= F_REV_NZ;      // adj[c,p] = deprel[c] + 42
            const int p = h[u] > 0 ? (int)min(h[u] - 1, (int64_t)(PW_MASK - 2)) : -1;      // range-checked against len below
            pw[i] = (p + 2) | (f << PW_SHIFT);
            lab[i] = (int)d[u]; cnt[i] = 0; deg[i] = 0; degT[i] = 0;
        }
    }
    // sentence length = number of non-pad slots (gcn.py:96)
    int len;
    if (pad_mask) {
        const int np = team_sum(npad, 0);
        if constexpr (ALLW) { __syncthreads(); len = T - s_red[0]; } else { len = T - np; }
    } else {
        len = min(max(len_in[b], 0), T);
    }
    // (the longest sentence: workgroup 0's idle waves scan for it, see the kernel; batches too big for that / sentences that leave no
    // idle wave: one atomicMax per sentence on a word prune_impl cleared)
    if (threadIdx.x == 0 && ((long long)B * T > PRUNE_SCAN_MAX || (ALLW && T + WAVE > PRUNE_THREADS))) atomicMax(&status[B], len);
    int nsubj = 0, nent = 0;
    for (int i = t0; i < T; i += NT) {
        const int w = pw[i];
        int f = w >> PW_SHIFT, p = pw_par(w);
        if (i >= len) { f = 0; p = -1; }
        else if (p >= len) p = -2;
        nsubj += (f & F_SUBJ) ? 1 : 0;
        nent += ((f & F_SUBJ) ? 1 : 0) + ((f & F_OBJ) ? 1 : 0);
        pw[i] = (p + 2) | (f << PW_SHIFT);
    }
    nsubj = team_sum(nsubj, 1);
    nent = team_sum(nent, 2);
    phase_sync();
    if constexpr (ALLW) { nsubj = s_red[1]; nent = s_red[2]; }
    GCNPT_STAMP(stamps, 1);

    // ---- every entity token walks to the root, counting visits (tree.py:86-109)
    for (int i = t0; i < len; i += NT) {
        const int f = pw[i] >> PW_SHIFT;
        const int w = ((f & F_SUBJ) ? 1 : 0) + ((f & F_OBJ) ? 1 : 0);
        if (!w) continue;
        int a = i, steps = 0;
        while (a >= 0) {
            atomicAdd(&cnt[a], w);                     // result unused: no-return LDS add, nothing to wait for
            a = pw_par(pw[a]);
            if (++steps > len) { atomicOr(s_err, ERR_CHAIN_CYCLE); a = -1; }
        }
        if (a == -2) atomicOr(s_err, ERR_CHAIN_BADHEAD);
    }
    phase_sync();
    int err = 0;
    if (*s_err & ERR_CHAIN_BADHEAD) err = GCNPT_E_BAD_HEAD;
    else if (*s_err & ERR_CHAIN_CYCLE) err = GCNPT_E_CYCLE;
    else if (nsubj == 0) err = GCNPT_E_NO_SUBJECT;
    GCNPT_STAMP(stamps, 2);

    // ---- common ancestors and the lowest of them (tree.py:112-124)
    int lca = 0x7fffffff;
    if (!err) {                                          // (err is the same in every thread: the syncs below are uniform)
        for (int i = t0; i < len; i += NT)
            if (cnt[i] == nent) pw[i] |= F_CA << PW_SHIFT;
        phase_sync();
        for (int i = t0; i < len; i += NT) {
            const int w = pw[i], p = pw_par(w);
            if ((w & (F_CA << PW_SHIFT)) && p >= 0 && (pw[p] & (F_CA << PW_SHIFT))) atomicOr(&pw[p], F_CA_HASCHILD << PW_SHIFT);
        }
        phase_sync();
        for (int i = t0; i < len; i += NT)
            if (((pw[i] >> PW_SHIFT) & (F_CA | F_CA_HASCHILD)) == F_CA) lca = min(lca, i);
        lca = wave_min(lca);
        if constexpr (ALLW) {
            if (lane == 0 && lca != 0x7fffffff) atomicMin(&s_red[3], lca);
            __syncthreads();
            lca = s_red[3];
        }
        if (lca == 0x7fffffff) err = GCNPT_E_NO_LCA;
    }
    GCNPT_STAMP(stamps, 3);

    // ---- path nodes, distance to the path, kept tokens (tree.py:126-147)
    if (!err) {
        for (int i = t0; i < len; i += NT)
            if ((cnt[i] > 0 && !(pw[i] & (F_CA << PW_SHIFT))) || i == lca) pw[i] |= F_PATH << PW_SHIFT;
        phase_sync();
        for (int i = t0; i < len; i += NT) {
            int a = i, d = 0, w = pw[i];
            while (a >= 0 && !(w & (F_PATH << PW_SHIFT))) {          // one LDS read per step
                a = pw_par(w);
                if (a >= 0) w = pw[a];
                if (++d > len) { atomicOr(s_err, ERR_CYCLE); a = -1; }
            }
            if (a == -2) atomicOr(s_err, ERR_BADHEAD);
            const bool keep = a >= 0 && d <= prune_k;
            const bool child = keep && i != lca && pw_par(pw[i]) >= 0;
            cnt[i] = (keep ? K_KEEP : 0) | (child ? K_CHILD : 0);   // cnt is free from here on
        }
        phase_sync();
        if (*s_err & ERR_BADHEAD) err = GCNPT_E_BAD_HEAD;
        else if (*s_err & ERR_CYCLE) err = GCNPT_E_CYCLE;
    }
    GCNPT_STAMP(stamps, 4);

    // ---- degrees of the labelled adjacency tree_to_adj would write (tree.py:182-192)
    if (!err) {
        for (int i = t0; i < len; i += NT) {
            if (!(cnt[i] & K_CHILD)) continue;
            const int w = pw[i], p = pw_par(w), f = w >> PW_SHIFT;
            if (!(cnt[p] & K_KEEP)) atomicOr(s_err, ERR_ASSERT);   // tree.py:159
            if (f & F_FWD_NZ) { atomicAdd(&deg[p], 1); atomicAdd(&degT[i], 1); }
            if (f & F_REV_NZ) { atomicAdd(&deg[i], 1); atomicAdd(&degT[p], 1); }
            atomicOr(&pw[p], F_HASEDGE << PW_SHIFT);
            atomicOr(&pw[i], F_HASEDGE << PW_SHIFT);
        }
        phase_sync();
        if (*s_err & ERR_ASSERT) err = GCNPT_E_ASSERT;
    }
    int n_edge_rows = 0;
    if (!err) {
        for (int i = t0; i < T; i += NT) {
            const int he = (i < len && (pw[i] & (F_HASEDGE << PW_SHIFT))) ? 1 : 0;
            deg[i] += he; degT[i] += he;                             // the 84 on the diagonal
            rank[i] = he;
            uint8_t* pm = po.cu ? po.pool_mask_padded : pool_mask;
            if (pm) pm[base + i] = (deg[i] + degT[i]) == 0;                      // gcn.py:262
        }
        phase_sync();
        if (!ALLW || threadIdx.x < WAVE) {                           // the three scans: one wave (lane l owns a contiguous segment)
            int tot, totT;
            wave_exclusive_scan3(deg, degT, rank, T, lane, tot, totT, n_edge_rows);
            if (lane == 0) { deg[T] = tot; degT[T] = totT; rank[T] = n_edge_rows; }
        }
        phase_sync();
        n_edge_rows = rank[T];
        if (deg[T] > cap || degT[T] > cap) err = GCNPT_E_CAPACITY;
    }
    GCNPT_STAMP(stamps, 5);

    // ---- where this sentence's rows and entries go.  Padded layout: slots of its own (rows b*T.., entries b*cap..).  Packed: behind
    //      the sentences before it (see PackedOut): the counts are published HERE, the offsets are taken as late as possible -- in
    //      emit_rows(), after the rows have been sorted in registers --, so that waiting for a slower sentence costs nothing
    const size_t rbase = base;              // first row
    const int ebase = b * cap, eTbase = b * cap;   // first entry of the two patterns
    if (po.cu) {
        const int my_nnz = err ? 0 : deg[T], my_nnzT = err ? 0 : degT[T];
        if (threadIdx.x == 0) {
            __hip_atomic_store(po.sync + b, PK_VALID | ((unsigned long long)len << 40) | ((unsigned long long)my_nnz << 20) | (unsigned long long)my_nnzT,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_out[0] = len; s_out[1] = my_nnz; s_out[2] = my_nnzT;
            status[b] = err; *s_status = err; *s_nrows = err ? 0 : n_edge_rows;
        }
        if (err) {
            if (po.pool_mask_padded)
                for (int i = t0; i < T; i += NT) po.pool_mask_padded[base + i] = 1;
            return;
        }
        for (int i = t0; i < len; i += NT) {                         // the compacted edge rows (see below); every global write waits
            const int w = pw[i];
            if (w & (F_HASEDGE << PW_SHIFT)) {
                const int f = w >> PW_SHIFT;
                einfo[rank[i]] = i | ((w & PW_MASK) << 12) | ((cnt[i] & K_CHILD) ? 1 << 24 : 0) |
                                 ((f & F_FWD_NZ) ? 1 << 25 : 0) | ((f & F_REV_NZ) ? 1 << 26 : 0);
            }
        }
        if constexpr (!ALLW) wave_lds_fence();
        GCNPT_STAMP(stamps, 6);
        return;
    }
    // row_ptr index of this sentence's row 0 (the padded layout keeps T + 1 offsets per sentence)
    const size_t rp0 = (size_t)b * (T + 1);
    const int n_off = T + 1;
    if (threadIdx.x == 0) { s_out[0] = (int)rbase; s_out[1] = ebase; s_out[2] = eTbase; s_out[3] = 0; }

    if (err) {   // the sentence contributes no edges; every row is empty and masked
        for (int i = t0; i < n_off; i += NT) {
            row_ptr[rp0 + i] = ebase;
            if (rowT_ptr) rowT_ptr[rp0 + i] = eTbase;
        }
        for (int i = t0; i < T; i += NT)
            if (pool_mask) pool_mask[base + i] = 1;
        for (int i = t0; i < T * 8; i += NT) {
            ell[rbase * 8 + i] = 0;
            if (ellT) ellT[rbase * 8 + i] = 0;
        }
        if (threadIdx.x == 0) { status[b] = err; *s_status = err; }
        return;
    }

    // ---- emit both patterns, columns ascending (same order a dense -> CSR conversion gives).
    // Only rows that carry an edge have entries, and only such rows appear as columns: compact them (ascending)
    // with everything the inner loop needs in ONE word, so that loop is a stream of broadcast LDS reads.
    for (int i = t0; i < n_off; i += NT) {
        row_ptr[rp0 + i] = ebase + deg[i];
        if (rowT_ptr) rowT_ptr[rp0 + i] = eTbase + degT[i];
    }
    for (int i = t0; i < T; i += NT) {
        const int w = i < len ? pw[i] : 0;
        if (w & (F_HASEDGE << PW_SHIFT)) {
            const int f = w >> PW_SHIFT;
            einfo[rank[i]] = i | ((w & PW_MASK) << 12) | ((cnt[i] & K_CHILD) ? 1 << 24 : 0) |
                             ((f & F_FWD_NZ) ? 1 << 25 : 0) | ((f & F_REV_NZ) ? 1 << 26 : 0);
        } else {                                                     // no entries: an all-zero ELL head
            int4* e = reinterpret_cast<int4*>(ell + (rbase + i) * 8);
            e[0] = make_int4(0, 0, 0, 0); e[1] = make_int4(0, 0, 0, 0);
            if (ellT) {
                int4* eT = reinterpret_cast<int4*>(ellT + (rbase + i) * 8);
                eT[0] = make_int4(0, 0, 0, 0); eT[1] = make_int4(0, 0, 0, 0);
            }
        }
    }
    if constexpr (!ALLW) wave_lds_fence();                          // (ALLW: the kernel's barrier in front of emit_rows)
    GCNPT_STAMP(stamps, 6);
    if (threadIdx.x == 0) { *s_nrows = n_edge_rows; status[b] = 0; }
}

// One row of both patterns by ONE WAVE: one candidate column per lane (the edge rows, ascending), ballot + prefix popcount give every
// entry its slot -- no sorting, any number of entries.  rw / rdeg / rdegT / rlab: the chunk's row info in the lanes' registers (row qq of
// the chunk is read with v_readlane).
__device__ __forceinline__ void emit_row_scan(size_t base, int ebase, int eTbase, int cshift, const int* lab, const int* einfo, int n_edge_rows, int lane, int me, int labr_raw,
                                              int o_rel, int oT_rel, int32_t* __restrict__ col_idx, int32_t* __restrict__ label,
                                              int32_t* __restrict__ colT_idx, int32_t* __restrict__ ell, int32_t* __restrict__ ellT) {
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int r = me & 0xfff, rp = ((me >> 12) & 0xfff) - 2;
    const bool rchild = me & (1 << 24), rfwd = me & (1 << 25), rrev = me & (1 << 26);
    const int labr = labr_raw + FWD_BOUND;
    const int o = ebase + o_rel, oT = eTbase + oT_rel;
    int32_t* hd = ell + (base + r) * 8;                       // ELL head: [0] = count, [1..7] = first 7 columns
    int32_t* hdT = ellT ? ellT + (base + r) * 8 : nullptr;
    int n_e = 0, n_eT = 0;
    for (int k0 = 0; k0 < n_edge_rows; k0 += WAVE) {          // candidate columns: the edge rows, ascending
        const bool valid = k0 + lane < n_edge_rows;
        const int w = valid ? einfo[k0 + lane] : 0;
        const int j = w & 0xfff, jp = ((w >> 12) & 0xfff) - 2;
        const bool is_child = valid && (w & (1 << 24)) && jp == r;      // j is a kept child of r
        const bool is_self = valid && j == r;
        const bool is_par = valid && rchild && j == rp;
        const bool e = (is_child && (w & (1 << 25))) || is_self || (is_par && rrev);      // adj[r,j] != 0
        const bool eT = (is_child && (w & (1 << 26))) || is_self || (is_par && rfwd);     // adj[j,r] != 0
        const unsigned long long m = __ballot(e), mT = __ballot(eT);
        if (e) {
            const int pos = n_e + __popcll(m & lt);
            col_idx[o + pos] = j + cshift;
            if (label) label[o + pos] = is_self ? SELF_LOOP_ID : (is_child ? lab[j] : labr);
            if (pos < 7) hd[1 + pos] = j + cshift;
        }
        if (eT) {
            const int pos = n_eT + __popcll(mT & lt);
            if (colT_idx) colT_idx[oT + pos] = j + cshift;
            if (hdT && pos < 7) hdT[1 + pos] = j + cshift;
        }
        n_e += __popcll(m);
        n_eT += __popcll(mT);
    }
    if (lane < 8) {                                           // count, and zeros in the unused slots
        if (lane == 0) hd[0] = n_e; else if (lane > n_e) hd[lane] = 0;
        if (hdT) { if (lane == 0) hdT[0] = n_eT; else if (lane > n_eT) hdT[lane] = 0; }
    }
}

// All 16 waves: the CSR entries and ELL heads of the rows that carry an edge.
// STAGED (sentences whose entry lists fit LDS beside the pruning arrays): every edge row is a THREAD.  It appends its own entries
// (diagonal; for a kept child also the pair with its parent) to the rows' LDS segments through per-row fill counters, and after one
// barrier sorts its row's segment by column (an insertion sort: a row of a pruned tree has 2-4 entries) and writes it out with its
// ELL head.  Rows with more than EMIT_SORT_MAX entries (a star-shaped parse) go to the scan form below, one wave each.
// Otherwise: one row at a time per wave, one candidate column per lane (emit_row_scan) -- O(rows^2 / 64) wave iterations, which at
// T = 300, K = 2 was 11-25 k cycles of the sentence's 29-43 k.
constexpr int EMIT_SORT_MAX = 12;
// What emit_rows() needs beyond the entry arrays when the output is token-packed: the sentence's offsets are resolved INSIDE it.
struct PackedEmit {
    PackedOut po;
    int b, B;
    int32_t *row_ptr, *rowT_ptr, *status;
    uint8_t* pool_mask;
    int *s_red, *s_status;
};
__device__ void emit_rows(int* s_out, int T, int* smem, int err, int n_edge_rows, int lane, int wave, bool staged,
                          int32_t* __restrict__ col_idx, int32_t* __restrict__ label, int32_t* __restrict__ colT_idx,
                          int32_t* __restrict__ ell, int32_t* __restrict__ ellT, unsigned long long* stamps, const PackedEmit& pe) {
    const bool packed = pe.po.cu != nullptr;
    if (err && !packed) return;
    GCNPT_STAMP(stamps, 8);
    int* cnt = smem + T;               // free after the pruning phases: fill counter of the forward rows
    int* deg = smem + 2 * T;
    int* degT = deg + T + 1;
    int* rank = degT + T + 1;          // free as well: fill counter of the transposed rows
    int* lab = rank + T + 1;
    int* einfo = lab + T;
    size_t base = (size_t)s_out[0];                        // first row, first entries of the two patterns, column shift (prune_sentence)
    int ebase = s_out[1], eTbase = s_out[2], cshift = s_out[3];
    int done = -1;                                         // (thread 0, packed) how many workgroups had resolved before this one

    // Token-packed output: the look-back of PackedOut.  Sums the words of the sentences before this one (spinning on their valid bits),
    // then writes everything of the sentence that is not an entry of an edge row: cu, row offsets, empty ELL heads, masks, sentence ids.
    // All threads; false = the sentence does not fit the caller's arrays (nothing of it is written: gcnpt_pack_trees' rule).
    auto resolve = [&]() -> bool {
        const PackedOut& po = pe.po;
        const int b = pe.b, B = pe.B;
        const int len = s_out[0], my_nnz = s_out[1], my_nnzT = s_out[2];
        int a0 = 0, a1 = 0, a2 = 0;
        for (int j = threadIdx.x; j < b; j += PRUNE_THREADS) {
            unsigned long long v = __hip_atomic_load(po.sync + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (!(v & PK_VALID)) {
                __builtin_amdgcn_s_sleep(2);
                v = __hip_atomic_load(po.sync + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            a0 += (int)((v >> 40) & 0xffff); a1 += (int)((v >> 20) & 0xfffff); a2 += (int)(v & 0xfffff);
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
        __syncthreads();                                               // (s_out and the sums of the pruning phases are consumed)
        if (threadIdx.x == 0) { pe.s_red[0] = 0; pe.s_red[1] = 0; pe.s_red[2] = 0; }
        __syncthreads();
        if (lane == 0 && (a0 | a1 | a2)) { atomicAdd(&pe.s_red[0], a0); atomicAdd(&pe.s_red[1], a1); atomicAdd(&pe.s_red[2], a2); }
        __syncthreads();
        a0 = pe.s_red[0]; a1 = pe.s_red[1]; a2 = pe.s_red[2];
        // this workgroup has polled every sentence before it: the last one to get here leaves the workspace zeroed (finish(), below; the
        // counter's old value is only looked at there, so its round trip is off the sentence's critical path)
        if (threadIdx.x == 0) done = (int)atomicAdd(po.sync + B + 1, 1ull);
        base = (size_t)a0; ebase = a1; eTbase = a2; cshift = a0;
        const bool fits = a0 + len <= po.n_rows && a1 + my_nnz <= po.nnz_cap && a2 + my_nnzT <= po.nnz_cap;
        if (threadIdx.x == 0) {
            po.cu[b] = a0;
            if (b == B - 1) {                                          // the totals: the one writer of the batch-level results
                po.cu[B] = a0 + len;
                po.pk_status[0] = fits ? 0 : GCNPT_E_CAPACITY;         // (offsets grow with b: the last sentence fits iff all do)
                po.pk_status[1] = a0 + len;
                if (fits) {
                    pe.row_ptr[a0 + len] = a1 + my_nnz;
                    if (pe.rowT_ptr) pe.rowT_ptr[a0 + len] = a2 + my_nnzT;
                }
            }
            if (!fits) *pe.s_status = GCNPT_E_CAPACITY;
        }
        if (!fits) return false;
        for (int i = threadIdx.x; i < len; i += PRUNE_THREADS) {
            po.row_sent[a0 + i] = b;
            pe.row_ptr[a0 + i] = a1 + (err ? 0 : deg[i]);
            if (pe.rowT_ptr) pe.rowT_ptr[a0 + i] = a2 + (err ? 0 : degT[i]);
            if (pe.pool_mask) pe.pool_mask[a0 + i] = err ? 1 : ((deg[i + 1] - deg[i]) + (degT[i + 1] - degT[i])) == 0;
            if (err || !(smem[i] & (F_HASEDGE << PW_SHIFT))) {         // no entries: an all-zero ELL head
                int4* e = reinterpret_cast<int4*>(ell + (size_t)(a0 + i) * 8);
                e[0] = make_int4(0, 0, 0, 0); e[1] = make_int4(0, 0, 0, 0);
                if (ellT) {
                    int4* eT = reinterpret_cast<int4*>(ellT + (size_t)(a0 + i) * 8);
                    eT[0] = make_int4(0, 0, 0, 0); eT[1] = make_int4(0, 0, 0, 0);
                }
            }
        }
        return true;
    };
    auto finish = [&]() {
        if (packed && threadIdx.x == 0 && done == pe.B - 1)
            for (int j = 0; j < pe.B + 2; ++j) pe.po.sync[j] = 0ull;
    };
    if (err) {                                                         // (packed) an empty sentence still has its rows
        resolve();
        finish();
        return;
    }

    if (staged && (n_edge_rows > WAVE || packed)) {          // (up to 64 edge rows the scan form is as fast: 4.7 k against 5.3 k cycles at T = 100, K = 1)
        int* entF = einfo + T;         // [nnz]  forward entries: column (the label follows from the pair, see below)
        int* entT = entF + 3 * T;      // [nnzT] transposed entries: column
        int* longrows = entT + 3 * T;  // [<= T / EMIT_SORT_MAX + 1] edge rows left to the scan form; [T-1] (end of the region) = their count
        int* n_long = longrows + T - 1;
        for (int i = threadIdx.x; i < T; i += PRUNE_THREADS) { cnt[i] = 0; rank[i] = 0; }
        if (threadIdx.x == 0) *n_long = 0;
        __syncthreads();
        for (int q = threadIdx.x; q < n_edge_rows; q += PRUNE_THREADS) {
            const int me = einfo[q];
            const int r = me & 0xfff, rp = ((me >> 12) & 0xfff) - 2;
            entF[deg[r] + atomicAdd(&cnt[r], 1)] = r;                                               // adj[r,r] = 84 (tree.py:186-187)
            entT[degT[r] + atomicAdd(&rank[r], 1)] = r;
            if (me & (1 << 24)) {                                                                   // r is a kept child of rp
                if (me & (1 << 25)) {                                                               // adj[rp,r] = deprel[r]
                    entF[deg[rp] + atomicAdd(&cnt[rp], 1)] = r;
                    entT[degT[r] + atomicAdd(&rank[r], 1)] = rp;
                }
                if (me & (1 << 26)) {                                                               // adj[r,rp] = deprel[r] + 42
                    entF[deg[r] + atomicAdd(&cnt[r], 1)] = rp;
                    entT[degT[rp] + atomicAdd(&rank[rp], 1)] = r;
                }
            }
        }
        __syncthreads();
        // a thread's edge row: its entries sorted by column in registers (sort_row), then written with its ELL head (write_row).  Packed
        // output, one row per thread at most: the offsets are resolved BETWEEN the two, so a sentence that has to wait for a slower one
        // before it waits with its rows sorted
        struct Row { int r, o, n, oT, nT; int v[EMIT_SORT_MAX], vT[EMIT_SORT_MAX]; };
        auto sort_row = [&](int q, Row& w) -> bool {
            w.r = einfo[q] & 0xfff;
            w.o = deg[w.r]; w.n = deg[w.r + 1] - w.o; w.oT = degT[w.r]; w.nT = degT[w.r + 1] - w.oT;
            if (w.n > EMIT_SORT_MAX || w.nT > EMIT_SORT_MAX) { longrows[atomicAdd(n_long, 1)] = q; return false; }
#pragma unroll
            for (int k = 0; k < EMIT_SORT_MAX; ++k) {
                w.v[k] = k < w.n ? entF[w.o + k] : 0x7fffffff;
                w.vT[k] = k < w.nT ? entT[w.oT + k] : 0x7fffffff;
            }
            // (a fixed odd-even transposition network keeps the arrays in registers; unused slots hold INT_MAX and stay behind)
#pragma unroll
            for (int pass = 0; pass < EMIT_SORT_MAX; ++pass)
#pragma unroll
                for (int k = pass & 1; k + 1 < EMIT_SORT_MAX; k += 2) {
                    const bool sw = w.v[k] > w.v[k + 1];
                    const int lo = sw ? w.v[k + 1] : w.v[k], hi = sw ? w.v[k] : w.v[k + 1];
                    w.v[k] = lo; w.v[k + 1] = hi;
                    const bool swT = w.vT[k] > w.vT[k + 1];
                    const int loT = swT ? w.vT[k + 1] : w.vT[k], hiT = swT ? w.vT[k] : w.vT[k + 1];
                    w.vT[k] = loT; w.vT[k + 1] = hiT;
                }
            return true;
        };
        auto write_row = [&](const Row& w) {
            const int r = w.r, o = w.o, n = w.n, oT = w.oT, nT = w.nT;
            int hd[8], hdT[8];
            hd[0] = n; hdT[0] = nT;
#pragma unroll
            for (int k = 0; k < EMIT_SORT_MAX; ++k) {
                if (k < n) {
                    const int j = w.v[k];
                    col_idx[ebase + o + k] = j + cshift;
                    // the value tree_to_adj wrote there (tree.py:184-192): 84 on the diagonal, deprel[j] for a child j, deprel[r] + 42 for the parent
                    if (label) label[ebase + o + k] = j == r ? SELF_LOOP_ID : (pw_par(smem[j]) == r ? lab[j] : lab[r] + FWD_BOUND);
                }
                if (k < nT && colT_idx) colT_idx[eTbase + oT + k] = w.vT[k] + cshift;
                if (k < 7) { hd[1 + k] = k < n ? w.v[k] + cshift : 0; hdT[1 + k] = k < nT ? w.vT[k] + cshift : 0; }
            }
            int4* e = reinterpret_cast<int4*>(ell + (base + r) * 8);
            e[0] = make_int4(hd[0], hd[1], hd[2], hd[3]); e[1] = make_int4(hd[4], hd[5], hd[6], hd[7]);
            if (ellT) {
                int4* eT = reinterpret_cast<int4*>(ellT + (base + r) * 8);
                eT[0] = make_int4(hdT[0], hdT[1], hdT[2], hdT[3]); eT[1] = make_int4(hdT[4], hdT[5], hdT[6], hdT[7]);
            }
        };
        if (packed && n_edge_rows <= PRUNE_THREADS) {
            Row w;
            const bool mine = (int)threadIdx.x < n_edge_rows && sort_row(threadIdx.x, w);
            if (!resolve()) { finish(); return; }
            if (mine) write_row(w);
        } else {
            if (packed && !resolve()) { finish(); return; }
            for (int q = threadIdx.x; q < n_edge_rows; q += PRUNE_THREADS) {
                Row w;
                if (sort_row(q, w)) write_row(w);
            }
        }
        __syncthreads();
        const int nl = *n_long;
        for (int x = wave; x < nl; x += PRUNE_THREADS / WAVE) {           // the few long rows: one wave each, scan form
            const int me = einfo[longrows[x]];
            const int r = me & 0xfff;
            emit_row_scan(base, ebase, eTbase, cshift, lab, einfo, n_edge_rows, lane, me, lab[r], deg[r], degT[r], col_idx, label, colT_idx, ell, ellT);
        }
        GCNPT_STAMP(stamps, 7);
        finish();
        return;
    }
    if (packed && !resolve()) { finish(); return; }
    for (int q0 = 0; q0 < n_edge_rows; q0 += WAVE) {                  // rows, a chunk of 64 at a time (registers)
        const int rw = q0 + lane < n_edge_rows ? einfo[q0 + lane] : 0;
        const int rdeg = deg[rw & 0xfff], rdegT = degT[rw & 0xfff], rlab = lab[rw & 0xfff];
        const int nq = min(WAVE, n_edge_rows - q0);
        for (int qq = wave; qq < nq; qq += PRUNE_THREADS / WAVE) {    // this wave's rows of the chunk
            emit_row_scan(base, ebase, eTbase, cshift, lab, einfo, n_edge_rows, lane, __builtin_amdgcn_readlane(rw, qq), __builtin_amdgcn_readlane(rlab, qq),
                          __builtin_amdgcn_readlane(rdeg, qq), __builtin_amdgcn_readlane(rdegT, qq), col_idx, label, colT_idx, ell, ellT);
        }
    }
    GCNPT_STAMP(stamps, 7);
    finish();
}

__global__ __launch_bounds__(PRUNE_THREADS) void prune_to_csr_kernel(
    const int64_t* __restrict__ head, const int64_t* __restrict__ subj_pos, const int64_t* __restrict__ obj_pos,
    const int64_t* __restrict__ deprel, const uint8_t* __restrict__ pad_mask, const int32_t* __restrict__ len_in,
    int B, int T, int prune_k, int cap, int32_t* __restrict__ row_ptr, int32_t* __restrict__ col_idx,
    int32_t* __restrict__ label, int32_t* __restrict__ rowT_ptr, int32_t* __restrict__ colT_idx,
    int32_t* __restrict__ ell, int32_t* __restrict__ ellT, uint8_t* __restrict__ pool_mask,
    int32_t* __restrict__ status, unsigned long long* stamps, const PackParams pk, int pk_dtype, int staged, const PackedOut po) {
    extern __shared__ int smem[];  // carved in prune_sentence() (+ the entry lists of emit_rows when staged)
    __shared__ int s_err;
    if ((int)blockIdx.x >= B) {    // side job of gcnpt_prune_to_csr_pack: the launch leaves most CUs idle, these workgroups pack the weights
        pack_side_job(pk, pk_dtype, B);
        return;
    }

    __shared__ int s_status, s_nrows, s_red[4], s_out[4], s_ticket;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // packed output: sentences in ticket order (PackedOut), else the workgroup's own
    if (threadIdx.x == 0) s_ticket = po.cu ? (int)atomicAdd(po.sync + B, 1ull) : (int)blockIdx.x;
    if (threadIdx.x == 0) { s_err = 0; s_status = 0; s_nrows = 0; s_red[0] = s_red[1] = s_red[2] = 0; s_red[3] = 0x7fffffff; }
    GCNPT_STAMP_REAL(stamps);
    GCNPT_STAMP(stamps, 0);
    __shared__ int s_maxlen;
    if (threadIdx.x == 0) s_maxlen = 0;
    __syncthreads();
    const int b = s_ticket;
    const bool allw = T > PRUNE_WAVE0_T;
    // status[B] = longest sentence of the batch (gcn.py:97): waves of workgroup 0 that have no token to prune count every sentence's
    // non-pad slots while the others prune (wave 0 alone, or the waves holding the T tokens), so the launch needs neither a memset
    // of that word nor one atomic per sentence.  (Big batches, or sentences that occupy every wave: atomicMax, see prune_impl.)
    const int first_idle = allw ? (T + WAVE - 1) / WAVE : 1;
    if (b == 0 && (long long)B * T <= PRUNE_SCAN_MAX && first_idle < PRUNE_THREADS / WAVE && wave >= first_idle) {
        const int n_idle = PRUNE_THREADS / WAVE - first_idle;
        int m = 0;
        for (int sidx = wave - first_idle; sidx < B; sidx += n_idle) {
            int n;
            if (pad_mask) {
                n = 0;
                for (int i0 = 0; i0 < T; i0 += WAVE) {
                    const int i = i0 + lane;
                    n += __popcll(__ballot(i < T && pad_mask[(size_t)sidx * T + min(i, T - 1)] == 0));
                }
            } else {
                n = min(max(len_in[sidx], 0), T);
            }
            m = max(m, n);
        }
        if (lane == 0) atomicMax(&s_maxlen, m);
    }
    if (allw) prune_sentence<true>(head, subj_pos, obj_pos, deprel, pad_mask, len_in, b, B, T, prune_k, cap, smem, &s_err, &s_status,
                                   &s_nrows, s_red, row_ptr, rowT_ptr, ell, ellT, pool_mask, status, stamps, po, s_out);
    else if (wave == 0) prune_sentence<false>(head, subj_pos, obj_pos, deprel, pad_mask, len_in, b, B, T, prune_k, cap, smem, &s_err, &s_status,
                                              &s_nrows, s_red, row_ptr, rowT_ptr, ell, ellT, pool_mask, status, stamps, po, s_out);
    __syncthreads();
    if (b == 0 && threadIdx.x == 0 && (long long)B * T <= PRUNE_SCAN_MAX && first_idle < PRUNE_THREADS / WAVE) status[B] = s_maxlen;
    const PackedEmit pe{po, b, B, row_ptr, rowT_ptr, status, pool_mask, s_red, &s_status};
    emit_rows(s_out, T, smem, s_status, s_nrows, lane, wave, staged != 0, col_idx, label, colT_idx, ell, ellT, stamps, pe);
    // (packed output that did not fit: the per-sentence code stays what the pruning found; the batch-level status says E_CAPACITY)
}

// ---- dense float adjacency -> CSR of (adj != 0) and of its transpose (gcn.py:260-262) -----------------
__global__ __launch_bounds__(ADJ_THREADS) void adj_to_csr_kernel(
    const float* __restrict__ adj, int B, int T, int cap, int32_t* __restrict__ row_ptr, int32_t* __restrict__ col_idx,
    int32_t* __restrict__ label, int32_t* __restrict__ rowT_ptr, int32_t* __restrict__ colT_idx,
    int32_t* __restrict__ ell, int32_t* __restrict__ ellT, uint8_t* __restrict__ pool_mask,
    int32_t* __restrict__ status) {
    extern __shared__ int smem[];
    int* deg = smem;              // [T+1]
    int* degT = deg + T + 1;      // [T+1]
    int* scratch = degT + T + 1;  // [ADJ_THREADS]
    const int b = blockIdx.x, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, NW = ADJ_THREADS / WAVE;
    const float* A = adj + (size_t)b * T * T;

    for (int r = wave; r < T; r += NW) {
        int n = 0, nT = 0;
        for (int c0 = 0; c0 < T; c0 += WAVE) {
            const int c = c0 + lane;
            const bool nz = c < T && A[(size_t)r * T + c] != 0.0f;
            const bool nzT = c < T && A[(size_t)c * T + r] != 0.0f;
            n += __popcll(__ballot(nz));
            nT += __popcll(__ballot(nzT));
        }
        if (lane == 0) { deg[r] = n; degT[r] = nT; }
    }
    __syncthreads();
    if (pool_mask)
        for (int i = t; i < T; i += ADJ_THREADS) pool_mask[(size_t)b * T + i] = (deg[i] + degT[i]) == 0;
    __syncthreads();
    block_exclusive_scan(deg, T, scratch);
    block_exclusive_scan(degT, T, scratch);
    const bool over = deg[T] > cap;
    for (int i = t; i <= T; i += ADJ_THREADS) {
        row_ptr[(size_t)b * (T + 1) + i] = b * cap + (over ? 0 : deg[i]);
        if (rowT_ptr) rowT_ptr[(size_t)b * (T + 1) + i] = b * cap + (over ? 0 : degT[i]);
    }
    if (t == 0) {
        status[b] = over ? GCNPT_E_CAPACITY : 0;
        atomicMax(&status[B], T);
    }
    // ELL heads start as "count = 0"; the fill below overwrites the rows that have entries
    for (int i = t; i < T * 8; i += ADJ_THREADS) {
        ell[(size_t)b * T * 8 + i] = 0;
        if (ellT) ellT[(size_t)b * T * 8 + i] = 0;
    }
    __syncthreads();
    if (over) return;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int r = wave; r < T; r += NW) {
        const int o0 = b * cap + deg[r], oT0 = b * cap + degT[r];
        int o = o0, oT = oT0;
        int32_t* e = ell + ((size_t)b * T + r) * 8;
        int32_t* eT = ellT ? ellT + ((size_t)b * T + r) * 8 : nullptr;
        for (int c0 = 0; c0 < T; c0 += WAVE) {
            const int c = c0 + lane;
            const float v = c < T ? A[(size_t)r * T + c] : 0.0f;
            const float vT = c < T ? A[(size_t)c * T + r] : 0.0f;
            const bool nz = c < T && v != 0.0f, nzT = c < T && vT != 0.0f;
            const unsigned long long m = __ballot(nz), mT = __ballot(nzT);
            if (nz) {
                const int k = o + __popcll(m & lt);
                col_idx[k] = c; if (label) label[k] = (int)v;
                if (k - o0 < 7) e[1 + k - o0] = c;
            }
            if (nzT) {
                const int k = oT + __popcll(mT & lt);
                if (colT_idx) colT_idx[k] = c;
                if (eT && k - oT0 < 7) eT[1 + k - oT0] = c;
            }
            o += __popcll(m);
            oT += __popcll(mT);
        }
        if (lane == 0) { e[0] = o - o0; if (eT) eT[0] = oT - oT0; }
    }
}

__global__ void csr_to_adj_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                  const int32_t* __restrict__ label, int B, int T, float* __restrict__ adj) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B * T) return;
    const int b = r / T, i = r - b * T;
    const int beg = row_ptr[(size_t)b * (T + 1) + i], end = row_ptr[(size_t)b * (T + 1) + i + 1];
    for (int e = beg; e < end; ++e)
        adj[((size_t)b * T + i) * T + col_idx[e]] = label ? (float)label[e] : 1.0f;
}

// ---- N4: a batch assembled from a dataset that was pruned once (loader.py:81-141 builds batches from cached features) ----
// Sentence idx[b] of the cached PrunedTrees (S sentences padded to Ts, capacity cap_s) becomes sentence b of a PrunedTrees
// for [B, T].  Columns are sentence-local, so entries and ELL heads are copied as they are; only offsets are re-based.
constexpr int GATHER_THREADS = 256;
struct TreeArrays {
    int32_t *row_ptr, *col_idx, *label, *rowT_ptr, *colT_idx, *ell, *ellT;
    uint8_t* pool_mask;
    int32_t* status;
};
__global__ __launch_bounds__(GATHER_THREADS) void gather_trees_kernel(const TreeArrays src, const int32_t* __restrict__ src_len,
                                                                     int S, int Ts, int cap_s, const int64_t* __restrict__ idx,
                                                                     int B, int T, int cap, const TreeArrays dst, const PackParams pk,
                                                                     int pk_dtype) {
    __shared__ int s_max[GATHER_THREADS / WAVE];
    if ((int)blockIdx.x >= B) {    // side job of gcnpt_gather_trees_pack
        pack_side_job(pk, pk_dtype, B);
        return;
    }
    const int b = blockIdx.x, t = threadIdx.x;
    const int64_t s64 = idx[b];
    const bool known = s64 >= 0 && s64 < S;
    const size_t s = known ? (size_t)s64 : 0;
    const int len = known ? src_len[s] : 0;
    const int nnz = src.row_ptr[s * (Ts + 1) + Ts] - (int)s * cap_s;
    const int nnzT = src.rowT_ptr ? src.rowT_ptr[s * (Ts + 1) + Ts] - (int)s * cap_s : 0;
    int code = known ? src.status[s] : GCNPT_E_INVALID;
    if (code == 0 && len > T) code = GCNPT_E_LENGTH;
    if (code == 0 && (nnz > cap || nnzT > cap)) code = GCNPT_E_CAPACITY;
    const bool ok = code == 0;                           // otherwise: an empty, fully masked sentence, as the pruner leaves it
    for (int i = t; i <= T; i += GATHER_THREADS) {
        const size_t o = s * (Ts + 1) + min(i, Ts);
        dst.row_ptr[(size_t)b * (T + 1) + i] = b * cap + (ok ? src.row_ptr[o] - (int)s * cap_s : 0);
        if (dst.rowT_ptr) dst.rowT_ptr[(size_t)b * (T + 1) + i] = b * cap + (ok ? src.rowT_ptr[o] - (int)s * cap_s : 0);
    }
    if (ok) {
        for (int k = t; k < nnz; k += GATHER_THREADS) {
            dst.col_idx[(size_t)b * cap + k] = src.col_idx[s * cap_s + k];
            if (dst.label) dst.label[(size_t)b * cap + k] = src.label[s * cap_s + k];
        }
        if (dst.colT_idx)
            for (int k = t; k < nnzT; k += GATHER_THREADS) dst.colT_idx[(size_t)b * cap + k] = src.colT_idx[s * cap_s + k];
    }
    typedef int i32x4_t __attribute__((ext_vector_type(4)));       // (a first-class vector: a select between int4 STRUCTS went through scratch memory)
    const i32x4_t z = {0, 0, 0, 0};
    for (int q = t; q < 2 * T; q += GATHER_THREADS) {    // ELL heads, 16 bytes at a time
        const int i = q >> 1;
        const bool have = ok && i < Ts;
        const size_t o = (s * Ts + min(i, Ts - 1)) * 2 + (q & 1);
        reinterpret_cast<i32x4_t*>(dst.ell)[(size_t)b * T * 2 + q] = have ? reinterpret_cast<const i32x4_t*>(src.ell)[o] : z;
        if (dst.ellT) reinterpret_cast<i32x4_t*>(dst.ellT)[(size_t)b * T * 2 + q] = have ? reinterpret_cast<const i32x4_t*>(src.ellT)[o] : z;
    }
    if (dst.pool_mask)
        for (int i = t; i < T; i += GATHER_THREADS)
            dst.pool_mask[(size_t)b * T + i] = (ok && i < Ts) ? src.pool_mask[s * Ts + i] : (uint8_t)1;
    if (t == 0) dst.status[b] = code;
    if (b == 0) {                                        // status[B] = longest sentence of the batch (gcn.py:97): no memset, no atomics
        int m = 0;
        for (int j = t; j < B; j += GATHER_THREADS) {
            const int64_t sj = idx[j];
            if (sj >= 0 && sj < S) m = max(m, src_len[sj]);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d));
        if ((t & 63) == 0) s_max[t >> 6] = m;
        __syncthreads();
        if (t == 0) {
            for (int w = 1; w < GATHER_THREADS / WAVE; ++w) m = max(m, s_max[w]);
            dst.status[B] = m;
        }
    }
}

// The same assembly straight into the TOKEN-PACKED layout (what gcnpt_pack_trees makes of gcnpt_gather_trees' arrays, bit for bit): a
// sentence's offsets are prefix sums over the batch sentences before it, which every workgroup takes from the CACHE itself (lengths and
// offsets of the cached sentences are all there: no inter-workgroup hand-off as in the pruner's packed form).
struct GatherPackedDst {
    int32_t *cu, *row_ptr, *col_idx, *label, *rowT_ptr, *colT_idx, *ell, *ellT;
    uint8_t *pool_mask, *pool_mask_padded;
    int32_t *row_sent, *status, *sent_status;
    int n_rows, nnz_cap;
};
__global__ __launch_bounds__(GATHER_THREADS) void gather_trees_packed_kernel(const TreeArrays src, const int32_t* __restrict__ src_len,
                                                                            int S, int Ts, int cap_s, const int64_t* __restrict__ idx,
                                                                            int B, int T, const GatherPackedDst dst, const PackParams pk,
                                                                            int pk_dtype) {
    __shared__ int s_red[GATHER_THREADS / WAVE][4];
    if ((int)blockIdx.x >= B) {
        pack_side_job(pk, pk_dtype, B);
        return;
    }
    const int b = blockIdx.x, t = threadIdx.x;
    // what sentence j of the batch contributes: its code, rows and entries (a failed sentence: its rows, no entries -- as the padded
    // assembly leaves it)
    auto look = [&](int j, int& code, int& len, int& nnz, int& nnzT, size_t& s) {
        const int64_t s64 = idx[j];
        const bool known = s64 >= 0 && s64 < S;
        s = known ? (size_t)s64 : 0;
        const int l = known ? src_len[s] : 0;
        nnz = src.row_ptr[s * (Ts + 1) + Ts] - (int)s * cap_s;
        nnzT = src.rowT_ptr ? src.rowT_ptr[s * (Ts + 1) + Ts] - (int)s * cap_s : 0;
        code = known ? src.status[s] : GCNPT_E_INVALID;
        if (code == 0 && l > T) code = GCNPT_E_LENGTH;
        if (code == 0 && (nnz > 3 * T || nnzT > 3 * T)) code = GCNPT_E_CAPACITY;
        len = min(l, T);
        if (code != 0) { nnz = 0; nnzT = 0; }
    };
    int a[4] = {0, 0, 0, 0};                                        // rows, entries, transposed entries before b; longest sentence
    for (int j = t; j < B; j += GATHER_THREADS) {
        int code, len, nnz, nnzT;
        size_t s;
        look(j, code, len, nnz, nnzT, s);
        if (j < b) { a[0] += len; a[1] += nnz; a[2] += nnzT; }
        const int64_t sj = idx[j];
        a[3] = max(a[3], (sj >= 0 && sj < S) ? src_len[sj] : 0);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a[0] += __shfl_xor(a[0], d); a[1] += __shfl_xor(a[1], d); a[2] += __shfl_xor(a[2], d); a[3] = max(a[3], __shfl_xor(a[3], d)); }
    if ((t & 63) == 0) { s_red[t >> 6][0] = a[0]; s_red[t >> 6][1] = a[1]; s_red[t >> 6][2] = a[2]; s_red[t >> 6][3] = a[3]; }
    __syncthreads();
    int cu = 0, eo = 0, eoT = 0, longest = 0;
#pragma unroll
    for (int w = 0; w < GATHER_THREADS / WAVE; ++w) { cu += s_red[w][0]; eo += s_red[w][1]; eoT += s_red[w][2]; longest = max(longest, s_red[w][3]); }
    int code, len, nnz, nnzT;
    size_t s;
    look(b, code, len, nnz, nnzT, s);
    const bool ok = code == 0;
    const bool fits = cu + len <= dst.n_rows && eo + nnz <= dst.nnz_cap && eoT + nnzT <= dst.nnz_cap;
    if (t == 0) {
        dst.cu[b] = cu;
        dst.sent_status[b] = code;
        if (b == 0) dst.sent_status[B] = longest;
        if (b == B - 1) {
            dst.cu[B] = cu + len;
            dst.status[0] = fits ? 0 : GCNPT_E_CAPACITY;
            dst.status[1] = cu + len;
            if (fits) {
                dst.row_ptr[cu + len] = eo + nnz;
                if (dst.rowT_ptr) dst.rowT_ptr[cu + len] = eoT + nnzT;
            }
        }
    }
    if (dst.pool_mask_padded)
        for (int i = t; i < T; i += GATHER_THREADS)
            dst.pool_mask_padded[(size_t)b * T + i] = (ok && i < Ts) ? src.pool_mask[s * Ts + i] : (uint8_t)1;
    if (!fits) return;
    const int rp0 = (int)s * cap_s;
    for (int i = t; i < len; i += GATHER_THREADS) {
        dst.row_ptr[cu + i] = eo + (ok ? src.row_ptr[s * (Ts + 1) + i] - rp0 : 0);
        if (dst.rowT_ptr) dst.rowT_ptr[cu + i] = eoT + (ok ? src.rowT_ptr[s * (Ts + 1) + i] - rp0 : 0);
        dst.pool_mask[cu + i] = ok ? src.pool_mask[s * Ts + i] : (uint8_t)1;
        dst.row_sent[cu + i] = b;
    }
    for (int k = t; k < nnz; k += GATHER_THREADS) {
        dst.col_idx[eo + k] = src.col_idx[s * cap_s + k] + cu;
        if (dst.label) dst.label[eo + k] = src.label[s * cap_s + k];
    }
    if (dst.colT_idx)
        for (int k = t; k < nnzT; k += GATHER_THREADS) dst.colT_idx[eoT + k] = src.colT_idx[s * cap_s + k] + cu;
    for (int q = t; q < 2 * len; q += GATHER_THREADS) {             // ELL heads, 16 bytes at a time: [count, c0..c2] / [c3..c6], live slots + cu
        const int i = q >> 1, half = q & 1;
        const size_t o = (s * Ts + i) * 2 + half;
        auto shift = [&](const int32_t* e_src, int32_t* e_dst) {
            const int cnt = ok ? e_src[(s * Ts + i) * 8] : 0;
            int4 v = ok ? reinterpret_cast<const int4*>(e_src)[o] : make_int4(0, 0, 0, 0);
            int* e = reinterpret_cast<int*>(&v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int slot = half * 4 + j - 1;                   // entry number of this word (-1 = the count)
                if (slot >= 0) e[j] = slot < cnt ? e[j] + cu : 0;
            }
            reinterpret_cast<int4*>(e_dst)[((size_t)(cu + i)) * 2 + half] = v;
        };
        shift(src.ell, dst.ell);
        if (dst.ellT) shift(src.ellT, dst.ellT);
    }
}

// ---- N1: "pooled-only" rows (gcn.py:116-121 pools h over the tokens of the pruned tree only, and a tree token's row of
// every layer depends on tree tokens only -- the adjacency has no entry outside the tree).  The kept tokens of a sentence
// are renumbered 0..kept-1 in token order and the pattern is rewritten in those numbers for a [B, Tc] batch: the layer
// kernels then run on B*Tc rows instead of B*T (a pruned tree keeps ~1 token in 4-8), with identical values in every
// kept row.  One workgroup per sentence; slot numbers of the sentence's tokens live in LDS.
constexpr int COMPACT_THREADS = 256;
__global__ __launch_bounds__(COMPACT_THREADS) void compact_trees_kernel(const TreeArrays src, int B, int T, int cap, int Tc, int cap_c,
                                                                       const TreeArrays dst, int64_t* __restrict__ tok,
                                                                       int32_t* __restrict__ kept_out) {
    extern __shared__ int c_slot[];                      // [T] slot of token i, or -1
    __shared__ int s_wave[COMPACT_THREADS / WAVE];
    __shared__ int s_base;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_base = 0;
    __syncthreads();
    // exclusive scan of the in-tree flags, 256 tokens per round
    for (int i0 = 0; i0 < T; i0 += COMPACT_THREADS) {
        const int i = i0 + t;
        const bool in = i < T && src.pool_mask[(size_t)b * T + i] == 0;
        const unsigned long long m = __ballot(in);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (i < T) c_slot[i] = in ? before + __popcll(m & ((1ull << lane) - 1ull)) : -1;
        __syncthreads();
        if (t == 0) {
            int tot = s_base;
            for (int w = 0; w < COMPACT_THREADS / WAVE; ++w) tot += s_wave[w];
            s_base = tot;
        }
        __syncthreads();
    }
    const int kept = s_base;
    const int nnz = src.row_ptr[(size_t)b * (T + 1) + T] - b * cap;
    const int nnzT = src.rowT_ptr ? src.rowT_ptr[(size_t)b * (T + 1) + T] - b * cap : 0;
    int code = src.status[b];
    if (code == 0 && kept > Tc) code = GCNPT_E_LENGTH;
    if (code == 0 && (nnz > cap_c || nnzT > cap_c)) code = GCNPT_E_CAPACITY;
    const bool ok = code == 0;
    if (t == 0) {
        dst.status[b] = code;
        kept_out[b] = ok ? kept : 0;
        if (ok) atomicMax(&dst.status[B], kept);
    }
    // slots past the kept tokens (or the whole sentence when it failed): no entries, excluded from pooling
    const int first_free = ok ? kept : 0;
    for (int j = first_free + t; j <= Tc; j += COMPACT_THREADS) {
        dst.row_ptr[(size_t)b * (Tc + 1) + j] = b * cap_c + (ok ? nnz : 0);
        if (dst.rowT_ptr) dst.rowT_ptr[(size_t)b * (Tc + 1) + j] = b * cap_c + (ok ? nnzT : 0);
        if (j < Tc) {
            tok[(size_t)b * Tc + j] = -1;
            dst.pool_mask[(size_t)b * Tc + j] = 1;
            int4* e = reinterpret_cast<int4*>(dst.ell + ((size_t)b * Tc + j) * 8);
            e[0] = make_int4(0, 0, 0, 0); e[1] = make_int4(0, 0, 0, 0);
            if (dst.ellT) {
                int4* eT = reinterpret_cast<int4*>(dst.ellT + ((size_t)b * Tc + j) * 8);
                eT[0] = make_int4(0, 0, 0, 0); eT[1] = make_int4(0, 0, 0, 0);
            }
        }
    }
    if (!ok) return;
    auto renumber_head = [&](const int32_t* in, int32_t* out) {       // [count, 7 columns] -> the same in slot numbers
        const int n = in[0];
        out[0] = n;
#pragma unroll
        for (int k = 1; k < 8; ++k) out[k] = k <= n ? c_slot[min(max(in[k], 0), T - 1)] : 0;
    };
    for (int i = t; i < T; i += COMPACT_THREADS) {
        const int j = c_slot[i];
        if (j < 0) continue;
        tok[(size_t)b * Tc + j] = i;
        dst.pool_mask[(size_t)b * Tc + j] = 0;
        dst.row_ptr[(size_t)b * (Tc + 1) + j] = b * cap_c + src.row_ptr[(size_t)b * (T + 1) + i] - b * cap;
        renumber_head(src.ell + ((size_t)b * T + i) * 8, dst.ell + ((size_t)b * Tc + j) * 8);
        if (dst.rowT_ptr) {
            dst.rowT_ptr[(size_t)b * (Tc + 1) + j] = b * cap_c + src.rowT_ptr[(size_t)b * (T + 1) + i] - b * cap;
            renumber_head(src.ellT + ((size_t)b * T + i) * 8, dst.ellT + ((size_t)b * Tc + j) * 8);
        }
    }
    for (int k = t; k < nnz; k += COMPACT_THREADS) {
        dst.col_idx[(size_t)b * cap_c + k] = c_slot[min(max(src.col_idx[(size_t)b * cap + k], 0), T - 1)];
        if (dst.label) dst.label[(size_t)b * cap_c + k] = src.label[(size_t)b * cap + k];
    }
    if (dst.colT_idx)
        for (int k = t; k < nnzT; k += COMPACT_THREADS)
            dst.colT_idx[(size_t)b * cap_c + k] = c_slot[min(max(src.colT_idx[(size_t)b * cap + k], 0), T - 1)];
}

}  // namespace gcnpt

using namespace gcnpt;

// pack_blocks > 0: that many extra workgroups of the launch pack the weights described by pk (side job)
static int prune_impl(void* stream, const int64_t* head, const int64_t* subj_pos, const int64_t* obj_pos,
                      const int64_t* deprel, const uint8_t* pad_mask, const int32_t* len, int B, int T,
                      int prune_k, int cap, int32_t* row_ptr, int32_t* col_idx, int32_t* label,
                      int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell, int32_t* ellT, uint8_t* pool_mask,
                      int32_t* status, const PackParams& pk, int pk_dtype, int pack_blocks, const PackedOut& po = PackedOut{}) {
    GCNPT_REQUIRE(head && subj_pos && obj_pos && deprel && (pad_mask || len), "prune_to_csr: null input pointer");
    GCNPT_REQUIRE(row_ptr && col_idx && ell && status, "prune_to_csr: null output pointer");
    GCNPT_REQUIRE((rowT_ptr == nullptr) == (ellT == nullptr), "prune_to_csr: rowT_ptr, colT_idx and ellT go together");
    GCNPT_REQUIRE(B > 0 && T > 0 && cap > 0, "prune_to_csr: B, T, cap must be positive (B=%d T=%d cap=%d)", B, T, cap);
    GCNPT_REQUIRE((rowT_ptr == nullptr) == (colT_idx == nullptr), "prune_to_csr: rowT_ptr and colT_idx go together");
    if (prune_k < 0)
        return fail(GCNPT_E_PRUNE_NEGATIVE, "prune_k=%d: the reference fork only works with prune_k >= 0 "
                    "(model/tree.py:194 reads Tree.head, which the unpruned branch never sets)", prune_k);
    if ((long long)B * cap > 0x7fffffffLL) return fail(GCNPT_E_UNSUPPORTED, "prune_to_csr: B*cap overflows int32");
    // 7 words per token for the pruning phases; + 7 more for the staged row emission (3T forward + 3T transposed entries, T for the
    // long-row list) when that still fits LDS (T <= ~2800)
    const size_t lds_base = sizeof(int) * ((size_t)7 * T + 4), lds_staged = sizeof(int) * ((size_t)14 * T + 4);
    const int staged = lds_staged <= 160 * 1024 - 256 ? 1 : 0;
    const size_t lds = staged ? lds_staged : lds_base;
    if (T > PRUNE_MAX_T) return fail(GCNPT_E_UNSUPPORTED, "prune_to_csr: T=%d exceeds the %d tokens a sentence may have", T, PRUNE_MAX_T);
    hipStream_t s = (hipStream_t)stream;
    // sentences beyond ~2300 tokens need more than the default 64 KB of LDS (7 words per token); minus the kernel's few static words
    if (lds > 64 * 1024) GCNPT_LDS_ATTR_ONCE(prune_to_csr_kernel, 160 * 1024 - 256);
    // big batches, and sentences that occupy every wave of their workgroup: atomicMax per sentence (see the kernel)
    if ((long long)B * T > PRUNE_SCAN_MAX || (T > PRUNE_WAVE0_T && T + WAVE > PRUNE_THREADS))
        GCNPT_HIP_CHECK(hipMemsetAsync(status + B, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(prune_to_csr_kernel, dim3(B + pack_blocks), dim3(PRUNE_THREADS), lds, s, head, subj_pos, obj_pos, deprel,
                       pad_mask, len, B, T, prune_k, cap, row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, status,
                       static_cast<unsigned long long*>(g_debug_stamps), pk, pk_dtype, staged, po);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_prune_to_csr(void* stream, const int64_t* head, const int64_t* subj_pos, const int64_t* obj_pos,
                                  const int64_t* deprel, const uint8_t* pad_mask, const int32_t* len, int B, int T,
                                  int prune_k, int cap, int32_t* row_ptr, int32_t* col_idx, int32_t* label,
                                  int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell, int32_t* ellT, uint8_t* pool_mask,
                                  int32_t* status) {
    return prune_impl(stream, head, subj_pos, obj_pos, deprel, pad_mask, len, B, T, prune_k, cap, row_ptr, col_idx, label, rowT_ptr, colT_idx,
                      ell, ellT, pool_mask, status, PackParams{}, GCNPT_F32, 0);
}

// workgroups a side-job pack adds to a launch of `threads` per workgroup: enough for one fragment per thread, at most 192
static int pack_side_blocks(const PackParams& pk, int threads) {
    return (int)std::min<long long>(192, (pk.first[pk.n_layers] + threads - 1) / threads);
}

extern "C" int gcnpt_prune_to_csr_pack(void* stream, const int64_t* head, const int64_t* subj_pos, const int64_t* obj_pos,
                                       const int64_t* deprel, const uint8_t* pad_mask, const int32_t* len, int B, int T,
                                       int prune_k, int cap, int32_t* row_ptr, int32_t* col_idx, int32_t* label,
                                       int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell, int32_t* ellT, uint8_t* pool_mask,
                                       int32_t* status, int n_layers, const float* const* W, const int* H, const int* Din, int dtype,
                                       void* const* w_fwd, void* const* w_bwd) {
    PackParams pk;
    const int rc = fill_pack_params(pk, n_layers, W, H, Din, dtype, w_fwd, w_bwd);
    if (rc != GCNPT_OK) return rc;
    return prune_impl(stream, head, subj_pos, obj_pos, deprel, pad_mask, len, B, T, prune_k, cap, row_ptr, col_idx, label, rowT_ptr, colT_idx,
                      ell, ellT, pool_mask, status, pk, dtype, pack_side_blocks(pk, PRUNE_THREADS));
}

// The pruner writing the token-packed layout itself (what gcnpt_pack_trees makes of gcnpt_prune_to_csr's arrays, bit for bit), optionally
// with the weight pack as a side job: one launch instead of two (three with the pack).
extern "C" int gcnpt_prune_to_csr_packed(void* stream, const int64_t* head, const int64_t* subj_pos, const int64_t* obj_pos,
                                         const int64_t* deprel, const uint8_t* pad_mask, const int32_t* len, int B, int T, int prune_k,
                                         int32_t* cu_seqlens, int32_t* row_ptr, int32_t* col_idx, int32_t* label, int32_t* rowT_ptr,
                                         int32_t* colT_idx, int32_t* ell, int32_t* ellT, uint8_t* pool_mask, int32_t* row_sent, int n_rows,
                                         int nnz_cap, int32_t* status, int32_t* sent_status, uint8_t* pool_mask_padded, uint64_t* sync_ws,
                                         int n_layers, const float* const* W, const int* H, const int* Din, int dtype, void* const* w_fwd,
                                         void* const* w_bwd) {
    GCNPT_REQUIRE(cu_seqlens && row_sent && status && sent_status && sync_ws && pool_mask, "prune_to_csr_packed: null pointer");
    GCNPT_REQUIRE(n_rows > 0 && nnz_cap > 0, "prune_to_csr_packed: n_rows and nnz_cap must be positive");
    GCNPT_REQUIRE(T <= 0xffff && 3 * (long long)T < (1 << 20), "prune_to_csr_packed: T=%d too long", T);
    PackParams pk{};
    int blocks = 0;
    if (n_layers > 0) {
        const int rc = fill_pack_params(pk, n_layers, W, H, Din, dtype, w_fwd, w_bwd);
        if (rc != GCNPT_OK) return rc;
        blocks = pack_side_blocks(pk, PRUNE_THREADS);
    }
    PackedOut po{cu_seqlens, row_sent, pool_mask_padded, reinterpret_cast<unsigned long long*>(sync_ws), status, n_rows, nnz_cap};
    return prune_impl(stream, head, subj_pos, obj_pos, deprel, pad_mask, len, B, T, prune_k, 3 * T, row_ptr, col_idx, label, rowT_ptr, colT_idx,
                      ell, ellT, pool_mask, sent_status, pk, dtype, blocks, po);
}

extern "C" int gcnpt_adj_to_csr(void* stream, const float* adj, int B, int T, int cap, int32_t* row_ptr,
                                int32_t* col_idx, int32_t* label, int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell,
                                int32_t* ellT, uint8_t* pool_mask, int32_t* status) {
    GCNPT_REQUIRE(adj && row_ptr && col_idx && ell && status, "adj_to_csr: null pointer");
    GCNPT_REQUIRE((rowT_ptr == nullptr) == (ellT == nullptr), "adj_to_csr: rowT_ptr, colT_idx and ellT go together");
    GCNPT_REQUIRE(B > 0 && T > 0 && cap > 0, "adj_to_csr: B, T, cap must be positive");
    GCNPT_REQUIRE((rowT_ptr == nullptr) == (colT_idx == nullptr), "adj_to_csr: rowT_ptr and colT_idx go together");
    if ((long long)B * cap > 0x7fffffffLL) return fail(GCNPT_E_UNSUPPORTED, "adj_to_csr: B*cap overflows int32");
    const size_t lds = sizeof(int) * ((size_t)2 * T + 2 + ADJ_THREADS);
    if (lds > 150 * 1024) return fail(GCNPT_E_UNSUPPORTED, "adj_to_csr: T=%d needs %zu B of LDS", T, lds);
    hipStream_t s = (hipStream_t)stream;
    GCNPT_HIP_CHECK(hipMemsetAsync(status + B, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(adj_to_csr_kernel, dim3(B), dim3(ADJ_THREADS), lds, s, adj, B, T, cap, row_ptr, col_idx, label,
                       rowT_ptr, colT_idx, ell, ellT, pool_mask, status);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_csr_to_adj(void* stream, const int32_t* row_ptr, const int32_t* col_idx, const int32_t* label,
                                int B, int T, float* adj) {
    GCNPT_REQUIRE(row_ptr && col_idx && adj, "csr_to_adj: null pointer");
    GCNPT_REQUIRE(B > 0 && T > 0, "csr_to_adj: B and T must be positive");
    hipStream_t s = (hipStream_t)stream;
    GCNPT_HIP_CHECK(hipMemsetAsync(adj, 0, sizeof(float) * (size_t)B * T * T, s));
    const int rows = B * T;
    hipLaunchKernelGGL(csr_to_adj_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, s, row_ptr, col_idx, label, B, T, adj);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

static int gather_impl(void* stream, const int32_t* src_row_ptr, const int32_t* src_col_idx, const int32_t* src_label,
                       const int32_t* src_rowT_ptr, const int32_t* src_colT_idx, const int32_t* src_ell,
                       const int32_t* src_ellT, const uint8_t* src_pool_mask, const int32_t* src_status,
                       const int32_t* src_len, int S, int Ts, int cap_s, const int64_t* idx, int B, int T, int cap,
                       int32_t* row_ptr, int32_t* col_idx, int32_t* label, int32_t* rowT_ptr, int32_t* colT_idx,
                       int32_t* ell, int32_t* ellT, uint8_t* pool_mask, int32_t* status, const PackParams& pk, int pk_dtype, int pack_blocks) {
    GCNPT_REQUIRE(src_row_ptr && src_col_idx && src_ell && src_status && src_len && idx, "gather_trees: null cache pointer");
    GCNPT_REQUIRE(row_ptr && col_idx && ell && status, "gather_trees: null output pointer");
    GCNPT_REQUIRE(S > 0 && Ts > 0 && cap_s > 0 && B > 0 && T > 0 && cap > 0, "gather_trees: sizes must be positive");
    GCNPT_REQUIRE(!label || src_label, "gather_trees: labels wanted but the cache holds none");
    GCNPT_REQUIRE((rowT_ptr == nullptr) == (colT_idx == nullptr) && (rowT_ptr == nullptr) == (ellT == nullptr),
                  "gather_trees: rowT_ptr, colT_idx and ellT go together");
    GCNPT_REQUIRE(!rowT_ptr || (src_rowT_ptr && src_colT_idx && src_ellT), "gather_trees: transposed pattern wanted but the cache holds none");
    GCNPT_REQUIRE(!pool_mask || src_pool_mask, "gather_trees: pool mask wanted but the cache holds none");
    if ((long long)B * cap > 0x7fffffffLL) return fail(GCNPT_E_UNSUPPORTED, "gather_trees: B*cap overflows int32");
    const TreeArrays src{const_cast<int32_t*>(src_row_ptr), const_cast<int32_t*>(src_col_idx), const_cast<int32_t*>(src_label),
                         const_cast<int32_t*>(rowT_ptr ? src_rowT_ptr : nullptr), const_cast<int32_t*>(src_colT_idx),
                         const_cast<int32_t*>(src_ell), const_cast<int32_t*>(src_ellT), const_cast<uint8_t*>(src_pool_mask),
                         const_cast<int32_t*>(src_status)};
    const TreeArrays dst{row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, status};
    hipLaunchKernelGGL(gather_trees_kernel, dim3(B + pack_blocks), dim3(GATHER_THREADS), 0, (hipStream_t)stream, src, src_len, S, Ts, cap_s, idx,
                       B, T, cap, dst, pk, pk_dtype);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_gather_trees(void* stream, const int32_t* src_row_ptr, const int32_t* src_col_idx, const int32_t* src_label,
                                  const int32_t* src_rowT_ptr, const int32_t* src_colT_idx, const int32_t* src_ell,
                                  const int32_t* src_ellT, const uint8_t* src_pool_mask, const int32_t* src_status,
                                  const int32_t* src_len, int S, int Ts, int cap_s, const int64_t* idx, int B, int T, int cap,
                                  int32_t* row_ptr, int32_t* col_idx, int32_t* label, int32_t* rowT_ptr, int32_t* colT_idx,
                                  int32_t* ell, int32_t* ellT, uint8_t* pool_mask, int32_t* status) {
    return gather_impl(stream, src_row_ptr, src_col_idx, src_label, src_rowT_ptr, src_colT_idx, src_ell, src_ellT, src_pool_mask, src_status,
                       src_len, S, Ts, cap_s, idx, B, T, cap, row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, status,
                       PackParams{}, GCNPT_F32, 0);
}

extern "C" int gcnpt_gather_trees_pack(void* stream, const int32_t* src_row_ptr, const int32_t* src_col_idx, const int32_t* src_label,
                                       const int32_t* src_rowT_ptr, const int32_t* src_colT_idx, const int32_t* src_ell,
                                       const int32_t* src_ellT, const uint8_t* src_pool_mask, const int32_t* src_status,
                                       const int32_t* src_len, int S, int Ts, int cap_s, const int64_t* idx, int B, int T, int cap,
                                       int32_t* row_ptr, int32_t* col_idx, int32_t* label, int32_t* rowT_ptr, int32_t* colT_idx,
                                       int32_t* ell, int32_t* ellT, uint8_t* pool_mask, int32_t* status, int n_layers,
                                       const float* const* W, const int* H, const int* Din, int dtype, void* const* w_fwd,
                                       void* const* w_bwd) {
    PackParams pk;
    const int rc = fill_pack_params(pk, n_layers, W, H, Din, dtype, w_fwd, w_bwd);
    if (rc != GCNPT_OK) return rc;
    return gather_impl(stream, src_row_ptr, src_col_idx, src_label, src_rowT_ptr, src_colT_idx, src_ell, src_ellT, src_pool_mask, src_status,
                       src_len, S, Ts, cap_s, idx, B, T, cap, row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, status,
                       pk, dtype, pack_side_blocks(pk, GATHER_THREADS));
}

extern "C" int gcnpt_gather_trees_packed(void* stream, const int32_t* src_row_ptr, const int32_t* src_col_idx, const int32_t* src_label,
                                        const int32_t* src_rowT_ptr, const int32_t* src_colT_idx, const int32_t* src_ell,
                                        const int32_t* src_ellT, const uint8_t* src_pool_mask, const int32_t* src_status,
                                        const int32_t* src_len, int S, int Ts, int cap_s, const int64_t* idx, int B, int T,
                                        int32_t* cu_seqlens, int32_t* row_ptr, int32_t* col_idx, int32_t* label, int32_t* rowT_ptr,
                                        int32_t* colT_idx, int32_t* ell, int32_t* ellT, uint8_t* pool_mask, int32_t* row_sent, int n_rows,
                                        int nnz_cap, int32_t* status, int32_t* sent_status, uint8_t* pool_mask_padded, int n_layers,
                                        const float* const* W, const int* H, const int* Din, int dtype, void* const* w_fwd,
                                        void* const* w_bwd) {
    GCNPT_REQUIRE(src_row_ptr && src_col_idx && src_ell && src_status && src_len && src_pool_mask && idx, "gather_trees_packed: null cache pointer");
    GCNPT_REQUIRE(cu_seqlens && row_ptr && col_idx && ell && pool_mask && row_sent && status && sent_status, "gather_trees_packed: null output pointer");
    GCNPT_REQUIRE(S > 0 && Ts > 0 && cap_s > 0 && B > 0 && T > 0 && n_rows > 0 && nnz_cap > 0, "gather_trees_packed: sizes must be positive");
    GCNPT_REQUIRE(!label || src_label, "gather_trees_packed: labels wanted but the cache holds none");
    GCNPT_REQUIRE((rowT_ptr == nullptr) == (colT_idx == nullptr) && (rowT_ptr == nullptr) == (ellT == nullptr),
                  "gather_trees_packed: rowT_ptr, colT_idx and ellT go together");
    GCNPT_REQUIRE(!rowT_ptr || (src_rowT_ptr && src_colT_idx && src_ellT), "gather_trees_packed: transposed pattern wanted but the cache holds none");
    PackParams pk{};
    int blocks = 0;
    if (n_layers > 0) {
        const int rc = fill_pack_params(pk, n_layers, W, H, Din, dtype, w_fwd, w_bwd);
        if (rc != GCNPT_OK) return rc;
        blocks = pack_side_blocks(pk, GATHER_THREADS);
    }
    const TreeArrays src{const_cast<int32_t*>(src_row_ptr), const_cast<int32_t*>(src_col_idx), const_cast<int32_t*>(src_label),
                         const_cast<int32_t*>(rowT_ptr ? src_rowT_ptr : nullptr), const_cast<int32_t*>(src_colT_idx),
                         const_cast<int32_t*>(src_ell), const_cast<int32_t*>(src_ellT), const_cast<uint8_t*>(src_pool_mask),
                         const_cast<int32_t*>(src_status)};
    const GatherPackedDst dst{cu_seqlens, row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, pool_mask_padded, row_sent, status,
                              sent_status, n_rows, nnz_cap};
    hipLaunchKernelGGL(gather_trees_packed_kernel, dim3(B + blocks), dim3(GATHER_THREADS), 0, (hipStream_t)stream, src, src_len, S, Ts, cap_s, idx, B,
                       T, dst, pk, dtype);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

extern "C" int gcnpt_compact_trees(void* stream, const int32_t* src_row_ptr, const int32_t* src_col_idx, const int32_t* src_label,
                                   const int32_t* src_rowT_ptr, const int32_t* src_colT_idx, const int32_t* src_ell,
                                   const int32_t* src_ellT, const uint8_t* src_pool_mask, const int32_t* src_status, int B, int T,
                                   int cap, int Tc, int cap_c, int32_t* row_ptr, int32_t* col_idx, int32_t* label,
                                   int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell, int32_t* ellT, uint8_t* pool_mask,
                                   int32_t* status, int64_t* tok, int32_t* kept) {
    GCNPT_REQUIRE(src_row_ptr && src_col_idx && src_ell && src_pool_mask && src_status, "compact_trees: null source pointer");
    GCNPT_REQUIRE(row_ptr && col_idx && ell && pool_mask && status && tok && kept, "compact_trees: null output pointer");
    GCNPT_REQUIRE(B > 0 && T > 0 && cap > 0 && Tc > 0 && cap_c > 0, "compact_trees: sizes must be positive");
    GCNPT_REQUIRE(!label || src_label, "compact_trees: labels wanted but the source holds none");
    GCNPT_REQUIRE((rowT_ptr == nullptr) == (colT_idx == nullptr) && (rowT_ptr == nullptr) == (ellT == nullptr),
                  "compact_trees: rowT_ptr, colT_idx and ellT go together");
    GCNPT_REQUIRE(!rowT_ptr || (src_rowT_ptr && src_colT_idx && src_ellT), "compact_trees: transposed pattern wanted but the source holds none");
    if ((long long)B * cap_c > 0x7fffffffLL) return fail(GCNPT_E_UNSUPPORTED, "compact_trees: B*cap_c overflows int32");
    const size_t lds = sizeof(int) * (size_t)T;
    if (lds > 60 * 1024) return fail(GCNPT_E_UNSUPPORTED, "compact_trees: T=%d exceeds the %d tokens a sentence may have", T, 60 * 1024 / 4);
    const TreeArrays src{const_cast<int32_t*>(src_row_ptr), const_cast<int32_t*>(src_col_idx), const_cast<int32_t*>(src_label),
                         const_cast<int32_t*>(rowT_ptr ? src_rowT_ptr : nullptr), const_cast<int32_t*>(src_colT_idx),
                         const_cast<int32_t*>(src_ell), const_cast<int32_t*>(src_ellT), const_cast<uint8_t*>(src_pool_mask),
                         const_cast<int32_t*>(src_status)};
    const TreeArrays dst{row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, status};
    hipStream_t s = (hipStream_t)stream;
    GCNPT_HIP_CHECK(hipMemsetAsync(status + B, 0, sizeof(int32_t), s));         // status[B] = most kept tokens in a sentence (atomicMax)
    hipLaunchKernelGGL(compact_trees_kernel, dim3(B), dim3(COMPACT_THREADS), lds, s, src, B, T, cap, Tc, cap_c, dst, tok, kept);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}
