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

namespace gcnpt {

constexpr int PRUNE_THREADS = 64;    // ONE wave per sentence: phases are separated by wave-local LDS ordering only
constexpr int ADJ_THREADS = 256;

enum : int {
    F_SUBJ = 1, F_OBJ = 2, F_FWD_NZ = 4, F_REV_NZ = 8, F_CA = 16, F_CA_HASCHILD = 32, F_PATH = 64, F_HASEDGE = 128
};
enum : int { K_KEEP = 1, K_CHILD = 2 };
enum : int { ERR_CHAIN_BADHEAD = 1, ERR_CHAIN_CYCLE = 2, ERR_BADHEAD = 4, ERR_CYCLE = 8, ERR_ASSERT = 16 };

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

__global__ __launch_bounds__(PRUNE_THREADS) void prune_to_csr_kernel(
    const int64_t* __restrict__ head, const int64_t* __restrict__ subj_pos, const int64_t* __restrict__ obj_pos,
    const int64_t* __restrict__ deprel, const uint8_t* __restrict__ pad_mask, const int32_t* __restrict__ len_in,
    int B, int T, int prune_k, int cap, int32_t* __restrict__ row_ptr, int32_t* __restrict__ col_idx,
    int32_t* __restrict__ label, int32_t* __restrict__ rowT_ptr, int32_t* __restrict__ colT_idx,
    int32_t* __restrict__ ell, int32_t* __restrict__ ellT, uint8_t* __restrict__ pool_mask,
    int32_t* __restrict__ status) {
    extern __shared__ int smem[];
    int* par = smem;               // [T]   parent token (-1 root / none, -2 head points past the sentence)
    int* cnt = par + T;            // [T]   #entity chains through the token; later K_KEEP/K_CHILD bits
    int* flg = cnt + T;            // [T]   F_* bits
    int* deg = flg + T;            // [T+1] row degree -> row offsets
    int* degT = deg + T + 1;       // [T+1] column degree -> transposed row offsets
    int* klist = degT + T + 1;     // [T+1] tokens that carry an edge, ascending (compacted)
    int* lab = klist + T + 1;      // [T]   deprel id of the token
    __shared__ int s_err;

    const int b = blockIdx.x, lane = threadIdx.x;
    const size_t base = (size_t)b * T;
    if (lane == 0) s_err = 0;

    // ---- stage the parse (tree.py:60-63, 82-83): every load is unconditional and issued before the first use
    //      (a load behind `if (i < len)` would be a dependent round trip per token)
    int npad = 0;
    for (int i = lane; i < T; i += WAVE) {
        const int64_t h = head[base + i];
        const int64_t sp = subj_pos[base + i], op = obj_pos[base + i], d = deprel[base + i];
        const bool pad = pad_mask ? pad_mask[base + i] != 0 : false;
        npad += pad ? 1 : 0;
        int f = 0;
        if (sp == 0) f |= F_SUBJ;
        if (op == 0) f |= F_OBJ;
        if (d != 0) f |= F_FWD_NZ;                     // adj[p,c] = deprel[c]        survives `adj != 0`
        if (d + FWD_BOUND != 0) f |= F_REV_NZ;         // adj[c,p] = deprel[c] + 42
        par[i] = h > 0 ? (int)min(h - 1, (int64_t)0x3fffffff) : -1;   // range-checked against len below
        flg[i] = f; lab[i] = (int)d; cnt[i] = 0; deg[i] = 0; degT[i] = 0;
    }
    // sentence length = number of non-pad slots (gcn.py:96)
    const int len = pad_mask ? T - wave_sum(npad) : min(max(len_in[b], 0), T);
    if (lane == 0) atomicMax(&status[B], len);
    int nsubj = 0, nent = 0;
    for (int i = lane; i < T; i += WAVE) {
        int f = flg[i], p = par[i];
        if (i >= len) { f = 0; p = -1; }
        else if (p >= len) p = -2;
        nsubj += (f & F_SUBJ) ? 1 : 0;
        nent += ((f & F_SUBJ) ? 1 : 0) + ((f & F_OBJ) ? 1 : 0);
        flg[i] = f; par[i] = p;
    }
    nsubj = wave_sum(nsubj);
    nent = wave_sum(nent);
    __syncthreads();

    // ---- every entity token walks to the root, counting visits (tree.py:86-109)
    for (int i = lane; i < len; i += WAVE) {
        const int f = flg[i];
        const int w = ((f & F_SUBJ) ? 1 : 0) + ((f & F_OBJ) ? 1 : 0);
        if (!w) continue;
        int a = i, steps = 0;
        while (a >= 0) {
            atomicAdd(&cnt[a], w);
            a = par[a];
            if (++steps > len) { atomicOr(&s_err, ERR_CHAIN_CYCLE); a = -1; }
        }
        if (a == -2) atomicOr(&s_err, ERR_CHAIN_BADHEAD);
    }
    __syncthreads();
    int err = 0;
    if (s_err & ERR_CHAIN_BADHEAD) err = GCNPT_E_BAD_HEAD;
    else if (s_err & ERR_CHAIN_CYCLE) err = GCNPT_E_CYCLE;
    else if (nsubj == 0) err = GCNPT_E_NO_SUBJECT;

    // ---- common ancestors and the lowest of them (tree.py:112-124)
    int lca = 0x7fffffff;
    if (!err) {
        for (int i = lane; i < len; i += WAVE)
            if (cnt[i] == nent) flg[i] |= F_CA;
        __syncthreads();
        for (int i = lane; i < len; i += WAVE) {
            const int p = par[i];
            if ((flg[i] & F_CA) && p >= 0 && (flg[p] & F_CA)) atomicOr(&flg[p], F_CA_HASCHILD);
        }
        __syncthreads();
        for (int i = lane; i < len; i += WAVE)
            if ((flg[i] & (F_CA | F_CA_HASCHILD)) == F_CA) lca = min(lca, i);
        lca = wave_min(lca);
        if (lca == 0x7fffffff) err = GCNPT_E_NO_LCA;
    }

    // ---- path nodes, distance to the path, kept tokens (tree.py:126-147)
    if (!err) {
        for (int i = lane; i < len; i += WAVE)
            if ((cnt[i] > 0 && !(flg[i] & F_CA)) || i == lca) flg[i] |= F_PATH;
        __syncthreads();
        for (int i = lane; i < len; i += WAVE) {
            int a = i, d = 0;
            while (a >= 0 && !(flg[a] & F_PATH)) {
                a = par[a];
                if (++d > len) { atomicOr(&s_err, ERR_CYCLE); a = -1; }
            }
            if (a == -2) atomicOr(&s_err, ERR_BADHEAD);
            const bool keep = a >= 0 && d <= prune_k;
            const bool child = keep && i != lca && par[i] >= 0;
            cnt[i] = (keep ? K_KEEP : 0) | (child ? K_CHILD : 0);   // cnt is free from here on
        }
        __syncthreads();
        if (s_err & ERR_BADHEAD) err = GCNPT_E_BAD_HEAD;
        else if (s_err & ERR_CYCLE) err = GCNPT_E_CYCLE;
    }

    // ---- degrees of the labelled adjacency tree_to_adj would write (tree.py:182-192)
    if (!err) {
        for (int i = lane; i < len; i += WAVE) {
            if (!(cnt[i] & K_CHILD)) continue;
            const int p = par[i], f = flg[i];
            if (!(cnt[p] & K_KEEP)) atomicOr(&s_err, ERR_ASSERT);   // tree.py:159
            if (f & F_FWD_NZ) { atomicAdd(&deg[p], 1); atomicAdd(&degT[i], 1); }
            if (f & F_REV_NZ) { atomicAdd(&deg[i], 1); atomicAdd(&degT[p], 1); }
            atomicOr(&flg[p], F_HASEDGE);
            atomicOr(&flg[i], F_HASEDGE);
        }
        __syncthreads();
        if (s_err & ERR_ASSERT) err = GCNPT_E_ASSERT;
    }
    int n_edge_rows = 0;
    if (!err) {
        for (int i = lane; i < T; i += WAVE) {
            const int he = (i < len && (flg[i] & F_HASEDGE)) ? 1 : 0;
            deg[i] += he; degT[i] += he;                             // the 84 on the diagonal
            klist[i] = he;
            if (pool_mask) pool_mask[base + i] = (deg[i] + degT[i]) == 0;   // gcn.py:262
        }
        __syncthreads();
        const int tot = wave_exclusive_scan(deg, T, lane);
        const int totT = wave_exclusive_scan(degT, T, lane);
        n_edge_rows = wave_exclusive_scan(klist, T, lane);           // klist[i] = rank of token i among the edge rows
        if (lane == 0) { deg[T] = tot; degT[T] = totT; }
        __syncthreads();
        if (tot > cap || totT > cap) err = GCNPT_E_CAPACITY;
    }

    if (err) {   // the sentence contributes no edges; every row is empty and masked
        for (int i = lane; i <= T; i += WAVE) {
            row_ptr[(size_t)b * (T + 1) + i] = b * cap;
            if (rowT_ptr) rowT_ptr[(size_t)b * (T + 1) + i] = b * cap;
        }
        for (int i = lane; i < T; i += WAVE)
            if (pool_mask) pool_mask[base + i] = 1;
        for (int i = lane; i < T * 8; i += WAVE) {
            ell[base * 8 + i] = 0;
            if (ellT) ellT[base * 8 + i] = 0;
        }
        if (lane == 0) status[b] = err;
        return;
    }

    // ---- emit both patterns, columns ascending (same order a dense -> CSR conversion gives)
    for (int i = lane; i <= T; i += WAVE) {
        row_ptr[(size_t)b * (T + 1) + i] = b * cap + deg[i];
        if (rowT_ptr) rowT_ptr[(size_t)b * (T + 1) + i] = b * cap + degT[i];
    }
    // compact the rows that carry an edge (ascending): only they, and only columns among them, have entries.
    // klist[i] is the rank of token i among them; every such token writes itself at its rank.
    int* elist = lab + T;          // [T]
    for (int i = lane; i < len; i += WAVE)
        if (flg[i] & F_HASEDGE) elist[klist[i]] = i;
    __syncthreads();

    for (int r = lane; r < T; r += WAVE) {
        int hd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hdT[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ELL heads: count, first 7 columns
        const int rf = r < len ? flg[r] : 0;
        if (rf & F_HASEDGE) {
            const bool rchild = cnt[r] & K_CHILD;
            const int rp = par[r];
            const int o0 = b * cap + deg[r], oT0 = b * cap + degT[r];
            int o = o0, oT = oT0;
            auto put = [&](int j, int lb) {
                col_idx[o] = j; if (label) label[o] = lb;
                if (o - o0 < 7) hd[1 + o - o0] = j;
                ++o;
            };
            auto putT = [&](int j) {
                if (colT_idx) colT_idx[oT] = j;
                if (oT - oT0 < 7) hdT[1 + oT - oT0] = j;
                ++oT;
            };
            for (int q = 0; q < n_edge_rows; ++q) {
                const int j = elist[q];
                const bool child = (cnt[j] & K_CHILD) && par[j] == r;
                const int fj = flg[j];
                if (child) {
                    if (fj & F_FWD_NZ) put(j, lab[j]);
                    if (fj & F_REV_NZ) putT(j);
                } else if (j == r) {
                    put(r, SELF_LOOP_ID);
                    putT(r);
                } else if (rchild && j == rp) {
                    if (rf & F_REV_NZ) put(j, lab[r] + FWD_BOUND);
                    if (rf & F_FWD_NZ) putT(j);
                }
            }
            hd[0] = o - o0; hdT[0] = oT - oT0;
        }
        int4* e = reinterpret_cast<int4*>(ell + (base + r) * 8);
        e[0] = make_int4(hd[0], hd[1], hd[2], hd[3]);
        e[1] = make_int4(hd[4], hd[5], hd[6], hd[7]);
        if (ellT) {
            int4* eT = reinterpret_cast<int4*>(ellT + (base + r) * 8);
            eT[0] = make_int4(hdT[0], hdT[1], hdT[2], hdT[3]);
            eT[1] = make_int4(hdT[4], hdT[5], hdT[6], hdT[7]);
        }
    }
    if (lane == 0) status[b] = 0;
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

}  // namespace gcnpt

using namespace gcnpt;

extern "C" int gcnpt_prune_to_csr(void* stream, const int64_t* head, const int64_t* subj_pos, const int64_t* obj_pos,
                                  const int64_t* deprel, const uint8_t* pad_mask, const int32_t* len, int B, int T,
                                  int prune_k, int cap, int32_t* row_ptr, int32_t* col_idx, int32_t* label,
                                  int32_t* rowT_ptr, int32_t* colT_idx, int32_t* ell, int32_t* ellT, uint8_t* pool_mask,
                                  int32_t* status) {
    GCNPT_REQUIRE(head && subj_pos && obj_pos && deprel && (pad_mask || len), "prune_to_csr: null input pointer");
    GCNPT_REQUIRE(row_ptr && col_idx && ell && status, "prune_to_csr: null output pointer");
    GCNPT_REQUIRE((rowT_ptr == nullptr) == (ellT == nullptr), "prune_to_csr: rowT_ptr, colT_idx and ellT go together");
    GCNPT_REQUIRE(B > 0 && T > 0 && cap > 0, "prune_to_csr: B, T, cap must be positive (B=%d T=%d cap=%d)", B, T, cap);
    GCNPT_REQUIRE((rowT_ptr == nullptr) == (colT_idx == nullptr), "prune_to_csr: rowT_ptr and colT_idx go together");
    if (prune_k < 0)
        return fail(GCNPT_E_PRUNE_NEGATIVE, "prune_k=%d: the reference fork only works with prune_k >= 0 "
                    "(model/tree.py:194 reads Tree.head, which the unpruned branch never sets)", prune_k);
    if ((long long)B * cap > 0x7fffffffLL) return fail(GCNPT_E_UNSUPPORTED, "prune_to_csr: B*cap overflows int32");
    const size_t lds = sizeof(int) * ((size_t)8 * T + 4);
    if (lds > 150 * 1024) return fail(GCNPT_E_UNSUPPORTED, "prune_to_csr: T=%d needs %zu B of LDS", T, lds);
    hipStream_t s = (hipStream_t)stream;
    GCNPT_HIP_CHECK(hipMemsetAsync(status + B, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(prune_to_csr_kernel, dim3(B), dim3(PRUNE_THREADS), lds, s, head, subj_pos, obj_pos, deprel,
                       pad_mask, len, B, T, prune_k, cap, row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, status);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
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
