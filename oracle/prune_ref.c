/*
 * oracle/prune_ref.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's pruned-tree
 * adjacency builder.  It exists so that tests, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg can check / time the HIP path against the
 * reference's algorithm on a box where /root/reference does not exist.
 * Nothing under gcn-over-pruned-trees_amd/ may import, link or call it.
 *
 * Parity pin: tests/golden/trees_*.npz (generated from the live reference by
 * tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.
 *
 * It follows the reference statement by statement (python sets become byte
 * flags, python lists become arrays), citing model/tree.py line numbers:
 *   head_to_tree   model/tree.py:58-165
 *   tree_to_adj    model/tree.py:167-204   (directed=False, self_loop=True as
 *                                           called from model/gcn.py:106)
 * Constants: DEPREL_FORWARD_BOUND = 42 (utils/constant.py:14),
 *            DEPREL_TO_ID['self_loop'] = 84 (utils/constant.py:12,29).
 *
 * Behaviours of the reference that are Python exceptions or hangs become
 * negative return codes here (the same codes the product's C-ABI reports):
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GCNPT_OK 0
#define GCNPT_E_PRUNE_NEGATIVE -2 /* tree.py:67-79 then tree.py:194 AttributeError: prune<0 trees have no .head */
#define GCNPT_E_NO_SUBJECT -3     /* tree.py:109 / 113: cas is None when no token has subj_pos==0 */
#define GCNPT_E_NO_LCA -4         /* tree.py:112-124: empty common-ancestor set (forest) -> UnboundLocalError */
#define GCNPT_E_CYCLE -5          /* tree.py:91-94: head cycle never terminates in the reference */
#define GCNPT_E_BAD_HEAD -6       /* tree.py:94: head[h-1] IndexError when a head points past len */
#define GCNPT_E_ASSERT -7         /* tree.py:159 assert nodes[h-1] is not None */

#define FWD_BOUND 42.0f
#define SELF_LOOP_ID 84.0f
#define DIST_INF 10000

/* walk from token t to the root, marking `anc` and filling chain[] (tree.py:88-94 / 102-108) */
static int walk_chain(const int64_t* head, int len, int t, uint8_t* anc, int* chain, int* n_chain) {
    int n = 0;
    int64_t h = head[t];
    chain[n++] = t;
    while (h > 0) {
        if (h - 1 >= len) return GCNPT_E_BAD_HEAD;
        if (n > len) return GCNPT_E_CYCLE;
        chain[n++] = (int)(h - 1);
        anc[h - 1] = 1;
        h = head[h - 1];
    }
    *n_chain = n;
    return GCNPT_OK;
}

/*
 * One sentence.  adj is a caller-zeroed T*T float32 row-major matrix (tree.py:171).
 * kept[i] (len T, may be NULL) = 1 for tokens that became Tree nodes; *root_out = LCA.
 */
int gcnpt_oracle_head_to_adj(const int64_t* head, const int64_t* subj_pos, const int64_t* obj_pos,
                             const int64_t* deprel, int len, int T, int prune, float* adj,
                             uint8_t* kept, int* root_out) {
    if (prune < 0) return GCNPT_E_PRUNE_NEGATIVE;
    if (len <= 0 || len > T) return GCNPT_E_BAD_HEAD;
    int rc = GCNPT_OK;
    uint8_t* subj_anc = (uint8_t*)calloc((size_t)len, 1);
    uint8_t* obj_anc = (uint8_t*)calloc((size_t)len, 1);
    uint8_t* cas = (uint8_t*)calloc((size_t)len, 1);
    uint8_t* tmp = (uint8_t*)calloc((size_t)len, 1);
    uint8_t* path = (uint8_t*)calloc((size_t)len, 1);
    uint8_t* node = (uint8_t*)calloc((size_t)len, 1);
    int* chain = (int*)malloc(sizeof(int) * (size_t)(len + 2));
    int* dist = (int*)malloc(sizeof(int) * (size_t)len);
    int* stack = (int*)malloc(sizeof(int) * (size_t)(len + 2));
    int* queue = (int*)malloc(sizeof(int) * (size_t)len);
    int* child_count = (int*)calloc((size_t)len, sizeof(int));
    int have_cas = 0, n_cas = 0, lca = -1, n_chain = 0;

    /* tree.py:82-83 entity token lists; tree.py:87,101 ancestors start as the tokens themselves */
    for (int s = 0; s < len; ++s) {
        if (subj_pos[s] != 0) continue;
        subj_anc[s] = 1;
        if ((rc = walk_chain(head, len, s, subj_anc, chain, &n_chain)) != GCNPT_OK) goto done;
        memset(tmp, 0, (size_t)len);
        for (int j = 0; j < n_chain; ++j) tmp[chain[j]] = 1;
        if (!have_cas) { /* tree.py:96-97 */
            memcpy(cas, tmp, (size_t)len);
            have_cas = 1;
        } else { /* tree.py:98-99 */
            for (int j = 0; j < len; ++j) cas[j] &= tmp[j];
        }
    }
    for (int o = 0; o < len; ++o) {
        if (obj_pos[o] != 0) continue;
        obj_anc[o] = 1;
        if ((rc = walk_chain(head, len, o, obj_anc, chain, &n_chain)) != GCNPT_OK) goto done;
        if (!have_cas) { rc = GCNPT_E_NO_SUBJECT; goto done; } /* tree.py:109 on None */
        memset(tmp, 0, (size_t)len);
        for (int j = 0; j < n_chain; ++j) tmp[chain[j]] = 1;
        for (int j = 0; j < len; ++j) cas[j] &= tmp[j];
    }
    if (!have_cas) { rc = GCNPT_E_NO_SUBJECT; goto done; } /* tree.py:113 len(None) */

    /* tree.py:112-124 lowest common ancestor */
    for (int j = 0; j < len; ++j) n_cas += cas[j];
    if (n_cas == 1) {
        for (int j = 0; j < len; ++j) if (cas[j]) lca = j;
    } else {
        for (int ca = 0; ca < len; ++ca)
            if (cas[ca] && head[ca] > 0 && head[ca] - 1 < len && cas[head[ca] - 1]) child_count[head[ca] - 1] += 1;
        for (int ca = 0; ca < len; ++ca)
            if (cas[ca] && child_count[ca] == 0) { lca = ca; break; }
    }
    if (lca < 0) { rc = GCNPT_E_NO_LCA; goto done; }

    /* tree.py:126-127 */
    for (int j = 0; j < len; ++j) path[j] = (uint8_t)((subj_anc[j] | obj_anc[j]) & !cas[j]);
    path[lca] = 1;

    /* tree.py:130-144 distance to the path */
    for (int i = 0; i < len; ++i) dist[i] = path[i] ? 0 : -1;
    for (int i = 0; i < len; ++i) {
        if (dist[i] >= 0) continue;
        int n = 0;
        stack[n++] = i;
        while (stack[n - 1] >= 0 && !path[stack[n - 1]]) {
            if (n > len) { rc = GCNPT_E_CYCLE; goto done; }
            int64_t up = head[stack[n - 1]] - 1;
            if (up >= len) { rc = GCNPT_E_BAD_HEAD; goto done; }
            stack[n++] = (int)(up < 0 ? -1 : up);
        }
        if (stack[n - 1] >= 0) {
            for (int d = 0; d < n; ++d) dist[stack[n - 1 - d]] = d;
        } else {
            for (int j = 0; j < n; ++j)
                if (stack[j] >= 0 && dist[stack[j]] < 0) dist[stack[j]] = DIST_INF;
        }
    }

    /* tree.py:146-160 which tokens become nodes; children hang off head-1 (checked only) */
    for (int i = 0; i < len; ++i) node[i] = (uint8_t)(dist[i] <= prune);
    for (int i = 0; i < len; ++i) {
        if (!node[i]) continue;
        if (head[i] > 0 && i != lca && !node[head[i] - 1]) { rc = GCNPT_E_ASSERT; goto done; }
    }

    /* tree.py:173-196 breadth-first walk from the root; children are visited in the order
       add_child appended them = increasing token index (tree.py:149-160) */
    {
        int qh = 0, qt = 0;
        queue[qt++] = lca;
        while (qh < qt) {
            int t = queue[qh++];
            for (int c = 0; c < len; ++c) {
                if (!node[c] || c == lca || head[c] <= 0 || head[c] - 1 != t) continue;
                adj[(size_t)t * T + c] = (float)deprel[c];             /* tree.py:184 */
                adj[(size_t)c * T + t] = (float)deprel[c] + FWD_BOUND; /* tree.py:188 */
                adj[(size_t)t * T + t] = SELF_LOOP_ID;                 /* tree.py:191 */
                adj[(size_t)c * T + c] = SELF_LOOP_ID;                 /* tree.py:192 */
                queue[qt++] = c;
            }
        }
    }
    if (kept) {
        memset(kept, 0, (size_t)T);
        for (int i = 0; i < len; ++i) kept[i] = node[i];
    }
    if (root_out) *root_out = lca;

done:
    free(subj_anc); free(obj_anc); free(cas); free(tmp); free(path); free(node);
    free(chain); free(dist); free(stack); free(queue); free(child_count);
    return rc;
}

/*
 * A batch, as model/gcn.py:96-108 drives it: len[b] = number of non-pad tokens, every
 * sentence padded to T, adj = float32 [B,T,T] (zeroed here).  status[b] receives the
 * per-sentence return code; the function returns the first non-zero one.
 */
int gcnpt_oracle_batch_adj(const int64_t* head, const int64_t* subj_pos, const int64_t* obj_pos,
                           const int64_t* deprel, const int32_t* len, int B, int T, int prune,
                           float* adj, uint8_t* kept, int32_t* root, int32_t* status) {
    int first = GCNPT_OK;
    memset(adj, 0, sizeof(float) * (size_t)B * T * T);
    for (int b = 0; b < B; ++b) {
        const size_t o = (size_t)b * T;
        int r = -1;
        int rc = gcnpt_oracle_head_to_adj(head + o, subj_pos + o, obj_pos + o, deprel + o, len[b], T, prune,
                                          adj + o * T, kept ? kept + o : NULL, &r);
        if (status) status[b] = rc;
        if (root) root[b] = r;
        if (rc != GCNPT_OK && first == GCNPT_OK) first = rc;
    }
    return first;
}
