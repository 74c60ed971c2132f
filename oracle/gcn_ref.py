"""
oracle/gcn_ref.py  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy CPU restatement of the reference's GCN layer loop (`adj_type == 'regular'`) and of
its autograd, used only as the checker by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py.  Nothing under gcn-over-pruned-trees_amd/ imports it.

Parity pin: tests/golden/layers_*.npz, generated from the live reference
(/root/reference, torch CPU) by tests/golden/make_golden.py; checked in
tests/test_oracle_golden.py.

Reference lines followed (all model/gcn.py):
  260   adj_matrix = where(adj != 0, 1, 0)
  261   denom = adj_matrix.sum(2) + 1
  262   mask  = (adj_matrix.sum(2) + adj_matrix.sum(1)).eq(0)
  264-265 no_adj ablation zeroes adj_matrix AFTER denom/mask were taken
  269   Ax  = adj_matrix.bmm(gcn_inputs)
  270   AxW = W[l](Ax)
  271   AxW = AxW + W[l](gcn_inputs)          (bias therefore enters twice)
  390   AxW = AxW / denom
  392   gAxW = relu(AxW)
  393   dropout on every layer but the last
  473-483 pool()
"""
import numpy as np

INFINITY_NUMBER = 1e12  # utils/constant.py:35


def adjacency_prep(adj):
    """gcn.py:260-262.  adj: float32 [B,T,T] with deprel labels -> (A, denom, mask)."""
    A = (adj != 0).astype(np.float32)
    row = A.sum(2)
    denom = (row + 1.0)[..., None].astype(np.float32)
    mask = ((row + A.sum(1)) == 0)[..., None]
    return A, denom, mask


def gcn_forward(adj, x, weights, biases, drop_masks=None, drop_p=0.0, no_adj=False, return_saved=False):
    """
    Layer loop of GCN.forward (gcn.py:258-395) in float32.

    adj      float32 [B,T,T]   labelled adjacency from tree_to_adj
    x        float32 [B,T,Din] gcn_inputs (embeddings / BiLSTM output)
    weights  list of [H,Din_l] (nn.Linear layout, gcn.py:176), biases list of [H]
    drop_masks  optional list (len L-1) of {0,1} arrays [B,T,H]; the kept values are scaled by
                1/(1-drop_p) exactly as nn.Dropout does (gcn.py:393).  None = eval mode / p = 0.
    returns (h_L [B,T,H], mask bool [B,T,1])
    """
    A, denom, mask = adjacency_prep(adj)
    if no_adj:
        A = np.zeros_like(A)
    h = np.asarray(x, dtype=np.float32)
    saved = []
    L = len(weights)
    for l in range(L):
        W = np.asarray(weights[l], dtype=np.float32)
        b = np.asarray(biases[l], dtype=np.float32)
        Ax = np.matmul(A, h)                       # gcn.py:269
        AxW = np.matmul(Ax, W.T) + b               # gcn.py:270
        AxW = AxW + (np.matmul(h, W.T) + b)        # gcn.py:271
        AxW = AxW / denom                          # gcn.py:390
        g = np.maximum(AxW, 0.0)                   # gcn.py:392
        if l < L - 1 and drop_masks is not None and drop_p > 0.0:
            scale = np.float32(1.0 / (1.0 - drop_p))
            out = g * drop_masks[l].astype(np.float32) * scale   # gcn.py:393
        else:
            out = g
        saved.append((h, Ax, out))
        h = out.astype(np.float32)
    if return_saved:
        return h, mask, (A, denom, saved)
    return h, mask


def gcn_backward(adj, x, weights, biases, gy, drop_masks=None, drop_p=0.0, no_adj=False, acts=None):
    """
    What torch autograd produces for gcn_forward:
      dZ = dY * 1[out != 0] * dropscale / denom ;  dS = dZ W ;  dh = A^T dS + dS
      dW = dZ^T (A h + h) ;  db = 2 * sum dZ
    returns (dx, [dW_l], [db_l])

    acts: optional list of the L layer OUTPUTS to differentiate through instead of recomputing them
    (layer l then reads acts[l-1] as its input and takes its relu/dropout mask from acts[l] > 0).
    Used to check a reduced-precision backward on its own: relu' is a step function, so a forward that
    differs in the last bit near zero legitimately flips whole gradient terms.
    """
    h_L, _, (A, denom, saved) = gcn_forward(adj, x, weights, biases, drop_masks, drop_p, no_adj, True)
    if acts is not None:
        ins = [np.asarray(x, dtype=np.float32)] + [np.asarray(a, dtype=np.float32) for a in acts[:-1]]
        saved = [(ins[l], np.matmul(A, ins[l]), np.asarray(acts[l], dtype=np.float32)) for l in range(len(weights))]
    L = len(weights)
    g = np.asarray(gy, dtype=np.float32)
    dWs, dbs = [None] * L, [None] * L
    At = np.transpose(A, (0, 2, 1))
    for l in reversed(range(L)):
        h_in, Ax, out = saved[l]
        W = np.asarray(weights[l], dtype=np.float32)
        scale = np.float32(1.0)
        if l < L - 1 and drop_masks is not None and drop_p > 0.0:
            scale = np.float32(1.0 / (1.0 - drop_p))
        dZ = g * (out > 0) * scale / denom          # relu', dropout and /denom in one factor
        S = Ax + h_in
        H, Din = W.shape
        dWs[l] = np.matmul(dZ.reshape(-1, H).T, S.reshape(-1, Din)).astype(np.float32)
        dbs[l] = (2.0 * dZ.reshape(-1, H).sum(0)).astype(np.float32)
        dS = np.matmul(dZ, W)
        g = (np.matmul(At, dS) + dS).astype(np.float32)
    return g, dWs, dbs


def pool(h, mask, type="max"):
    """gcn.py:473-483."""
    if type == "max":
        return np.where(mask, np.float32(-INFINITY_NUMBER), h).max(1)
    hz = np.where(mask, np.float32(0), h)
    if type == "avg":
        return hz.sum(1) / (mask.shape[1] - mask.astype(np.float32).sum(1))
    return hz.sum(1)


# ----------------------------------------------------------------------------------------------
# bf16-storage variant: what the HIP path computes when `dtype = bf16` (bf16 operands and stored
# activations, fp32 accumulation).  Used for the tight comparison; the fp32 functions above are
# the loose one.  Not part of the reference -- the reference is fp32 only.
# ----------------------------------------------------------------------------------------------
def round_bf16(a):
    """round-to-nearest-even float32 -> bfloat16 -> float32 (finite inputs)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def gcn_forward_bf16(adj, x, weights, biases, drop_masks=None, drop_p=0.0):
    A, denom, mask = adjacency_prep(adj)
    h = round_bf16(x)
    L = len(weights)
    acts = [h]
    for l in range(L):
        W = round_bf16(weights[l])
        b = np.asarray(biases[l], dtype=np.float32)
        S = round_bf16(np.matmul(A, h) + h)                      # gather sum in fp32, staged as bf16
        Z = np.matmul(S.astype(np.float64), W.T.astype(np.float64)).astype(np.float32) + 2.0 * b
        g = np.maximum(Z / denom, 0.0).astype(np.float32)
        if l < L - 1 and drop_masks is not None and drop_p > 0.0:
            g = g * drop_masks[l].astype(np.float32) * np.float32(1.0 / (1.0 - drop_p))
        h = round_bf16(g)
        acts.append(h)
    return h, mask, acts


# ----------------------------------------------------------------------------------------------
# adj_type == 'diagonal_deprel' (SURVEY.md 8f row N2): reference model/gcn.py:153-155, 255-257, 272-294.
# No per-layer weight matrix: a Linear "preprocessor" in front, then every layer scales each neighbour
# row ELEMENT-WISE by the embedding of a dependency relation:
#   forward edges  (0 < adj[r,c] < 42):   E[deprel[c]]      * h[c]
#   reverse edges  (42 < adj[r,c] < 84):  E[deprel[c] + 42] * h[c]     (the COLUMN token's relation, as the reference does)
#   self           E[84] * h[r]           (the diagonal 84 of adj is in neither range)
# ----------------------------------------------------------------------------------------------
DEPREL_FORWARD_BOUND, DEPREL_REVERSE_BOUND, SELF_LOOP_INDEX = 42, 84, 84   # utils/constant.py:14-17


def diag_forward(adj, x, deprel, Wp, bp, E, layers, drop_masks=None, drop_p=0.0, return_saved=False):
    A, denom, mask = adjacency_prep(adj)
    F = ((adj > 0) & (adj < DEPREL_FORWARD_BOUND)).astype(np.float32)                        # gcn.py:276-279
    R = ((adj > DEPREL_FORWARD_BOUND) & (adj < DEPREL_REVERSE_BOUND)).astype(np.float32)     # gcn.py:281-285
    E = np.asarray(E, dtype=np.float32)
    fe, re, se = E[deprel], E[deprel + DEPREL_FORWARD_BOUND], E[SELF_LOOP_INDEX]             # gcn.py:274, 287, 289-292
    h = (np.matmul(np.asarray(x, np.float32), np.asarray(Wp, np.float32).T) + np.asarray(bp, np.float32)).astype(np.float32)   # gcn.py:257
    saved = []
    for l in range(layers):
        z = np.matmul(F, fe * h) + np.matmul(R, re * h) + h * se                              # gcn.py:280, 288, 293-294
        g = np.maximum(z / denom, 0.0).astype(np.float32)                                     # gcn.py:390-392
        if l < layers - 1 and drop_masks is not None and drop_p > 0.0:
            g = g * drop_masks[l].astype(np.float32) * np.float32(1.0 / (1.0 - drop_p))
        saved.append((h, g))
        h = g
    if return_saved:
        return h, mask, (F, R, fe, re, se, denom, saved)
    return h, mask


def diag_backward(adj, x, deprel, Wp, bp, E, layers, gy, drop_masks=None, drop_p=0.0, acts=None):
    """returns (dx, dWp, dbp, dE); dE[0] = 0 as nn.Embedding(padding_idx=0) does (gcn.py:56)."""
    _, _, (F, R, fe, re, se, denom, saved) = diag_forward(adj, x, deprel, Wp, bp, E, layers, drop_masks, drop_p, True)
    if acts is not None:
        ins = [saved[0][0]] + [np.asarray(a, np.float32) for a in acts[:-1]]
        saved = [(ins[l], np.asarray(acts[l], np.float32)) for l in range(layers)]
    E = np.asarray(E, dtype=np.float32)
    dE = np.zeros_like(E, dtype=np.float64)
    g = np.asarray(gy, dtype=np.float32)
    Ft, Rt = np.transpose(F, (0, 2, 1)), np.transpose(R, (0, 2, 1))
    for l in reversed(range(layers)):
        h_in, out = saved[l]
        scale = np.float32(1.0 / (1.0 - drop_p)) if (l < layers - 1 and drop_masks is not None and drop_p > 0.0) else np.float32(1.0)
        dZ = g * (out > 0) * scale / denom
        uf, ur = np.matmul(Ft, dZ), np.matmul(Rt, dZ)                  # gradient wrt (fe*h) and (re*h)
        np.add.at(dE, deprel, (uf * h_in).astype(np.float64))
        np.add.at(dE, deprel + DEPREL_FORWARD_BOUND, (ur * h_in).astype(np.float64))
        dE[SELF_LOOP_INDEX] += (dZ * h_in).reshape(-1, E.shape[1]).sum(0)
        g = (uf * fe + ur * re + dZ * se).astype(np.float32)
    dE[0] = 0.0
    Wp = np.asarray(Wp, np.float32)
    din = Wp.shape[1]
    dWp = np.matmul(g.reshape(-1, Wp.shape[0]).T, np.asarray(x, np.float32).reshape(-1, din)).astype(np.float32)
    dbp = g.reshape(-1, Wp.shape[0]).sum(0).astype(np.float32)
    dx = np.matmul(g, Wp).astype(np.float32)
    return dx, dWp, dbp, dE.astype(np.float32)


# ----------------------------------------------------------------------------------------------
# adj_type == 'full_deprel' (SURVEY.md 8f row N3): reference model/gcn.py:156-167 (one Linear reused by every layer),
# 296-388 (layer body), 400-434 (traverse_deprel / traverse_self_loop).  Eval-mode semantics: edge dropout
# (maybe_drop_edges, 436-449) and relation forgetting (maybe_forget_deprels, 451-470) are training-time RNG.
#   trav(x, e)[n] = sum_d e[n,d] * (x[n] @ W3[d] + b3[d]),   W3 = W.weight.reshape(D, Tin, H), b3 = W.bias.reshape(D, H)
#   z = F trav(x, E[deprel]) + [not directed] R trav(x, E[deprel+42]) + [self_loop] trav(x, E[84]);  out = relu(z / denom)
#   layers l >= deprel_max_depth use all-ones relation vectors (gcn.py:323-324, 355-356, 371-374)
# ----------------------------------------------------------------------------------------------
def _full_setup(adj, deprel, W, b, E, D, layer, max_depth, Tin):
    F = ((adj > 0) & (adj < DEPREL_FORWARD_BOUND)).astype(np.float32)
    R = ((adj > DEPREL_FORWARD_BOUND) & (adj < DEPREL_REVERSE_BOUND)).astype(np.float32)
    W3 = np.asarray(W, np.float32).reshape(D, Tin, -1)                 # gcn.py:301 (a reinterpretation, not a transpose)
    b3 = np.asarray(b, np.float32).reshape(D, -1)                      # gcn.py:303
    E = np.asarray(E, np.float32)
    plain = layer >= max_depth
    ones = np.ones(deprel.shape + (D,), np.float32)
    fe = ones if plain else E[deprel]
    re = ones if plain else E[deprel + DEPREL_FORWARD_BOUND]
    se = np.ones((D,), np.float32) if plain else E[SELF_LOOP_INDEX]
    return F, R, W3, b3, fe, re, se, plain


def _trav(x, e, W3, b3):
    return np.einsum("bnd,bnt,dth->bnh", e, x, W3, optimize=True) + np.matmul(e, b3)          # gcn.py:408-414


def full_forward(adj, x, deprel, W, b, E, layers, max_depth=2, directed=False, self_loop=True, return_saved=False):
    _, denom, mask = adjacency_prep(adj)
    D = np.asarray(E).shape[1]
    h = np.asarray(x, np.float32)
    saved = []
    for l in range(layers):
        F, R, W3, b3, fe, re, se, _ = _full_setup(adj, deprel, W, b, E, D, l, max_depth, h.shape[-1])
        z = np.matmul(F, _trav(h, fe, W3, b3))
        if not directed:
            z = z + np.matmul(R, _trav(h, re, W3, b3))
        if self_loop:
            z = z + np.matmul(h, np.einsum("d,dth->th", se, W3)) + se @ b3                    # gcn.py:426-433
        g = np.maximum(z / denom, 0.0).astype(np.float32)
        saved.append((h, g))
        h = g
    return (h, mask, saved) if return_saved else (h, mask)


def full_backward(adj, x, deprel, W, b, E, layers, gy, max_depth=2, directed=False, self_loop=True):
    """returns (dx, dW, db, dE) with dW/db in the nn.Linear layout and dE[0] = 0 (padding_idx)."""
    _, denom, _ = adjacency_prep(adj)
    D = np.asarray(E).shape[1]
    _, _, saved = full_forward(adj, x, deprel, W, b, E, layers, max_depth, directed, self_loop, True)
    dW3 = np.zeros((D, saved[0][0].shape[-1], saved[0][1].shape[-1]), np.float64)
    db3 = np.zeros((D, dW3.shape[2]), np.float64)
    dE = np.zeros(np.asarray(E).shape, np.float64)
    g = np.asarray(gy, np.float32)
    for l in reversed(range(layers)):
        h, out = saved[l]
        F, R, W3, b3, fe, re, se, plain = _full_setup(adj, deprel, W, b, E, D, l, max_depth, h.shape[-1])
        dz = g * (out > 0) / denom
        dh = np.zeros_like(h)
        for A, e, ids in ((F, fe, deprel), (None if directed else R, re, deprel + DEPREL_FORWARD_BOUND)):
            if A is None:
                continue
            dy = np.matmul(np.transpose(A, (0, 2, 1)), dz)
            dh += np.einsum("bnh,bnd,dth->bnt", dy, e, W3, optimize=True)
            dW3 += np.einsum("bnd,bnt,bnh->dth", e, h, dy, optimize=True)
            db3 += np.einsum("bnd,bnh->dh", e, dy)
            if not plain:
                de = np.einsum("bnh,bnt,dth->bnd", dy, h, W3, optimize=True) + np.matmul(dy, b3.T)
                np.add.at(dE, ids, de.astype(np.float64))
        if self_loop:
            Wsl = np.einsum("d,dth->th", se, W3)
            dh += np.matmul(dz, Wsl.T)
            dWsl = np.einsum("bnt,bnh->th", h, dz)
            dbsl = dz.reshape(-1, dz.shape[-1]).sum(0)
            dW3 += se[:, None, None] * dWsl[None]
            db3 += se[:, None] * dbsl[None]
            if not plain:
                dE[SELF_LOOP_INDEX] += np.einsum("th,dth->d", dWsl, W3) + b3 @ dbsl
        g = dh.astype(np.float32)
    dE[0] = 0.0
    return g, dW3.reshape(np.asarray(W).shape).astype(np.float32), db3.reshape(-1).astype(np.float32), dE.astype(np.float32)
