"""
Pruned dependency trees on the device -- host-side mirror of reference model/tree.py for the hot path.

The reference builds one Python `Tree` per sentence on the host (model/tree.py:58 head_to_tree) and
turns it into a dense float32 [T,T] matrix (model/tree.py:167 tree_to_adj), after copying six tensors
back from the GPU (model/gcn.py:96-110).  Here the whole batch is pruned by one HIP kernel launch
(csrc/tree_kernels.hip) straight from the loader tensors that already sit in HBM, and what comes out
is the CSR pattern of that matrix (`PrunedTrees`).  `PrunedTrees.to_dense()` gives the reference's
labelled adjacency back for callers that want it.

No CPU path exists: these functions raise if the tensors are not on a GPU or libgcnpt.so is missing.
"""
import torch

from .. import _lib

# what the reference raises for each per-sentence status code (informational, see TreeError)
REFERENCE_EXCEPTION = {
    _lib.E_PRUNE_NEGATIVE: "AttributeError (model/tree.py:194, Tree has no .head when prune < 0)",
    _lib.E_NO_SUBJECT: "AttributeError/TypeError (model/tree.py:109,113, no subject token)",
    _lib.E_NO_LCA: "UnboundLocalError (model/tree.py:124, entities under different roots)",
    _lib.E_CYCLE: "never returns (model/tree.py:91-94, head cycle)",
    _lib.E_BAD_HEAD: "IndexError (model/tree.py:94, head points past the sentence)",
    _lib.E_ASSERT: "AssertionError (model/tree.py:159)",
    _lib.E_CAPACITY: "n/a (adjacency capacity exceeded)",
    _lib.E_INVALID: "IndexError (sentence index outside the cached dataset)",
    _lib.E_LENGTH: "n/a (cached sentence longer than the batch is padded to)",
}


class TreeError(ValueError):
    """A sentence of the batch has no valid pruned tree; `.sentence`, `.code`, `.reference` say which and why."""

    def __init__(self, sentence, code):
        self.sentence, self.code = int(sentence), int(code)
        self.reference = REFERENCE_EXCEPTION.get(self.code, "unknown")
        super().__init__("sentence %d: pruned-tree build failed with code %d; the reference raises %s"
                         % (self.sentence, self.code, self.reference))


class PrunedTrees(object):
    """
    CSR pattern of the batch adjacency in HBM (layout: include/gcnpt.h).

    row_ptr  int32 [B*(T+1)]   col_idx / label  int32 [B*cap]   rowT_ptr / colT_idx: transposed pattern
    ell / ellT int32 [B*T*8]   ELL heads: entry count + first 7 columns of every row (what the layer kernels read first)
    pool_mask bool [B,T,1]     the `mask` GCN.forward returns (model/gcn.py:262)
    status   int32 [B+1]       per-sentence code, [B] = longest sentence seen
    """

    packed = False

    def __init__(self, B, T, cap, row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, status):
        self.B, self.T, self.cap = B, T, cap
        self.row_ptr, self.col_idx, self.label = row_ptr, col_idx, label
        self.rowT_ptr, self.colT_idx = rowT_ptr, colT_idx
        self.ell, self.ellT = ell, ellT
        self.pool_mask, self.status = pool_mask, status
        self._empty = None

    @property
    def device(self):
        return self.row_ptr.device

    def empty_ell(self):
        """ELL head of an adjacency with no entries (the `no_adj` ablation, model/gcn.py:264-265)."""
        if self._empty is None:
            self._empty = torch.zeros_like(self.ell)
        return self._empty

    def check(self, expect_maxlen=None):
        """
        Synchronises and raises what the reference would have raised while building the trees.
        expect_maxlen: the padded length T; the reference's bmm fails when max(len) != T (gcn.py:97,269).
        """
        st = self.status.cpu()
        bad = torch.nonzero(st[:-1] != 0)
        if bad.numel():
            b = int(bad[0])
            raise TreeError(b, int(st[b]))
        if expect_maxlen is not None and int(st[-1]) != expect_maxlen:
            raise ValueError("longest sentence has %d tokens but the batch is padded to %d: the reference builds "
                             "adj as [B,%d,%d] and its bmm with [B,%d,D] inputs fails (model/gcn.py:97,269)"
                             % (int(st[-1]), expect_maxlen, int(st[-1]), int(st[-1]), expect_maxlen))
        return self

    def nnz(self):
        rp = self.row_ptr.view(self.B, self.T + 1)
        return (rp[:, -1] - rp[:, 0]).to(torch.int64)

    def compact(self, Tc=None, also_keep=None):
        """
        "Pooled-only" rows: the same pattern for a [B, Tc] batch that holds only the tokens of the pruned
        trees (pool_mask False), renumbered in token order -- see gcnpt_compact_trees in include/gcnpt.h.  Tc=None takes the
        most tokens any sentence keeps (ONE host sync; pass a fixed Tc, e.g. a TreeCache's, to stay asynchronous: a sentence
        that keeps more gets status E_LENGTH).  also_keep: bool [B,T], tokens to keep although they are outside the tree --
        the subject / object tokens: the reference pools h over them through subj_mask / obj_mask whether or not they are in the
        tree (gcn.py:116-119), and a one-node tree (tree.py:182-192 writes no self loop for a childless root) has NO token in
        pool_mask's sense.  Such tokens get empty rows and stay excluded from the tree pooling.  Returns a CompactTrees.
        """
        src_mask = self.pool_mask
        if also_keep is not None:
            src_mask = (self.pool_mask.view(self.B, self.T) & ~also_keep.view(self.B, self.T).bool()).view(self.B, self.T, 1).contiguous()
        if Tc is None:
            Tc = max(int((~src_mask.view(self.B, self.T)).sum(1).max()), 1)
        Tc = int(Tc)
        cap_c = 3 * Tc if self.cap == 3 * self.T else min(self.cap, Tc * Tc)
        bufs = _alloc(self.B, Tc, cap_c, self.device, self.label is not None, self.rowT_ptr is not None)
        row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, status = bufs
        tok = torch.empty((self.B, Tc), dtype=torch.int64, device=self.device)
        kept = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        P = _lib.ptr
        _lib.check(_lib.lib().gcnpt_compact_trees(
            _lib.stream(), P(self.row_ptr), P(self.col_idx), P(self.label), P(self.rowT_ptr), P(self.colT_idx), P(self.ell), P(self.ellT),
            P(src_mask), P(self.status), self.B, self.T, self.cap, Tc, cap_c, P(row_ptr), P(col_idx), P(label), P(rowT_ptr),
            P(colT_idx), P(ell), P(ellT), P(pool_mask), P(status), P(tok), P(kept)))
        if also_keep is not None:
            # the extra tokens are rows of the compact batch but NOT members of the tree: the mask GCN.forward returns stays the reference's
            orig = torch.gather(self.pool_mask.view(self.B, self.T), 1, tok.clamp(min=0))
            pool_mask.copy_((orig | (tok < 0)).view(self.B, Tc, 1))
        return CompactTrees(PrunedTrees(self.B, Tc, cap_c, *bufs), tok, kept, self.T)

    def pack(self, lens, n_rows=None):
        """
        The same adjacency over TOKEN-PACKED rows (north_star "packed"; include/gcnpt.h gcnpt_pack_trees): sentence b's token i
        becomes row cu_seqlens[b] + i, padding slots do not exist.  lens: tokens per sentence (int sequence / tensor; a host
        list avoids the one sync that sizing the packed buffers otherwise needs -- the reference itself syncs for the lengths
        at this point, model/gcn.py:96); n_rows: rows to allocate if the caller knows sum(lens).  Returns a PackedTrees.
        """
        dev = self.device
        lens_dev = torch.as_tensor(lens, device=dev).to(torch.int32).contiguous()
        if lens_dev.numel() != self.B:
            raise ValueError("pack: %d lengths for %d sentences" % (lens_dev.numel(), self.B))
        if n_rows is None:
            n_rows = int(sum(int(v) for v in lens)) if not torch.is_tensor(lens) or not lens.is_cuda else int(lens_dev.clamp(max=self.T).sum().item())
        n_rows = max(int(n_rows), 1)
        nnz_cap = min(self.B * self.cap, max(3 * n_rows, 1)) if self.cap == 3 * self.T else self.B * self.cap
        i32 = dict(dtype=torch.int32, device=dev)
        tr = self.rowT_ptr is not None
        cu = torch.empty((self.B + 1,), **i32)
        row_ptr = torch.empty((n_rows + 1,), **i32)
        col_idx = torch.empty((nnz_cap,), **i32)
        label = torch.empty((nnz_cap,), **i32) if self.label is not None else None
        rowT_ptr = torch.empty((n_rows + 1,), **i32) if tr else None
        colT_idx = torch.empty((nnz_cap,), **i32) if tr else None
        ell = torch.zeros((n_rows * 8,), **i32)
        ellT = torch.zeros((n_rows * 8,), **i32) if tr else None
        pool_mask = torch.ones((n_rows, 1), dtype=torch.bool, device=dev)
        row_sent = torch.zeros((n_rows,), **i32)
        status = torch.empty((2,), **i32)
        P = _lib.ptr
        _lib.check(_lib.lib().gcnpt_pack_trees(
            _lib.stream(), P(self.row_ptr), P(self.col_idx), P(self.label), P(self.rowT_ptr), P(self.colT_idx), P(self.ell), P(self.ellT),
            P(self.pool_mask), P(lens_dev), self.B, self.T, self.cap, P(cu), P(row_ptr), P(col_idx), P(label), P(rowT_ptr), P(colT_idx),
            P(ell), P(ellT), P(pool_mask), P(row_sent), n_rows, nnz_cap, P(status)))
        return PackedTrees(self, n_rows, nnz_cap, cu, row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, row_sent, status)

    def to_dense(self):
        """float32 [B,T,T] with the labels tree_to_adj writes (deprel id, +42 for the reverse edge, 84 on the diagonal)."""
        adj = torch.empty((self.B, self.T, self.T), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().gcnpt_csr_to_adj(_lib.stream(), _lib.ptr(self.row_ptr), _lib.ptr(self.col_idx),
                                               _lib.ptr(self.label), self.B, self.T, _lib.ptr(adj)))
        return adj


class PackedTrees(object):
    """
    The batch adjacency over token-packed rows (PrunedTrees.pack; layout in include/gcnpt.h under gcnpt_pack_trees).

    N rows = sum(len); cu_seqlens int32 [B+1]; row_ptr / rowT_ptr int32 [N+1]; col_idx / colT_idx / label with ABSOLUTE packed
    row numbers; ell / ellT int32 [N*8]; pool_mask bool [N,1]; row_sent int32 [N]; status int32 [2].  `padded` is the [B,T]
    PrunedTrees it was made from (its pool_mask is what GCN.forward returns).  The layer ops take it in place of a PrunedTrees
    together with packed activations [N, width] (pack_rows / unpack_rows move between the layouts).
    """
    packed = True

    def __init__(self, padded, n_rows, nnz_cap, cu_seqlens, row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, row_sent, status):
        self.padded, self.N, self.nnz_cap = padded, int(n_rows), int(nnz_cap)
        self.B, self.T = padded.B, padded.T
        self.cu_seqlens, self.row_ptr, self.col_idx, self.label = cu_seqlens, row_ptr, col_idx, label
        self.rowT_ptr, self.colT_idx, self.ell, self.ellT = rowT_ptr, colT_idx, ell, ellT
        self.pool_mask, self.row_sent, self.status = pool_mask, row_sent, status
        self._empty = None

    @property
    def device(self):
        return self.row_ptr.device

    def empty_ell(self):
        if self._empty is None:
            self._empty = torch.zeros_like(self.ell)
        return self._empty

    def check(self):
        """Synchronises: raises if the packed buffers were too small or sum(len) differs from the rows allocated."""
        self.padded.check()
        st = self.status.cpu()
        if int(st[0]) != 0:
            raise _lib.GcnptError(int(st[0]), "pack_trees: %d rows / %d entries allocated do not hold the batch (sum(len) = %d)"
                                  % (self.N, self.nnz_cap, int(st[1])))
        if int(st[1]) != self.N:
            raise ValueError("pack_trees: sum(len) = %d but %d rows were allocated" % (int(st[1]), self.N))
        return self

    def pack_rows(self, t):
        """[B,T,W] (or [B,T]) -> packed [N,W] ([N]); differentiable."""
        return _PackRowsFn.apply(t, self)

    def unpack_rows(self, t):
        """packed [N,W] -> [B,T,W], zero in the slots past each sentence's end; differentiable."""
        return _UnpackRowsFn.apply(t, self)


def _move_rows(t, trees, unpack):
    lib = _lib.lib()
    if unpack:
        W = t.shape[-1]
        src = t.contiguous()
        dst = torch.empty((trees.B, trees.T, W), dtype=t.dtype, device=t.device)
        fn = lib.gcnpt_unpack_rows
    else:
        flat = t.dim() == 2
        src = (t.unsqueeze(-1) if flat else t).contiguous()
        W = src.shape[-1]
        dst = torch.empty((trees.N, W), dtype=t.dtype, device=t.device)
        fn = lib.gcnpt_pack_rows
    if src.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("pack/unpack_rows: float32 or bfloat16 rows (integer fields: use PackedTrees.row_sent / cu_seqlens)")
    _lib.check(fn(_lib.stream(), _lib.ptr(src), _lib.dtype_code(src.dtype), _lib.ptr(trees.cu_seqlens), trees.B, trees.T, W, _lib.ptr(dst)))
    return dst


class _PackRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, trees):
        ctx.trees, ctx.flat = trees, t.dim() == 2
        out = _move_rows(t, trees, unpack=False)
        return out.squeeze(-1) if ctx.flat else out

    @staticmethod
    def backward(ctx, g):
        g = _move_rows(g.unsqueeze(-1) if ctx.flat else g, ctx.trees, unpack=True)
        return (g.squeeze(-1) if ctx.flat else g), None


class _UnpackRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, trees):
        ctx.trees = trees
        return _move_rows(t, trees, unpack=True)

    @staticmethod
    def backward(ctx, g):
        return _move_rows(g, ctx.trees, unpack=False), None


class CompactTrees(object):
    """
    PrunedTrees over the kept tokens only (PrunedTrees.compact / TreeCache.batch(compact=True)).

    trees  PrunedTrees for [B, Tc]       tok  int64 [B,Tc] token position of each slot (-1 = empty slot)
    kept   int32 [B] tokens per sentence  T    width of the batch the token positions refer to

    GCNRelationModel / GCNClassifier.forward(inputs, trees=<CompactTrees>) run the layers on [B,Tc] rows and pool them;
    logits are the ones of the full batch (the reference pools over exactly these tokens, gcn.py:116-121).
    GCN.forward(<CompactTrees>, inputs) returns ([B,Tc,H], mask [B,Tc,1]).
    """

    def __init__(self, trees, tok, kept, T):
        self.trees, self.tok, self.kept, self.T = trees, tok, kept, int(T)
        self.B, self.Tc = trees.B, trees.T
        self._index = None

    @property
    def device(self):
        return self.trees.device

    @property
    def valid(self):
        return self.tok >= 0

    def check(self, expect_maxlen=None):
        self.trees.check()
        return self

    def take(self, t, fill=None):
        """Rows of t [B,T,...] (or [B,T]) at the kept tokens -> [B,Tc,...]; empty slots read token 0, or `fill` if given."""
        if t.shape[0] != self.B or t.shape[1] != self.T:
            raise ValueError("take: tensor is %s, the trees are for a [%d,%d] batch" % (tuple(t.shape), self.B, self.T))
        if self._index is None:
            self._index = self.tok.clamp(min=0)
        idx = self._index
        if t.dim() > 2:
            idx = idx.view(self.B, self.Tc, *([1] * (t.dim() - 2))).expand(-1, -1, *t.shape[2:])
        out = torch.gather(t, 1, idx)
        if fill is not None:
            v = self.valid
            out = torch.where(v.view(self.B, self.Tc, *([1] * (t.dim() - 2))) if t.dim() > 2 else v, out, torch.full_like(out, fill))
        return out


def _alloc(B, T, cap, device, want_label, want_transpose):
    i32 = dict(dtype=torch.int32, device=device)
    row_ptr = torch.empty((B * (T + 1),), **i32)
    col_idx = torch.empty((B * cap,), **i32)
    label = torch.empty((B * cap,), **i32) if want_label else None
    rowT_ptr = torch.empty((B * (T + 1),), **i32) if want_transpose else None
    colT_idx = torch.empty((B * cap,), **i32) if want_transpose else None
    ell = torch.empty((B * T * 8,), **i32)
    ellT = torch.empty((B * T * 8,), **i32) if want_transpose else None
    pool_mask = torch.empty((B, T, 1), dtype=torch.bool, device=device)
    status = torch.empty((B + 1,), **i32)
    return row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, status


def _i64(t, name):
    if t.dtype != torch.int64:
        raise TypeError("%s must be an int64 tensor (as data/loader.py produces), got %s" % (name, t.dtype))
    return t.contiguous()


def prune_to_csr(head, subj_pos, obj_pos, deprel, prune_k, masks=None, lens=None, want_label=True, pack=None, want_transpose=True):
    """
    Batch version of `tree_to_adj(maxlen, head_to_tree(head[i], words[i], l[i], prune, subj_pos[i],
    obj_pos[i], deprel[i]), directed=False, self_loop=True)` for every sentence i (model/gcn.py:105-106),
    on the device.  head/subj_pos/obj_pos/deprel: int64 [B,T] CUDA tensors straight from the loader;
    masks: bool [B,T] (True = pad, model/gcn.py:96) or lens: int32 [B].
    Asynchronous; call `.check()` on the result to surface per-sentence errors.
    pack: a model.gcn.WeightPack -- the same launch then also packs the layer weights (gcnpt_prune_to_csr_pack).
    want_transpose=False: no transposed pattern (rowT_ptr / colT_idx / ellT stay None): inference, which never runs a backward.
    """
    _lib.require_gpu(head)
    head, subj_pos, obj_pos, deprel = (_i64(t, n) for t, n in ((head, "head"), (subj_pos, "subj_pos"),
                                                                (obj_pos, "obj_pos"), (deprel, "deprel")))
    B, T = head.shape
    if masks is None and lens is None:
        raise ValueError("give masks (True = pad) or lens")
    if masks is not None:
        masks = masks.contiguous()
        if masks.dtype not in (torch.bool, torch.uint8) or tuple(masks.shape) != (B, T):
            raise TypeError("masks must be bool/uint8 [B,T]")
    if lens is not None:
        lens = lens.to(device=head.device, dtype=torch.int32).contiguous()
    cap = 3 * T
    bufs = _alloc(B, T, cap, head.device, want_label, bool(want_transpose))
    row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, status = bufs
    args = (_lib.stream(), _lib.ptr(head), _lib.ptr(subj_pos), _lib.ptr(obj_pos), _lib.ptr(deprel),
            _lib.ptr(masks) if masks is not None else None, _lib.ptr(lens) if masks is None else None,
            B, T, int(prune_k), cap, _lib.ptr(row_ptr), _lib.ptr(col_idx), _lib.ptr(label), _lib.ptr(rowT_ptr),
            _lib.ptr(colT_idx), _lib.ptr(ell), _lib.ptr(ellT), _lib.ptr(pool_mask), _lib.ptr(status))
    if pack is None:
        _lib.check(_lib.lib().gcnpt_prune_to_csr(*args))
    else:
        _lib.check(_lib.lib().gcnpt_prune_to_csr_pack(*(args + pack.c_args())))
        pack.launched = True
    return PrunedTrees(B, T, cap, *bufs)


class _PaddedInfo(PrunedTrees):
    """What a batch pruned straight into the packed layout keeps of the padded one: its shape, the per-sentence status and the
    padded pool mask (what GCN.forward returns); no adjacency arrays."""

    def __init__(self, B, T, pool_mask, status):
        super().__init__(B, T, 3 * T, None, None, None, None, None, None, None, pool_mask, status)

    @property
    def device(self):
        return self.status.device


_SYNC_WS = {}        # (device, stream) -> zeroed uint64 workspace of gcnpt_prune_to_csr_packed (left zero by every launch)


def prune_to_csr_packed(head, subj_pos, obj_pos, deprel, prune_k, lens, masks=None, want_label=False, want_transpose=True, pack=None,
                        n_rows=None, want_padded_mask=True):
    """
    prune_to_csr(...).pack(lens) in ONE launch (gcnpt_prune_to_csr_packed): the pruner writes the token-packed layout itself, bit
    for bit what the two-step form gives.  lens: tokens per sentence, a host sequence (the loader has them, data/loader.py:109-121)
    or a tensor; n_rows: sum(lens) if the caller knows it (avoids summing a device tensor).  masks: the pad mask, if the lengths
    should come from it as the reference's do (gcn.py:96); default: `lens`.  pack: a model.gcn.WeightPack to fill in the same launch.
    Returns a PackedTrees whose `.padded` holds the batch's shape, per-sentence status and padded pool mask only.
    """
    _lib.require_gpu(head)
    head, subj_pos, obj_pos, deprel = (_i64(t, n) for t, n in ((head, "head"), (subj_pos, "subj_pos"), (obj_pos, "obj_pos"), (deprel, "deprel")))
    B, T = head.shape
    dev = head.device
    lens_dev = torch.as_tensor(lens, device=dev).to(torch.int32).contiguous()
    if lens_dev.numel() != B:
        raise ValueError("prune_to_csr_packed: %d lengths for %d sentences" % (lens_dev.numel(), B))
    if n_rows is None:
        n_rows = int(sum(int(v) for v in lens)) if not torch.is_tensor(lens) or not lens.is_cuda else int(lens_dev.clamp(max=T).sum().item())
    n_rows = max(int(n_rows), 1)
    nnz_cap = max(3 * n_rows, 1)
    if masks is not None:
        masks = masks.contiguous()
        if masks.dtype not in (torch.bool, torch.uint8) or tuple(masks.shape) != (B, T):
            raise TypeError("masks must be bool/uint8 [B,T]")
    i32 = dict(dtype=torch.int32, device=dev)
    tr = bool(want_transpose)
    cu = torch.empty((B + 1,), **i32)
    row_ptr = torch.empty((n_rows + 1,), **i32)
    col_idx = torch.empty((nnz_cap,), **i32)
    label = torch.empty((nnz_cap,), **i32) if want_label else None
    rowT_ptr = torch.empty((n_rows + 1,), **i32) if tr else None
    colT_idx = torch.empty((nnz_cap,), **i32) if tr else None
    ell = torch.zeros((n_rows * 8,), **i32)
    ellT = torch.zeros((n_rows * 8,), **i32) if tr else None
    pool_mask = torch.ones((n_rows, 1), dtype=torch.bool, device=dev)
    row_sent = torch.zeros((n_rows,), **i32)
    status = torch.empty((2,), **i32)
    sent_status = torch.empty((B + 1,), **i32)
    pm_padded = torch.empty((B, T, 1), dtype=torch.bool, device=dev) if want_padded_mask else None
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    ws = _SYNC_WS.get(key)
    if ws is None or ws.numel() < B + 2:
        ws = _SYNC_WS[key] = torch.zeros((max(B + 2, 1024),), dtype=torch.int64, device=dev)
    P = _lib.ptr
    tail = pack.c_args() if pack is not None else (0, None, None, None, 0, None, None)
    _lib.check(_lib.lib().gcnpt_prune_to_csr_packed(
        _lib.stream(), P(head), P(subj_pos), P(obj_pos), P(deprel), P(masks) if masks is not None else None, P(lens_dev) if masks is None else None,
        B, T, int(prune_k), P(cu), P(row_ptr), P(col_idx), P(label), P(rowT_ptr), P(colT_idx), P(ell), P(ellT), P(pool_mask), P(row_sent),
        n_rows, nnz_cap, P(status), P(sent_status), P(pm_padded), P(ws), *tail))
    if pack is not None:
        pack.launched = True
    padded = _PaddedInfo(B, T, pm_padded, sent_status)
    return PackedTrees(padded, n_rows, nnz_cap, cu, row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, row_sent, status)


def adj_to_csr(adj, want_label=True):
    """CSR of an explicit dense adjacency [B,T,T] (the `adj` argument of GCN.forward, model/gcn.py:229,260-262)."""
    if adj.dtype != torch.float32 or adj.dim() != 3 or adj.shape[1] != adj.shape[2]:
        raise TypeError("adj must be float32 [B,T,T]")
    adj = _lib.require_gpu(adj).contiguous()
    B, T, _ = adj.shape
    cap = T * T
    bufs = _alloc(B, T, cap, adj.device, want_label, True)
    row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, status = bufs
    _lib.check(_lib.lib().gcnpt_adj_to_csr(_lib.stream(), _lib.ptr(adj), B, T, cap, _lib.ptr(row_ptr), _lib.ptr(col_idx),
                                           _lib.ptr(label), _lib.ptr(rowT_ptr), _lib.ptr(colT_idx), _lib.ptr(ell), _lib.ptr(ellT),
                                           _lib.ptr(pool_mask), _lib.ptr(status)))
    return PrunedTrees(B, T, cap, *bufs)


class TreeCache(object):
    """
    Loader-side pre-pruning (reference data/loader.py:81-141 + model/gcn.py:96-110): pruning depends on
    the parse alone, so the whole dataset is pruned once on the device and stays in HBM; `batch(idx, T)` assembles the
    PrunedTrees of a batch from the cached rows (one small copy kernel), bit-identical to pruning that batch directly.
    """

    def __init__(self, trees, lens, prune_k, compact=None):
        self.trees, self.lens, self.prune_k = trees, lens, int(prune_k)
        self.compact = compact            # CompactTrees of the whole dataset, or None

    def __len__(self):
        return self.trees.B

    @classmethod
    def build(cls, head, subj_pos, obj_pos, deprel, prune_k, masks=None, lens=None, want_label=True, compact=False):
        """head/subj_pos/obj_pos/deprel: int64 [S,Ts] CUDA tensors of the WHOLE dataset padded to its longest sentence;
        masks (True = pad) or lens as in prune_to_csr.  Per-sentence errors stay in the cache and surface in the
        batches that contain the sentence (PrunedTrees.check).  compact=True also keeps the kept-token form of every
        sentence (PrunedTrees.compact; one host sync here for the dataset's widest tree), for batch(..., compact=True)."""
        trees = prune_to_csr(head, subj_pos, obj_pos, deprel, prune_k, masks=masks, lens=lens, want_label=want_label)
        if lens is None:
            lens = (masks == 0).sum(1)
        lens = lens.to(device=head.device, dtype=torch.int32).contiguous()
        # compact form: entity tokens are kept even when they are outside the tree (one-node trees), see PrunedTrees.compact
        return cls(trees, lens, prune_k, trees.compact(also_keep=(subj_pos == 0) | (obj_pos == 0)) if compact else None)

    def batch(self, idx, T, want_label=None, compact=False, Tc=None, pack=None):
        """idx: int64 [B] sentence numbers (CUDA tensor; repeats allowed); T: the width the batch tensors are padded to
        (the reference pads to the longest sentence of the batch, gcn.py:97).  compact=True: a CompactTrees of width Tc
        (default: the widest tree of the dataset) whose token positions refer to the [B,T] batch.
        pack: a model.gcn.WeightPack (GCN.weight_pack()) -- the gather launch then also packs the layer weights."""
        idx = _lib.require_gpu(idx).to(torch.int64).contiguous()
        if compact:
            if self.compact is None:
                raise ValueError("the cache was built without compact=True")
            width = int(Tc) if Tc is not None else self.compact.Tc
            ct = self._gather(self.compact.trees, self.compact.kept, idx, width, want_label, pack)
            n = min(width, self.compact.Tc)
            tok = torch.full((idx.numel(), width), -1, dtype=torch.int64, device=idx.device)
            tok[:, :n] = self.compact.tok.index_select(0, idx.clamp(0, len(self) - 1))[:, :n]
            tok = torch.where((ct.status[:-1] == 0).unsqueeze(1), tok, torch.full_like(tok, -1))
            return CompactTrees(ct, tok, self.compact.kept.index_select(0, idx.clamp(0, len(self) - 1)), int(T))
        return self._gather(self.trees, self.lens, idx, int(T), want_label, pack)

    def batch_packed(self, idx, T, want_label=None, want_transpose=True, pack=None, n_rows=None, want_padded_mask=True):
        """batch(idx, T).pack(lens[idx]) in ONE launch (gcnpt_gather_trees_packed), bit for bit.  n_rows: the batch's token count
        (sum of min(len, T)) when the caller knows it -- the loader does; otherwise it is summed on the device (one sync)."""
        idx = _lib.require_gpu(idx).to(torch.int64).contiguous()
        src, B, T = self.trees, int(idx.numel()), int(T)
        want_label = (src.label is not None) if want_label is None else want_label
        if want_label and src.label is None:
            raise ValueError("the cache was built without labels")
        tr = bool(want_transpose)
        if tr and src.rowT_ptr is None:
            raise ValueError("the cache was built without the transposed pattern")
        if n_rows is None:
            ok = (idx >= 0) & (idx < len(self))
            n_rows = int((self.lens.index_select(0, idx.clamp(0, len(self) - 1)).clamp(max=T) * ok).sum().item())
        n_rows = max(int(n_rows), 1)
        nnz_cap = max(3 * n_rows, 1)
        dev = src.device
        i32 = dict(dtype=torch.int32, device=dev)
        cu = torch.empty((B + 1,), **i32)
        row_ptr = torch.empty((n_rows + 1,), **i32)
        col_idx = torch.empty((nnz_cap,), **i32)
        label = torch.empty((nnz_cap,), **i32) if want_label else None
        rowT_ptr = torch.empty((n_rows + 1,), **i32) if tr else None
        colT_idx = torch.empty((nnz_cap,), **i32) if tr else None
        ell = torch.zeros((n_rows * 8,), **i32)
        ellT = torch.zeros((n_rows * 8,), **i32) if tr else None
        pool_mask = torch.ones((n_rows, 1), dtype=torch.bool, device=dev)
        row_sent = torch.zeros((n_rows,), **i32)
        status = torch.empty((2,), **i32)
        sent_status = torch.empty((B + 1,), **i32)
        pm_padded = torch.empty((B, T, 1), dtype=torch.bool, device=dev) if want_padded_mask else None
        P = _lib.ptr
        tail = pack.c_args() if pack is not None else (0, None, None, None, 0, None, None)
        _lib.check(_lib.lib().gcnpt_gather_trees_packed(
            _lib.stream(), P(src.row_ptr), P(src.col_idx), P(src.label), P(src.rowT_ptr), P(src.colT_idx), P(src.ell), P(src.ellT),
            P(src.pool_mask), P(src.status), P(self.lens), src.B, src.T, src.cap, P(idx), B, T,
            P(cu), P(row_ptr), P(col_idx), P(label), P(rowT_ptr), P(colT_idx), P(ell), P(ellT), P(pool_mask), P(row_sent),
            n_rows, nnz_cap, P(status), P(sent_status), P(pm_padded), *tail))
        if pack is not None:
            pack.launched = True
        return PackedTrees(_PaddedInfo(B, T, pm_padded, sent_status), n_rows, nnz_cap, cu, row_ptr, col_idx, label, rowT_ptr, colT_idx,
                           ell, ellT, pool_mask, row_sent, status)

    @staticmethod
    def _gather(src, lens, idx, T, want_label, pack=None):
        B = int(idx.numel())
        want_label = (src.label is not None) if want_label is None else want_label
        if want_label and src.label is None:
            raise ValueError("the cache was built without labels")
        cap = 3 * T
        bufs = _alloc(B, T, cap, src.device, want_label, True)
        row_ptr, col_idx, label, rowT_ptr, colT_idx, ell, ellT, pool_mask, status = bufs
        P = _lib.ptr
        args = (_lib.stream(), P(src.row_ptr), P(src.col_idx), P(src.label), P(src.rowT_ptr), P(src.colT_idx), P(src.ell), P(src.ellT),
                P(src.pool_mask), P(src.status), P(lens), src.B, src.T, src.cap, P(idx), B, T, cap,
                P(row_ptr), P(col_idx), P(label), P(rowT_ptr), P(colT_idx), P(ell), P(ellT), P(pool_mask), P(status))
        if pack is None:
            _lib.check(_lib.lib().gcnpt_gather_trees(*args))
        else:
            _lib.check(_lib.lib().gcnpt_gather_trees_pack(*(args + pack.c_args())))
            pack.launched = True
        return PrunedTrees(B, T, cap, *bufs)


def inputs_to_tree_reps(head, words, l, prune, subj_pos, obj_pos, deprel):
    """
    Same signature and result as the closure of that name in GCNRelationModel.forward (model/gcn.py:102-110):
    the dense labelled adjacency float32 [B,maxlen,maxlen], but built on the device.  `words` is unused (the
    reference only stores it on the Tree nodes); `l` are the sentence lengths (any int sequence/tensor).
    """
    lens = torch.as_tensor(l, dtype=torch.int32, device=head.device)
    trees = prune_to_csr(head, subj_pos, obj_pos, deprel, prune, lens=lens)
    trees.check(expect_maxlen=head.shape[1])
    return trees.to_dense()
