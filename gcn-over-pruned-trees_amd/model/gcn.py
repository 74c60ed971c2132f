"""
Host-side mirror of the reference's model/gcn.py for the `regular` GCN path, running on MI355X kernels.

Same class names, constructor arguments, `opt` keys, parameter / state_dict names and forward()
signatures as the reference, so `from model.gcn import GCNClassifier` can be swapped for
`from gcn_over_pruned_trees_amd.model.gcn import GCNClassifier` in its trainer (model/trainer.py:82):

  GCNClassifier(opt, emb_matrix=None).forward(inputs) -> (logits [B,C], pooling_output [B,H])   gcn.py:15-36
  GCNRelationModel(opt, emb_matrix=None).forward(inputs) -> (outputs [B,H], h_out [B,H])        gcn.py:38-126
  GCN(opt, embeddings, mem_dim, num_layers).forward(adj, inputs) -> (h [B,T,H], mask [B,T,1])   gcn.py:128-395
  pool(h, mask, type)                                                                            gcn.py:473-483

What is different underneath: the tree pruning + adjacency build (gcn.py:96-110) is one device kernel
(model/tree.py here) and every iteration of the layer loop (gcn.py:266-393) is one fused HIP kernel per
direction, reached through the C-ABI of include/gcnpt.h.  Embeddings, the optional BiLSTM, pooling and
the output MLP stay ordinary PyTorch-ROCm modules, as in the reference.

New optional `opt` keys (defaults reproduce the reference): `gcn_dtype` = 'fp32' | 'bf16' (MFMA operand /
activation storage type inside the layer stack), `gcn_check_trees` = True (synchronise once per forward to
raise on malformed trees the way the reference does; False keeps the step free of host syncs), `gcn_packed` = False (True: the
layer loop runs on token-packed rows, sum(len) instead of B*T, padding only at the module boundary), `gcn_graph_rng` = False
(True: dropout seeds that survive hipGraph capture), `gcn_pool_handover` = True (layer stack + poolings as one op whose backward
hands the top layer its dZ), `gcn_pack_with_trees` = True (the tree launch also packs the weights in a training step),
`gcn_sparse_emb_grad` = False (True: the word-embedding table's gradient is a row-sparse tensor of the batch's token rows, for
shard.SparseRowExchange in a data-parallel loop), `gcn_reuse_packed_weights` = False (True: a forward may reuse the packed weight images of the previous one while the weights'
version counters stand still -- gradient accumulation, a frozen model; off by default because `p.data` updates do not bump them, in eval() either) -- with `gcn_check_trees=False` a whole training step of the no-LSTM
model can be captured with torch.cuda.graph and replayed, see tests/test_gpu_parity.py::test_training_step_graph_capture).
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..utils import constant
from .tree import CompactTrees, PackedTrees, PrunedTrees, adj_to_csr, prune_to_csr, prune_to_csr_packed


def _row_dims(h, trees, what):
    """(B, T) as the C-ABI wants them, rows, leading shape.  Padded: h is [B,T,W] and the trees are for [B,T].  Token-packed
    (PackedTrees): h is [N,W]; the entry points then take B = N, T = 0 (include/gcnpt.h, gcnpt_pack_trees)."""
    if getattr(trees, "packed", False):
        if h.dim() != 2 or h.shape[0] != trees.N:
            raise RuntimeError("%s: packed trees hold %d rows but the inputs are %s" % (what, trees.N, tuple(h.shape)))
        return trees.N, 0, trees.N, (trees.N,)
    if h.dim() != 3 or (h.shape[0], h.shape[1]) != (trees.B, trees.T):
        raise RuntimeError("%s: inputs are %s but the adjacency is for [%d,%d]" % (what, tuple(h.shape), trees.B, trees.T))
    return trees.B, trees.T, trees.B * trees.T, (trees.B, trees.T)


# ------------------------------------------------------------------------------------------------------
# one GCN layer as an autograd op over the C-ABI
# ------------------------------------------------------------------------------------------------------
class _GCNLayerFn(torch.autograd.Function):
    """out = dropout(relu((((A+I) h) W^T + 2 b) / (deg + 1)))  -- model/gcn.py:269-271, 390-393."""

    @staticmethod
    def forward(ctx, h, weight, bias, trees, drop_p, seed, compute, out_dtype, no_adj, seed_dev=None):
        B, T, rows, lead = _row_dims(h, trees, "GCN layer")
        Din = h.shape[-1]
        H = weight.shape[0]
        if weight.shape[1] != Din:
            raise RuntimeError("GCN layer: input width %d does not match weight %s" % (Din, tuple(weight.shape)))
        L = _lib.lib()
        h = h.contiguous()
        w32 = weight.detach().to(torch.float32).contiguous()
        b32 = bias.detach().to(torch.float32).contiguous()
        w_fwd = torch.empty((L.gcnpt_packed_bytes(H, Din, compute),), dtype=torch.uint8, device=h.device)
        w_bwd = torch.empty((L.gcnpt_packed_bytes(Din, H, compute),), dtype=torch.uint8, device=h.device)
        st = _lib.stream()
        _lib.check(L.gcnpt_pack_weights(st, _lib.ptr(w32), H, Din, compute, _lib.ptr(w_fwd), _lib.ptr(w_bwd)))
        out = torch.empty(lead + (H,), dtype=out_dtype, device=h.device)
        g_ell = trees.empty_ell() if no_adj else trees.ell
        # the gathered tile S = (A+I)h is saved in MFMA fragment order for the weight gradient (include/gcnpt.h)
        want_wgrad = weight.requires_grad or bias.requires_grad
        s_frag = torch.empty((L.gcnpt_frag_bytes(rows, Din, compute),), dtype=torch.uint8, device=h.device) if want_wgrad else None
        _lib.check(L.gcnpt_layer_fwd(st, _lib.ptr(h), _lib.dtype_code(h.dtype), _lib.ptr(w_fwd), _lib.ptr(b32),
                                     _lib.ptr(trees.row_ptr), _lib.ptr(trees.col_idx), _lib.ptr(g_ell), _lib.ptr(trees.ell), B, T, Din, H,
                                     _lib.ptr(out), _lib.dtype_code(out_dtype), compute, float(drop_p), int(seed), _lib.ptr(s_frag), _lib.ptr(seed_dev)))
        ctx.save_for_backward(out, w_bwd, s_frag)
        ctx.h_dtype = h.dtype
        ctx.trees, ctx.no_adj, ctx.compute = trees, no_adj, compute
        ctx.scale = 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0
        ctx.dims = (B, T, Din, H)
        ctx.rows, ctx.lead = rows, lead
        ctx.param_dtypes = (weight.dtype, bias.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        out, w_bwd, s_frag = ctx.saved_tensors
        trees, B, T, Din, H = ctx.trees, *ctx.dims
        L, st = _lib.lib(), _lib.stream()
        dev = out.device
        gout = gout.to(out.dtype).contiguous()
        gd = _lib.dtype_code(out.dtype)
        dh = dW = db = z_frag = None
        g_ellT = trees.empty_ell() if ctx.no_adj else trees.ellT
        want_w = (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and s_frag is not None
        if ctx.needs_input_grad[0]:
            dh = torch.empty(ctx.lead + (Din,), dtype=ctx.h_dtype, device=dev)
        if want_w:
            z_frag = torch.empty((L.gcnpt_frag_bytes(ctx.rows, H, ctx.compute),), dtype=torch.uint8, device=dev)
            dW = torch.empty((H, Din), dtype=torch.float32, device=dev)     # cleared by bwd_data, filled by bwd_weight
            db = torch.empty((H,), dtype=torch.float32, device=dev)
        if dh is not None or want_w:
            _lib.check(L.gcnpt_layer_bwd_data(st, _lib.ptr(gout), _lib.ptr(out), gd, _lib.ptr(w_bwd), _lib.ptr(trees.ell),
                                              _lib.ptr(trees.rowT_ptr), _lib.ptr(trees.colT_idx), _lib.ptr(g_ellT), B, T, Din, H, _lib.ptr(dh),
                                              _lib.dtype_code(ctx.h_dtype), ctx.compute, ctx.scale, _lib.ptr(z_frag),
                                              _lib.ptr(dW), _lib.ptr(db), None, 1.0, 0))
        if want_w:
            _lib.check(L.gcnpt_layer_bwd_weight(st, _lib.ptr(z_frag), _lib.ptr(s_frag), B, T, Din, H, _lib.ptr(dW), _lib.ptr(db),
                                                ctx.compute))
            dW, db = dW.to(ctx.param_dtypes[0]), db.to(ctx.param_dtypes[1])
        return dh, dW, db, None, None, None, None, None, None, None


def gcn_layer(h, weight, bias, trees, drop_p=0.0, seed=0, compute_dtype=torch.float32, out_dtype=None, no_adj=False, seed_dev=None):
    """
    Functional form of one iteration of the reference's layer loop.  h [B,T,Din] (float32 or bfloat16, CUDA),
    weight [H,Din], bias [H] (nn.Linear layout), trees: PrunedTrees.  compute_dtype float32 = exact fp32 MFMA,
    bfloat16 = bf16 operands with fp32 accumulation.  seed_dev: optional int64 [1] CUDA tensor added to `seed` on the device
    (advance it between replays of a captured graph to get fresh dropout masks).
    """
    if not isinstance(trees, (PrunedTrees, PackedTrees)):
        raise TypeError("trees must be a PrunedTrees (see model.tree.prune_to_csr / adj_to_csr) or a PackedTrees (PrunedTrees.pack)")
    compute = _lib.dtype_code(compute_dtype)
    _lib.require_gpu(h)
    if compute == _lib.F32 and h.dtype != torch.float32:
        h = h.float()
    out_dtype = out_dtype or h.dtype
    if compute == _lib.F32:
        out_dtype = torch.float32
    return _GCNLayerFn.apply(h, weight, bias, trees, float(drop_p), int(seed), compute, out_dtype, bool(no_adj), seed_dev)


class WeightPack(object):
    """The layer weights' MFMA-order images (gcnpt_pack_weights_multi) as a request that another launch can carry as a side job:
    `prune_to_csr(..., pack=wp)` / `TreeCache.batch(..., pack=wp)` build the trees and fill wp.wf / wp.wb in ONE launch
    (include/gcnpt.h, gcnpt_prune_to_csr_pack); the layer op then takes the images from the request when they still belong to the
    current weights (same storage, same version counters), and packs as usual otherwise."""

    def __init__(self, Ws, compute, dev):
        lib = _lib.lib()
        self.key = WeightPack.key_of(Ws, compute, dev)
        self.compute = compute
        self.dims = [tuple(w.shape) for w in Ws]
        self.w32 = [w.detach().to(torch.float32).contiguous() for w in Ws]
        u8 = dict(dtype=torch.uint8, device=dev)
        self.wf = [torch.empty((lib.gcnpt_packed_bytes(h, k, compute),), **u8) for h, k in self.dims]
        self.wb = [torch.empty((lib.gcnpt_packed_bytes(k, h, compute),), **u8) for h, k in self.dims]
        self.launched = False                      # set by the launch that carried the request

    @staticmethod
    def key_of(Ws, compute, dev):
        return (compute, str(dev)) + tuple((w.data_ptr(), w._version, tuple(w.shape)) for w in Ws)

    def c_args(self):
        """The trailing arguments of gcnpt_pack_weights_multi / gcnpt_*_pack."""
        L = len(self.dims)
        ints = lambda v: (ctypes.c_int * L)(*v)  # noqa: E731
        return (L, _lib.ptr_array(self.w32), ints([h for h, _ in self.dims]), ints([k for _, k in self.dims]), self.compute,
                _lib.ptr_array(self.wf), _lib.ptr_array(self.wb))


class _GCNLayersFn(torch.autograd.Function):
    """The reference's layer loop (model/gcn.py:266-393) as ONE autograd op over the per-layer kernels: one launch packs
    every layer's weights, one launch per layer and direction does the layer, one launch at the end of the backward sweep
    computes every layer's weight gradient (2L + 2 launches instead of 4L)."""

    @staticmethod
    def forward(ctx, x, trees, cfg, *params):
        Ws, bs = params[0::2], params[1::2]
        L = len(Ws)
        B, T, rows, lead = _row_dims(x, trees, "GCN layers")
        Din = x.shape[-1]
        dims = [tuple(w.shape) for w in Ws]                           # (H_l, Din_l)
        for l, (h, k) in enumerate(dims):
            if k != (Din if l == 0 else dims[l - 1][0]):
                raise RuntimeError("GCN layer %d: weight %s does not fit an input of width %d" % (l, (h, k), Din if l == 0 else dims[l - 1][0]))
        lib, st, dev, compute = _lib.lib(), _lib.stream(), x.device, cfg["compute"]
        x = x.contiguous()
        b32 = [b.detach().to(torch.float32).contiguous() for b in bs]
        u8 = dict(dtype=torch.uint8, device=dev)
        ints = lambda v: (ctypes.c_int * L)(*v)  # noqa: E731
        # the fragment-order images only change when an optimizer step (or a load_state_dict) rewrites a weight in place, which bumps
        # the tensor's version counter: a module passes its cache and eval() / gradient-accumulation forwards skip the pack launch;
        # a training step that builds its trees in the same forward gets them from that launch's side job (WeightPack)
        cache = None if torch.cuda.is_current_stream_capturing() else cfg.get("wcache")     # (a captured step must contain its pack;
        # the module passes its cache only where reuse is safe, see GCN._cache_usable)
        key = WeightPack.key_of(Ws, compute, dev)
        pre = cfg.get("prepacked")
        if pre is not None and pre.key == key and pre.launched:
            wf, wb = pre.wf, pre.wb
            if cache is not None:
                cache.update(key=key, wf=wf, wb=wb)
        elif cache is not None and cache.get("key") == key:
            wf, wb = cache["wf"], cache["wb"]
        else:
            pk = WeightPack(Ws, compute, dev)
            _lib.check(lib.gcnpt_pack_weights_multi(st, *pk.c_args()))
            wf, wb = pk.wf, pk.wb
            if cache is not None:
                cache.update(key=key, wf=wf, wb=wb)
        need_w = any(p.requires_grad for p in params)
        g_ell = trees.empty_ell() if cfg["no_adj"] else trees.ell
        outs = [torch.empty(lead + (H,), dtype=cfg["out_dtype"] if l == L - 1 else cfg["mid_dtype"], device=dev) for l, (H, _) in enumerate(dims)]
        s_frag = [torch.empty((lib.gcnpt_frag_bytes(rows, K, compute),), **u8) if need_w else None for _, K in dims]
        # every layer's launch from ONE native call (gcnpt_layers_fwd): no interpreter time between the launches
        _lib.check(lib.gcnpt_layers_fwd(
            st, L, _lib.ptr(x), _lib.dtype_code(x.dtype), _lib.ptr_array(wf), _lib.ptr_array(b32), _lib.ptr(trees.row_ptr),
            _lib.ptr(trees.col_idx), _lib.ptr(g_ell), _lib.ptr(trees.ell), B, T, ints([k for _, k in dims]), ints([h for h, _ in dims]),
            _lib.ptr_array(outs), ints([_lib.dtype_code(o.dtype) for o in outs]), compute, (ctypes.c_float * L)(*cfg["drop_p"]),
            (ctypes.c_uint64 * L)(*cfg["seed"]), _lib.ptr_array(s_frag), _lib.ptr(cfg.get("seed_dev"))))
        if cfg.get("acts") is not None:
            cfg["acts"].extend(outs)          # every layer's stored output (what the backward reads): tests drive the oracle's backward with them
        ctx.trees, ctx.cfg, ctx.dims, ctx.shape, ctx.need_w = trees, cfg, dims, (B, T, Din, L), need_w
        ctx.rows, ctx.lead = rows, lead
        ctx.x_dtype = x.dtype
        ctx.param_dtypes = [p.dtype for p in params]
        pool = cfg.get("pool")
        if pool is None:
            ctx.save_for_backward(*outs, *wb, *[f for f in s_frag if f is not None])
            return outs[-1]
        # the consumer of the stack, fused into the op (gcn.py:114-121): the three masked poolings in one pass over h_L.  Its backward
        # then hands the top layer dZ instead of dh (gcnpt_pool3_bwd_dz), and that layer gathers one row per neighbour, not three
        sp, op, kind = pool
        H = dims[-1][0]
        pm = trees.pool_mask.contiguous()
        pooled = torch.empty((B, 3 * H), dtype=torch.float32, device=dev)
        argmax = torch.empty((B, 3, H), dtype=torch.int32, device=dev) if kind == 0 else None
        _lib.check(lib.gcnpt_pool3_fwd(st, _lib.ptr(outs[-1]), _lib.dtype_code(outs[-1].dtype), _lib.ptr(pm), _lib.ptr(sp), _lib.ptr(op), B, T, H, kind,
                                       _lib.ptr(pooled), _lib.ptr(argmax)))
        ctx.save_for_backward(*outs, *wb, *[f for f in s_frag if f is not None])
        ctx.pool = (pm, sp, op, argmax, kind)
        return pooled

    @staticmethod
    def backward(ctx, gout):
        B, T, Din, L = ctx.shape
        saved = ctx.saved_tensors
        outs, wb, s_frag = saved[:L], saved[L:2 * L], saved[2 * L:]
        trees, cfg, dims = ctx.trees, ctx.cfg, ctx.dims
        lib, st, dev, compute = _lib.lib(), _lib.stream(), gout.device, cfg["compute"]
        want_w = ctx.need_w and any(ctx.needs_input_grad[3:])
        g_ellT = trees.empty_ell() if cfg["no_adj"] else trees.ellT
        u8 = dict(dtype=torch.uint8, device=dev)
        z_frag, dWs, dbs = [None] * L, [None] * L, [None] * L
        scales = [1.0 / (1.0 - p) if p > 0 else 1.0 for p in cfg["drop_p"]]
        pool = getattr(ctx, "pool", None) if cfg.get("pool") is not None else None
        if pool is None:
            g = gout.to(outs[-1].dtype).contiguous()
        else:
            pm, sp, op, argmax, kind = pool
            H = dims[-1][0]
            g = torch.empty_like(outs[-1])                       # dZ of the top layer, straight from the pooling's backward
            _lib.check(lib.gcnpt_pool3_bwd_dz(st, _lib.ptr(gout.to(torch.float32).contiguous()), _lib.ptr(argmax), _lib.ptr(pm), _lib.ptr(sp),
                                              _lib.ptr(op), B, T, H, kind, _lib.ptr(outs[-1]), _lib.ptr(trees.ell), scales[-1], _lib.ptr(g),
                                              _lib.dtype_code(g.dtype)))
        in_dtypes = [ctx.x_dtype if l == 0 else outs[l - 1].dtype for l in range(L)]
        dhs = [torch.empty(ctx.lead + (K,), dtype=in_dtypes[l], device=dev) if (l > 0 or ctx.needs_input_grad[0]) else None
               for l, (_, K) in enumerate(dims)]
        if want_w:
            z_frag = [torch.empty((lib.gcnpt_frag_bytes(ctx.rows, H, compute),), **u8) for H, _ in dims]
            dWs = [torch.empty((H, K), dtype=torch.float32, device=dev) for H, K in dims]       # cleared by the sweep's launches
            dbs = [torch.empty((H,), dtype=torch.float32, device=dev) for H, _ in dims]
        ints = lambda v: (ctypes.c_int * L)(*v)  # noqa: E731
        # the backward sweep and all weight gradients from ONE native call (gcnpt_layers_bwd; _bwd_dz: its first tensor already is dZ)
        _lib.check((lib.gcnpt_layers_bwd if pool is None else lib.gcnpt_layers_bwd_dz)(
            st, L, _lib.ptr(g), _lib.ptr_array(list(outs)), ints([_lib.dtype_code(o.dtype) for o in outs]), _lib.ptr_array(list(wb)),
            _lib.ptr(trees.ell), _lib.ptr(trees.rowT_ptr), _lib.ptr(trees.colT_idx), _lib.ptr(g_ellT), B, T, ints([k for _, k in dims]),
            ints([h for h, _ in dims]), _lib.ptr_array(dhs), ints([_lib.dtype_code(t) for t in in_dtypes]), compute,
            (ctypes.c_float * L)(*scales), _lib.ptr_array(z_frag) if want_w else None, _lib.ptr_array(list(s_frag)) if want_w else None,
            _lib.ptr_array(dWs) if want_w else None, _lib.ptr_array(dbs) if want_w else None))
        g = dhs[0]
        grads = [None] * (2 * L)
        if want_w:
            for l in range(L):
                grads[2 * l] = dWs[l].to(ctx.param_dtypes[2 * l])
                grads[2 * l + 1] = dbs[l].to(ctx.param_dtypes[2 * l + 1])
        return (g if ctx.needs_input_grad[0] else None, None, None) + tuple(grads)


def gcn_layers(x, weights, biases, trees, drop_p=None, seeds=None, compute_dtype=torch.float32, out_dtype=torch.float32, no_adj=False,
               seed_dev=None, wcache=None, pool=None, prepacked=None, acts=None):
    """
    The reference's whole layer loop (model/gcn.py:266-393) over the per-layer kernels.  x [B,T,Din] float32/bfloat16 CUDA;
    weights / biases: lists of the nn.Linear parameters (any widths that chain); drop_p[l]: dropout applied to the output of
    layer l (0 for the last); acts: None or a list that receives every layer's stored output; wcache: an empty dict the caller keeps -- the packed weight images are reused while the weights'
    version counters stand still; pool: see below.  compute_dtype float32 = exact fp32 MFMA (activations stay float32), bfloat16 = bf16 operands
    and bf16 activations between the layers, fp32 accumulation; the last layer's output has out_dtype.
    """
    if not isinstance(trees, (PrunedTrees, PackedTrees)):
        raise TypeError("trees must be a PrunedTrees (see model.tree.prune_to_csr / adj_to_csr) or a PackedTrees (PrunedTrees.pack)")
    _lib.require_gpu(x)
    L = len(weights)
    compute = _lib.dtype_code(compute_dtype)
    if compute == _lib.F32:
        x, out_dtype = (x.float() if x.dtype != torch.float32 else x), torch.float32
    cfg = dict(drop_p=[float(p) for p in (drop_p or [0.0] * L)], seed=[int(s) for s in (seeds or [0] * L)], compute=compute,
               mid_dtype=torch.float32 if compute == _lib.F32 else torch.bfloat16, out_dtype=out_dtype, no_adj=bool(no_adj),
               seed_dev=seed_dev, wcache=wcache, pool=None, prepacked=prepacked, acts=acts)
    if pool is not None:
        # pool = (subj_pos, obj_pos, 'max' | 'avg' | 'sum'): the op returns float32 [B, 3H] = pool3(h_L, trees.pool_mask, subj_pos, obj_pos)
        # instead of h_L (padded layout only)
        if getattr(trees, "packed", False):
            raise ValueError("gcn_layers(pool=...) needs the padded layout")
        sp, op, kind = pool
        cfg["pool"] = (sp.contiguous(), op.contiguous(), 0 if kind == 'max' else (1 if kind == 'avg' else 2))
    params = [t for wb in zip(weights, biases) for t in wb]
    return _GCNLayersFn.apply(x, trees, cfg, *params)


def gcn_layers_with_acts(x, weights, biases, trees, **kw):
    """gcn_layers plus the list of every layer's stored output (detached; the last one is the op's result before any pooling)."""
    acts = []
    out = gcn_layers(x, weights, biases, trees, acts=acts, **kw)
    return out, [a.detach() for a in acts]


class _EmbedFn(torch.autograd.Function):
    """nn.Embedding lookup (model/gcn.py:235-239) whose backward is ONE index_add_ of float atomics.  PyTorch-ROCm 2.10's own
    embedding backward takes ~140 us per table at the 1.5-5 k indices of a batch (embedding_backward_feature_kernel: three
    tables = half of the no-LSTM training step) and above 3072 indices a rocPRIM sort path that faults when replayed in a
    hipGraph; same sums, same zero row for padding_idx.
    sparse = True: the gradient is a ROW-SPARSE tensor instead (indices = the batch's token ids, values = their gradient rows; rows
    >= topn and the padding row zeroed, gcn.py:84-88): a data-parallel loop then exchanges (row ids, rows) with
    shard.SparseRowExchange instead of all-reducing the dense [V, E] table."""

    @staticmethod
    def forward(ctx, weight, idx, padding_idx, sparse=False, topn=None):
        ctx.save_for_backward(idx)
        ctx.table, ctx.pad, ctx.sparse, ctx.topn = tuple(weight.shape), padding_idx, sparse, topn
        return torch.nn.functional.embedding(idx, weight)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        flat, rows = idx.reshape(-1), g.reshape(-1, g.shape[-1])
        if ctx.sparse:
            keep = torch.ones_like(flat, dtype=torch.bool)
            if ctx.pad is not None:
                keep &= flat != ctx.pad
            if ctx.topn is not None:
                keep &= flat < ctx.topn
            rows = rows * keep.unsqueeze(1).to(rows.dtype)
            return torch.sparse_coo_tensor(flat.unsqueeze(0), rows, size=ctx.table), None, None, None, None
        gw = torch.zeros(ctx.table, dtype=g.dtype, device=g.device)
        gw.index_add_(0, flat, rows)
        if ctx.pad is not None:
            gw[ctx.pad].zero_()
        return gw, None, None, None, None


def _embed(table, idx, sparse=False, topn=None):
    """table: nn.Embedding.  Same result as table(idx)."""
    if idx.is_cuda and table.weight.requires_grad and torch.is_grad_enabled():
        return _EmbedFn.apply(table.weight, idx, table.padding_idx, sparse, topn)
    return table(idx)


# ------------------------------------------------------------------------------------------------------
# modules with the reference's names and state_dict layout
# ------------------------------------------------------------------------------------------------------
class GCNClassifier(nn.Module):
    """Reference model/gcn.py:15-36."""

    def __init__(self, opt, emb_matrix=None):
        super().__init__()
        self.gcn_model = GCNRelationModel(opt, emb_matrix=emb_matrix)
        self.classifier = nn.Linear(opt['hidden_dim'], opt['num_class'])
        self.opt = opt

    def conv_l2(self):
        return self.gcn_model.gcn.conv_l2()

    def forward(self, inputs, trees=None):
        outputs, pooling_output = self.gcn_model(inputs, trees)
        return self.classifier(outputs), pooling_output

    def get_deprel_emb(self):
        return self.gcn_model.get_deprel_embedding()

    def get_gcn_parameters(self):
        return self.gcn_model.get_gcn_parameters()


class GCNRelationModel(nn.Module):
    """Reference model/gcn.py:38-126; the tree build of lines 96-112 runs on the device."""

    def __init__(self, opt, emb_matrix=None):
        super().__init__()
        self.opt = opt
        self.emb_matrix = emb_matrix
        self.adj_type = opt.get('adj_type', 'regular')
        if self.adj_type not in ('regular', 'diagonal_deprel', 'full_deprel'):
            raise ValueError('Adjacency aggregation type not supported.')            # gcn.py:388 (concat_deprel is dead code there)
        self.emb = nn.Embedding(opt['vocab_size'], opt['emb_dim'], padding_idx=constant.PAD_ID)
        self.pos_emb = nn.Embedding(constant.N_POS, opt['pos_dim']) if opt['pos_dim'] > 0 else None
        self.ner_emb = nn.Embedding(constant.N_NER, opt['ner_dim']) if opt['ner_dim'] > 0 else None
        # gcn.py:48-56: the reference keeps a 1-wide dummy table on the regular path (kept for checkpoints);
        # diagonal_deprel scales hidden vectors element-wise, so its table is hidden_dim wide; full_deprel mixes deprel_emb_dim matrices
        width = {'regular': 1, 'diagonal_deprel': opt['hidden_dim']}.get(self.adj_type) or opt['deprel_emb_dim']
        self.deprel_emb = nn.Embedding(constant.N_DEPREL, width, padding_idx=0)
        self.init_embeddings()
        self.gcn = GCN(opt, (self.emb, self.pos_emb, self.ner_emb, self.deprel_emb), opt['hidden_dim'], opt['num_layers'])
        mlp = [nn.Linear(opt['hidden_dim'] * 3, opt['hidden_dim']), nn.ReLU()]
        for _ in range(opt['mlp_layers'] - 1):
            mlp += [nn.Linear(opt['hidden_dim'], opt['hidden_dim']), nn.ReLU()]
        self.out_mlp = nn.Sequential(*mlp)

    def get_deprel_embedding(self):
        return self.deprel_emb.weight

    def get_gcn_parameters(self):
        return self.gcn.get_gcn_parameters()

    def init_embeddings(self):
        """gcn.py:73-88: uniform(-1,1) or the given matrix; topn decides which rows are fine-tuned."""
        with torch.no_grad():
            if self.emb_matrix is None:
                self.emb.weight[1:, :].uniform_(-1.0, 1.0)
            else:
                m = self.emb_matrix
                self.emb.weight.copy_(torch.from_numpy(m) if isinstance(m, np.ndarray) else m)
        topn = self.opt.get('topn', self.opt['vocab_size'])
        if topn <= 0:
            self.emb.weight.requires_grad = False
        elif topn < self.opt['vocab_size'] and not self.opt.get('gcn_sparse_emb_grad', False):
            def keep_top(grad, n=topn):
                grad = grad.clone()
                grad[n:].zero_()
                return grad
            self.emb.weight.register_hook(keep_top)

    def forward(self, inputs, trees=None):
        """`trees`: optional PrunedTrees of this batch from a TreeCache (model.tree, loader-side pre-pruning); by default
        the batch is pruned here."""
        if self.opt['dataset'] == 'tacred':
            words, masks, pos, ner, deprel, head, subj_pos, obj_pos = inputs
        else:
            words, masks, pos, deprel, head, subj_pos, obj_pos = inputs
        pack = None
        if trees is None:
            # lengths, head_to_tree, tree_to_adj and the upload (gcn.py:96-112) in one launch, no host round trip; in a training step the
            # same launch packs the layer weights on the CUs the tree build leaves idle (one launch boundary less)
            pack = self.gcn.weight_pack(head.shape[0], head.shape[1]) if self.opt.get('gcn_pack_with_trees', True) else None
            if self.opt.get('gcn_packed', False) and self.adj_type == 'regular' and not self.opt.get('gcn_pooled_only', False):
                # token-packed rows: the layers run on sum(len) rows instead of B*T (one host sync for sum(len), where the reference syncs
                # for the lengths anyway, gcn.py:96); the SAME launch writes the packed layout (and packs the weights)
                trees = prune_to_csr_packed(head, subj_pos, obj_pos, deprel, self.opt['prune_k'], masks.eq(0).sum(1), masks=masks, pack=pack)
                if self.opt.get('gcn_check_trees', True):
                    trees.padded.check(expect_maxlen=head.shape[1])
                    trees.check()
                return self._forward_trees(inputs, trees, pack, subj_pos, obj_pos)
            trees = prune_to_csr(head, subj_pos, obj_pos, deprel, self.opt['prune_k'], masks=masks,
                                 want_label=self.adj_type != 'regular', pack=pack)
            if self.opt.get('gcn_check_trees', True):
                trees.check(expect_maxlen=head.shape[1])
            if self.opt.get('gcn_pooled_only', False):
                # one host sync for the width (a TreeCache avoids it); entity tokens are kept even when the tree does not hold them
                trees = trees.compact(also_keep=(subj_pos == 0) | (obj_pos == 0))
        else:
            if (trees.B, trees.T) != tuple(head.shape):
                raise ValueError("trees are for a [%d,%d] batch, the inputs are %s" % (trees.B, trees.T, tuple(head.shape)))
            if self.opt.get('gcn_check_trees', True):
                trees.check() if isinstance(trees, PackedTrees) else trees.check(expect_maxlen=head.shape[1])
        return self._forward_trees(inputs, trees, pack, subj_pos, obj_pos)

    def _forward_trees(self, inputs, trees, pack, subj_pos, obj_pos):
        # regular adjacency on the padded layout: the GCN returns the three pooled vectors itself (stack + pooling as ONE autograd op, so the
        # pooling's backward hands the top layer its dZ); opt['gcn_pool_handover'] = False keeps the two ops apart
        handover = (self.adj_type == 'regular' and isinstance(trees, (PrunedTrees, CompactTrees)) and self.opt.get('gcn_pool_handover', True))
        if isinstance(trees, CompactTrees):
            # only the tokens of the pruned trees and the entity tokens are computed ([B,Tc,H]); the three poolings never look at
            # any other (gcn.py:116-119), so the pooled vectors are the full batch's
            subj_pos, obj_pos = trees.take(subj_pos, fill=150), trees.take(obj_pos, fill=150)      # 150: the loader's pad value (loader.py:120-121)
        self.gcn._pool_req = (subj_pos, obj_pos, self.opt['pooling']) if handover else None
        if pack is not None:
            self.gcn.use_weight_pack(pack)
        try:
            h, pool_mask = self.gcn(trees, inputs)
        finally:
            self.gcn._pool_req = None
            if pack is not None:
                self.gcn.use_weight_pack(None)
        if isinstance(h, _Pooled):
            pooled = h.value
            return self.out_mlp(pooled), pooled[:, :self.opt['hidden_dim']]
        # gcn.py:116-121: three masked poolings and the concat, one pass over h (masks straight from the position tensors)
        pooled = pool3(h, pool_mask, subj_pos, obj_pos, type=self.opt['pooling'])
        h_out = pooled[:, :self.opt['hidden_dim']]
        return self.out_mlp(pooled), h_out


class _Pooled(object):
    """Marker: GCN.forward was asked (by GCNRelationModel) for pool3(h_L, ...) and returns it in place of h_L."""

    def __init__(self, value):
        self.value = value


class GCN(nn.Module):
    """Reference model/gcn.py:128-395, `regular` and `diagonal_deprel` adjacency types.  `adj` may be a dense
    float32 [B,T,T] tensor (as the reference passes) or a PrunedTrees from model.tree."""

    def __init__(self, opt, embeddings, mem_dim, num_layers):
        super().__init__()
        self.opt = opt
        self.layers = num_layers
        self.use_cuda = opt.get('cuda', True)
        self.mem_dim = mem_dim
        tacred = opt['dataset'] == 'tacred'
        self.in_dim = opt['emb_dim'] + opt['pos_dim'] + (opt['ner_dim'] if tacred else 0)
        self.emb, self.pos_emb, self.ner_emb, self.deprel_emb = embeddings
        self.adj_type = opt.get('adj_type', 'regular')
        if self.adj_type not in ('regular', 'diagonal_deprel', 'full_deprel'):
            raise ValueError('Adjacency aggregation type not supported.')            # gcn.py:388
        if opt.get('rnn', False):
            self.rnn = nn.LSTM(self.in_dim, opt['rnn_hidden'], opt['rnn_layers'], batch_first=True,
                               dropout=opt['rnn_dropout'], bidirectional=True)
            self.in_dim = opt['rnn_hidden'] * 2
            self.rnn_drop = nn.Dropout(opt['rnn_dropout'])
        self.in_drop = nn.Dropout(opt['input_dropout'])
        self.gcn_drop = nn.Dropout(opt['gcn_dropout'])
        self.emb_dropout = opt.get('emb_dropout', 0.0)
        if self.adj_type == 'diagonal_deprel':                  # gcn.py:153-155: no per-layer weights in this variant
            self.preprocessor = nn.Linear(self.in_dim, mem_dim)
            self.in_dim = mem_dim
        elif self.adj_type == 'full_deprel':                    # gcn.py:156-167: ONE Linear, reshaped to [D, in_dim, H], for every layer
            self.W = nn.Linear(self.in_dim, opt['deprel_emb_dim'] * mem_dim, bias=True)
        else:
            self.W = nn.ModuleList(nn.Linear(self.in_dim if l == 0 else mem_dim, mem_dim) for l in range(num_layers))
        kind = opt.get('gcn_dtype', 'fp32')
        if kind not in ('fp32', 'bf16'):
            raise ValueError("gcn_dtype must be 'fp32' or 'bf16'")
        self.compute_dtype = torch.float32 if kind == 'fp32' else torch.bfloat16
        self._wcache = {}             # packed weight images, valid while the weights' version counters stand still
        # graph-safe dropout: a by-value seed is frozen into a captured hipGraph, so with opt['gcn_graph_rng'] the per-layer
        # seeds are fixed at construction and a device counter (advanced by one captured op per forward) is added in the kernel
        self.graph_rng = bool(opt.get('gcn_graph_rng', False))
        if self.graph_rng:
            self.register_buffer('_rng_step', torch.zeros(1, dtype=torch.int64), persistent=False)
            self._base_seeds = [int(torch.randint(0, 2 ** 62, (1,)).item()) for _ in range(num_layers)]

    def conv_l2(self):
        # the reference's diagonal_deprel model has no W list, so its conv_l2() / get_gcn_parameters() raise AttributeError
        # (gcn.py:180-184, 397-398), and full_deprel's single nn.Linear is not iterable (TypeError); the same happens here
        return sum(p.pow(2).sum() for lin in self.W for p in (lin.weight, lin.bias))

    def get_gcn_parameters(self):
        return self.W

    def _word_embeddings(self, words):
        """EmbeddingDropout of the reference (model/dropouts.py:23-39): in training, every word TYPE of a
        sentence is dropped with probability emb_dropout and the rest is scaled by 1/(1-p)."""
        # opt['gcn_sparse_emb_grad']: the word table's gradient as (token ids, rows) -- what a data-parallel loop exchanges with
        # shard.SparseRowExchange instead of the dense [V, 300] all-reduce; topn is then applied to the rows here, not by the hook
        sparse = bool(self.opt.get('gcn_sparse_emb_grad', False))
        topn = self.opt.get('topn', self.opt['vocab_size'])
        embs = _embed(self.emb, words, sparse, topn if (sparse and 0 < topn < self.opt['vocab_size']) else None)
        p = self.emb_dropout
        if not self.training or p <= 0.0:
            return embs
        keep = torch.empty((words.shape[0], self.emb.num_embeddings), device=words.device).bernoulli_(1 - p)
        return embs * torch.gather(keep, 1, words).unsqueeze(-1) / (1 - p)

    def encode_with_rnn(self, rnn_inputs, masks, batch_size):
        lens = masks.eq(0).long().sum(1).cpu()
        shape = (self.opt['rnn_layers'] * 2, batch_size, self.opt['rnn_hidden'])
        h0 = rnn_inputs.new_zeros(shape)
        packed = nn.utils.rnn.pack_padded_sequence(rnn_inputs, lens, batch_first=True, enforce_sorted=False)
        out, _ = self.rnn(packed, (h0, h0.clone()))
        out, _ = nn.utils.rnn.pad_packed_sequence(out, batch_first=True)
        return out

    def weight_pack(self, B=None, T=None):
        """A WeightPack for the launch that builds this step's trees (`prune_to_csr(..., pack=)`, `TreeCache.batch(..., pack=)`), or None
        when the next forward would not use one: only the `regular` per-layer kernels pack this way, and only when the cached images do
        not already belong to the current weights (eval(), gradient accumulation).  Hand the request back with `use_weight_pack`."""
        if self.adj_type != 'regular':
            return None
        dev = self.W[0].weight.device
        if dev.type != 'cuda':
            return None
        Ws = [lin.weight for lin in self.W]
        compute = _lib.dtype_code(self.compute_dtype)
        if self._cache_usable() and self._wcache.get("key") == WeightPack.key_of(Ws, compute, dev):
            return None
        return WeightPack(Ws, compute, dev)

    def use_weight_pack(self, pack):
        """The next forward takes the weight images from `pack` (a WeightPack a tree launch has filled); None clears it."""
        self._prepacked = pack

    def _cache_usable(self):
        """Whether this forward may reuse the packed weight images of an earlier one.  The cache is keyed on the weights' storage and
        version counters, and an in-place update through `p.data` (the reference's own MyAdagrad does that, utils/torch_utils.py:84-88;
        so do EMA swaps and hand-written clipping) changes the values WITHOUT bumping the counter -- in train() and in eval() alike
        (`eval -> p.data.copy_(ema) -> eval` is the usual evaluation of an averaged model).  So every forward re-packs -- one 4 us launch,
        none at all when the tree launch of the same forward carries it (gcn_pack_with_trees) -- unless opt['gcn_reuse_packed_weights']
        says the caller knows its weights stand still between forwards (gradient accumulation, a frozen model serving requests); such a
        caller calls invalidate_weight_cache() after writing to `p.data` by hand."""
        if torch.cuda.is_current_stream_capturing():
            return False                                                  # a captured step must contain its pack
        ok = bool(self.opt.get('gcn_reuse_packed_weights', False))
        if not ok:
            self._wcache.clear()
        return ok

    def invalidate_weight_cache(self):
        self._wcache.clear()

    def _dropout_plan(self):
        """(p per layer, seed per layer, device seed word or None) for this forward -- gcn.py:393: every layer but the last."""
        ps = [self.gcn_drop.p if (self.training and l < self.layers - 1) else 0.0 for l in range(self.layers)]
        if self.graph_rng:
            if any(p > 0 for p in ps):
                self._rng_step.add_(1)                                # on the stream: part of a captured graph
            return ps, [s if p > 0 else 0 for s, p in zip(self._base_seeds, ps)], self._rng_step
        return ps, [int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0 for p in ps], None      # CPU generator: no GPU sync

    def _forward_diagonal(self, adj, gcn_inputs, deprel):
        """gcn.py:255-257, 272-294: Linear preprocessor (host BLAS), then the element-wise relation-scaled layers.
        no_adj has no effect on this variant in the reference either (it only zeroes the matrix the regular path uses)."""
        trees = adj if isinstance(adj, PrunedTrees) else adj_to_csr(adj, want_label=True)
        x = self.preprocessor(gcn_inputs).to(self.compute_dtype)
        table = self.deprel_emb.weight            # padding_idx=0: nn.Embedding's backward would zero row 0, the hook below does
        if table.requires_grad and not getattr(self, '_pad_hooked', False):
            table.register_hook(_zero_pad_row)
            self._pad_hooked = True
        ps, seeds, seed_dev = self._dropout_plan()
        for l in range(self.layers):
            x = diag_layer(x, table, deprel, trees, ps[l], seeds[l], seed_dev)
        return x.float(), trees.pool_mask

    def _forward_full(self, adj, gcn_inputs, deprel, all_tokens=False):
        """adj_type='full_deprel', gcn.py:296-388 + traverse_deprel / traverse_self_loop 400-434; the traversal's contraction runs on
        csrc/bilinear_kernels.hip in the module's precision: exact fp32 MFMA (default) or, with opt['gcn_dtype']='bf16', bf16 operands.
        trav(x, e)[n] = sum_d e[n,d] (x[n] W3[d] + b3[d]) is only needed for tokens that sit in a pruned tree, so those are
        compacted (their number is read back together with the tree check's status word; with opt['gcn_check_trees']=False or
        CompactTrees, whose rows are those tokens already, every row is traversed and there is no host sync at all).  The
        contraction runs on the hand-written kernels in both precisions (a library GEMM only when Tin > 256).  Everything around it -- aggregation of the traversed rows over the forward / reverse entries of the
        device pruner's CSR (picked by label range), edge dropout, the self-loop term, /(deg+1), ReLU, dropout -- is ONE kernel
        (csrc/full_kernels.hip, gcnpt_full_agg_fwd / _bwd).  The reference materialises [B,T,D,Tin] for all tokens and multiplies
        dense [B,T,T] matrices instead."""
        opt = self.opt
        trees = adj if isinstance(adj, PrunedTrees) else adj_to_csr(adj, want_label=True)
        if trees.label is None:
            raise ValueError("full_deprel needs the adjacency values: build the trees with want_label=True")
        B, T, cap = trees.B, trees.T, trees.cap
        N, D, H = B * T, opt['deprel_emb_dim'], self.mem_dim
        dev = trees.device
        max_depth = opt.get('deprel_max_depth', 2)
        directed, self_loop = bool(opt.get('deprel_directed', False)), bool(opt.get('deprel_self_loop', True))
        if all_tokens or not opt.get('gcn_check_trees', True):
            # no compaction, no sync: every row is traversed (CompactTrees: the rows ARE the tokens of the pruned trees)
            tok = torch.arange(N, device=dev)
            pos = tok.to(torch.int32)
        else:
            tok = torch.nonzero(~trees.pool_mask.view(-1)).squeeze(1)                            # tokens of the pruned trees
            pos = torch.full((N,), -1, dtype=torch.int32, device=dev)
            pos[tok] = torch.arange(tok.numel(), device=dev, dtype=torch.int32)
        M = int(tok.numel())
        deprel_tok = deprel.reshape(-1)[tok]
        b3 = self.W.bias.reshape(D, H)                                                            # gcn.py:303
        x = gcn_inputs.to(torch.float32)
        ps, seeds, seed_dev = self._dropout_plan()
        for l in range(self.layers):
            Tin = x.shape[-1]
            if self.W.weight.shape[1] != Tin:       # the reference's einsum fails the same way when in_dim != mem_dim (layer 1)
                raise RuntimeError("full_deprel: layer %d input has %d features but the shared weight is [D, %d, H] (gcn.py:167, 301)"
                                   % (l, Tin, self.W.weight.shape[1]))
            Wk = self.W.weight.reshape(D * Tin, H)                                                # W3[d,t,h] flattened over (d,t), gcn.py:301
            xf = x.reshape(N, Tin)
            xt = xf[tok]
            plain = l >= max_depth                                                                # gcn.py:323-324, 355-356, 371-374
            ys, keeps = [None, None], [None, None]
            for k, (shift, on) in enumerate(((0, True), (constant.DEPREL_FORWARD_BOUND, not directed))):
                if not on or M == 0:
                    continue
                e = _embed(self.deprel_emb, deprel_tok + shift)
                keep_prop = opt.get('deprel_keep_prop', 1.0)
                if self.training and keep_prop < 1.0:                                             # maybe_forget_deprels, gcn.py:451-470
                    kept = torch.empty((M, 1), device=dev).bernoulli_(keep_prop) == 1
                    e = torch.where(kept, e, torch.ones_like(e))
                if plain:
                    e = torch.ones_like(e)
                if bilinear_supported(D, Tin, H, self.compute_dtype):
                    # hand-written MFMA contraction in the module's precision (exact fp32 MFMA by default): e (x) x never exists
                    ys[k] = bilinear_traverse(xt, e, self.W.weight, self.W.bias, self.compute_dtype)
                else:                                                                             # Tin > 256: one library GEMM, gcn.py:408-414
                    ys[k] = torch.mm((e.unsqueeze(2) * xt.unsqueeze(1)).reshape(-1, D * Tin), Wk) + torch.mm(e, b3)
                edge_keep = opt.get('edge_keep_prob', 1.0)
                if self.training and edge_keep < 1.0:                                             # maybe_drop_edges, gcn.py:436-449
                    keeps[k] = torch.empty((B * cap,), device=dev).bernoulli_(edge_keep).to(torch.uint8)
            self_term = None
            if self_loop:                                                                         # gcn.py:366-385, 417-434 (plain GEMMs)
                se = torch.ones((D,), device=dev) if plain else self.deprel_emb.weight[constant.SELF_LOOP_INDEX]
                self_term = torch.mm(xf, torch.mm(se.unsqueeze(0), self.W.weight.reshape(D, Tin * H)).reshape(Tin, H)) + torch.mv(b3.t(), se)
            if ys[0] is None:
                ys[0] = torch.zeros((1, H), dtype=torch.float32, device=dev)                      # no token sits in a tree
            x = _FullAggFn.apply(ys[0], ys[1], self_term, trees, pos, keeps[0], keeps[1], M if M > 0 else 0, ps[l], seeds[l], seed_dev)
            x = x.view(B, T, H)
        return x, trees.pool_mask

    def forward(self, adj, inputs):
        if self.opt['dataset'] == 'tacred':
            words, masks, pos, ner, deprel, head, subj_pos, obj_pos = inputs
        else:
            words, masks, pos, deprel, head, subj_pos, obj_pos = inputs
            ner = None
        ct = adj if isinstance(adj, CompactTrees) else None
        use_rnn = bool(self.opt.get('rnn', False))
        if ct is not None:
            # "pooled-only" rows: the layers run on the kept tokens.  Without an LSTM in front the lookups themselves are
            # done for those tokens only; with one, its outputs are gathered
            adj, deprel = ct.trees, ct.take(deprel)
            if not use_rnn:
                words, pos = ct.take(words), ct.take(pos)
                ner = ct.take(ner) if ner is not None else None
        parts = [words if words.dim() > 2 else self._word_embeddings(words)]    # gcn.py:235-239
        if self.opt['pos_dim'] > 0:
            parts.append(_embed(self.pos_emb, pos))
        if self.opt['ner_dim'] > 0 and ner is not None:
            parts.append(_embed(self.ner_emb, ner))
        embs = self.in_drop(torch.cat(parts, dim=2))
        if use_rnn:
            gcn_inputs = self.rnn_drop(self.encode_with_rnn(embs, masks, words.size(0)))
            if ct is not None:
                gcn_inputs = ct.take(gcn_inputs)
        else:
            gcn_inputs = embs

        packed = adj if isinstance(adj, PackedTrees) else None
        if packed is not None and self.adj_type != 'regular':
            adj = packed.padded                                  # the deprel variants run on the padded layout
        if self.adj_type == 'diagonal_deprel':
            return self._forward_diagonal(adj, gcn_inputs, deprel)
        if self.adj_type == 'full_deprel':
            return self._forward_full(adj, gcn_inputs, deprel, all_tokens=ct is not None)
        if packed is not None:
            # pad / unpad only here, at the module boundary: the layer loop sees [sum(len), width] rows, the caller [B,T,H] (gcn.py:395)
            no_adj = bool(self.opt.get('no_adj', False))
            ps, seeds, seed_dev = self._dropout_plan()
            Ws, bs = [lin.weight for lin in self.W], [lin.bias for lin in self.W]
            xp = packed.pack_rows(gcn_inputs if gcn_inputs.dtype in (torch.float32, torch.bfloat16) else gcn_inputs.float())
            pre, self._prepacked = getattr(self, "_prepacked", None), None
            wcache = self._wcache if self._cache_usable() else None
            hp = gcn_layers(xp, Ws, bs, packed, ps, seeds, self.compute_dtype, torch.float32, no_adj, seed_dev, wcache, prepacked=pre)
            return packed.unpack_rows(hp), packed.padded.pool_mask
        trees = adj if isinstance(adj, PrunedTrees) else adj_to_csr(adj, want_label=False)   # gcn.py:260-262
        no_adj = bool(self.opt.get('no_adj', False))                                           # gcn.py:264-265
        x = gcn_inputs
        B, T, Din = x.shape
        ps, seeds, seed_dev = self._dropout_plan()
        Ws, bs = [lin.weight for lin in self.W], [lin.bias for lin in self.W]
        pre, self._prepacked = getattr(self, "_prepacked", None), None      # a request serves ONE forward
        wcache = self._wcache if self._cache_usable() else None
        req = getattr(self, "_pool_req", None)
        if req is not None:
            # GCNRelationModel asked for the pooled vectors directly (it would pool h next, gcn.py:116-121): stack + pooling as one op
            return _Pooled(gcn_layers(x, Ws, bs, trees, ps, seeds, self.compute_dtype, torch.float32, no_adj, seed_dev, wcache, pool=req,
                                      prepacked=pre)), trees.pool_mask
        x = gcn_layers(x, Ws, bs, trees, ps, seeds, self.compute_dtype, torch.float32, no_adj, seed_dev, wcache, prepacked=pre)
        return x, trees.pool_mask


class _FullAggFn(torch.autograd.Function):
    """Aggregation + self loop + /(deg+1) + ReLU + dropout of a full_deprel layer in one kernel (csrc/full_kernels.hip); reference
    model/gcn.py:308-311, 331, 340-344, 362, 385, 390-393."""

    @staticmethod
    def forward(ctx, yf, yr, self_term, trees, pos, keep_f, keep_r, M, drop_p, seed, seed_dev):
        B, T = trees.B, trees.T
        H = yf.shape[1]
        P = _lib.ptr
        yf = yf.to(torch.float32).contiguous()
        yr = yr.to(torch.float32).contiguous() if yr is not None else None
        st = self_term.to(torch.float32).contiguous() if self_term is not None else None
        out = torch.empty((B * T, H), dtype=torch.float32, device=yf.device)
        _lib.check(_lib.lib().gcnpt_full_agg_fwd(_lib.stream(), P(yf), P(yr), P(st), P(pos), P(trees.row_ptr), P(trees.col_idx), P(trees.label),
                                                 P(keep_f), P(keep_r), B, T, H, int(M), P(out), float(drop_p), int(seed), P(seed_dev)))
        ctx.save_for_backward(out, pos, keep_f, keep_r)
        ctx.trees, ctx.M, ctx.rows = trees, int(M), (yf.shape[0], yr is not None, st is not None)
        ctx.scale = 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0
        return out

    @staticmethod
    def backward(ctx, g):
        out, pos, keep_f, keep_r = ctx.saved_tensors
        trees = ctx.trees
        B, T, H = trees.B, trees.T, out.shape[1]
        rows, has_r, has_self = ctx.rows
        P = _lib.ptr
        g = g.to(torch.float32).contiguous()
        dagg = torch.empty_like(out)
        dyf = torch.zeros((rows, H), dtype=torch.float32, device=out.device)
        dyr = torch.zeros((rows, H), dtype=torch.float32, device=out.device) if has_r else None
        _lib.check(_lib.lib().gcnpt_full_agg_bwd(_lib.stream(), P(g), P(out), P(pos), P(trees.row_ptr), P(trees.col_idx), P(trees.label),
                                                 P(keep_f), P(keep_r), B, T, H, ctx.M, ctx.scale, P(dagg), P(dyf), P(dyr)))
        return dyf, dyr, (dagg if has_self else None), None, None, None, None, None, None, None, None


class _BilinearFn(torch.autograd.Function):
    """y = sum_d e[:,d] * (x @ W3[d]) + e @ b3 (reference traverse_deprel, model/gcn.py:400-415) with the contraction on
    csrc/bilinear_kernels.hip in either precision: exact fp32 MFMA (the module's default) or bf16 operands with fp32 accumulation.
    The outer product e (x) x the reference materialises never exists.  The op is linear in each argument: dx is the same kernel on
    the transposed weight image, de the same per-relation products dotted with the upstream gradient, dW the token contraction
    of row-contraction fragment images with the gy fragments scaled by e in registers (gcnpt_bilinear_bwd_w)."""

    @staticmethod
    def forward(ctx, xt, e, weight, bias, compute):
        M, Tin = xt.shape
        D = e.shape[1]
        H = weight.shape[0] // D
        lib, st, dev = _lib.lib(), _lib.stream(), xt.device
        k = 32 if compute == _lib.BF16 else 16
        op_dtype = torch.bfloat16 if compute == _lib.BF16 else torch.float32
        w32 = weight.detach().to(torch.float32).contiguous()
        img = torch.empty((lib.gcnpt_bilinear_packed_bytes(D, Tin, H, compute),), dtype=torch.uint8, device=dev)
        _lib.check(lib.gcnpt_bilinear_pack(st, _lib.ptr(w32), D, Tin, H, _lib.ptr(img), 0, compute))
        xb = torch.zeros((M, (Tin + k - 1) // k * k), dtype=op_dtype, device=dev)
        xb[:, :Tin] = xt.detach()
        e32 = e.detach().to(torch.float32).contiguous()
        planes = torch.empty((lib.gcnpt_bilinear_planes(M, D, Tin, H, compute), M, H), dtype=torch.float32, device=dev)
        _lib.check(lib.gcnpt_bilinear_fwd(st, _lib.ptr(xb), _lib.ptr(e32), _lib.ptr(img), M, D, Tin, H, _lib.ptr(planes), compute))
        y = torch.addmm(planes.sum(0), e32, bias.detach().to(torch.float32).reshape(D, H))      # + e @ b3, gcn.py:413
        ctx.save_for_backward(xt, e, weight, bias, xb, img)
        ctx.compute = compute
        return y

    @staticmethod
    def backward(ctx, gy):
        xt, e, weight, bias, xb, img = ctx.saved_tensors
        compute = ctx.compute
        M, Tin = xt.shape
        D = e.shape[1]
        H = weight.shape[0] // D
        lib, st, dev = _lib.lib(), _lib.stream(), xt.device
        k = 32 if compute == _lib.BF16 else 16
        op_dtype = torch.bfloat16 if compute == _lib.BF16 else torch.float32
        gy = gy.to(torch.float32).contiguous()
        x32, e32 = xt.to(torch.float32), e.to(torch.float32).contiguous()
        w32 = weight.detach().to(torch.float32).contiguous()
        b3 = bias.to(torch.float32).reshape(D, H)
        dx = de = dW = db = None
        on_kernel = bool(lib.gcnpt_bilinear_supported(D, H, Tin, compute))         # the transposed problem contracts over H
        if ctx.needs_input_grad[0]:
            if on_kernel:       # dx = sum_d e_d (gy @ W3[d]^T): the forward kernel on the transposed weight image
                imgT = torch.empty((lib.gcnpt_bilinear_packed_bytes(D, H, Tin, compute),), dtype=torch.uint8, device=dev)
                _lib.check(lib.gcnpt_bilinear_pack(st, _lib.ptr(w32), D, Tin, H, _lib.ptr(imgT), 1, compute))
                gyb = torch.zeros((M, (H + k - 1) // k * k), dtype=op_dtype, device=dev)
                gyb[:, :H] = gy
                planes = torch.empty((lib.gcnpt_bilinear_planes(M, D, H, Tin, compute), M, Tin), dtype=torch.float32, device=dev)
                _lib.check(lib.gcnpt_bilinear_fwd(st, _lib.ptr(gyb), _lib.ptr(e32), _lib.ptr(imgT), M, D, H, Tin, _lib.ptr(planes), compute))
                dx = planes.sum(0)
            else:
                dx = (torch.mm(gy, w32.reshape(D * Tin, H).t()).view(M, D, Tin) * e32.unsqueeze(2)).sum(1)
        if ctx.needs_input_grad[1]:
            # de[m,d] = (x[m] @ W3[d]) . gy[m] + gy[m] . b3[d]: the forward's per-relation products, dotted instead of summed
            planes = torch.empty((lib.gcnpt_bilinear_de_planes(M, D, Tin, H, compute), M, D), dtype=torch.float32, device=dev)
            _lib.check(lib.gcnpt_bilinear_bwd_e(st, _lib.ptr(xb), _lib.ptr(gy), _lib.ptr(img), M, D, Tin, H, _lib.ptr(planes), compute))
            de = torch.addmm(planes.sum(0), gy, b3.t())
        if ctx.needs_input_grad[2]:                                       # dW3[d] = (e_d * x)^T gy without the [M, D*Tin] outer product
            u8 = dict(dtype=torch.uint8, device=dev)
            xI = torch.empty((lib.gcnpt_rows_image_bytes(M, Tin, compute),), **u8)
            gI = torch.empty((lib.gcnpt_rows_image_bytes(M, H, compute),), **u8)
            x32c = x32.contiguous()
            _lib.check(lib.gcnpt_rows_pack(st, _lib.ptr(x32c), M, Tin, _lib.ptr(xI), compute))
            _lib.check(lib.gcnpt_rows_pack(st, _lib.ptr(gy), M, H, _lib.ptr(gI), compute))
            eT = torch.zeros((D, (M + k - 1) // k * k), dtype=torch.float32, device=dev)
            eT[:, :M] = e32.t()
            dW32 = torch.empty(weight.shape, dtype=torch.float32, device=dev)
            _lib.check(lib.gcnpt_bilinear_bwd_w(st, _lib.ptr(xI), _lib.ptr(gI), _lib.ptr(eT), M, D, Tin, H, _lib.ptr(dW32), compute))
            dW = dW32.to(weight.dtype)
        if ctx.needs_input_grad[3]:
            db = torch.mm(e32.t(), gy).reshape(-1).to(bias.dtype)
        return dx, de, dW, db, None


def bilinear_traverse(xt, e, weight, bias, compute_dtype=torch.bfloat16):
    """traverse_deprel of the reference for M compacted token rows: xt [M,Tin], e [M,D] relation vectors, weight [D*H,Tin] and
    bias [D*H] of the shared nn.Linear (read as W3 [D,Tin,H] / b3 [D,H], gcn.py:301-303).  float32 [M,H].  compute_dtype: float32 =
    exact fp32 MFMA, bfloat16 = bf16 operands with fp32 accumulation."""
    for t in (xt, e, weight, bias):
        _lib.require_gpu(t)
    return _BilinearFn.apply(xt, e, weight, bias, _lib.dtype_code(compute_dtype))


def bilinear_supported(D, Tin, H, compute_dtype=torch.bfloat16):
    return bool(_lib.lib().gcnpt_bilinear_supported(int(D), int(Tin), int(H), _lib.dtype_code(compute_dtype)))


def _zero_pad_row(grad):
    grad = grad.clone()
    grad[0].zero_()
    return grad


class _DiagLayerFn(torch.autograd.Function):
    """One diagonal_deprel layer (csrc/diag_kernels.hip): reference model/gcn.py:272-294 + 390-393."""

    @staticmethod
    def forward(ctx, h, emb, deprel, trees, drop_p, seed, seed_dev=None):
        for t in (h, emb, deprel):
            _lib.require_gpu(t)
        if trees.label is None:
            raise ValueError("diagonal_deprel needs the adjacency values: build the trees with want_label=True")
        B, T = trees.B, trees.T
        H = h.shape[-1]
        if h.shape[0] * h.shape[1] != B * T or emb.shape != (constant.N_DEPREL, H) or deprel.numel() != B * T:
            raise ValueError("diag layer: h %s, table %s, deprel %s do not fit %d x %d trees" %
                             (tuple(h.shape), tuple(emb.shape), tuple(deprel.shape), B, T))
        h = h.contiguous()
        E = emb.detach().float().contiguous()
        deprel = deprel.contiguous()
        out = torch.empty_like(h)
        _lib.check(_lib.lib().gcnpt_diag_layer_fwd(_lib.stream(), _lib.ptr(h), _lib.dtype_code(h.dtype), _lib.ptr(E), _lib.ptr(deprel),
                                                   _lib.ptr(trees.row_ptr), _lib.ptr(trees.col_idx), _lib.ptr(trees.label), B, T, H,
                                                   _lib.ptr(out), float(drop_p), int(seed), _lib.ptr(seed_dev)))
        ctx.trees, ctx.scale = trees, (1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0)
        ctx.save_for_backward(h, out, E, deprel)
        return out

    @staticmethod
    def backward(ctx, gout):
        h, out, E, deprel = ctx.saved_tensors
        trees = ctx.trees
        B, T, H = trees.B, trees.T, h.shape[-1]
        gout = gout.to(h.dtype).contiguous()
        dh = torch.empty_like(h)
        dE = torch.zeros_like(E)
        _lib.check(_lib.lib().gcnpt_diag_layer_bwd(_lib.stream(), _lib.ptr(gout), _lib.ptr(out), _lib.ptr(h), _lib.dtype_code(h.dtype),
                                                   _lib.ptr(E), _lib.ptr(deprel), _lib.ptr(trees.row_ptr), _lib.ptr(trees.col_idx),
                                                   _lib.ptr(trees.label), _lib.ptr(trees.rowT_ptr), _lib.ptr(trees.colT_idx), B, T, H,
                                                   _lib.ptr(dh), _lib.ptr(dE), ctx.scale))
        return dh, dE, None, None, None, None, None


def diag_layer(h, deprel_table, deprel, trees, drop_p=0.0, seed=0, seed_dev=None):
    """dropout(relu(((F (E[deprel] * h)) + (R (E[deprel+42] * h)) + E[84] * h) / (deg + 1))) for h [B,T,H] (fp32 or bf16)."""
    return _DiagLayerFn.apply(h, deprel_table, deprel, trees, drop_p, seed, seed_dev)


class _Pool3Fn(torch.autograd.Function):
    """[pool(h, pool_mask) | pool(h, subj_pos != 0) | pool(h, obj_pos != 0)] in one pass (csrc/pool_kernels.hip)."""

    @staticmethod
    def forward(ctx, h, pool_mask, subj_pos, obj_pos, kind):
        B, T, H = h.shape
        lib, st = _lib.lib(), _lib.stream()
        h = h.contiguous()
        pm = pool_mask.contiguous()
        sp, op = subj_pos.contiguous(), obj_pos.contiguous()
        out = torch.empty((B, 3 * H), dtype=torch.float32, device=h.device)
        argmax = torch.empty((B, 3, H), dtype=torch.int32, device=h.device) if kind == 0 else None
        _lib.check(lib.gcnpt_pool3_fwd(st, _lib.ptr(h), _lib.dtype_code(h.dtype), _lib.ptr(pm), _lib.ptr(sp), _lib.ptr(op), B, T, H, kind,
                                       _lib.ptr(out), _lib.ptr(argmax)))
        ctx.save_for_backward(pm, sp, op, argmax)
        ctx.kind, ctx.shape, ctx.h_dtype = kind, (B, T, H), h.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        pm, sp, op, argmax = ctx.saved_tensors
        B, T, H = ctx.shape
        dh = torch.empty((B, T, H), dtype=ctx.h_dtype, device=g.device)
        g = g.to(torch.float32).contiguous()
        _lib.check(_lib.lib().gcnpt_pool3_bwd(_lib.stream(), _lib.ptr(g), _lib.ptr(argmax), _lib.ptr(pm), _lib.ptr(sp), _lib.ptr(op), B, T, H,
                                              ctx.kind, _lib.ptr(dh), _lib.dtype_code(ctx.h_dtype)))
        return dh, None, None, None, None


def pool3(h, pool_mask, subj_pos, obj_pos, type='max'):
    """
    The three poolings of GCNRelationModel.forward (model/gcn.py:116-121) fused: returns float32 [B, 3H] =
    cat([pool(h, pool_mask), pool(h, subj_pos != 0), pool(h, obj_pos != 0)], dim=1) with pool() of model/gcn.py:473-483.
    h [B,T,H] float32/bfloat16 CUDA, pool_mask bool [B,T,1] (True = excluded), subj_pos / obj_pos int64 [B,T].
    """
    _lib.require_gpu(h)
    if pool_mask.dtype not in (torch.bool, torch.uint8) or subj_pos.dtype != torch.int64 or obj_pos.dtype != torch.int64:
        raise TypeError("pool3: pool_mask must be bool/uint8, subj_pos / obj_pos int64")
    kind = 0 if type == 'max' else (1 if type == 'avg' else 2)
    return _Pool3Fn.apply(h, pool_mask, subj_pos, obj_pos, kind)


def pool(h, mask, type='max'):
    """Reference model/gcn.py:473-483 (mask True = excluded)."""
    if type == 'max':
        return h.masked_fill(mask, -constant.INFINITY_NUMBER).max(1)[0]
    h = h.masked_fill(mask, 0)
    if type == 'avg':
        return h.sum(1) / (mask.size(1) - mask.float().sum(1))
    return h.sum(1)
