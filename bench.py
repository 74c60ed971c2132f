#!/usr/bin/env python3
"""
bench.py -- BASELINE.json metric: GCN-layer fwd+bwd sentences/sec at batch=50 seq=100 h=200 on N MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: launched by torch.distributed.run)

One "step" = the hot path over one synthetic TACRED-shaped batch of 50 sentences x 100 tokens
(BASELINE.json configs[1]: 2-layer GCN, no LSTM, Din 360 -> 200 -> 200, prune_k 1, bf16 storage,
fp32 accumulation, dropout 0.5 between the layers), entirely through the C-ABI of include/gcnpt.h:
    pack W0+W1 (one launch) -> layer0 fwd -> layer1 fwd -> layer1 bwd-data -> [layer1 bwd-weight || layer0 bwd-data] -> layer0 bwd-weight
Inputs (x, gy, weights, the loader's integer tensors) are resident in HBM before the timed region.
`value` is the layer stack alone, as the metric says; `with_prune` repeats the measurement with the
pruned-tree adjacency build (gcnpt_prune_to_csr) inside every step.
For N > 1 every rank runs its own 50-sentence shard (weak scaling, no data-path collective) and the
flat gradient bucket [dW0,db0,dW1,db1] is all-reduced over RCCL once per step, overlapped with the
next step's compute (BASELINE.json configs[3]).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel and
`cpu_baseline` (the CPU oracle -- a numpy port of the reference's dense-bmm layer loop -- on this box's cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=50)
    ap.add_argument("--seq", type=int, default=100)
    ap.add_argument("--din", type=int, default=360)
    ap.add_argument("--hidden", type=int, default=200)
    ap.add_argument("--prune-k", type=int, default=1)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--lengths", choices=["full", "tacred"], default="full")
    ap.add_argument("--layout", choices=["padded", "packed"], default="padded",
                    help="padded: the reference's [B,T,*] rows (the BASELINE metric's shape); packed: token-packed sum(len) rows + cu_seqlens "
                         "(gcnpt_pack_trees), what a variable-length batch costs without its padding (use with --lengths tacred)")
    ap.add_argument("--drop", type=float, default=0.5)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--fused", action="store_true",
                    help="sentence-resident stack kernels (all layers in one launch per direction) instead of one launch per layer; "
                         "measured slower on MI355X at this size (109 vs 76 us/step): 50 workgroups carry every elementwise phase")
    ap.add_argument("--fused2", action="store_true",
                    help="both layers' forward in ONE launch (gcnpt_fused2_fwd, halo recompute, no inter-workgroup wait) instead of one launch "
                         "per layer; same bits; measured slower on MI355X at this size (fwd 32 us against 18 us)")
    ap.add_argument("--split-weight-grad", action="store_true",
                    help="one weight-gradient launch per layer (right after that layer's backward-data) instead of one launch "
                         "for all layers at the end of the backward sweep")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="2 = run the tree build beside the weight pack on a side stream (only matters for with_prune / with_cached_trees); "
                         "measured slower on MI355X (85 -> 88 us with the pruner, 68 -> 82 with cached trees): parallel graph branches cost more than they hide")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="seconds of CPU work per cpu_baseline leg (three legs)")
    ap.add_argument("--cpu-threads", type=int, default=8,
                    help="threads of the headline cpu_baseline leg (8 = what BASELINE.md timed the reference itself with); an all-cores "
                         "leg and the NumPy port (4 BLAS threads, its measured optimum) are reported beside it")
    ap.add_argument("--launch", choices=["auto", "graph", "native"], default="auto",
                    help="graph: the step's launches replayed as one hipGraph; native: eager launches from three native calls per step "
                         "(gcnpt_pack_weights_multi, gcnpt_layers_fwd, gcnpt_layers_bwd), which keeps the queue fed as long as the host is "
                         "fast enough; auto: both are tried during warm-up and the faster one is timed")
    ap.add_argument("--exchange", choices=["auto", "overlap", "inline"], default="auto",
                    help="N > 1: gradient all-reduce overlapped with the following steps on the communication stream, in line with the "
                         "compute (synchronous call), or whichever a warm-up trial finds faster")
    ap.add_argument("--no-kernel-breakdown", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements on rank 0 (with_prune, with_cached_trees, the fp32 "
                                                                "strict-parity block): quick runs and tests")
    ap.add_argument("--separate-pack", action="store_true", help="A/B: keep the weight pack a launch of its own in the with_prune / "
                    "with_cached_trees steps instead of a side job of the tree launch")
    ap.add_argument("--no-pooled-only", action="store_true", help="skip the secondary pooled-only-rows measurement (profiling runs: its launches "
                                                                  "would mix into the per-kernel statistics of the headline step)")
    return ap.parse_args()


N_BUCKETS = 4       # gradient buckets in flight for N > 1: the all-reduce of a bucket is long over when its turn comes again


class Stack(object):
    """Device buffers of one rank's shard and the C-ABI calls of one step."""

    def __init__(self, args, dev, seed, pooled_only=False, packed=False):
        from gcn_over_pruned_trees_amd import _lib
        from gcn_over_pruned_trees_amd.model import tree
        from gcn_over_pruned_trees_amd.utils import synthetic
        self.L, self._lib, self.tree = _lib.lib(), _lib, tree
        self.args, self.dev = args, dev
        B, T, Din, H = args.batch, args.seq, args.din, args.hidden
        self.B, self.T, self.Din, self.H = B, T, Din, H
        act = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        self.compute = _lib.BF16 if args.dtype == "bf16" else _lib.F32
        self.act = _lib.dtype_code(act)
        tb = synthetic.random_tree_batch(seed, B, T, args.lengths)
        self.tb = tb
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        self.head, self.subj, self.obj, self.deprel, self.masks = (t(tb[k]) for k in ("head", "subj_pos", "obj_pos", "deprel", "masks"))
        Ws, bs = synthetic.layer_params(seed + 1, [Din, H, H])
        self.W = [t(w) for w in Ws]
        self.b = [t(b) for b in bs]
        self.x = t(synthetic.normal(seed + 2, (B, T, Din))).to(act)
        self.gy = t(synthetic.normal(seed + 3, (B, T, H))).to(act)
        self.trees = tree.prune_to_csr(self.head, self.subj, self.obj, self.deprel, args.prune_k, masks=self.masks, want_label=False)
        self.trees.check(expect_maxlen=T)
        self.nnz = int(self.trees.nnz().sum())
        self.rows_full = B * T
        if pooled_only:
            # N1 "pooled-only" rows: the same step on the tokens of the pruned trees only (what a pooling consumer needs,
            # gcn.py:116-121): every buffer below is [B, Tc, *]
            ct = self.trees.compact()
            self.trees, self.x, self.gy = ct.trees, ct.take(self.x).contiguous(), ct.take(self.gy).contiguous()
            self.T = T = ct.Tc
        self.rows = B * T
        self.packed = packed
        if packed:
            # token-packed rows (north_star "packed"): sum(len) rows, pattern with absolute columns; the C-ABI then takes B = rows, T = 0
            pk = self.trees.pack(tb["lens"].tolist())
            keep = ~self.masks
            self.trees, self.x, self.gy = pk, self.x[keep].contiguous(), self.gy[keep].contiguous()
            self.rows = pk.N
        R = self.rows
        self.h1 = torch.empty((R, H), dtype=act, device=dev)
        self.h2 = torch.empty((R, H), dtype=act, device=dev)
        self.dh1 = torch.empty((R, H), dtype=act, device=dev)
        self.dx = torch.empty((R, Din), dtype=act, device=dev)
        dims = [(H, Din), (H, H)]
        self.wf = [torch.empty((self.L.gcnpt_packed_bytes(h, d, self.compute),), dtype=torch.uint8, device=dev) for h, d in dims]
        self.wb = [torch.empty((self.L.gcnpt_packed_bytes(d, h, self.compute),), dtype=torch.uint8, device=dev) for h, d in dims]
        # saved operands in MFMA fragment order: S_l = (A+I)h_l written by fwd, dZ_l written by bwd_data
        self.sf = [torch.empty((self.L.gcnpt_frag_bytes(R, d, self.compute),), dtype=torch.uint8, device=dev) for h, d in dims]
        self.zf = [torch.empty((self.L.gcnpt_frag_bytes(R, h, self.compute),), dtype=torch.uint8, device=dev) for h, d in dims]
        # a ring of flat gradient buckets [dW0, db0, dW1, db1] so that the all-reduce of step i overlaps the following steps
        self.n_grad = H * Din + H + H * H + H
        self.buckets = [torch.zeros((self.n_grad,), dtype=torch.float32, device=dev) for _ in range(N_BUCKETS)]
        # loader-side pre-pruning (N4): a "dataset" of 20 batches pruned once; a step then only gathers its batch's rows
        reps = 20
        rep = lambda a: a.repeat(reps, 1)  # noqa: E731
        self.cache = tree.TreeCache.build(rep(self.head), rep(self.subj), rep(self.obj), rep(self.deprel), args.prune_k,
                                          masks=rep(self.masks), want_label=False)
        self.cache_idx = (torch.arange(B, device=dev) + B * (reps // 2)).to(torch.int64)
        self.scale = 1.0 / (1.0 - args.drop) if args.drop > 0 else 1.0
        # big batches: scratch for the gather + matrix form of a layer (csrc/rowsplit_kernels.hip); 0 bytes below 16 384 rows
        Bc, Tc = (self.rows, 0) if packed else (B, T)
        nws = self.L.gcnpt_layers_workspace_bytes(2, Bc, Tc, (ctypes.c_int * 2)(Din, H), (ctypes.c_int * 2)(H, H), self.act)
        self.ws = torch.empty((nws,), dtype=torch.uint8, device=dev) if nws else None
        self.ws_n = nws
        self.side = torch.cuda.Stream(device=dev)
        self.fused = args.fused and args.dtype == "bf16" and not pooled_only and not packed and bool(self.L.gcnpt_stack_supported(T, Din, H, 2, self.compute))
        if self.fused:       # fragment images with per-sentence k-steps: layer inputs h_l (fwd) and G_l = (A+I)^T dZ_l (bwd)
            fb = self.L.gcnpt_stack_frag_bytes
            self.hf = [torch.empty((fb(B, T, d),), dtype=torch.uint8, device=dev) for h, d in dims]
            self.gf = [torch.empty((fb(B, T, h),), dtype=torch.uint8, device=dev) for h, d in dims]
        if packed:
            self.B, self.T = self.rows, 0                   # what the C-ABI takes for packed rows (include/gcnpt.h, gcnpt_pack_trees)

    def grads(self, k):
        H, Din = self.H, self.Din
        f = self.buckets[k]
        o = [0, H * Din, H * Din + H, H * Din + H + H * H]
        return f[o[0]:o[1]], f[o[1]:o[2]], f[o[2]:o[3]], f[o[3]:]

    # ---- individual C-ABI calls (each only enqueues on the current stream) ----
    def prune(self, pack=False):
        """pack: the same launch also packs the weights (gcnpt_prune_to_csr_pack: side job on the CUs the tree build leaves idle)."""
        tr, P, st = self.trees, self._lib.ptr, self._lib.stream()
        a = (st, P(self.head), P(self.subj), P(self.obj), P(self.deprel), P(self.masks), None,
             self.B, self.T, self.args.prune_k, tr.cap, P(tr.row_ptr), P(tr.col_idx), None,
             P(tr.rowT_ptr), P(tr.colT_idx), P(tr.ell), P(tr.ellT), P(tr.pool_mask), P(tr.status))
        self._lib.check(self.L.gcnpt_prune_to_csr_pack(*(a + self._native_args(0)[0])) if pack else self.L.gcnpt_prune_to_csr(*a))

    def gather(self, pack=False):
        """The batch's PrunedTrees from the cached dataset (gcnpt_gather_trees) into the same buffers prune() fills."""
        tr, src, P = self.trees, self.cache.trees, self._lib.ptr
        a = (self._lib.stream(), P(src.row_ptr), P(src.col_idx), None, P(src.rowT_ptr), P(src.colT_idx), P(src.ell), P(src.ellT),
             P(src.pool_mask), P(src.status), P(self.cache.lens), src.B, src.T, src.cap, P(self.cache_idx), self.B, self.T, tr.cap,
             P(tr.row_ptr), P(tr.col_idx), None, P(tr.rowT_ptr), P(tr.colT_idx), P(tr.ell), P(tr.ellT), P(tr.pool_mask), P(tr.status))
        self._lib.check(self.L.gcnpt_gather_trees_pack(*(a + self._native_args(0)[0])) if pack else self.L.gcnpt_gather_trees(*a))

    def pack(self, l):
        P = self._lib.ptr
        H, Din = self.W[l].shape
        self._lib.check(self.L.gcnpt_pack_weights(self._lib.stream(), P(self.W[l]), H, Din, self.compute, P(self.wf[l]), P(self.wb[l])))

    def pack_all(self):
        """Both layers' weights -> MFMA fragment order in ONE launch."""
        n = len(self.W)
        arr = lambda vals, ty: (ty * n)(*vals)  # noqa: E731
        vp = ctypes.c_void_p
        self._lib.check(self.L.gcnpt_pack_weights_multi(
            self._lib.stream(), n, arr([w.data_ptr() for w in self.W], vp), arr([w.shape[0] for w in self.W], ctypes.c_int),
            arr([w.shape[1] for w in self.W], ctypes.c_int), self.compute, arr([t.data_ptr() for t in self.wf], vp),
            arr([t.data_ptr() for t in self.wb], vp)))

    def fwd(self, l):
        P, tr = self._lib.ptr, self.trees
        src, dst = (self.x, self.h1) if l == 0 else (self.h1, self.h2)
        H, Din = self.W[l].shape
        p = self.args.drop if l == 0 else 0.0
        self._lib.check(self.L.gcnpt_layer_fwd_ws(self._lib.stream(), P(src), self.act, P(self.wf[l]), P(self.b[l]), P(tr.row_ptr), P(tr.col_idx),
                                                  P(tr.ell), None, self.B, self.T, Din, H, P(dst), self.act, self.compute, p, 0x5eed, P(self.sf[l]), None,
                                                  P(self.ws), self.ws_n))

    def _dw_db(self, l, k):
        g = self.grads(k)
        return (g[0], g[1]) if l == 0 else (g[2], g[3])

    def bwd_data(self, l, k=0):
        """dh, the dZ fragment image, and cleared dW/db accumulators for bwd_weight(l, k)."""
        P, tr = self._lib.ptr, self.trees
        dy, y, dst = (self.gy, self.h2, self.dh1) if l == 1 else (self.dh1, self.h1, self.dx)
        H, Din = self.W[l].shape
        sc = self.scale if l == 0 else 1.0
        dW, db = self._dw_db(l, k)
        # the hand-over gcnpt_layers_bwd uses: layer 1 leaves dZ of layer 0 in dh1 (it has h1's rows at hand), layer 0 takes it as is
        relu, nsc, is_dz = (self.h1, self.scale, 0) if l == 1 else (None, 1.0, 1)
        self._lib.check(self.L.gcnpt_layer_bwd_data_ws(self._lib.stream(), P(dy), P(y), self.act, P(self.wb[l]), P(tr.ell), P(tr.rowT_ptr),
                                                       P(tr.colT_idx), P(tr.ellT), self.B, self.T, Din, H, P(dst), self.act, self.compute, sc,
                                                       P(self.zf[l]), P(dW), P(db), P(relu), nsc, is_dz, P(self.ws), self.ws_n))

    def bwd_data_wgrad(self, k=0):
        """Layer 0's backward-data launch carrying layer 1's weight gradient on the CUs without a row tile (what gcnpt_layers_bwd does
        for batches of up to 192 row tiles; bigger ones: two launches)."""
        P, tr = self._lib.ptr, self.trees
        H, Din = self.W[0].shape
        H1, Din1 = self.W[1].shape
        dW, db = self._dw_db(0, k)
        dW1, db1 = self._dw_db(1, k)
        self._lib.check(self.L.gcnpt_layer_bwd_data_wgrad(
            self._lib.stream(), P(self.dh1), P(self.h1), self.act, P(self.wb[0]), P(tr.ell), P(tr.rowT_ptr), P(tr.colT_idx), P(tr.ellT), self.B,
            self.T, Din, H, P(self.dx), self.act, self.compute, self.scale, P(self.zf[0]), P(dW), P(db), None, 1.0, 1,
            P(self.zf[1]), P(self.sf[1]), Din1, H1, P(dW1), P(db1)))

    def carries_wgrad(self):
        return self.rows <= 192 * 32 and self.ws is None and not self.args.split_weight_grad

    def bwd_weight(self, l, k=0):
        P = self._lib.ptr
        H, Din = self.W[l].shape
        dW, db = self._dw_db(l, k)
        self._lib.check(self.L.gcnpt_layer_bwd_weight(self._lib.stream(), P(self.zf[l]), P(self.sf[l]), self.B, self.T, Din, H,
                                                      P(dW), P(db), self.compute))

    def bwd_weight_all(self, k=0):
        """Both layers' weight gradients in ONE launch (they only need the fragment images the sweep has left behind)."""
        A, n = self._lib.ptr_array, len(self.W)
        g = self.grads(k)
        ints = lambda vals: (ctypes.c_int * n)(*vals)  # noqa: E731
        self._lib.check(self.L.gcnpt_layer_bwd_weight_multi(
            self._lib.stream(), n, A(self.zf), A(self.sf), self.B, self.T, ints([w.shape[1] for w in self.W]),
            ints([w.shape[0] for w in self.W]), A([g[0], g[2]]), A([g[1], g[3]]), self.compute))

    # ---- the whole layer loop / backward sweep from one native call each (gcnpt_layers_fwd / gcnpt_layers_bwd) ----
    def _native_args(self, k):
        if not hasattr(self, "_nargs"):
            self._nargs = {}
        if k not in self._nargs:
            P, A, tr, n = self._lib.ptr, self._lib.ptr_array, self.trees, len(self.W)
            ints = lambda vals: (ctypes.c_int * n)(*vals)  # noqa: E731
            Din, H = ints([w.shape[1] for w in self.W]), ints([w.shape[0] for w in self.W])
            g = self.grads(k)
            act = ints([self.act] * n)
            fwd = (n, P(self.x), self.act, A(self.wf), A(self.b), P(tr.row_ptr), P(tr.col_idx), P(tr.ell), None, self.B, self.T, Din, H,
                   A([self.h1, self.h2]), act, self.compute, (ctypes.c_float * n)(self.args.drop, 0.0), (ctypes.c_uint64 * n)(0x5eed, 0),
                   A(self.sf), None, P(self.ws), self.ws_n)
            bwd = (n, P(self.gy), A([self.h1, self.h2]), act, A(self.wb), P(tr.ell), P(tr.rowT_ptr), P(tr.colT_idx), P(tr.ellT), self.B, self.T,
                   Din, H, A([self.dx, self.dh1]), act, self.compute, (ctypes.c_float * n)(self.scale, 1.0), A(self.zf), A(self.sf),
                   A([g[0], g[2]]), A([g[1], g[3]]), 0, P(self.ws), self.ws_n)
            vp = ctypes.c_void_p
            pack = (n, (vp * n)(*[w.data_ptr() for w in self.W]), H, Din, self.compute, A(self.wf), A(self.wb))
            self._nargs[k] = (pack, fwd, bwd)
        return self._nargs[k]

    def step_native(self, k=0, with_prune=False):
        """One step as three native calls (pack, all forward layers, backward sweep + weight gradients): eager launches, no graph.
        The argument lists are built once: the host has ~50 us per step for six launches and must not spend them marshalling."""
        pack, fwd, bwd = self._native_args(k)
        st = self._lib.stream()
        merged = bool(with_prune) and not self.args.separate_pack and not self.args.fused2       # the tree launch carries the weight pack
        if with_prune == "cached":
            self.gather(pack=merged)
        elif with_prune:
            self.prune(pack=merged)
        L = self.L
        if self.args.fused2:                    # opt-in A/B: the step with the two-layer forward launch (eager launches from Python)
            for _, call in self.calls(k):
                call()
            return
        rc = (0 if merged else L.gcnpt_pack_weights_multi(st, *pack)) or L.gcnpt_layers_fwd_ws(st, *fwd) or L.gcnpt_layers_bwd_ws(st, *bwd)
        if rc:
            self._lib.check(rc)

    def step_with_pooling(self, handover, k=0):
        """The step with its consumer inside (model/gcn.py:116-121): pack, layers forward, the three poolings, their backward, the backward
        sweep.  handover: the pooling's backward writes dZ of the top layer (gcnpt_pool3_bwd_dz + gcnpt_layers_bwd_dz) instead of dh."""
        pack, fwd, bwd = self._native_args(k)
        P, st, L = self._lib.ptr, self._lib.stream(), self.L
        if not hasattr(self, "pooled"):
            B, H = self.B, self.H
            self.pooled = torch.empty((B, 3 * H), dtype=torch.float32, device=self.dev)
            self.gpool = torch.randn((B, 3 * H), dtype=torch.float32, device=self.dev)
            self.amax = torch.empty((B, 3, H), dtype=torch.int32, device=self.dev)
            self.dtop = torch.empty_like(self.h2)
        tr = self.trees
        bwd = (bwd[0], P(self.dtop)) + bwd[2:-3] + (1 if handover else 0,) + bwd[-2:]
        rc = (L.gcnpt_pack_weights_multi(st, *pack) or L.gcnpt_layers_fwd_ws(st, *fwd)
              or L.gcnpt_pool3_fwd(st, P(self.h2), self.act, P(tr.pool_mask), P(self.subj), P(self.obj), self.B, self.T, self.H, 0, P(self.pooled),
                                   P(self.amax)))
        if rc == 0 and handover:
            rc = (L.gcnpt_pool3_bwd_dz(st, P(self.gpool), P(self.amax), P(tr.pool_mask), P(self.subj), P(self.obj), self.B, self.T, self.H, 0,
                                       P(self.h2), P(tr.ell), 1.0, P(self.dtop), self.act) or L.gcnpt_layers_bwd_ws(st, *bwd))
        elif rc == 0:
            rc = (L.gcnpt_pool3_bwd(st, P(self.gpool), P(self.amax), P(tr.pool_mask), P(self.subj), P(self.obj), self.B, self.T, self.H, 0,
                                    P(self.dtop), self.act) or L.gcnpt_layers_bwd_ws(st, *bwd))
        if rc:
            self._lib.check(rc)

    # ---- sentence-resident stack: every layer in one launch per direction ----
    def stack_fwd(self, k=0):
        P, A, tr, L = self._lib.ptr, self._lib.ptr_array, self.trees, 2
        g = self.grads(k)
        self._lib.check(self.L.gcnpt_stack_fwd(
            self._lib.stream(), L, P(self.x), self.act, A(self.wf), A(self.b), P(tr.row_ptr), P(tr.col_idx), P(tr.ell), None,
            self.B, self.T, self.Din, self.H, A([self.h1, self.h2]), self.act, (ctypes.c_float * L)(self.args.drop, 0.0),
            (ctypes.c_uint64 * L)(0x5eed, 0), A(self.hf), A([g[0], g[2]]), A([g[1], g[3]]), None))

    def stack_bwd(self, k=0):
        P, A, tr, L = self._lib.ptr, self._lib.ptr_array, self.trees, 2
        g = self.grads(k)
        self._lib.check(self.L.gcnpt_stack_bwd(
            self._lib.stream(), L, P(self.gy), A([self.h1, self.h2]), self.act, A(self.wb), P(tr.ell), P(tr.rowT_ptr), P(tr.colT_idx),
            P(tr.ellT), self.B, self.T, self.Din, self.H, P(self.dx), self.act, (ctypes.c_float * L)(self.scale, 1.0), A(self.gf),
            A([g[1], g[3]])))

    def stack_bwd_weight(self, k=0):
        A, L = self._lib.ptr_array, 2
        g = self.grads(k)
        self._lib.check(self.L.gcnpt_stack_bwd_weight(self._lib.stream(), L, A(self.gf), A(self.hf), self.B, self.T, self.Din, self.H,
                                                      A([g[0], g[2]])))

    def fwd_all(self):
        """Both layers' forward in ONE launch (csrc/fused_kernels.hip, opt-in)."""
        P, A, tr = self._lib.ptr, self._lib.ptr_array, self.trees
        self._lib.check(self.L.gcnpt_fused2_fwd(
            self._lib.stream(), P(self.x), A(self.wf), A(self.b), P(tr.row_ptr), P(tr.col_idx), P(tr.ell), None, self.B, self.T, self.Din,
            (ctypes.c_int * 2)(self.H, self.H), P(self.h1), P(self.h2), self.act, (ctypes.c_float * 2)(self.args.drop, 0.0),
            (ctypes.c_uint64 * 2)(0x5eed, 0), A(self.sf), None))

    def two_layer_launches(self):
        return (self.args.fused2 and not self.packed and len(self.W) == 2 and self.args.dtype == "bf16" and
                bool(self.L.gcnpt_fused2_supported(self.T, self.Din, self.H, self.H, self.act, self.compute)))

    def calls(self, k=0):
        """The launches of one step, in order (name, callable)."""
        if self.fused:
            return [("pack", self.pack_all), ("stack_fwd", lambda: self.stack_fwd(k)), ("stack_bwd", lambda: self.stack_bwd(k)),
                    ("stack_bwd_weight", lambda: self.stack_bwd_weight(k))]
        if self.args.split_weight_grad:
            return [("pack", self.pack_all), ("fwd0", lambda: self.fwd(0)), ("fwd1", lambda: self.fwd(1)),
                    ("bwd_data1", lambda: self.bwd_data(1, k)), ("bwd_weight1", lambda: self.bwd_weight(1, k)),
                    ("bwd_data0", lambda: self.bwd_data(0, k)), ("bwd_weight0", lambda: self.bwd_weight(0, k))]
        if self.two_layer_launches():
            return [("pack", self.pack_all), ("fwd", self.fwd_all), ("bwd_data1", lambda: self.bwd_data(1, k)),
                    ("bwd_data0", lambda: self.bwd_data(0, k)), ("bwd_weight", lambda: self.bwd_weight_all(k))]
        if self.carries_wgrad():
            return [("pack", self.pack_all), ("fwd0", lambda: self.fwd(0)), ("fwd1", lambda: self.fwd(1)),
                    ("bwd_data1", lambda: self.bwd_data(1, k)), ("bwd_data0+wgrad1", lambda: self.bwd_data_wgrad(k)),
                    ("bwd_weight0", lambda: self.bwd_weight(0, k))]
        return [("pack", self.pack_all), ("fwd0", lambda: self.fwd(0)), ("fwd1", lambda: self.fwd(1)),
                ("bwd_data1", lambda: self.bwd_data(1, k)), ("bwd_data0", lambda: self.bwd_data(0, k)),
                ("bwd_weight", lambda: self.bwd_weight_all(k))]

    def step(self, k=0, with_prune=False):
        """
        One step.  With --streams 2 the tree build, which only needs the loader tensors, runs beside the weight pack on a
        side stream (fork and join are inside the step, so they become parallel branches of the captured graph).
        """
        if self.args.streams == 1 or self.fused:
            merged = bool(with_prune) and not self.args.separate_pack and not self.fused and not self.args.fused2
            if with_prune == "cached":
                self.gather(pack=merged)
            elif with_prune:
                self.prune(pack=merged)
            for _, call in self.calls(k)[1 if merged else 0:]:
                call()
            return
        main = torch.cuda.current_stream()
        side = self.side
        if with_prune:                                   # fork: the tree build beside the weight pack
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self.gather() if with_prune == "cached" else self.prune()
        calls = self.calls(k)
        calls[0][1]()                                    # pack
        if with_prune:
            main.wait_stream(side)                       # join before the first layer
        for _, call in calls[1:]:
            call()

    # ---- algorithmic bytes per launch (DESIGN.md "Measurement") ----
    def algorithmic_bytes(self):
        e = 2 if self.args.dtype == "bf16" else 4
        N, B, T = self.rows, self.B, self.T
        csr = 32 * N            # one ELL head (count + 7 columns) per row; the CSR arrays are only touched by rows with > 7 entries
        out = {}
        for l, (H, Din) in enumerate([tuple(w.shape) for w in self.W]):
            wp = self.wf[l].numel()
            out["fwd%d" % l] = e * N * (Din + H) + wp + 4 * H + csr + self.sf[l].numel()
            # the top layer reads dY and Y, the layers below read the dZ the layer above left them; every layer but the bottom one
            # also reads its input rows to leave dZ for the layer below (hand-over of gcnpt_layers_bwd)
            top, bottom = l == len(self.W) - 1, l == 0
            out["bwd_data%d" % l] = e * N * ((2 if top else 1) * H + (1 if bottom else 2) * Din) + self.wb[l].numel() + 2 * csr + \
                self.zf[l].numel() + 4 * (H * Din + H)
            out["bwd_weight%d" % l] = self.zf[l].numel() + self.sf[l].numel() + 4 * (H * Din + H)
            out["bwd_weight"] = out.get("bwd_weight", 0) + out["bwd_weight%d" % l]
            if l == 1:
                out["bwd_data0+wgrad1"] = out["bwd_data0"] + out["bwd_weight1"]
            out["pack"] = out.get("pack", 0) + 4 * H * Din + self.wf[l].numel() + self.wb[l].numel()
        out["fwd"] = sum(out["fwd%d" % l] for l in range(len(self.W)))
        out["bwd_data"] = sum(out["bwd_data%d" % l] for l in range(len(self.W)))
        out["prune"] = 4 * 8 * N + N + 2 * (csr + 4 * B * (T + 1) + 4 * self.nnz) + N + 4 * (B + 1)
        if self.fused:
            (H, Din), grads = tuple(self.W[0].shape), 4 * self.n_grad
            hf, gf = sum(t.numel() for t in self.hf), sum(t.numel() for t in self.gf)
            # x in, h1 + h2 out, weights, bias, ELL, fragment images of the layer inputs, cleared accumulators
            out["stack_fwd"] = e * N * (Din + 2 * H) + sum(t.numel() for t in self.wf) + 8 * H + csr + hf + grads
            # gy, h2, h1 in, dx out, weights, ELL (both patterns), fragment images of G, bias gradients
            out["stack_bwd"] = e * N * (3 * H + Din) + sum(t.numel() for t in self.wb) + 2 * csr + gf + 8 * H
            out["stack_bwd_weight"] = hf + gf + 4 * (H * Din + H * H)
        return out

    def survey_bytes(self):
        """SURVEY.md 8(d)'s ALGORITHMIC bytes of the layer math each launch covers (padded layout, S recomputed in backward, no saved
        images, no pack): fwd = e N (Din+H) + e Din H + 4H + CSR; bwd = e N (2H+2Din) + (e+4) Din H + 4H + CSR, of which the data
        half reads dY, Y, W, CSR and writes dh, and the weight half reads h and writes dW, db.  Launches that cover several of these
        get their sum; `pack` covers none (0)."""
        e = 2 if self.args.dtype == "bf16" else 4
        N = self.rows
        csr = 4 * (N + 1) + 4 * self.nnz
        per = {}
        for l, (H, Din) in enumerate([tuple(w.shape) for w in self.W]):
            per["fwd%d" % l] = e * N * (Din + H) + e * Din * H + 4 * H + csr
            per["bwd_data%d" % l] = e * N * (2 * H + Din) + e * Din * H + csr
            per["bwd_weight%d" % l] = e * N * Din + 4 * Din * H + 4 * H
        L = len(self.W)
        per["bwd_weight"] = sum(per["bwd_weight%d" % l] for l in range(L))
        if L > 1:
            per["bwd_data0+wgrad1"] = per["bwd_data0"] + per["bwd_weight1"]
        per["fwd"] = per["stack_fwd"] = sum(per["fwd%d" % l] for l in range(L))
        per["bwd_data"] = per["stack_bwd"] = sum(per["bwd_data%d" % l] for l in range(L))
        per["stack_bwd_weight"] = per["bwd_weight"]
        per["pack"] = 0
        per["prune"] = 0
        return per


def capture(fn, use_graph):
    """Returns a callable that replays `fn` (a hipGraph when possible)."""
    if not use_graph:
        return fn, False
    try:
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        torch.cuda.synchronize()
        return g.replay, True
    except Exception as exc:  # pragma: no cover - depends on the runtime
        print("[bench] hipGraph capture failed (%s); launching eagerly" % exc, file=sys.stderr)
        torch.cuda.synchronize()
        return fn, False


def timed(run, steps, warmup, barrier):
    """Contract: W untimed steps, then exactly K steps bracketed by barrier + synchronize; also HIP events on the stream."""
    for i in range(warmup):
        run(i)
    barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    return wall, e0.elapsed_time(e1) * 1e-3


def kernel_breakdown(stack, use_graph, rounds=100, reps=8):
    """
    Duration of each kernel IN the step: the step is replayed truncated after its first k launches, HIP events around the
    replays, and kernel k is charged t(k) - t(k-1).  Unlike timing a kernel alone back to back, this keeps the producer ->
    consumer cache state of the real step (each kernel reads what the previous one wrote from other XCDs) and includes its
    launch boundary.  One hipGraph holds `reps` copies of the truncated step (every launch of the step may be repeated: the
    backward clears the accumulators the weight gradient adds into), so the 5 us a graph replay costs on the device is
    spread over `reps` prefixes and the durations are those of back-to-back launches, as in the timed native-launch mode
    and in the rocprofv3 summaries.  `prune` is timed as the step with the tree build minus the step without.
    """
    calls = stack.calls(0)
    stack.step()
    torch.cuda.synchronize()

    def timed_replay(fn):
        def many():
            for _ in range(reps):
                fn()
        run, _ = capture(many, use_graph)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(rounds):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / (rounds * reps)

    out, prev = {}, 0.0
    for k in range(1, len(calls) + 1):
        def prefix(k=k):
            for _, c in calls[:k]:
                c()
        t = timed_replay(prefix)
        out[calls[k - 1][0]] = max(t - prev, 1e-9)
        prev = t
    if stack.B * stack.T == stack.rows_full:      # (a pooled-only stack holds [B, Tc] trees: the pruner's arrays do not fit them)
        out["prune"] = max(timed_replay(lambda: stack.step(0, with_prune=True)) - prev, 1e-9)
    stack.step()          # leave consistent buffers behind
    torch.cuda.synchronize()
    return out


def cpu_baseline(args, seconds):
    """
    The reference's CPU path on this box's host cores, same workload, three legs (bounded: `seconds` each):
      value      oracle/gcn_ref_torch.py -- the reference's own library ops (dense bmm over [B,T,T], 2 x linear per layer, autograd),
                 pinned to the reference's recorded outputs in tests/test_oracle_golden.py -- at --cpu-threads (8: BASELINE.md's setting)
      all_cores  the same at torch's default thread count for this box
      numpy_port oracle/gcn_ref.py (explicit NumPy backward), 4 BLAS threads
    The tree build is excluded (as in `value` of the GPU line); the C oracle's time for it is in `sample`.
    """
    from gcn_over_pruned_trees_amd.utils import synthetic
    from oracle import gcn_ref, gcn_ref_torch, prune_ref
    B, T, Din, H = args.batch, args.seq, args.din, args.hidden
    tb = synthetic.random_tree_batch(1234, B, T, args.lengths)
    Ws, bs = synthetic.layer_params(1235, [Din, H, H])
    x, gy = synthetic.normal(1236, (B, T, Din)), synthetic.normal(1237, (B, T, H))
    t0 = time.perf_counter()
    adj = prune_ref.batch_adj(tb["head"], tb["subj_pos"], tb["obj_pos"], tb["deprel"], tb["lens"], args.prune_k)["adj"]
    t_prune = time.perf_counter() - t0

    def loop(fn):
        fn()                                              # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            fn()
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds and n >= 3:
                return n, el

    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    targs = (tt(adj), tt(x), [tt(w) for w in Ws], [tt(b) for b in bs], tt(gy))
    default_threads = torch.get_num_threads()
    legs = {}
    for key, nthr in (("t", max(1, min(args.cpu_threads, os.cpu_count() or 1))), ("all", default_threads)):
        torch.set_num_threads(nthr)
        n, el = loop(lambda: gcn_ref_torch.forward_backward(*targs))
        legs[key] = dict(value=B * n / el, cores=nthr, steps=n, seconds=el)
    torch.set_num_threads(default_threads)
    np_cores = 4
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=np_cores):
            n, el = loop(lambda: gcn_ref.gcn_backward(adj, x, Ws, bs, gy))
    except Exception:
        np_cores = os.cpu_count() or 1
        n, el = loop(lambda: gcn_ref.gcn_backward(adj, x, Ws, bs, gy))
    t = legs["t"]
    return dict(value=t["value"], unit="sentences/s", cores=int(t["cores"]), kind="port",
                sample="%d fwd+bwd steps of the same %dx%d workload in %.1f s: torch-CPU restatement of the reference's layer loop with its own ops "
                       "(dense bmm + 2 x linear per layer + autograd, oracle/gcn_ref_torch.py, fp32); tree build excluded (%.1f ms per batch in the C "
                       "oracle on 1 core); host has %d logical CPUs" % (t["steps"], B, T, t["seconds"], t_prune * 1e3, os.cpu_count() or 0),
                all_cores=dict(value=legs["all"]["value"], cores=int(legs["all"]["cores"]), steps=legs["all"]["steps"]),
                numpy_port=dict(value=B * n / el, cores=int(np_cores), steps=n,
                                note="oracle/gcn_ref.py, explicit NumPy/BLAS backward (round 1's baseline)"))


def self_launch(args):
    """
    `python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start N fresh rank processes, one per
    GPU, BEFORE anything in this process touches the GPU (this parent never does: it only waits), and relay rank 0's single
    JSON line.  The children are this same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly what
    `python -m torch.distributed.run --nproc-per-node N` would give them.  A rank that fails ends the others and the run
    exits non-zero; nothing is restarted.
    """
    import socket
    import subprocess
    n = args.gpus
    one_dev = bool(os.environ.get("GCNPT_BENCH_ONE_DEVICE"))
    have = torch.cuda.device_count()                # counting devices does not initialise the GPU on this image
    if have < n and not one_dev:
        print("[bench] --gpus %d but only %d GPU(s) visible (GCNPT_BENCH_ONE_DEVICE=1 rehearses N ranks on one GPU)" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GCNPT_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        out = subprocess.PIPE if r == 0 else sys.stderr           # only rank 0 owns the JSON line
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    line0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for pr in procs[1:]:
        if rc != 0 and pr.poll() is None:
            pr.kill()                                               # by handle: these are exactly the processes started above
        rc = pr.wait() or rc
    text = line0.decode("utf-8", "replace") if line0 else ""
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    if rc != 0 or len(lines) != 1:
        sys.stderr.write(text)
        print("[bench] self-launched %d ranks: exit code %d, %d JSON line(s) from rank 0" % (n, rc, len(lines)), file=sys.stderr)
        return rc or 1
    print(lines[0], flush=True)
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # N comes from the launcher (WORLD_SIZE); a mismatch with --gpus is a caller error, not something to guess about
        if rank == 0:
            print("[bench] --gpus %d but WORLD_SIZE %d: they must agree" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    force_dist = bool(os.environ.get("GCNPT_BENCH_FORCE_DIST"))      # rehearsal: the N > 1 code path (RCCL init, overlapped all-reduce) with one rank
    saved_stdout = None
    if world > 1 or force_dist:
        # RCCL prints a version banner on stdout (C stdio) when the communicator is created; the contract is ONE JSON line on
        # rank 0's stdout, so everything but that line goes to stderr: fd 1 points at fd 2 until the result is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("GCNPT_BENCH_ONE_DEVICE"):          # rehearsal of the N > 1 code path on a 1-GPU box (with --dist-backend gloo)
            local = 0
        torch.cuda.set_device(local)
        dist.init_process_group(args.dist_backend)
        barrier = dist.barrier
    else:
        dist = None
        torch.cuda.set_device(0)
        barrier = lambda: None  # noqa: E731
    dev = torch.device("cuda", local if world > 1 else 0)
    use_graph = not args.no_graph

    # every rank draws its own shard of the global batch; GCNPT_BENCH_SAME_SHARD=1 (tests) gives all ranks rank 0's shard, so that
    # the all-reduced bucket must be exactly world x the one-rank bucket
    shard_seed = 1234 + (0 if os.environ.get("GCNPT_BENCH_SAME_SHARD") else 17 * rank)
    stack = Stack(args, dev, seed=shard_seed, packed=args.layout == "packed")
    from gcn_over_pruned_trees_amd.shard import OverlappedAllReduce

    # ---- the timed step: layer stack fwd+bwd (+ overlapped gradient all-reduce when N > 1)
    multi = world > 1 or force_dist
    # SUM, not AVG: the 1/world factor belongs to the optimizer's learning rate, and SUM is supported by every backend
    reducer = OverlappedAllReduce(stack.buckets, dist, average=False) if multi else None

    def runner(mode, with_prune=False, n_buckets=1):
        """[(callable, is_graph)] per gradient bucket for a launch mode."""
        if mode == "native":
            fns = [((lambda k=k: stack.step_native(k, with_prune)), False) for k in range(n_buckets)]
            for f, _ in fns:
                f()
            torch.cuda.synchronize()
            return fns
        return [capture(lambda k=k: stack.step(k, with_prune=with_prune), use_graph) for k in range(n_buckets)]

    def make_run(replays, exchange="overlap"):
        """A step + (N > 1) the all-reduce of the bucket it wrote.  exchange "overlap": asynchronous, on the communication stream,
        beside the following steps (ring of buckets); "inline": a synchronous all_reduce, in line with the compute."""
        def run(i):
            k = i % N_BUCKETS if multi else 0
            if reducer and exchange == "overlap":
                reducer.before_write(k)
            replays[k][0]()
            if reducer:
                if exchange == "overlap":
                    reducer.after_write(k)
                else:
                    dist.all_reduce(stack.buckets[k])
        return run

    def drain():
        if reducer:
            reducer.finish()
        torch.cuda.synchronize()

    modes = ["graph"] if (stack.fused or args.launch == "graph") else (["native"] if args.launch == "native" else ["graph", "native"])
    cands = {m: runner(m, n_buckets=N_BUCKETS if multi else 1) for m in modes}
    exchanges = (["overlap", "inline"] if args.exchange == "auto" else [args.exchange]) if multi else ["overlap"]
    combos = [(m, x) for m in modes for x in exchanges]
    (launch, exchange), trial = combos[0], {}
    if len(combos) > 1:
        # part of the warm-up: a short trial of each launch mode (and, for N > 1, of each way to run the exchange); every rank
        # must take the same one: MAX over ranks decides
        for m, x in combos:
            w, _ = timed(make_run(cands[m], x), 200, 20, barrier)
            drain()
            if multi:
                tm = torch.tensor([w], dtype=torch.float64, device=dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                w = float(tm.item())
            trial[m + ("/" + x if multi else "")] = w / 200
        launch, exchange = min(combos, key=lambda c: trial[c[0] + ("/" + c[1] if multi else "")])
    replays = cands[launch]
    graphed = replays[0][1]
    wall, ev = timed(make_run(replays, exchange), args.steps, args.warmup, barrier)
    drain()
    rank_ms = [wall / args.steps * 1e3]
    if multi:
        mine = torch.tensor([wall], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        wall = max(float(t.item()) for t in every)                 # contract: MAX over ranks
    assert torch.isfinite(stack.buckets[0]).all() and torch.isfinite(stack.dx.float()).all()
    # the bucket the LAST timed step wrote, after its all-reduce (drain() above): rank-sum of [dW0, db0, dW1, db1]
    last_bucket = stack.buckets[(args.steps - 1) % N_BUCKETS if multi else 0]
    grad_abs_sum = float(last_bucket.double().abs().sum().item())

    result = None
    if rank == 0:
        sent = args.batch * world * args.steps
        # second measurement on rank 0 only: tree build inside the step
        if args.layout == "packed":
            args.no_secondary = True                       # (the pruner's and the cache's arrays are [B,T]: not part of a packed step)
        if not args.no_secondary:
            run_p = runner(launch, with_prune=True)[0][0]
            wall_p, _ = timed(lambda i: run_p(), args.steps, min(args.warmup, 50), lambda: None)
            run_c = runner(launch, with_prune="cached")[0][0]
            wall_c, _ = timed(lambda i: run_c(), args.steps, min(args.warmup, 50), lambda: None)
        result = {
            "metric": "GCN-layer fwd+bwd sentences/sec at batch=50 seq=100 h=200",
            "value": sent / wall, "unit": "sentences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "2-layer GCN stack fwd+bwd (no LSTM), batch=%d seq_len=%d Din=%d hidden=%d prune_k=%d, %s storage / fp32 accumulate, "
                                   "synthetic TACRED-shaped random trees (lengths=%s, layout=%s: %d rows), dropout %.1f between layers"
                                   % (args.batch, args.seq, args.din, args.hidden, args.prune_k, args.dtype, args.lengths, args.layout, stack.rows,
                                      args.drop),
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "parallelism": "dp%d" % world,
                       "world_size": dist.get_world_size() if multi else 1, "dist_backend": dist.get_backend() if multi else None,
                       "ranks_started_by": ("bench.py itself (--gpus N without a launcher)" if os.environ.get("GCNPT_BENCH_SELF_LAUNCHED")
                                            else "torch.distributed.run / caller") if world > 1 else None,
                       "ms_per_step_by_rank": [round(t, 6) for t in rank_ms], "grad_bucket_abs_sum": grad_abs_sum,
                       "launch": "hipGraph replay" if graphed else ("eager launches from 3 native calls per step (pack, gcnpt_layers_fwd_ws, gcnpt_layers_bwd_ws)"
                                                                     if launch == "native" else "eager"),
                       "layer_form": ("gather + matrix launch per layer and direction (csrc/rowsplit_kernels.hip, %d B workspace)" % stack.ws_n) if stack.ws_n and not stack.fused
                                     else "one row-tile launch per layer and direction",
                       "allreduce_stream_waits": reducer.stream_waits if reducer else None,
                       "launch_trial_us_per_step": {m: round(t * 1e6, 2) for m, t in trial.items()} or None, "nnz_per_batch": stack.nnz,
                       "grad_allreduce": ("flat fp32 bucket %d B per step over RCCL, " % (4 * stack.n_grad)) +
                                         ("overlapped with the following steps (ring of %d buckets)" % N_BUCKETS if exchange == "overlap"
                                          else "synchronous, in line with the compute") if world > 1 else "none (1 GPU)"},
            "event_ms_per_step": ev / args.steps * 1e3,
        }
        if not args.no_secondary:
            result["with_prune"] = {"value": args.batch * args.steps / wall_p, "unit": "sentences/s", "ms_per_step": wall_p / args.steps * 1e3,
                                    "note": "rank 0, pruned-tree adjacency build inside every step; the tree launch carries the weight pack as a "
                                            "side job (gcnpt_prune_to_csr_pack: one launch boundary less)"}
            result["with_cached_trees"] = {"value": args.batch * args.steps / wall_c, "unit": "sentences/s", "ms_per_step": wall_c / args.steps * 1e3,
                                           "note": "rank 0, dataset pruned once; every step assembles its batch's adjacency with gcnpt_gather_trees_pack (the weight "
                                                   "pack rides in the same launch)"}
            if args.dtype == "bf16":
                # the reference's own arithmetic: fp32 activations, exact fp32 MFMA (the strict-parity mode of the tests), same step
                import copy
                a32 = copy.copy(args)
                a32.dtype = "fp32"
                s32 = Stack(a32, dev, seed=shard_seed)
                if launch == "native":
                    run32 = s32.step_native
                    run32()
                    torch.cuda.synchronize()
                else:
                    run32, _ = capture(lambda: s32.step(0), use_graph)
                n32 = max(args.steps // 4, 50)
                wall32, _ = timed(lambda i: run32(), n32, min(args.warmup, 50), lambda: None)
                result["fp32"] = {"value": args.batch * n32 / wall32, "unit": "sentences/s", "ms_per_step": wall32 / n32 * 1e3, "dtype": "f32",
                                  "steps": n32, "note": "rank 0, same step with fp32 activations and exact fp32 MFMA (v_mfma_f32_16x16x4_f32): the "
                                                        "reference's own arithmetic, the mode the 1e-5 / 1e-4 parity tests run in"}
                del s32
        if not stack.fused and not args.no_secondary and args.layout == "padded":
            # the step with the reference's consumer in it (max pooling x3, gcn.py:116-121), with and without the dZ hand-over
            wp = {}
            for name, ho in (("two_ops", False), ("handover", True)):
                if launch == "native":
                    run_h = lambda ho=ho: stack.step_with_pooling(ho)  # noqa: E731
                    run_h()
                    torch.cuda.synchronize()
                else:
                    run_h, _ = capture(lambda ho=ho: stack.step_with_pooling(ho), use_graph)
                n_h = max(args.steps // 4, 50)
                wall_h, _ = timed(lambda i: run_h(), n_h, min(args.warmup, 50), lambda: None)
                wp[name] = wall_h / n_h * 1e3
            result["with_pooling"] = {"ms_per_step_pool3_then_layers_bwd": wp["two_ops"], "ms_per_step_pool3_bwd_dz_handover": wp["handover"],
                                      "note": "rank 0, the step plus the consumer's three max poolings forward and backward (gcnpt_pool3_fwd / _bwd); "
                                              "handover: the pooling's backward writes dZ of the top layer (gcnpt_pool3_bwd_dz, gcnpt_layers_bwd_dz), "
                                              "which then gathers one row per neighbour instead of dY, Y and the degree"}
        if not stack.fused and not args.no_pooled_only and args.layout == "padded" and args.lengths == "tacred":
            sp = Stack(args, dev, seed=shard_seed, packed=True)
            if launch == "native":
                run_q = sp.step_native
                run_q()
                torch.cuda.synchronize()
            else:
                run_q, _ = capture(lambda: sp.step(0), use_graph)
            wall_q, _ = timed(lambda i: run_q(), args.steps, min(args.warmup, 50), lambda: None)
            result["packed_rows"] = {
                "value": args.batch * args.steps / wall_q, "unit": "sentences/s", "ms_per_step": wall_q / args.steps * 1e3,
                "rows": sp.rows, "rows_full": sp.rows_full,
                "note": "rank 0, the same step on token-packed rows (gcnpt_pack_trees: sum(len) rows, no padding slots); every real token's "
                        "row is bit-identical to the padded batch's"}
            del sp
        if not stack.fused and not args.no_pooled_only and args.layout == "padded":
            sc = Stack(args, dev, seed=shard_seed, pooled_only=True)
            if launch == "native":
                run_k = sc.step_native
                run_k()
                torch.cuda.synchronize()
            else:
                run_k, _ = capture(lambda: sc.step(0), use_graph)
            wall_k, _ = timed(lambda i: run_k(), args.steps, min(args.warmup, 50), lambda: None)
            result["pooled_only_rows"] = {
                "value": args.batch * args.steps / wall_k, "unit": "sentences/s", "ms_per_step": wall_k / args.steps * 1e3,
                "rows": sc.B * sc.T, "rows_full": sc.rows_full,
                "note": "rank 0, NOT the headline workload: the same step on the tokens of the pruned trees only (gcnpt_compact_trees, "
                        "[B, %d] instead of [B, %d]) -- what the step costs when the consumer is the reference's pooling (gcn.py:116-121), "
                        "which never reads another row; kept rows are bit-identical to the full batch's" % (sc.T, args.seq)}
            del sc
        alg = stack.algorithmic_bytes()
        if not args.no_kernel_breakdown:
            kt = kernel_breakdown(stack, use_graph)
            step_keys = [k for k in kt if k != "prune"]
            result["config"]["kernels_per_step"] = ", ".join(k for k, _ in stack.calls(0))
            dom = max(step_keys, key=lambda k: kt[k])
            sv = stack.survey_bytes()
            # HBM-side bytes per launch from separate rocprofv3 --pmc passes (tools/profile_round.sh): only quoted when that recording was
            # made for THIS shape / dtype / launch list (its "meta"), otherwise null -- a replayed number for another workload is no evidence
            traffic, traffic_src = None, None
            tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tf):
                with open(tf) as f:
                    rec = json.load(f)
                meta = rec.get("meta", {})
                here = dict(batch=args.batch, seq=args.seq, din=args.din, hidden=args.hidden, prune_k=args.prune_k, dtype=args.dtype,
                            lengths=args.lengths, kernels=[k for k, _ in stack.calls(0)])
                if all(meta.get(k) == v for k, v in here.items()):
                    traffic = rec.get("traffic", {}).get(dom)
                    traffic_src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload at git %s, 2*FETCH+WRITE)" % meta.get("git_sha", "?")
            gbs = sv[dom] / kt[dom] / 1e9
            result["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                  "traffic": traffic, "traffic_source": traffic_src,
                                  "algorithmic_bytes": sv[dom], "avg_launch_us": kt[dom] * 1e6,
                                  "dataflow_bytes": alg[dom], "dataflow_frac": alg[dom] / kt[dom] / 1e9 / HBM_PEAK_GBS,
                                  "note": "achieved = SURVEY.md 8(d) bytes of the layer math this launch covers / its in-step duration (HIP events around replays of "
                                          "the step truncated after k launches, 8 copies per hipGraph, t(k)-t(k-1); includes the launch boundary); dataflow_* counts "
                                          "what this implementation moves on top (saved-operand fragment images, packed weights)"}
            result["kernels"] = {k: {"avg_us": kt[k] * 1e6, "survey_8d_bytes": sv.get(k), "dataflow_bytes": alg[k],
                                     "GBps_8d": (sv[k] / kt[k] / 1e9) if sv.get(k) else None} for k in kt}
            tot_b = sum(alg[k] for k in step_keys)
            survey = sum(sv.get(k, 0) for k in step_keys)
            t_step = wall / args.steps
            result["step_roofline"] = {"survey_8d_bytes": survey, "frac_of_hbm_peak": survey / t_step / 1e9 / HBM_PEAK_GBS,
                                       "sum_kernel_us": sum(kt[k] for k in step_keys) * 1e6,
                                       "dataflow_bytes": tot_b, "dataflow_frac_of_hbm_peak": tot_b / sum(kt[k] for k in step_keys) / 1e9 / HBM_PEAK_GBS,
                                       "note": "survey_8d_bytes = SURVEY.md 8(d)'s formula for the layer math alone (S recomputed, no saved images, no pack), "
                                               "over the WHOLE timed step (ms_per_step); dataflow_bytes = what this implementation moves, over the summed kernel times"}
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
        print(json.dumps(result), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)                                  # teardown messages of the process group: not on stdout either
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
