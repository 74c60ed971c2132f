#!/usr/bin/env python3
"""
bench.py -- BASELINE.json metric: GCN-layer fwd+bwd sentences/sec at batch=50 seq=100 h=200 on N MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: launched by torch.distributed.run, or self-launched)

One "step" = the hot path over one synthetic TACRED-shaped batch of 50 sentences x 100 tokens
(BASELINE.json configs[1]: 2-layer GCN, no LSTM, Din 360 -> 200 -> 200, prune_k 1, bf16 storage,
fp32 accumulation, dropout 0.5 between the layers), entirely through the C-ABI of include/gcnpt.h:
    pack W0+W1 -> layer0 fwd -> layer1 fwd -> layer1 bwd-data -> layer0 bwd-data with layer1's weight gradient riding -> layer0 bwd-weight
(six launches from ONE native call, gcnpt_layers_step, whose argument struct is marshalled once).
Inputs (x, gy, weights, the loader's integer tensors) are resident in HBM before the timed region.
`value` is the layer stack alone, as the metric says; `with_prune` / `with_cached_trees` repeat the measurement with the
pruned-tree adjacency build / its assembly from a pre-pruned dataset inside every step.

For N > 1 every rank runs its own 50-sentence shard (weak scaling, no data-path collective) as SYNCHRONOUS data-parallel SGD
(BASELINE.json configs[3]; the reference updates every step at batch 50, train.py:209,224-227): the flat gradient bucket
[dW0,db0,dW1,db1] is all-reduced over RCCL, W -= lr * g runs on the device, and the next step's weight pack reads the updated
weights -- step i+1 depends on all-reduce i.  (`async_upper_bound` in the line is the free-running ring of round 2, where nothing
consumes the reduced gradient: NOT synchronous SGD, reported as a labelled secondary only.)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel, `launch_floor_us` (the
same launches with empty bodies) and `cpu_baseline` (the CPU oracle on this box's cores).
"""
import argparse
import copy
import ctypes
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
SGD_LR = 1e-9                  # N > 1: the device-side update between the all-reduce and the next step's pack
MAX_GRAD_NORM = 5.0            # train.py:85 --max_grad_norm default


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
    ap.add_argument("--no-riders", action="store_true",
                    help="A/B: both weight gradients in one launch of their own at the end of the backward sweep (gcnpt_set_option SIDE_TILES 0) "
                         "instead of layer 1's riding in layer 0's backward-data launch")
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
                    help="graph: the step's launches replayed as one hipGraph; native: eager launches from one native call per step "
                         "(gcnpt_layers_step), which keeps the queue fed as long as the host is fast enough; auto: both are tried during "
                         "warm-up and the faster one is timed")
    ap.add_argument("--exchange", choices=["sync", "async"], default="sync",
                    help="N > 1.  sync (the headline): all-reduce, device-side SGD update, and the next step's pack waits for both. "
                         "async: the round-2 ring of buckets whose all-reduce nobody consumes -- an upper bound, not data-parallel SGD")
    ap.add_argument("--repeats", type=int, default=20, help="N = 1: repeats of the timed K-step region behind the contract's one, for ms_per_step_median")
    ap.add_argument("--no-clip", action="store_true", help="N > 1, A/B: plain W -= lr*g instead of the reference's clip_grad_norm_ + SGD update")
    ap.add_argument("--torch-update", action="store_true", help="N > 1, A/B: clip + SGD update through torch's element-wise kernels instead of gcnpt_sgd_clip_update")
    ap.add_argument("--two-step-pack", action="store_true", help="A/B, packed layout with the tree build: gcnpt_prune_to_csr + gcnpt_pack_trees "
                                                                   "instead of the pruner writing the packed layout itself")
    ap.add_argument("--no-kernel-breakdown", action="store_true")
    ap.add_argument("--no-floor", action="store_true", help="skip the launch-floor leg (the step's launches with empty bodies)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements on rank 0 (with_prune, with_cached_trees, the fp32 "
                                                                "strict-parity block, pooling): quick runs and tests")
    ap.add_argument("--no-secondary-shapes", action="store_true",
                    help="skip the C5-shaped blocks (B=128, T=300, 600->300->300, K=2 on one GPU, padded and packed; its per-GPU shard of 8)")
    ap.add_argument("--separate-pack", action="store_true", help="A/B: keep the weight pack a launch of its own in the with_prune / "
                    "with_cached_trees steps instead of a side job of the tree launch")
    ap.add_argument("--no-pooled-only", action="store_true", help="skip the secondary pooled-only-rows measurement (profiling runs: its launches "
                                                                  "would mix into the per-kernel statistics of the headline step)")
    return ap.parse_args()


N_BUCKETS = 4       # async mode only: gradient buckets in flight


def source_hash():
    """Hash of the kernel sources: a PMC traffic recording is only quoted for the code it was made with."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gcn-over-pruned-trees_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    return h.hexdigest()[:16]


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
        # parameters as views of ONE flat fp32 tensor laid out like the gradient bucket [W0, b0, W1, b1]: the N > 1 update is one add_
        self.n_grad = H * Din + H + H * H + H
        self.wflat = torch.empty((self.n_grad,), dtype=torch.float32, device=dev)
        self.sgd_scratch = torch.empty((65,), dtype=torch.float32, device=dev)      # gcnpt_sgd_clip_update's partial sums + coefficient
        o = [0, H * Din, H * Din + H, H * Din + H + H * H, self.n_grad]
        self.W = [self.wflat[o[0]:o[1]].view(H, Din), self.wflat[o[2]:o[3]].view(H, H)]
        self.b = [self.wflat[o[1]:o[2]], self.wflat[o[3]:o[4]]]
        for dst, src in zip(self.W + self.b, list(Ws) + list(bs)):
            dst.copy_(t(src))
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
            self.padded_trees, self.lens_dev = self.trees, t(tb["lens"]).to(torch.int32).contiguous()
            self.pad_B, self.pad_T = B, T
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
        # flat gradient buckets [dW0, db0, dW1, db1] (one; a ring of them in the async exchange mode)
        self.buckets = [torch.zeros((self.n_grad,), dtype=torch.float32, device=dev) for _ in range(N_BUCKETS)]
        # loader-side pre-pruning (N4): a "dataset" of 20 batches pruned once; a step then only gathers its batch's rows
        reps = 20
        rep = lambda a: a.repeat(reps, 1)  # noqa: E731
        self.cache = tree.TreeCache.build(rep(self.head), rep(self.subj), rep(self.obj), rep(self.deprel), args.prune_k,
                                          masks=rep(self.masks), want_label=False)
        self.cache_idx = (torch.arange(B, device=dev) + B * (reps // 2)).to(torch.int64)
        self.scale = 1.0 / (1.0 - args.drop) if args.drop > 0 else 1.0
        self.side = torch.cuda.Stream(device=dev)
        if packed:
            self.B, self.T = self.rows, 0                   # what the C-ABI takes for packed rows (include/gcnpt.h, gcnpt_pack_trees)
        # small batches: layer 1's weight gradient rides in layer 0's backward-data launch (gcnpt_layer_bwd_data_wgrad)
        self.riders = (self.rows + 31) // 32 <= self.L.gcnpt_get_option(_lib.OPT_SIDE_TILES)

    def grads(self, k):
        H, Din = self.H, self.Din
        f = self.buckets[k]
        o = [0, H * Din, H * Din + H, H * Din + H + H * H]
        return f[o[0]:o[1]], f[o[1]:o[2]], f[o[2]:o[3]], f[o[3]:]

    # ---- tree launches (each only enqueues on the current stream) ----
    def prune(self, pack=False):
        """pack: the same launch also packs the weights (gcnpt_prune_to_csr_pack: side job on the CUs the tree build leaves idle).
        Token-packed layout: the pruner writes the packed arrays itself (gcnpt_prune_to_csr_packed: one launch; lengths and row count
        known: the loader has them, data/loader.py:109-121); --two-step-pack keeps round 3's gcnpt_prune_to_csr + gcnpt_pack_trees."""
        P, st = self._lib.ptr, self._lib.stream()
        tr = self.padded_trees if self.packed else self.trees
        B, T = (self.pad_B, self.pad_T) if self.packed else (self.B, self.T)
        if self.packed and not self.args.two_step_pack:
            pk = self.trees
            if not hasattr(self, "_sync_ws"):
                self._sync_ws = torch.zeros((B + 2,), dtype=torch.int64, device=self.dev)
            tail = self._native_args(0)[0] if pack else (0, None, None, None, 0, None, None)
            self._lib.check(self.L.gcnpt_prune_to_csr_packed(
                st, P(self.head), P(self.subj), P(self.obj), P(self.deprel), P(self.masks), None, B, T, self.args.prune_k, P(pk.cu_seqlens),
                P(pk.row_ptr), P(pk.col_idx), None, P(pk.rowT_ptr), P(pk.colT_idx), P(pk.ell), P(pk.ellT), P(pk.pool_mask), P(pk.row_sent), pk.N,
                pk.nnz_cap, P(pk.status), P(tr.status), P(tr.pool_mask), P(self._sync_ws), *tail))
            return
        a = (st, P(self.head), P(self.subj), P(self.obj), P(self.deprel), P(self.masks), None,
             B, T, self.args.prune_k, tr.cap, P(tr.row_ptr), P(tr.col_idx), None,
             P(tr.rowT_ptr), P(tr.colT_idx), P(tr.ell), P(tr.ellT), P(tr.pool_mask), P(tr.status))
        self._lib.check(self.L.gcnpt_prune_to_csr_pack(*(a + self._native_args(0)[0])) if pack else self.L.gcnpt_prune_to_csr(*a))
        if self.packed:
            pk = self.trees
            self._lib.check(self.L.gcnpt_pack_trees(
                st, P(tr.row_ptr), P(tr.col_idx), None, P(tr.rowT_ptr), P(tr.colT_idx), P(tr.ell), P(tr.ellT), P(tr.pool_mask),
                P(self.lens_dev), B, T, tr.cap, P(pk.cu_seqlens), P(pk.row_ptr), P(pk.col_idx), None, P(pk.rowT_ptr), P(pk.colT_idx),
                P(pk.ell), P(pk.ellT), P(pk.pool_mask), P(pk.row_sent), pk.N, pk.nnz_cap, P(pk.status)))

    def gather(self, pack=False):
        """The batch's PrunedTrees from the cached dataset (gcnpt_gather_trees) into the same buffers prune() fills."""
        tr, src, P = self.trees, self.cache.trees, self._lib.ptr
        a = (self._lib.stream(), P(src.row_ptr), P(src.col_idx), None, P(src.rowT_ptr), P(src.colT_idx), P(src.ell), P(src.ellT),
             P(src.pool_mask), P(src.status), P(self.cache.lens), src.B, src.T, src.cap, P(self.cache_idx), self.B, self.T, tr.cap,
             P(tr.row_ptr), P(tr.col_idx), None, P(tr.rowT_ptr), P(tr.colT_idx), P(tr.ell), P(tr.ellT), P(tr.pool_mask), P(tr.status))
        self._lib.check(self.L.gcnpt_gather_trees_pack(*(a + self._native_args(0)[0])) if pack else self.L.gcnpt_gather_trees(*a))

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
                   A(self.sf), None)
            bwd = (n, P(self.gy), A([self.h1, self.h2]), act, A(self.wb), P(tr.ell), P(tr.rowT_ptr), P(tr.colT_idx), P(tr.ellT), self.B, self.T,
                   Din, H, A([self.dx, self.dh1]), act, self.compute, (ctypes.c_float * n)(self.scale, 1.0), A(self.zf), A(self.sf),
                   A([g[0], g[2]]), A([g[1], g[3]]))
            vp = ctypes.c_void_p
            pack = (n, (vp * n)(*[w.data_ptr() for w in self.W]), H, Din, self.compute, A(self.wf), A(self.wb))
            self._nargs[k] = (pack, fwd, bwd)
        return self._nargs[k]

    def step_struct(self, k=0, parts=7):
        """gcnpt_step_t for gradient bucket k (built once per (k, parts)): the whole step is then ONE native call."""
        if not hasattr(self, "_steps"):
            self._steps = {}
        if (k, parts) not in self._steps:
            L, tr, g, n = self._lib, self.trees, self.grads(k), len(self.W)
            st = L.Step()
            st.n_layers, st.B, st.T, st.compute_dtype, st.parts, st.gy_is_dz = n, self.B, self.T, self.compute, parts, 0
            outs, dhs = [self.h1, self.h2], [self.dx, self.dh1]
            for l in range(n):
                st.W[l], st.bias[l], st.Din[l], st.H[l] = self.W[l].data_ptr(), self.b[l].data_ptr(), self.W[l].shape[1], self.W[l].shape[0]
                st.w_fwd[l], st.w_bwd[l] = self.wf[l].data_ptr(), self.wb[l].data_ptr()
                st.out[l], st.out_dtype[l] = outs[l].data_ptr(), self.act
                st.drop_p[l], st.seed[l] = (self.args.drop, 0x5eed) if l == 0 else (0.0, 0)
                st.s_frag[l], st.z_frag[l] = self.sf[l].data_ptr(), self.zf[l].data_ptr()
                st.dh[l], st.dh_dtype[l], st.scale[l] = dhs[l].data_ptr(), self.act, self.scale if l == 0 else 1.0
                st.dW[l], st.db[l] = g[2 * l].data_ptr(), g[2 * l + 1].data_ptr()
            P = L.ptr
            st.row_ptr, st.col_idx, st.ell, st.deg_ell = P(tr.row_ptr), P(tr.col_idx), P(tr.ell), None
            st.rowT_ptr, st.colT_idx, st.ellT, st.ell_bwd = P(tr.rowT_ptr), P(tr.colT_idx), P(tr.ellT), P(tr.ell)
            st.x, st.x_dtype, st.gy, st.seed_dev = P(self.x), self.act, P(self.gy), None
            self._steps[(k, parts)] = (st, ctypes.byref(st))
        return self._steps[(k, parts)][1]

    def launch_names(self):
        """The launches of one step, in order (what gcnpt_layers_bwd enqueues follows csrc/rowtile_kernels.hip, layers_bwd_impl)."""
        if self.riders:
            return ["pack", "fwd0", "fwd1", "bwd_data1", "bwd_data0+wgrad1", "bwd_weight0"]
        return ["pack", "fwd0", "fwd1", "bwd_data1", "bwd_data0", "bwd_weight"]

    def step_native(self, k=0, with_prune=False):
        """One step as ONE native call (gcnpt_layers_step: pack, all forward layers, backward sweep + weight gradients): eager launches,
        no graph.  The argument struct is built once: the host has ~45 us per step for six launches and must not spend them marshalling."""
        st = self._lib.stream()
        merged = bool(with_prune) and not self.args.separate_pack       # the tree launch carries the weight pack
        if with_prune == "cached":
            self.gather(pack=merged)
        elif with_prune:
            self.prune(pack=merged)
        # pack + forward sweep + backward sweep from ONE native call (gcnpt_layers_step: the argument struct is built once)
        rc = self.L.gcnpt_layers_step(st, self.step_struct(k, 6 if merged else 7))
        if rc:
            self._lib.check(rc)

    def step_prefix(self, n_launches, k=0):
        """The first n_launches launches of the step, through the same native entry points (measurement: t(k) - t(k-1))."""
        pack, fwd, bwd = self._native_args(k)
        st, L, nl = self._lib.stream(), self.L, len(self.W)
        rc = 0
        if n_launches >= 1:
            rc = L.gcnpt_pack_weights_multi(st, *pack)
        nf = min(n_launches - 1, nl)
        if rc == 0 and nf >= 1:
            rc = L.gcnpt_layers_fwd(st, nf, *fwd[1:])
        nb = n_launches - 1 - nl
        if rc == 0 and nb >= 1:
            rc = L.gcnpt_layers_bwd_range(st, *(bwd + (0, 0, nb)))
        if rc:
            self._lib.check(rc)

    def last_launch(self):
        v = [ctypes.c_int(0) for _ in range(4)]
        self._lib.check(self.L.gcnpt_last_launch(*[ctypes.byref(x) for x in v]))
        return tuple(x.value for x in v)

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
        bwd = (bwd[0], P(self.dtop)) + bwd[2:]
        rc = (L.gcnpt_pack_weights_multi(st, *pack) or L.gcnpt_layers_fwd(st, *fwd)
              or L.gcnpt_pool3_fwd(st, P(self.h2), self.act, P(tr.pool_mask), P(self.subj), P(self.obj), self.B, self.T, self.H, 0, P(self.pooled),
                                   P(self.amax)))
        if rc == 0 and handover:
            rc = (L.gcnpt_pool3_bwd_dz(st, P(self.gpool), P(self.amax), P(tr.pool_mask), P(self.subj), P(self.obj), self.B, self.T, self.H, 0,
                                       P(self.h2), P(tr.ell), 1.0, P(self.dtop), self.act) or L.gcnpt_layers_bwd_dz(st, *bwd))
        elif rc == 0:
            rc = (L.gcnpt_pool3_bwd(st, P(self.gpool), P(self.amax), P(tr.pool_mask), P(self.subj), P(self.obj), self.B, self.T, self.H, 0,
                                    P(self.dtop), self.act) or L.gcnpt_layers_bwd(st, *bwd))
        if rc:
            self._lib.check(rc)

    def step(self, k=0, with_prune=False):
        """
        One step for hipGraph capture.  With --streams 2 the tree build, which only needs the loader tensors, runs beside the weight pack
        on a side stream (fork and join are inside the step, so they become parallel branches of the captured graph).
        """
        if self.args.streams == 1 or not with_prune:
            self.step_native(k, with_prune)
            return
        main = torch.cuda.current_stream()
        side = self.side
        side.wait_stream(main)                           # fork: the tree build beside the weight pack
        with torch.cuda.stream(side):
            self.gather() if with_prune == "cached" else self.prune()
        pack, fwd, bwd = self._native_args(k)
        st, L = self._lib.stream(), self.L
        self._lib.check(L.gcnpt_pack_weights_multi(st, *pack))
        main.wait_stream(side)                           # join before the first layer
        self._lib.check(L.gcnpt_layers_fwd(st, *fwd) or L.gcnpt_layers_bwd(st, *bwd))

    # ---- bytes per launch (DESIGN.md "Measurement") ----
    def algorithmic_bytes(self):
        """What THIS dataflow moves per launch (every input and output counted once, saved-operand images and packed weights included)."""
        e = 2 if self.args.dtype == "bf16" else 4
        N, B, T = self.rows, self.B, self.T
        csr = 32 * N            # one ELL head (count + 7 columns) per row; the CSR arrays are only touched by rows with > 7 entries
        out = {}
        nl = len(self.W)
        for l, (H, Din) in enumerate([tuple(w.shape) for w in self.W]):
            wp = self.wf[l].numel()
            out["fwd%d" % l] = e * N * (Din + H) + wp + 4 * H + csr + self.sf[l].numel()
            # the top layer reads dY and Y, the layers below read the dZ the layer above left them; every layer but the bottom one
            # also reads its input rows to leave dZ for the layer below (hand-over of gcnpt_layers_bwd)
            top, bottom = l == nl - 1, l == 0
            out["bwd_data%d" % l] = e * N * ((2 if top else 1) * H + (1 if bottom else 2) * Din) + self.wb[l].numel() + 2 * csr + \
                self.zf[l].numel() + 4 * (H * Din + H)
            out["bwd_weight%d" % l] = self.zf[l].numel() + self.sf[l].numel() + 4 * (H * Din + H)
            out["bwd_weight"] = out.get("bwd_weight", 0) + out["bwd_weight%d" % l]
            out["pack"] = out.get("pack", 0) + 4 * H * Din + self.wf[l].numel() + self.wb[l].numel()
        out["bwd_data0+wgrad1"] = out["bwd_data0"] + out["bwd_weight1"]
        out["prune"] = 4 * 8 * N + N + 2 * (csr + 4 * B * (T + 1) + 4 * self.nnz) + N + 4 * (B + 1)
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
        per["bwd_data0+wgrad1"] = per["bwd_data0"] + per["bwd_weight1"]
        per["pack"] = 0
        per["prune"] = 0
        return per

    def survey_step_bytes(self):
        sv = self.survey_bytes()
        return sum(sv["fwd%d" % l] + sv["bwd_data%d" % l] + sv["bwd_weight%d" % l] for l in range(len(self.W)))


def capture(fn, use_graph):
    """Returns a callable that replays `fn` (a hipGraph when possible)."""
    if not use_graph:
        return fn, False
    try:
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread_local: with a process group alive its watchdog thread polls events while this thread captures; under the default (global)
        # capture mode that poll is an illegal call, the watchdog throws and the process aborts (seen once in ~4 single-rank RCCL runs)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
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
    Duration of each launch IN the step: the step is replayed truncated after its first k launches (Stack.step_prefix: the same native
    entry points, gcnpt_layers_bwd_range for the backward sweep), HIP events around the replays, and launch k is charged t(k) - t(k-1).
    Unlike timing a kernel alone back to back, this keeps the producer -> consumer cache state of the real step (each kernel reads what
    the previous one wrote from other XCDs) and includes its launch boundary.  One hipGraph holds `reps` copies of the truncated step
    (every launch of the step may be repeated: the backward clears the accumulators the weight gradient adds into), so the 5 us a
    graph replay costs on the device is spread over `reps` prefixes and the durations are those of back-to-back launches, as in the
    timed native-launch mode and in the rocprofv3 summaries.  `prune` is timed as the step with the tree build minus the step without.
    Also returns each launch's (grid, workgroup size, LDS bytes, kernel-argument bytes) for the launch-floor leg.
    """
    names = stack.launch_names()
    stack.step_native()
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

    out, shapes, prev = {}, [], 0.0
    for k in range(1, len(names) + 1):
        stack.step_prefix(k)
        shapes.append(stack.last_launch())
        t = timed_replay(lambda k=k: stack.step_prefix(k))
        out[names[k - 1]] = max(t - prev, 1e-9)
        prev = t
    if stack.B * stack.T == stack.rows_full:      # (a pooled-only stack holds [B, Tc] trees: the pruner's arrays do not fit them)
        out["prune"] = max(timed_replay(lambda: stack.step_native(0, with_prune=True)) - prev, 1e-9)
    stack.step_native()          # leave consistent buffers behind
    torch.cuda.synchronize()
    return out, shapes


def launch_floor(stack, shapes, steps, use_graph):
    """The step's launches -- same grids, workgroup sizes, LDS and kernel-argument sizes -- with bodies that return at entry
    (gcnpt_launch_empty_seq, one native call per step like the real step).  Two numbers: `device` = per step when 8
    steps' worth of empty launches are replayed as ONE hipGraph (the replay's fixed cost is spread out: what the dispatches cost the
    device back to back), `native` = the same eager loop the timed step runs in (what the host can issue).  What is left of
    ms_per_step after subtracting the device floor is the kernels' own work (one workgroup's dependent chain per launch)."""
    L, lib = stack.L, stack._lib
    n = len(shapes)
    cols = [(ctypes.c_int * n)(*[s[i] for s in shapes]) for i in range(4)]
    groups = [(0, n)]                               # one native call per step, as gcnpt_layers_step
    calls = [(hi - lo, [ctypes.cast(ctypes.byref(c, 4 * lo), ctypes.POINTER(ctypes.c_int)) for c in cols]) for lo, hi in groups if hi > lo]

    def empty_step():
        st = lib.stream()
        for cnt, ptrs in calls:
            rc = L.gcnpt_launch_empty_seq(st, cnt, *ptrs)
            if rc:
                lib.check(rc)

    wall, _ = timed(lambda i: empty_step(), steps, 50, lambda: None)
    out = {"native": wall / steps, "device": None}
    if use_graph:
        reps = 8

        def many():
            for _ in range(reps):
                empty_step()
        run, graphed = capture(many, True)
        if graphed:
            w, _ = timed(lambda i: run(), max(steps // reps, 50), 10, lambda: None)
            out["device"] = w / (max(steps // reps, 50) * reps)
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


def secondary_shapes(args, dev, seed, steps=150):
    """BASELINE.json configs[4]'s shape on ONE GPU, driver-timed: B=128, T=300, 600 -> 300 -> 300, prune_k 2, bf16 -- padded with
    full-length sentences (the SURVEY 8(d) byte count applies: fraction of the HBM peak over the whole step), token-packed with
    TACRED-shaped lengths (north_star "packed"), and the per-GPU shard of an 8-way split (B=16), with and without its tree build."""
    out = {}

    def run_one(a, packed=False, with_prune=False):
        s = Stack(a, dev, seed=seed, packed=packed)
        fn = (lambda: s.step_native(0, with_prune)) if with_prune else s.step_native
        fn()
        torch.cuda.synchronize()
        wall, _ = timed(lambda i: fn(), steps, 20, lambda: None)
        t = wall / steps
        r = {"ms_per_step": t * 1e3, "sentences_per_s": a.batch / t, "rows": s.rows, "steps": steps,
             "survey_8d_bytes": s.survey_step_bytes(), "frac_of_hbm_peak_8d": s.survey_step_bytes() / t / 1e9 / HBM_PEAK_GBS,
             "launches": ", ".join(s.launch_names())}
        del s
        torch.cuda.empty_cache()
        return r

    a = copy.copy(args)
    a.batch, a.seq, a.din, a.hidden, a.prune_k, a.dtype, a.lengths = 128, 300, 600, 300, 2, "bf16", "full"
    out["c5_shape"] = {"workload": "B=128 T=300 600->300->300 prune_k=2 bf16, one GPU",
                       "padded_full_length": run_one(a)}
    a2 = copy.copy(a)
    a2.lengths = "tacred"
    out["c5_shape"]["padded_tacred_lengths"] = run_one(a2)
    out["c5_shape"]["packed_tacred_lengths"] = run_one(a2, packed=True)
    a3 = copy.copy(a)
    a3.batch = 16
    shard = {"workload": "B=16 T=300 600->300->300 prune_k=2 bf16: one rank's share of configs[4] split 8 ways",
             "layers_only": run_one(a3), "with_prune": run_one(a3, with_prune=True), "with_cached_trees": run_one(a3, with_prune="cached")}
    shard["prune_us"] = (shard["with_prune"]["ms_per_step"] - shard["layers_only"]["ms_per_step"]) * 1e3
    a5 = copy.copy(a3)
    a5.lengths = "tacred"                                            # configs[4] as it says: "packed variable-length batches"
    shard["packed_tacred_lengths"] = run_one(a5, packed=True)
    shard["packed_tacred_lengths_with_prune"] = run_one(a5, packed=True, with_prune=True)      # + tree build + gcnpt_pack_trees
    out["per_gpu_shard_of_8"] = shard
    a4 = copy.copy(args)
    a4.batch, a4.lengths, a4.dtype = 1024, "full", "bf16"
    out["b1024_c2_widths"] = run_one(a4)
    return out


def self_launch(args):
    """
    `python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start N fresh rank processes, one per
    GPU, BEFORE anything in this process touches the GPU (this parent never does: it only waits), and relay rank 0's single
    JSON line.  The children are this same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly what
    `python -m torch.distributed.run --nproc-per-node N` would give them.  ALL children are polled: the first one that exits
    non-zero ends the others (by handle) and the run exits non-zero; nothing is restarted.
    """
    import socket
    import subprocess
    import threading
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
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            for p in procs:
                if p.poll() is None:
                    p.kill()                                        # by handle: these are exactly the processes started above
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
    reader.join(timeout=10)
    text = (chunks[0] if chunks else b"").decode("utf-8", "replace")
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
    force_dist = bool(os.environ.get("GCNPT_BENCH_FORCE_DIST"))      # rehearsal: the N > 1 code path (RCCL init, all-reduce, update) with one rank
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
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (the host driver only supports dmabuf IPC: RCCL across processes needs it)
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("GCNPT_BENCH_ONE_DEVICE"):          # rehearsal of the N > 1 code path on a 1-GPU box (with --dist-backend gloo)
            local = 0
        torch.cuda.set_device(local)
        dist.init_process_group(args.dist_backend, device_id=torch.device("cuda", local) if args.dist_backend == "nccl" else None)
        barrier = dist.barrier
    else:
        dist = None
        torch.cuda.set_device(0)
        barrier = lambda: None  # noqa: E731
    dev = torch.device("cuda", local if world > 1 else 0)
    use_graph = not args.no_graph
    from gcn_over_pruned_trees_amd import _lib
    if args.no_riders:
        _lib.set_option(_lib.OPT_SIDE_TILES, 0)

    # every rank draws its own shard of the global batch; GCNPT_BENCH_SAME_SHARD=1 (tests) gives all ranks rank 0's shard, so that
    # the all-reduced bucket must be exactly world x the one-rank bucket
    shard_seed = 1234 + (0 if os.environ.get("GCNPT_BENCH_SAME_SHARD") else 17 * rank)
    stack = Stack(args, dev, seed=shard_seed, packed=args.layout == "packed")
    from gcn_over_pruned_trees_amd.shard import OverlappedAllReduce

    multi = world > 1 or force_dist
    w0 = stack.wflat.clone()                                         # N > 1: the weights every trial / the timed region starts from

    def runner(mode, with_prune=False, n_buckets=1):
        """[(callable, is_graph)] per gradient bucket for a launch mode."""
        if mode == "native":
            fns = [((lambda k=k: stack.step_native(k, with_prune)), False) for k in range(n_buckets)]
            for f, _ in fns:
                f()
            torch.cuda.synchronize()
            return fns
        return [capture(lambda k=k: stack.step(k, with_prune=with_prune), use_graph) for k in range(n_buckets)]

    def make_run(replays, exchange, reducer=None):
        """A step (+ for N > 1 the gradient exchange).  sync: the bucket is all-reduced on the compute stream's order (a synchronous
        all_reduce: the collective's latency is in line with the step) and W -= lr * g (SUM over ranks / world) is applied on the device;
        the next step's pack reads W, so step i+1 depends on all-reduce i: synchronous data-parallel SGD, as the reference's
        per-step update (train.py:224-227).  async: ring of buckets, asynchronous all-reduce nobody consumes (upper bound)."""
        def run_sync(i):
            replays[0][0]()
            g = stack.buckets[0]
            dist.all_reduce(g)                                       # SUM: supported by every backend; the 1/world goes into the step size
            if args.torch_update:                                    # A/B: the same update through six element-wise library kernels
                coef = (MAX_GRAD_NORM / (torch.linalg.vector_norm(g) / world + 1e-6)).clamp_(max=1.0)
                stack.wflat.addcmul_(g, coef, value=-SGD_LR / world)
                return
            # the reference's update: clip_grad_norm_(max_grad_norm = 5) then SGD (train.py:224-227) on the flat parameter tensor, one
            # native call (gcnpt_sgd_clip_update: two launches, no host sync); --no-clip: the plain update
            stack._lib.check(stack.L.gcnpt_sgd_clip_update(stack._lib.stream(), stack._lib.ptr(stack.wflat), stack._lib.ptr(g), g.numel(), 1.0 / world,
                                                           0.0 if args.no_clip else MAX_GRAD_NORM, SGD_LR, stack._lib.ptr(stack.sgd_scratch), None))

        def run_async(i):
            k = i % N_BUCKETS
            reducer.before_write(k)
            replays[k][0]()
            reducer.after_write(k)

        def run_single(i):
            replays[0][0]()
        return run_single if not multi else (run_sync if exchange == "sync" else run_async)

    modes = ["graph"] if args.launch == "graph" else (["native"] if args.launch == "native" else ["graph", "native"])
    exchange = args.exchange if multi else None
    nb = N_BUCKETS if (multi and exchange == "async") else 1
    cands = {m: runner(m, n_buckets=nb) for m in modes}
    reducer = OverlappedAllReduce(stack.buckets, dist, average=False) if (multi and exchange == "async") else None

    def drain():
        if reducer:
            reducer.finish()
        torch.cuda.synchronize()

    launch, trial = modes[0], {}
    if len(modes) > 1:
        # part of the warm-up: a short trial of each launch mode; every rank must take the same one: MAX over ranks decides
        for m in modes:
            stack.wflat.copy_(w0)
            w, _ = timed(make_run(cands[m], exchange, reducer), 200, 20, barrier)
            drain()
            if multi:
                tm = torch.tensor([w], dtype=torch.float64, device=dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                w = float(tm.item())
            trial[m] = w / 200
        launch = min(modes, key=lambda m: trial[m])
    replays = cands[launch]
    graphed = replays[0][1]
    stack.wflat.copy_(w0)
    wall, ev = timed(make_run(replays, exchange, reducer), args.steps, args.warmup, barrier)
    drain()
    # the contract's `value` / `ms_per_step` are that ONE K-step region; beside it the median over repeats of the same region (a
    # 20-step region is ~1 ms: one sample is at the mercy of whatever else the host does in that millisecond)
    reps_ms = []
    if not multi:
        for _ in range(args.repeats):
            w_r, _ = timed(make_run(replays, exchange, reducer), args.steps, 0, barrier)
            reps_ms.append(w_r / args.steps * 1e3)
        drain()
    rank_ms = [wall / args.steps * 1e3]
    if multi:
        mine = torch.tensor([wall], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        wall = max(float(t.item()) for t in every)                 # contract: MAX over ranks
    assert torch.isfinite(stack.buckets[0]).all() and torch.isfinite(stack.dx.float()).all() and torch.isfinite(stack.wflat).all()
    # the bucket the LAST timed step wrote, after its all-reduce: rank-sum of [dW0, db0, dW1, db1]
    last_bucket = stack.buckets[(args.steps - 1) % N_BUCKETS if (multi and exchange == "async") else 0]
    grad_abs_sum = float(last_bucket.double().abs().sum().item())
    weight_drift = float((stack.wflat - w0).double().abs().sum().item())      # > 0 iff the SGD updates were applied (N > 1, sync)

    async_bound = None
    if multi and exchange == "sync":
        # labelled secondary: the free-running ring (not synchronous SGD)
        stack.wflat.copy_(w0)
        ring = OverlappedAllReduce(stack.buckets, dist, average=False)
        fns = runner(launch, n_buckets=N_BUCKETS)

        def run_ring(i):
            k = i % N_BUCKETS
            ring.before_write(k)
            fns[k][0]()
            ring.after_write(k)
        n_a = min(args.steps, 400)
        w_a, _ = timed(run_ring, n_a, 20, barrier)
        ring.finish()
        torch.cuda.synchronize()
        tm = torch.tensor([w_a], dtype=torch.float64, device=dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        async_bound = {"ms_per_step": float(tm.item()) / n_a * 1e3, "value": args.batch * world * n_a / float(tm.item()), "steps": n_a,
                       "note": "NOT synchronous SGD: ring of %d buckets, the all-reduce of step i runs beside steps i+1.. and nothing consumes "
                               "it; what the exchange could hide at best" % N_BUCKETS}
    stack.wflat.copy_(w0)

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
        names = stack.launch_names()
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
                       "launch": "hipGraph replay" if graphed else "eager launches from 1 native call per step (gcnpt_layers_step)",
                       "kernels_per_step": ", ".join(names),
                       "launch_trial_us_per_step": {m: round(t * 1e6, 2) for m, t in trial.items()} or None, "nnz_per_batch": stack.nnz,
                       "dp_mode": ("synchronous SGD: step -> all_reduce(SUM) of the flat fp32 bucket (%d B) over %s -> clip to max_grad_norm %g + W -= lr*g/world on the device (gcnpt_sgd_clip_update: 2 launches) -> "
                                   "the next step's weight pack reads W" % (4 * stack.n_grad, args.dist_backend, MAX_GRAD_NORM)) if (multi and exchange == "sync")
                                  else ("ASYNC RING (not synchronous SGD; --exchange async)" if multi else "none (1 GPU)"),
                       "weight_abs_drift_after_timed_steps": weight_drift if multi else None},
            "event_ms_per_step": ev / args.steps * 1e3,
            "ms_per_step_median": float(np.median(reps_ms)) if reps_ms else None,
            "ms_per_step_repeats": {"n": len(reps_ms), "min": float(np.min(reps_ms)), "max": float(np.max(reps_ms))} if reps_ms else None,
            "launch_mode": "hipGraph replay" if graphed else "eager, one native call per step",
            "launch_trial_us_per_step": {m: round(t * 1e6, 2) for m, t in trial.items()} or None,
        }
        if async_bound is not None:
            result["async_upper_bound"] = async_bound
        if not args.no_secondary:
            result["with_prune"] = {"value": args.batch * args.steps / wall_p, "unit": "sentences/s", "ms_per_step": wall_p / args.steps * 1e3,
                                    "note": "rank 0, pruned-tree adjacency build inside every step; the tree launch carries the weight pack as a "
                                            "side job (gcnpt_prune_to_csr_pack: one launch boundary less)"}
            result["with_cached_trees"] = {"value": args.batch * args.steps / wall_c, "unit": "sentences/s", "ms_per_step": wall_c / args.steps * 1e3,
                                           "note": "rank 0, dataset pruned once; every step assembles its batch's adjacency with gcnpt_gather_trees_pack (the weight "
                                                   "pack rides in the same launch)"}
            if args.dtype == "bf16":
                # the reference's own arithmetic: fp32 activations, exact fp32 MFMA (the strict-parity mode of the tests), same step
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
        if not args.no_secondary and args.layout == "padded":
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
        if not args.no_pooled_only and args.layout == "padded" and args.lengths == "tacred":
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
        if not args.no_pooled_only and args.layout == "padded":
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
            kt, shapes = kernel_breakdown(stack, use_graph)
            step_keys = [k for k in kt if k != "prune"]
            dom = max(step_keys, key=lambda k: kt[k])
            sv = stack.survey_bytes()
            # HBM-side bytes per launch from separate rocprofv3 --pmc passes (tools/profile_round.sh): only quoted when that recording was
            # made for THIS shape / dtype / launch list AND these kernel sources (hash of csrc/*.hip, *.h), otherwise null
            traffic, traffic_src = None, None
            tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tf):
                with open(tf) as f:
                    rec = json.load(f)
                meta = rec.get("meta", {})
                here = dict(batch=args.batch, seq=args.seq, din=args.din, hidden=args.hidden, prune_k=args.prune_k, dtype=args.dtype,
                            lengths=args.lengths, kernels=names, source_hash=source_hash())
                if all(meta.get(k) == v for k, v in here.items()):
                    traffic = rec.get("traffic", {}).get(dom)
                    traffic_src = ("profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload and these kernel "
                                   "sources, hash %s, 2*FETCH+WRITE)" % meta.get("source_hash"))
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
            if not args.no_floor and world == 1:
                fl = launch_floor(stack, shapes, min(args.steps, 2000), use_graph)
                floor = fl["device"] if fl["device"] is not None else fl["native"]
                result["launch_floor_us"] = floor * 1e6
                result["chain_us"] = (wall / args.steps - floor) * 1e6
                result["launch_floor"] = {"launches": [{"name": n, "grid": s[0], "block": s[1], "lds_bytes": s[2], "kernarg_bytes": s[3]}
                                                       for n, s in zip(names, shapes)],
                                          "per_launch_us": floor * 1e6 / len(shapes),
                                          "native_loop_us_per_step": fl["native"] * 1e6,
                                          "note": "the step's %d launches with the same grid / workgroup size / LDS / kernel-argument size and bodies that return "
                                                  "at entry (gcnpt_launch_empty_seq): launch_floor_us = per step when 8 steps of them replay as one hipGraph "
                                                  "(device-side dispatch cost, back to back); native_loop_us_per_step = the same one native call per step "
                                                  "issued eagerly (host issue rate: with empty kernels the host, not the device, is the limit); "
                                                  "chain_us = ms_per_step - launch_floor_us is what the kernels' own work adds" % len(shapes)}
        if not args.no_secondary_shapes and world == 1 and args.layout == "padded":
            result["secondary_shapes"] = secondary_shapes(args, dev, shard_seed)
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
