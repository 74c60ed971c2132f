"""
The opt-in two-layer forward launch (csrc/fused_kernels.hip: gcnpt_fused2_fwd) against the per-layer kernel it replaces
(gcnpt_layer_fwd, itself pinned to the reference's goldens in test_gpu_parity.py).  The fused kernel recomputes layer 0 on the
halo rows instead of waiting for other workgroups, with the same summation order, k order, epilogue arithmetic and bf16
rounding points: every output must be BIT-IDENTICAL (reference arithmetic: model/gcn.py:266-271, 390-393).
"""
import ctypes

import numpy as np
import pytest
import torch

from gcn_over_pruned_trees_amd.utils import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def L():
    from gcn_over_pruned_trees_amd import _lib
    return _lib


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _bits(t):
    return t.contiguous().view(torch.uint8)


class Case(object):
    """Inputs, packed weights and output buffers of a 2-layer stack on given trees."""

    def __init__(self, L, dev, trees, dims, seed, out_dtype=torch.bfloat16, drop=0.5):
        self.L, self.lib, self.dev, self.trees = L, L.lib(), dev, trees
        self.B, self.T = trees.B, trees.T
        self.Din, self.H0, self.H1 = dims
        self.drop, self.out_dtype = drop, out_dtype
        B, T = self.B, self.T
        Ws, bs = synthetic.layer_params(seed + 1, list(dims))
        self.W = [_t(w, dev) for w in Ws]
        self.b = [_t(b, dev) for b in bs]
        self.x = _t(synthetic.normal(seed + 2, (B, T, dims[0])), dev).to(torch.bfloat16)
        self.gy = _t(synthetic.normal(seed + 3, (B, T, dims[2])), dev).to(out_dtype)
        lib, c = self.lib, L.BF16
        u8 = dict(dtype=torch.uint8, device=dev)
        shapes = [(dims[1], dims[0]), (dims[2], dims[1])]                        # (H_l, Din_l)
        self.shapes = shapes
        self.wf = [torch.empty((lib.gcnpt_packed_bytes(h, k, c),), **u8) for h, k in shapes]
        self.wb = [torch.empty((lib.gcnpt_packed_bytes(k, h, c),), **u8) for h, k in shapes]
        for l, (h, k) in enumerate(shapes):
            L.check(lib.gcnpt_pack_weights(L.stream(), L.ptr(self.W[l]), h, k, c, L.ptr(self.wf[l]), L.ptr(self.wb[l])))

    def bufs(self):
        B, T, dev = self.B, self.T, self.dev
        lib, c = self.lib, self.L.BF16
        u8 = dict(dtype=torch.uint8, device=dev)
        o = dict(h1=torch.full((B, T, self.H0), float("nan"), dtype=torch.bfloat16, device=dev),
                 h2=torch.full((B, T, self.H1), float("nan"), dtype=self.out_dtype, device=dev),
                 sf=[torch.full((lib.gcnpt_frag_bytes(B * T, k, c),), 0xA5, **u8) for _, k in self.shapes],
                 zf=[torch.full((lib.gcnpt_frag_bytes(B * T, h, c),), 0xA5, **u8) for h, _ in self.shapes],
                 dh1=torch.full((B, T, self.H0), float("nan"), dtype=torch.bfloat16, device=dev),
                 dx=torch.full((B, T, self.Din), float("nan"), dtype=torch.bfloat16, device=dev),
                 dW=[torch.full((h, k), 7.0, dtype=torch.float32, device=dev) for h, k in self.shapes],
                 db=[torch.full((h,), 7.0, dtype=torch.float32, device=dev) for h, _ in self.shapes])
        return o

    def fwd_layers(self, o, no_adj=False):
        L, lib, tr, P = self.L, self.lib, self.trees, self.L.ptr
        g_ell = tr.empty_ell() if no_adj else tr.ell
        ps, seeds = (self.drop, 0.0), (0x5eed, 0)
        src = self.x
        for l, dst in enumerate((o["h1"], o["h2"])):
            h, k = self.shapes[l]
            L.check(lib.gcnpt_layer_fwd(L.stream(), P(src), L.dtype_code(src.dtype), P(self.wf[l]), P(self.b[l]), P(tr.row_ptr), P(tr.col_idx),
                                        P(g_ell), P(tr.ell), self.B, self.T, k, h, P(dst), L.dtype_code(dst.dtype), L.BF16, ps[l], seeds[l],
                                        P(o["sf"][l]), None))
            src = dst

    def fwd_fused(self, o, no_adj=False):
        L, lib, tr, P, A = self.L, self.lib, self.trees, self.L.ptr, self.L.ptr_array
        g_ell = tr.empty_ell() if no_adj else tr.ell
        L.check(lib.gcnpt_fused2_fwd(L.stream(), P(self.x), A(self.wf), A(self.b), P(tr.row_ptr), P(tr.col_idx), P(g_ell), P(tr.ell),
                                     self.B, self.T, self.Din, (ctypes.c_int * 2)(self.H0, self.H1), P(o["h1"]), P(o["h2"]),
                                     L.dtype_code(self.out_dtype), (ctypes.c_float * 2)(self.drop, 0.0), (ctypes.c_uint64 * 2)(0x5eed, 0),
                                     A(o["sf"]), None))


def _trees(L, dev, seed, B, T, lengths, K):
    from gcn_over_pruned_trees_amd.model import tree
    tb = synthetic.random_tree_batch(seed, B, T, lengths)
    tr = tree.prune_to_csr(*(_t(tb[k], dev) for k in ("head", "subj_pos", "obj_pos", "deprel")), K, masks=_t(tb["masks"], dev), want_label=False)
    tr.check(expect_maxlen=T)
    return tr


def _dense_trees(L, dev, seed, B, T, density):
    """An explicit dense adjacency (GCN.forward(adj, ...)): rows with more than 7 entries (CSR tail) and halos of several passes."""
    from gcn_over_pruned_trees_amd.model import tree
    rng = np.random.RandomState(seed)
    adj = (rng.random_sample((B, T, T)) < density).astype(np.float32) * rng.randint(1, 84, size=(B, T, T))
    adj[:, -3:, :] = 0
    adj[:, :, -3:] = 0                                    # a few empty rows / columns as padding has
    return tree.adj_to_csr(_t(adj.astype(np.float32), dev), want_label=False)


CASES = [
    # name, trees factory, dims (Din, H0, H1), out dtype
    ("c2_full", lambda L, d: _trees(L, d, 1234, 50, 100, "full", 1), (360, 200, 200), torch.bfloat16),
    ("c2_tacred_f32out", lambda L, d: _trees(L, d, 77, 50, 100, "tacred", 1), (360, 200, 200), torch.float32),
    ("c3_k2_ragged_tail", lambda L, d: _trees(L, d, 5, 7, 45, "tacred", 2), (400, 200, 200), torch.bfloat16),       # 315 rows: last tile partial
    ("c1_shape", lambda L, d: _trees(L, d, 9, 4, 20, "full", 1), (200, 200, 200), torch.float32),
    ("dense_adj_multi_pass", lambda L, d: _dense_trees(L, d, 3, 3, 70, 0.25), (360, 200, 200), torch.bfloat16),
    ("dense_adj_sparse", lambda L, d: _dense_trees(L, d, 4, 5, 33, 0.05), (200, 200, 200), torch.bfloat16),
]


@pytest.mark.parametrize("name,make,dims,odt", CASES, ids=[c[0] for c in CASES])
def test_fused_forward_bit_identical(L, dev, name, make, dims, odt):
    tr = make(L, dev)
    assert L.lib().gcnpt_fused2_supported(tr.T, dims[0], dims[1], dims[2], L.dtype_code(odt), L.BF16) == 1
    for no_adj in (False, True):
        c = Case(L, dev, tr, dims, seed=11, out_dtype=odt)
        a, b = c.bufs(), c.bufs()
        c.fwd_layers(a, no_adj)
        c.fwd_fused(b, no_adj)
        torch.cuda.synchronize()
        assert torch.isfinite(a["h2"].float()).all()
        for key in ("h1", "h2"):
            assert torch.equal(_bits(a[key]), _bits(b[key])), (name, key, no_adj, int((_bits(a[key]) != _bits(b[key])).sum()))
        for l in range(2):
            assert torch.equal(a["sf"][l], b["sf"][l]), (name, "s_frag%d" % l, no_adj)
        if not no_adj and "dense" not in name:
            assert (a["h1"] == 0).float().mean() > 0.3             # dropout really ran in both


def test_fused_unsupported_shapes_are_refused(L, dev):
    lib = L.lib()
    assert lib.gcnpt_fused2_supported(300, 600, 304, 304, L.BF16, L.BF16) == 0            # window of 2 (T-1) rows does not fit
    assert lib.gcnpt_fused2_supported(100, 360, 200, 200, L.BF16, L.F32) == 0             # exact-fp32 mode stays per layer
    assert lib.gcnpt_fused2_supported(100, 363, 200, 200, L.BF16, L.BF16) == 0
