"""
Token-packed layout (north_star "the batch of variable-length pruned trees is packed"): the reference pads
every batch to its longest sentence (data/loader.py:109-121, model/gcn.py:96-97,106), here the layer loop runs on sum(len) rows.

  * gcnpt_pack_trees: the packed block-diagonal CSR / ELL heads hold exactly the reference's per-sentence matrices (integer: exact)
  * the layer kernels on packed rows (T = 0): every real token's row is BIT-IDENTICAL to the padded path's, forward and dx;
    fp32 forward <= 1e-5 and gradients <= 1e-4 against the CPU oracle (the tolerances of SURVEY.md 8c)
  * pad / unpad only at the module boundary: GCNClassifier with opt['gcn_packed'] gives the reference's recorded logits
"""
import json

import numpy as np
import pytest
import torch

from conftest import dense_from_coo, load_golden
from helpers import FWD_RTOL, GRAD_RTOL, max_rel
from gcn_over_pruned_trees_amd.utils import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def api():
    from gcn_over_pruned_trees_amd import _lib
    from gcn_over_pruned_trees_amd.model import gcn, tree
    _lib.lib()
    return gcn, tree


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_pack_trees_structure(api, dev):
    gcn, tree = api
    g = load_golden("trees_random.npz")
    S, Ts = g["head"].shape
    lens = g["lens"].astype(np.int64)
    for K in (0, 2):
        full = tree.prune_to_csr(_t(g["head"], dev), _t(g["subj_pos"], dev), _t(g["obj_pos"], dev), _t(g["deprel"], dev), K,
                                 lens=_t(lens.astype(np.int32), dev)).check()
        pk = full.pack(lens.tolist()).check()
        N = int(lens.sum())
        assert pk.N == N
        cu = pk.cu_seqlens.cpu().numpy()
        np.testing.assert_array_equal(cu, np.concatenate([[0], np.cumsum(lens)]))
        ref = dense_from_coo(g["coo_k%d" % K], S, Ts)
        for name_ptr, name_col, transpose in (("row_ptr", "col_idx", False), ("rowT_ptr", "colT_idx", True)):
            rp = getattr(pk, name_ptr).cpu().numpy()
            ci = getattr(pk, name_col).cpu().numpy()
            lab = pk.label.cpu().numpy() if not transpose else None
            ell = (pk.ellT if transpose else pk.ell).cpu().numpy().reshape(N, 8)
            assert rp[0] == 0 and (np.diff(rp) >= 0).all()
            for b in range(S):
                n = int(lens[b])
                want = ref[b, :n, :n].T if transpose else ref[b, :n, :n]
                got = np.zeros((n, n), np.float32)
                for i in range(n):
                    r = cu[b] + i
                    cols = ci[rp[r]:rp[r + 1]]
                    assert ((cols >= cu[b]) & (cols < cu[b + 1])).all()                   # block diagonal: never leaves the sentence
                    assert (np.diff(cols) > 0).all()                                       # ascending, as a dense -> CSR conversion gives
                    got[i, cols - cu[b]] = lab[rp[r]:rp[r + 1]] if lab is not None else 1.0
                    assert ell[r, 0] == len(cols)
                    k = min(len(cols), 7)
                    np.testing.assert_array_equal(ell[r, 1:1 + k], cols[:k])
                    assert (ell[r, 1 + k:] == 0).all()
                np.testing.assert_array_equal(got, want if lab is not None else (want != 0).astype(np.float32))
            assert rp[N] == (ref != 0).sum()
        pm = pk.pool_mask.cpu().numpy()[:, 0]
        in_tree = (ref != 0).any(2) | (ref != 0).any(1)
        np.testing.assert_array_equal(pm, np.concatenate([~in_tree[b, :lens[b]] for b in range(S)]))
        np.testing.assert_array_equal(pk.row_sent.cpu().numpy(), np.repeat(np.arange(S), lens))
    # too few rows allocated: reported, not overrun
    small = full.pack(lens.tolist(), n_rows=N - 5)
    with pytest.raises(Exception):
        small.check()


def _same_packed(a, b, what):
    """Two PackedTrees hold the same batch, array by array (only the allocated-but-unused tails may differ)."""
    N = int(a.cu_seqlens[-1])
    assert N == int(b.cu_seqlens[-1]) and a.N == b.N == max(N, 1), what        # (a batch without a token still allocates one row)
    assert torch.equal(a.cu_seqlens, b.cu_seqlens) and torch.equal(a.status, b.status), what
    assert torch.equal(a.row_ptr[:N + 1], b.row_ptr[:N + 1]) and torch.equal(a.ell, b.ell) and torch.equal(a.pool_mask, b.pool_mask) and torch.equal(a.row_sent, b.row_sent), what
    nnz = int(a.row_ptr[N])
    assert torch.equal(a.col_idx[:nnz], b.col_idx[:nnz]), what
    if a.label is not None:
        assert torch.equal(a.label[:nnz], b.label[:nnz]), what
    if a.rowT_ptr is not None:
        nnzT = int(a.rowT_ptr[N])
        assert torch.equal(a.rowT_ptr[:N + 1], b.rowT_ptr[:N + 1]) and torch.equal(a.ellT, b.ellT) and torch.equal(a.colT_idx[:nnzT], b.colT_idx[:nnzT]), what


@pytest.mark.parametrize("shape", ["golden_wave0", "long_allwaves", "many_sentences", "errors", "shard16"])
def test_prune_to_csr_packed_equals_pack_of_prune(api, dev, shape):
    """the pruner writing the packed layout itself (gcnpt_prune_to_csr_packed: offsets by a look-back over the
    sentences before, inside the launch) gives exactly gcnpt_pack_trees(gcnpt_prune_to_csr(...)) -- every array, bit for bit -- for the
    wave-0 form (T <= 64), the all-waves form, more sentences than the chip holds workgroups (ticket order), sentences that fail,
    with and without labels / the transposed pattern, repeatedly on the same workspace (it cleans up after itself)."""
    gcn, tree = api
    if shape == "golden_wave0":
        g = load_golden("trees_random.npz")
        head, subj, obj, dep, lens = g["head"], g["subj_pos"], g["obj_pos"], g["deprel"], g["lens"].astype(np.int64)
        Ks = (0, 2)
    elif shape == "errors":
        g = load_golden("trees_edge_cases.npz")
        head, subj, obj, dep, lens = g["head"], g["subj_pos"], g["obj_pos"], g["deprel"], g["lens"].astype(np.int64)
        Ks = (1,)
    else:
        B, T = {"long_allwaves": (24, 300), "many_sentences": (700, 40), "shard16": (16, 300)}[shape]      # shard16: configs[4]'s per-GPU share
        tb = synthetic.random_tree_batch(17, B, T, "tacred", overlap_frac=0.1)
        head, subj, obj, dep, lens = tb["head"], tb["subj_pos"], tb["obj_pos"], tb["deprel"], tb["lens"].astype(np.int64)
        Ks = (1, 2)
    args = [_t(a, dev) for a in (head, subj, obj, dep)]
    lens_dev = _t(lens.astype(np.int32), dev)
    for K in Ks:
        for want_label, want_T in ((True, True), (False, True), (False, False)):
            two = tree.prune_to_csr(*args, K, lens=lens_dev, want_label=want_label, want_transpose=want_T).pack(lens.tolist())
            for rep in range(2):                                         # twice: the second call finds the workspace as the first left it
                one = tree.prune_to_csr_packed(*args, K, lens.tolist(), want_label=want_label, want_transpose=want_T)
                torch.cuda.synchronize()
                _same_packed(one, two, (shape, K, want_label, want_T, rep))
                assert torch.equal(one.padded.status, two.padded.status)
                assert torch.equal(one.padded.pool_mask, two.padded.pool_mask)
    # the lengths from the pad mask, as the reference takes them (gcn.py:96), and too few rows allocated: reported, not overrun
    masks = _t(np.arange(head.shape[1])[None, :] >= lens[:, None], dev)
    one = tree.prune_to_csr_packed(*args, Ks[0], lens.tolist(), masks=masks)
    _same_packed(one, tree.prune_to_csr(*args, Ks[0], masks=masks, want_label=False).pack(lens.tolist()), (shape, "masks"))
    small = tree.prune_to_csr_packed(*args, Ks[0], lens.tolist(), n_rows=int(lens.sum()) - 3)
    torch.cuda.synchronize()
    assert int(small.status[0]) == -8                                   # GCNPT_E_CAPACITY
    again = tree.prune_to_csr_packed(*args, Ks[0], lens.tolist())        # and the workspace survived that, too
    _same_packed(again, one, (shape, "after capacity"))


def test_prune_to_csr_packed_random_shapes(api, dev):
    """The look-back of the packed pruner under many grid shapes: 40 random (B, T, K, length profile) batches -- one sentence to more
    sentences than the chip holds workgroups, widths on both sides of the wave-0 / all-waves switch, zero-length sentences, corrupted
    parses (a failing sentence still owns its rows) -- every array equal to the two-step form, the workspace reused throughout."""
    gcn, tree = api
    rng = np.random.RandomState(99)
    for case in range(40):
        B = int(rng.choice([1, 2, 3, 7, 16, 50, 129, 300, 513]))
        T = int(rng.choice([1, 2, 5, 33, 64, 65, 100, 128, 257, 400]))
        if B * T > 60000:
            B = max(60000 // T, 1)
        K = int(rng.randint(0, 4))
        profile = rng.choice(["full", "tacred", "ragged"])
        if profile == "ragged":
            lens = rng.randint(0, T + 1, size=B).astype(np.int32)
            lens[0] = T
            tb = synthetic.random_tree_batch(1000 + case, B, T, np.maximum(lens, 1)[np.argsort(-np.maximum(lens, 1), kind="stable")])
            lens = tb["lens"].astype(np.int64)
            short = rng.rand(B) < 0.1                                     # some sentences are empty: no row, no entry
            short[0] = False
            lens = np.where(short, 0, lens)
        else:
            tb = synthetic.random_tree_batch(1000 + case, B, T, str(profile), overlap_frac=0.2)
            lens = tb["lens"].astype(np.int64)
        head = tb["head"].copy()
        bad = np.flatnonzero(rng.rand(B) < 0.1)
        for b_ in bad:                                                    # a head past the sentence / a cycle: the sentence fails
            if lens[b_] >= 2:
                head[b_, 0], head[b_, 1] = (2, 1) if rng.rand() < 0.5 else (lens[b_] + 5, head[b_, 1])
        args = [_t(a, dev) for a in (head, tb["subj_pos"], tb["obj_pos"], tb["deprel"])]
        lens_dev = _t(lens.astype(np.int32), dev)
        want_label, want_T = bool(case & 1), bool(case & 2)
        two = tree.prune_to_csr(*args, K, lens=lens_dev, want_label=want_label, want_transpose=want_T).pack(lens.tolist())
        one = tree.prune_to_csr_packed(*args, K, lens.tolist(), want_label=want_label, want_transpose=want_T)
        torch.cuda.synchronize()
        _same_packed(one, two, (case, B, T, K, str(profile)))
        assert torch.equal(one.padded.status, two.padded.status), (case, B, T, K)
        assert torch.equal(one.padded.pool_mask, two.padded.pool_mask), (case, B, T, K)


@pytest.mark.parametrize("shape", ["golden", "errors", "long"])
def test_cache_batch_packed_equals_pack_of_batch(api, dev, shape):
    """VERDICT r3 item 2, the cache's half: TreeCache.batch_packed (gcnpt_gather_trees_packed) gives exactly batch(idx, T).pack(lens[idx])
    -- every array, bit for bit -- for batches with repeats, a width narrower than the cache's (sentences too long for it fail as
    they do in the padded form), failed sentences, sentence numbers out of range, with/without labels and the transposed pattern."""
    gcn, tree = api
    if shape == "long":
        tb = synthetic.random_tree_batch(23, 64, 280, "tacred", overlap_frac=0.1)
        head, subj, obj, dep, lens = tb["head"], tb["subj_pos"], tb["obj_pos"], tb["deprel"], tb["lens"].astype(np.int64)
    else:
        g = load_golden("trees_random.npz" if shape == "golden" else "trees_edge_cases.npz")
        head, subj, obj, dep, lens = g["head"], g["subj_pos"], g["obj_pos"], g["deprel"], g["lens"].astype(np.int64)
    S, Ts = head.shape
    cache = tree.TreeCache.build(*(_t(a, dev) for a in (head, subj, obj, dep)), 1, lens=_t(lens.astype(np.int32), dev), want_label=True)
    rng = np.random.RandomState(5)
    for trial in range(4):
        B = int(rng.randint(1, 2 * S))
        idx = rng.randint(0, S, size=B)
        if trial == 3:
            idx[rng.randint(0, B)] = S + 3                                # not a sentence of the cache
        T = int(max(lens[np.minimum(idx, S - 1)].max(), 1)) if trial != 2 else max(int(np.median(lens)), 1)   # trial 2: too narrow for some
        idx_t = _t(idx.astype(np.int64), dev)
        bl = [int(min(lens[i], T)) if i < S else 0 for i in idx]
        for want_label, want_T in ((True, True), (False, True), (False, False)):
            padded = cache.batch(idx_t, T, want_label=want_label)
            if not want_T:
                padded = tree.PrunedTrees(padded.B, padded.T, padded.cap, padded.row_ptr, padded.col_idx, padded.label, None, None, padded.ell, None,
                                          padded.pool_mask, padded.status)
            two = padded.pack(bl)
            one = cache.batch_packed(idx_t, T, want_label=want_label, want_transpose=want_T, n_rows=(None if trial & 1 else max(sum(bl), 1)))
            torch.cuda.synchronize()
            _same_packed(one, two, (shape, trial, want_label, want_T))
            assert torch.equal(one.padded.status, padded.status), (shape, trial)
            assert torch.equal(one.padded.pool_mask, padded.pool_mask), (shape, trial)
    whole = _t(np.arange(S, dtype=np.int64), dev)
    ok_rows = int(cache.batch_packed(whole, Ts).cu_seqlens[-1])
    if ok_rows > 3:
        small = cache.batch_packed(whole, Ts, n_rows=ok_rows - 2)
        torch.cuda.synchronize()
        assert int(small.status[0]) == -8                               # GCNPT_E_CAPACITY: reported, not overrun


@pytest.mark.parametrize("width,dtype", [(360, torch.float32), (300, torch.bfloat16), (200, torch.bfloat16), (7, torch.float32)])
def test_pack_unpack_rows_roundtrip(api, dev, width, dtype):
    gcn, tree = api
    tb = synthetic.random_tree_batch(3, 9, 40, "tacred")
    tr = tree.prune_to_csr(*(_t(tb[k], dev) for k in ("head", "subj_pos", "obj_pos", "deprel")), 1, masks=_t(tb["masks"], dev), want_label=False)
    pk = tr.pack(tb["lens"].tolist()).check()
    x = torch.randn((9, 40, width), device=dev).to(dtype).requires_grad_()
    xp = pk.pack_rows(x)
    assert tuple(xp.shape) == (int(tb["lens"].sum()), width)
    keep = ~_t(tb["masks"], dev)
    assert torch.equal(xp, x[keep])
    back = pk.unpack_rows(xp)
    assert torch.equal(back[keep], x[keep]) and (back[~keep] == 0).all()
    back.float().sum().backward()
    assert torch.equal(x.grad[keep], torch.ones_like(x.grad[keep])) and (x.grad[~keep] == 0).all()


def _stack(gcn, x, Ws, bs, trees, compute, drop=None, seeds=None):
    xt = x.clone().requires_grad_()
    Wt = [w.clone().requires_grad_() for w in Ws]
    bt = [b.clone().requires_grad_() for b in bs]
    h = gcn.gcn_layers(xt, Wt, bt, trees, drop, seeds, compute, torch.float32)
    return xt, Wt, bt, h


@pytest.mark.parametrize("shape", ["c2", "c5"])
@pytest.mark.parametrize("compute", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_packed_layers_equal_padded_rows(api, dev, shape, compute):
    """Same kernels, T = 0: a real token's row does not depend on the layout (values, dropout mask and all)."""
    gcn, tree = api
    B, T, Din, H, K = (50, 100, 360, 200, 1) if shape == "c2" else (12, 300, 600, 300, 2)
    tb = synthetic.random_tree_batch(41, B, T, "tacred")
    tr = tree.prune_to_csr(*(_t(tb[k], dev) for k in ("head", "subj_pos", "obj_pos", "deprel")), K, masks=_t(tb["masks"], dev), want_label=False)
    tr.check(expect_maxlen=T)
    pk = tr.pack(tb["lens"].tolist()).check()
    Ws, bs = synthetic.layer_params(42, [Din, H, H])
    Ws, bs = [_t(w, dev) for w in Ws], [_t(b, dev) for b in bs]
    x = _t(synthetic.normal(43, (B, T, Din)), dev)
    gy = _t(synthetic.normal(44, (B, T, H)), dev)
    keep = ~_t(tb["masks"], dev)
    drop, seeds = [0.5, 0.0], [1234567, 0]
    # dropout hashes the ROW NUMBER, which differs between the layouts: compare without it, then check it separately
    a = _stack(gcn, x, Ws, bs, tr, compute)
    a[3].backward(gy * keep.unsqueeze(-1))                 # the padded path: no gradient into padding slots, as pooling guarantees
    b = _stack(gcn, x[keep], Ws, bs, pk, compute)
    b[3].backward(gy[keep])
    torch.cuda.synchronize()
    assert tuple(b[3].shape) == (int(tb["lens"].sum()), H)
    assert torch.equal(a[3][keep], b[3]), "forward rows differ"
    assert torch.equal(a[0].grad[keep], b[0].grad), "dx rows differ"
    for l in range(2):                                     # float atomics reorder the row sums
        assert max_rel(b[1][l].grad.cpu().numpy(), a[1][l].grad.cpu().numpy()) <= 1e-5
        assert max_rel(b[2][l].grad.cpu().numpy(), a[2][l].grad.cpu().numpy()) <= 1e-5
    if shape == "c2":
        c = _stack(gcn, x[keep], Ws, bs, pk, compute, drop, seeds)
        c[3].backward(gy[keep])
        d = _stack(gcn, x[keep], Ws, bs, pk, compute, drop, seeds)
        assert torch.equal(c[3], d[3]) and not torch.equal(c[3], b[3])


def test_packed_layers_vs_oracle_fp32(api, dev):
    """fp32 mode on packed rows against the CPU restatement of model/gcn.py:260-271, 390-393 and its autograd."""
    from oracle import gcn_ref, prune_ref
    gcn, tree = api
    B, T, Din, H, K = 20, 64, 96, 80, 1
    tb = synthetic.random_tree_batch(51, B, T, "tacred")
    tr = tree.prune_to_csr(*(_t(tb[k], dev) for k in ("head", "subj_pos", "obj_pos", "deprel")), K, masks=_t(tb["masks"], dev), want_label=False)
    pk = tr.pack(tb["lens"].tolist()).check()
    Ws, bs = synthetic.layer_params(52, [Din, H, H])
    x, gy = synthetic.normal(53, (B, T, Din)), synthetic.normal(54, (B, T, H))
    keep = ~tb["masks"]
    gy = gy * keep[..., None]
    adj = prune_ref.batch_adj(tb["head"], tb["subj_pos"], tb["obj_pos"], tb["deprel"], tb["lens"], K)["adj"]
    href, _ = gcn_ref.gcn_forward(adj, x, Ws, bs)
    dx, dWs, dbs = gcn_ref.gcn_backward(adj, x, Ws, bs, gy)
    xt, Wt, bt, h = _stack(gcn, _t(x[keep], dev), [_t(w, dev) for w in Ws], [_t(b, dev) for b in bs], pk, torch.float32)
    h.backward(_t(gy[keep], dev))
    assert max_rel(h.detach().cpu().numpy(), href[keep]) <= FWD_RTOL
    assert max_rel(xt.grad.cpu().numpy(), dx[keep]) <= GRAD_RTOL
    # the reference's padding rows also feed dW / db -- with zero upstream gradient here, so the packed sums are the reference's
    for l in range(2):
        assert max_rel(Wt[l].grad.cpu().numpy(), dWs[l]) <= GRAD_RTOL and max_rel(bt[l].grad.cpu().numpy(), dbs[l]) <= GRAD_RTOL


@pytest.mark.parametrize("compute", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_c5_shard_packed_vs_oracle(api, dev, compute):
    """BASELINE.json configs[4]'s real per-GPU workload -- one rank's 16 sentences of the 8-way split, T = 300, 600 -> 300 -> 300,
    prune_k 2, TACRED-shaped lengths, token-packed -- straight against the oracle (this is the shape where the
    column-split form of the layer kernel is the product path by itself, and it was only covered transitively).  fp32: 1e-5 / 1e-4;
    bf16: forward 2e-2 against the fp32 oracle, gradients 2e-2 against the oracle differentiated through the device's own activations
    (the tolerances at the top of tests/test_gpu_parity.py), and the launches must have taken the split form."""
    import ctypes
    from gcn_over_pruned_trees_amd import _lib
    from oracle import gcn_ref, prune_ref
    gcn, tree = api
    B, T, Din, H, K = 16, 300, 600, 300, 2
    tb = synthetic.random_tree_batch(61, B, T, "tacred")
    tr = tree.prune_to_csr(*(_t(tb[k], dev) for k in ("head", "subj_pos", "obj_pos", "deprel")), K, masks=_t(tb["masks"], dev), want_label=False)
    pk = tr.pack(tb["lens"].tolist()).check()
    Ws, bs = synthetic.layer_params(62, [Din, H, H])
    x, gy = synthetic.normal(63, (B, T, Din)), synthetic.normal(64, (B, T, H))
    keep = ~tb["masks"]
    gy = gy * keep[..., None]
    adj = prune_ref.batch_adj(tb["head"], tb["subj_pos"], tb["obj_pos"], tb["deprel"], tb["lens"], K)["adj"]
    bf16 = compute == torch.bfloat16
    xd = _t(x[keep], dev).to(compute)
    xin = x if not bf16 else _t(x, dev).to(torch.bfloat16).float().cpu().numpy()       # what the device sees
    xt = xd.clone().requires_grad_()
    Wt = [_t(w, dev).requires_grad_() for w in Ws]
    bt = [_t(b, dev).requires_grad_() for b in bs]
    h, acts = gcn.gcn_layers_with_acts(xt, Wt, bt, pk, compute_dtype=compute, out_dtype=torch.float32)
    v = [ctypes.c_int(0) for _ in range(4)]
    _lib.check(_lib.lib().gcnpt_last_launch(*[ctypes.byref(q) for q in v]))
    tiles = (pk.N + 31) // 32
    if bf16:
        assert v[0].value >= 2 * tiles, "the last forward launch (grid %d for %d row tiles) did not take the column-split form" % (v[0].value, tiles)
    h.backward(_t(gy[keep], dev))
    torch.cuda.synchronize()
    href, _ = gcn_ref.gcn_forward(adj, xin, Ws, bs)
    assert max_rel(h.detach().cpu().numpy(), href[keep]) <= (2e-2 if bf16 else FWD_RTOL)
    dev_acts = None
    if bf16:                                            # relu' is a step function: differentiate the oracle through the device's activations
        dev_acts = []
        for a in acts:
            full = np.zeros((B, T, a.shape[-1]), np.float32)
            full[keep] = a.float().cpu().numpy()
            dev_acts.append(full)
    dx, dWs, dbs = gcn_ref.gcn_backward(adj, xin, Ws, bs, gy, acts=dev_acts)
    gtol = 2e-2 if bf16 else GRAD_RTOL
    assert max_rel(xt.grad.float().cpu().numpy(), dx[keep]) <= gtol
    for l in range(2):
        assert max_rel(Wt[l].grad.cpu().numpy(), dWs[l]) <= gtol and max_rel(bt[l].grad.cpu().numpy(), dbs[l]) <= gtol


@pytest.mark.parametrize("fixture", ["e2e_gcn.npz", "e2e_cgcn.npz", "e2e_avgpool.npz"])
def test_classifier_packed_matches_golden_logits(api, dev, fixture):
    """opt['gcn_packed']: the layer loop on packed rows, padding only at the module boundary -- the reference's recorded logits."""
    gcn, tree = api
    e = load_golden(fixture)
    opt = json.loads(str(e["opt"]))
    opt["cuda"] = True
    outs = {}
    for packed in (False, True):
        model = gcn.GCNClassifier(dict(opt, gcn_packed=packed))
        model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in e.items() if k.startswith("sd:")}, strict=True)
        model.to(dev).eval()
        keys = ("words", "masks", "pos", "ner", "deprel", "head", "subj_pos", "obj_pos") if opt["dataset"] == "tacred" else \
            ("words", "masks", "pos", "deprel", "head", "subj_pos", "obj_pos")
        inputs = tuple(_t(e[k], dev) for k in keys)
        with torch.no_grad():
            logits, pooled = model(inputs)
        outs[packed] = (logits.cpu().numpy(), pooled.cpu().numpy())
    assert max_rel(outs[True][0], e["logits"]) <= 1e-4
    assert max_rel(outs[True][0], outs[False][0]) <= 1e-5 and max_rel(outs[True][1], outs[False][1]) <= 1e-5
    # the same batch with its packed trees assembled from a dataset pruned once (TreeCache.batch_packed): the model takes them as `trees=`
    head, subj, obj, dep, masks = (_t(e[k], dev) for k in ("head", "subj_pos", "obj_pos", "deprel", "masks"))
    lens = (~masks.bool()).sum(1).to(torch.int32)
    cache = tree.TreeCache.build(head, subj, obj, dep, opt["prune_k"], lens=lens, want_label=False)
    ids = torch.arange(head.shape[0], device=dev)
    with torch.no_grad():
        logits_c, pooled_c = model(inputs, trees=cache.batch_packed(ids, head.shape[1], n_rows=int(lens.sum())))
    assert np.array_equal(logits_c.cpu().numpy(), outs[True][0]) and np.array_equal(pooled_c.cpu().numpy(), outs[True][1])
