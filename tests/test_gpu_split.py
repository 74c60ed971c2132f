"""
The experimental two-launch form of a GCN layer (csrc/rowsplit_kernels.hip: a gather launch that writes the aggregating rows + the weight
gradient's fragment image, and a matrix launch with 128/160 rows per workgroup; opt-in with GCNPT_ROWSPLIT=1 from 16 384 token rows on,
when the caller passes the workspace of gcnpt_layers_workspace_bytes; DESIGN.md section 5 has the measurements that keep it off by
default) against the row-tile kernels: the same arithmetic in the same order
(model/gcn.py:269-271, 390-393), so forward rows are bit-identical; with the pooling hand-over (both sweeps start from the same dZ
rows) the input gradient is bit-identical too, otherwise the top layer's dZ is rounded to bf16 rows once instead of being derived in
fp32 per neighbour.  The weight gradient sums with float atomics: compared to 1e-3 of its largest element.
"""
import os

import numpy as np
import pytest
import torch

from gcn_over_pruned_trees_amd.utils import synthetic

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rel(u, v):
    return float((u.float() - v.float()).abs().max() / v.float().abs().max().clamp_min(1e-30))


def _run(gcn, tr, x, Ws, bs, gy, split, pool=None, drop=(0.5, 0.0), out_dtype=torch.float32):
    if split:
        os.environ["GCNPT_ROWSPLIT"] = "1"
    else:
        os.environ.pop("GCNPT_ROWSPLIT", None)
    try:
        xt = x.clone().requires_grad_()
        Wt = [w.clone().requires_grad_() for w in Ws]
        bt = [b.clone().requires_grad_() for b in bs]
        h = gcn.gcn_layers(xt, Wt, bt, tr, list(drop), [99, 7], torch.bfloat16, out_dtype, pool=pool)
        h.backward(gy.to(h.dtype))
        torch.cuda.synchronize()
        return h.detach(), xt.grad, [w.grad for w in Wt], [b.grad for b in bt]
    finally:
        os.environ.pop("GCNPT_ROWSPLIT", None)


def _trees(tree, tb, K, dev):
    tr = tree.prune_to_csr(*(_t(tb[k], dev) for k in ("head", "subj_pos", "obj_pos", "deprel")), K, masks=_t(tb["masks"], dev), want_label=False)
    tr.check(expect_maxlen=tb["head"].shape[1])
    return tr


CASES = {
    #            B,   T,  Din, H,   K, lengths, x dtype,        out dtype
    "c5_widths": (64, 300, 600, 300, 2, "tacred", torch.bfloat16, torch.bfloat16),     # 16-byte rows in, 8-byte rows (300) between the layers
    "c2_widths": (167, 100, 360, 200, 1, "full", torch.bfloat16, torch.float32),        # 16 700 rows: not a multiple of 32 or 160
    "fp32_rows": (70, 240, 96, 104, 3, "tacred", torch.float32, torch.float32),         # the module's dtypes: fp32 in, bf16 between, fp32 out
    "whole_tree": (66, 250, 64, 72, 200, "tacred", torch.bfloat16, torch.bfloat16),     # K = 200 keeps the whole tree: rows with > 7 entries
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_split_layers_match_row_tile_kernels(case):
    from gcn_over_pruned_trees_amd import _lib
    from gcn_over_pruned_trees_amd.model import gcn, tree
    dev = torch.device("cuda:0")
    B, T, Din, H, K, lengths, xdt, odt = CASES[case]
    tb = synthetic.random_tree_batch(3, B, T, lengths)
    tr = _trees(tree, tb, K, dev)
    os.environ["GCNPT_ROWSPLIT"] = "1"
    try:
        assert _lib.lib().gcnpt_layers_workspace_bytes(1, B, T, (_lib.ctypes.c_int * 1)(Din), (_lib.ctypes.c_int * 1)(H), _lib.BF16) > 0
    finally:
        os.environ.pop("GCNPT_ROWSPLIT", None)
    Ws, bs = synthetic.layer_params(4, [Din, H, H])
    Ws, bs = [_t(w, dev) for w in Ws], [_t(b, dev) for b in bs]
    x = _t(synthetic.normal(5, (B, T, Din)), dev).to(xdt)
    gy = _t(synthetic.normal(6, (B, T, H)), dev)
    a = _run(gcn, tr, x, Ws, bs, gy, split=False, out_dtype=odt)
    b = _run(gcn, tr, x, Ws, bs, gy, split=True, out_dtype=odt)
    assert torch.isfinite(b[0].float()).all() and float(b[0].float().abs().max()) > 0
    assert torch.equal(a[0], b[0]), "forward rows differ"
    assert _rel(b[1], a[1]) <= 2e-2                           # bf16 rows of dZ instead of fp32 per-neighbour values in the top layer
    for l in range(2):
        assert _rel(b[2][l], a[2][l]) <= 2e-2 and _rel(b[3][l], a[3][l]) <= 2e-2
    # from the same dZ rows (the pooling's hand-over) the sweeps are the same arithmetic: bit-identical input gradient
    subj, obj = _t(tb["subj_pos"], dev), _t(tb["obj_pos"], dev)
    gp = _t(synthetic.normal(8, (B, 3 * H)), dev)
    a = _run(gcn, tr, x, Ws, bs, gp, split=False, pool=(subj, obj, "max"), out_dtype=odt)
    b = _run(gcn, tr, x, Ws, bs, gp, split=True, pool=(subj, obj, "max"), out_dtype=odt)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), "pooled rows / input gradient differ"
    for l in range(2):
        assert _rel(b[2][l], a[2][l]) <= 1e-3 and _rel(b[3][l], a[3][l]) <= 1e-3


def test_split_layers_on_token_packed_rows():
    """T = 0 (token-packed rows, absolute columns): same kernels, same values as the padded batch's real tokens."""
    from gcn_over_pruned_trees_amd.model import gcn, tree
    dev = torch.device("cuda:0")
    B, T, Din, H, K = 640, 120, 128, 136, 2
    tb = synthetic.random_tree_batch(11, B, T, "tacred")
    tr = _trees(tree, tb, K, dev)
    pk = tr.pack(tb["lens"].tolist())
    if pk.N < 16384:
        pytest.skip("batch too small for the split path")
    keep = ~_t(tb["masks"], dev)
    Ws, bs = synthetic.layer_params(4, [Din, H, H])
    Ws, bs = [_t(w, dev) for w in Ws], [_t(b, dev) for b in bs]
    x = _t(synthetic.normal(5, (B, T, Din)), dev).to(torch.bfloat16)[keep].contiguous()
    gy = _t(synthetic.normal(6, (B, T, H)), dev)[keep].contiguous()
    a = _run(gcn, pk, x, Ws, bs, gy, split=False, drop=(0.0, 0.0))
    b = _run(gcn, pk, x, Ws, bs, gy, split=True, drop=(0.0, 0.0))
    assert torch.equal(a[0], b[0])
    assert _rel(b[1], a[1]) <= 2e-2
    for l in range(2):
        assert _rel(b[2][l], a[2][l]) <= 2e-2
