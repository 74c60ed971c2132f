"""
The experimental streaming form of the layer kernels (csrc/rowstream_kernels.hip: persistent workgroups, loader waves gather the
next 64-row tile while matrix waves run the current one; opt-in with GCNPT_ROWSTREAM=1, see DESIGN.md section 5 for why it is not the
default) against the row-tile kernels: same arithmetic (model/gcn.py:269-271, 390-393), so forward rows must be bit-identical and the
backward within bf16 rounding of the gradient rows (the top layer's dZ is written out as bf16 rows there instead of being derived
in fp32 per neighbour).
"""
import os

import numpy as np
import pytest
import torch

from gcn_over_pruned_trees_amd.utils import synthetic

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _run(gcn, tr, x, Ws, bs, gy, stream):
    if stream:
        os.environ["GCNPT_ROWSTREAM"] = "1"
    else:
        os.environ.pop("GCNPT_ROWSTREAM", None)
    try:
        xt = x.clone().requires_grad_()
        Wt = [w.clone().requires_grad_() for w in Ws]
        bt = [b.clone().requires_grad_() for b in bs]
        h = gcn.gcn_layers(xt, Wt, bt, tr, [0.5, 0.0], [99, 0], torch.bfloat16, torch.float32)
        h.backward(gy)
        torch.cuda.synchronize()
        return h.detach(), xt.grad, [w.grad for w in Wt], [b.grad for b in bt]
    finally:
        os.environ.pop("GCNPT_ROWSTREAM", None)


@pytest.mark.parametrize("dims", [(600, 300), (360, 200)], ids=["c5_widths", "c2_widths"])
def test_streaming_kernels_match_row_tile_kernels(dims):
    from gcn_over_pruned_trees_amd.model import gcn, tree
    dev = torch.device("cuda:0")
    Din, H = dims
    B, T, K = 64, 300, 2                                      # 19 200 rows: above the streaming kernel's threshold
    tb = synthetic.random_tree_batch(3, B, T, "tacred")
    tr = tree.prune_to_csr(*(_t(tb[k], dev) for k in ("head", "subj_pos", "obj_pos", "deprel")), K, masks=_t(tb["masks"], dev), want_label=False)
    tr.check(expect_maxlen=T)
    Ws, bs = synthetic.layer_params(4, [Din, H, H])
    Ws, bs = [_t(w, dev) for w in Ws], [_t(b, dev) for b in bs]
    x = _t(synthetic.normal(5, (B, T, Din)), dev).to(torch.bfloat16)
    gy = _t(synthetic.normal(6, (B, T, H)), dev)
    a = _run(gcn, tr, x, Ws, bs, gy, stream=False)
    b = _run(gcn, tr, x, Ws, bs, gy, stream=True)
    assert torch.equal(a[0], b[0]), "forward rows differ"
    rel = lambda u, v: float((u.float() - v.float()).abs().max() / v.float().abs().max())  # noqa: E731
    assert rel(b[1], a[1]) <= 2e-2                            # bf16 rows of dZ instead of fp32 per-neighbour values in the top layer
    for l in range(2):
        assert rel(b[2][l], a[2][l]) <= 2e-2 and rel(b[3][l], a[3][l]) <= 2e-2
