#!/usr/bin/env python3
"""
Developer aid (GPU box): the sentence-slice form of the layer stack (GCNPT_OPT_DATAFLOW = 1) against the row-tile form (0) and the
CPU oracle on a few shapes, forward and backward, then the in-step launch durations of both forms at the bench shape.
Not part of the product; tests/test_gpu_sent.py holds the parity tests proper.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from gcn_over_pruned_trees_amd import _lib  # noqa: E402
from gcn_over_pruned_trees_amd.model import gcn, tree  # noqa: E402
from gcn_over_pruned_trees_amd.utils import synthetic  # noqa: E402
from oracle import gcn_ref, prune_ref  # noqa: E402

OPT_DATAFLOW = 4


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def run_case(B, T, dims, K, compute, lengths="tacred", drop=0.0, backward=True, seed=7):
    dev = torch.device("cuda:0")
    tb = synthetic.random_tree_batch(seed, B, T, lengths)
    Ws, bs = synthetic.layer_params(seed + 1, dims)
    x, gy = synthetic.normal(seed + 2, (B, T, dims[0])), synthetic.normal(seed + 3, (B, T, dims[-1]))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    trees = tree.prune_to_csr(t(tb["head"]), t(tb["subj_pos"]), t(tb["obj_pos"]), t(tb["deprel"]), K, masks=t(tb["masks"]))
    adj = prune_ref.batch_adj(tb["head"], tb["subj_pos"], tb["obj_pos"], tb["deprel"], tb["lens"], K)["adj"]
    res = {}
    for form in (0, 1):
        _lib.set_option(OPT_DATAFLOW, form)
        xt = t(x).to(torch.bfloat16 if compute == torch.bfloat16 else torch.float32).requires_grad_()
        Wt = [t(w).requires_grad_() for w in Ws]
        bt = [t(b).requires_grad_() for b in bs]
        L = len(Ws)
        drops = [drop] * (L - 1) + [0.0]
        with torch.set_grad_enabled(backward):
            h, acts = gcn.gcn_layers_with_acts(xt, Wt, bt, trees, drop_p=drops, seeds=[11 + l for l in range(L)], compute_dtype=compute,
                                               out_dtype=torch.float32)
        r = {"h": h.detach().float().cpu().numpy(), "acts": [a.float().cpu().numpy() for a in acts]}
        if backward:
            h.backward(t(gy))
            r.update(dx=xt.grad.float().cpu().numpy(), dW=[w.grad.cpu().numpy() for w in Wt], db=[b.grad.cpu().numpy() for b in bt])
        torch.cuda.synchronize()
        res[form] = r
    _lib.set_option(OPT_DATAFLOW, -1)
    out = ["B=%d T=%d dims=%s K=%d %s drop=%.1f" % (B, T, dims, K, str(compute).split(".")[-1], drop)]
    out.append("fwd sent-vs-rowtile %.2e" % rel(res[1]["h"], res[0]["h"]))
    if drop == 0.0:
        xin = x if compute == torch.float32 else t(x).to(torch.bfloat16).float().cpu().numpy()
        href, _ = gcn_ref.gcn_forward(adj, xin, Ws, bs)
        out.append("vs oracle: rowtile %.2e sent %.2e" % (rel(res[0]["h"], href), rel(res[1]["h"], href)))
    if backward:
        for form in (0, 1):
            r = res[form]
            if drop == 0.0:
                xin = x if compute == torch.float32 else t(x).to(torch.bfloat16).float().cpu().numpy()
                dx, dWs, dbs = gcn_ref.gcn_backward(adj, xin, Ws, bs, gy, acts=r["acts"])
                out.append("form %d bwd vs oracle(dev acts): dx %.2e dW %s db %s" % (
                    form, rel(r["dx"], dx), ["%.2e" % rel(a, b) for a, b in zip(r["dW"], dWs)], ["%.2e" % rel(a, b) for a, b in zip(r["db"], dbs)]))
    print("  ".join(out), flush=True)


def main():
    backward = "--fwd-only" not in sys.argv
    for compute in (torch.float32, torch.bfloat16):
        run_case(4, 20, [200, 200, 200], 1, compute, backward=backward)
        run_case(50, 100, [360, 200, 200], 1, compute, backward=backward)
        run_case(50, 100, [360, 200, 200], 1, compute, "full", backward=backward)
        run_case(50, 100, [400, 200, 200], 1, compute, backward=backward)
        run_case(7, 37, [52, 24, 40], 2, compute, backward=backward)
        run_case(3, 128, [64, 300, 300], 2, compute, backward=backward)
        run_case(16, 300, [600, 300, 300], 2, compute, backward=backward)
    run_case(50, 100, [360, 200, 200], 1, torch.bfloat16, drop=0.5, backward=backward)


if __name__ == "__main__":
    main()
