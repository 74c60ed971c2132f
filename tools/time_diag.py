#!/usr/bin/env python3
"""Times gcnpt_diag_layer_fwd / bwd (N2) at the BASELINE config-2 shape with HIP events; prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_over_pruned_trees_amd.model import gcn, tree  # noqa: E402
from gcn_over_pruned_trees_amd.utils import synthetic  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, T, H, K = 50, 100, 200, 1
    tb = synthetic.random_tree_batch(1, B, T, "tacred")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    masks = np.arange(T)[None, :] >= tb["lens"][:, None]
    trees = tree.prune_to_csr(t(tb["head"]), t(tb["subj_pos"]), t(tb["obj_pos"]), t(tb["deprel"]), K, masks=t(masks), want_label=True).check()
    deprel = t(tb["deprel"])
    out = {}
    for dtype in (torch.float32, torch.bfloat16):
        E = torch.rand(85, H, device=dev).requires_grad_()
        h = torch.randn(B, T, H, device=dev).to(dtype).requires_grad_()
        gy = torch.randn(B, T, H, device=dev).to(dtype)
        y = gcn.diag_layer(h, E, deprel, trees)
        y.backward(gy)
        torch.cuda.synchronize()
        n = 200
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        for _ in range(n):
            y = gcn._DiagLayerFn.apply(h.detach(), E.detach(), deprel, trees, 0.0, 0)
        ev[1].record()
        torch.cuda.synchronize()
        # whole fwd+bwd through autograd (includes the zero-fill of dE and torch's launch overheads)
        ev[1].record()
        for _ in range(n):
            y = gcn.diag_layer(h, E, deprel, trees)
            y.backward(gy)
        ev[2].record()
        torch.cuda.synchronize()
        es = 4 if dtype == torch.float32 else 2
        fwd_us = ev[0].elapsed_time(ev[1]) * 1e3 / n
        out[str(dtype)] = dict(fwd_us=round(fwd_us, 2), fwd_bwd_us=round(ev[1].elapsed_time(ev[2]) * 1e3 / n, 2),
                               fwd_alg_GBps=round(2 * B * T * H * es / fwd_us / 1e3, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
