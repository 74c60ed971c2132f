#!/usr/bin/env python3
"""Runs the auxiliary kernels (pool3, diagonal_deprel layer, tree gather) at the bench shape a few hundred times, for
`rocprofv3 --kernel-trace --stats` (developer tool: their durations are not part of the headline step)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_over_pruned_trees_amd.model import gcn, tree  # noqa: E402
from gcn_over_pruned_trees_amd.utils import synthetic  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, T, H = 50, 100, 200
    tb = synthetic.random_tree_batch(1, B, T, "tacred")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    head, subj, obj, deprel, masks = (t(tb[k]) for k in ("head", "subj_pos", "obj_pos", "deprel", "masks"))
    trees = tree.prune_to_csr(head, subj, obj, deprel, 1, masks=masks, want_label=True).check()
    cache = tree.TreeCache.build(head, subj, obj, deprel, 1, masks=masks)
    idx = torch.arange(B, device=dev)
    for dtype in (torch.float32, torch.bfloat16):
        E = torch.rand(85, H, device=dev).requires_grad_()
        h = torch.randn(B, T, H, device=dev).to(dtype).requires_grad_()
        for kind in ("max", "avg"):
            for _ in range(100):
                p = gcn.pool3(h, trees.pool_mask, subj, obj, type=kind)
                p.sum().backward()
        for _ in range(100):
            y = gcn.diag_layer(h, E, deprel, trees, 0.5, 3)
            y.backward(torch.ones_like(y))
    for _ in range(100):
        cache.batch(idx, T)
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
