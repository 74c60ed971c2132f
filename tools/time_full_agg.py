#!/usr/bin/env python3
"""
Developer timing: one full_deprel layer's aggregation epilogue (gcn.py:308-311, 331, 340-344, 362, 385, 390-393) as the fused kernel
gcnpt_full_agg_fwd/bwd against the torch ops it replaced in round 2 (searchsorted slot -> row map, two index_add_, divide, relu).
B=50, T=100, H=200, K=1, random trees; prints microseconds per forward and per forward+backward (HIP events, 200 repetitions).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gcn_over_pruned_trees_amd.model import gcn, tree  # noqa: E402
from gcn_over_pruned_trees_amd.utils import synthetic  # noqa: E402


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    dev = torch.device("cuda:0")
    B, T, H = 50, 100, 200
    tb = synthetic.random_tree_batch(7, B, T, "full")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    tr = tree.prune_to_csr(t(tb["head"]), t(tb["subj_pos"]), t(tb["obj_pos"]), t(tb["deprel"]), 1, masks=t(tb["masks"]), want_label=True).check()
    N, cap = B * T, tr.cap
    tok = torch.nonzero(~tr.pool_mask.view(-1)).squeeze(1)
    M = int(tok.numel())
    pos32 = torch.full((N,), -1, dtype=torch.int32, device=dev)
    pos32[tok] = torch.arange(M, device=dev, dtype=torch.int32)
    pos = pos32.to(torch.int64).clamp(min=0)
    yf = torch.randn((M, H), device=dev, requires_grad=True)
    yr = torch.randn((M, H), device=dev, requires_grad=True)
    st = torch.randn((N, H), device=dev, requires_grad=True)
    gy = torch.randn((N, H), device=dev)
    # the round-1 formulation
    rp = tr.row_ptr.view(B, T + 1).to(torch.int64)
    slot = torch.arange(B * cap, device=dev).view(B, cap)
    valid = (slot < rp[:, -1:]).view(-1)
    base = (torch.arange(B, device=dev) * T).view(B, 1)
    rows = (torch.searchsorted(rp[:, 1:].contiguous(), slot, right=True).clamp_(max=T - 1) + base).view(-1)
    cols = (tr.col_idx.view(B, cap).to(torch.int64).clamp(0, T - 1) + base).view(-1)
    lab = tr.label.view(-1)
    fwd_e = (valid & (lab > 0) & (lab < 42)).float().unsqueeze(1)
    rev_e = (valid & (lab > 42) & (lab < 84)).float().unsqueeze(1)
    denom = (tr.ell.view(N, 8)[:, 0] + 1).float().unsqueeze(1)

    def torch_fwd():
        agg = torch.zeros((N, H), device=dev)
        agg = agg.index_add(0, rows, yf[pos[cols]] * fwd_e).index_add(0, rows, yr[pos[cols]] * rev_e)
        return torch.relu((agg + st) / denom)

    def kern_fwd():
        return gcn._FullAggFn.apply(yf, yr, st, tr, pos32, None, None, M, 0.0, 0, None)
    a, b = torch_fwd(), kern_fwd()
    print("max |kernel - torch ops| = %.2e  (M = %d tokens in trees of %d)" % (float((a - b).abs().max()), M, N))
    print("forward          : torch ops %7.1f us   fused kernel %7.1f us" % (timeit(torch_fwd), timeit(kern_fwd)))
    print("forward+backward : torch ops %7.1f us   fused kernel %7.1f us" % (timeit(lambda: torch_fwd().backward(gy)), timeit(lambda: kern_fwd().backward(gy))))


if __name__ == "__main__":
    main()
