#!/usr/bin/env python3
"""Times the full_deprel traversal contraction: hand-written gcnpt_bilinear_fwd (+ its weight pack) against the library path
(materialised outer product + one hipBLASLt GEMM, fp32 and bf16).  Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_over_pruned_trees_amd import _lib  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
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
    lib, out = _lib.lib(), {}
    for M, D, Tin, H in ((1200, 50, 200, 200), (1200, 200, 200, 200), (5000, 50, 200, 200)):
        x = torch.randn(M, Tin, device=dev)
        e = torch.randn(M, D, device=dev)
        W = torch.randn(D * H, Tin, device=dev) / 30
        Wk = W.reshape(D * Tin, H)
        img = torch.empty((lib.gcnpt_bilinear_packed_bytes(D, Tin, H),), dtype=torch.uint8, device=dev)
        xb = torch.zeros((M, (Tin + 31) // 32 * 32), dtype=torch.bfloat16, device=dev)
        xb[:, :Tin] = x
        y = torch.empty((lib.gcnpt_bilinear_planes(M, D, Tin, H), M, H), device=dev)
        st = _lib.stream()
        pack = lambda: _lib.check(lib.gcnpt_bilinear_pack(st, _lib.ptr(W), D, Tin, H, _lib.ptr(img)))  # noqa: E731
        fwd = lambda: _lib.check(lib.gcnpt_bilinear_fwd(st, _lib.ptr(xb), _lib.ptr(e), _lib.ptr(img), M, D, Tin, H, _lib.ptr(y)))  # noqa: E731
        lib32 = lambda: torch.mm((e.unsqueeze(2) * x.unsqueeze(1)).reshape(M, D * Tin), Wk)  # noqa: E731
        Wk16, e16, x16 = Wk.bfloat16(), e.bfloat16(), x.bfloat16()
        lib16 = lambda: torch.mm((e16.unsqueeze(2) * x16.unsqueeze(1)).reshape(M, D * Tin), Wk16)  # noqa: E731
        flops = 2.0 * M * D * Tin * H
        t_pack, t_fwd, t32, t16 = timed(pack), timed(fwd), timed(lib32), timed(lib16)
        t_sum = timed(lambda: y.sum(0))
        out["M%d_D%d" % (M, D)] = dict(pack_us=round(t_pack, 1), kernel_us=round(t_fwd, 1), kernel_TFLOPs=round(flops / t_fwd / 1e6, 1), planes=int(y.shape[0]), plane_sum_us=round(t_sum, 1),
                                       library_fp32_us=round(t32, 1), library_bf16_us=round(t16, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
