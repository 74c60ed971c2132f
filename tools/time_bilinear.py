#!/usr/bin/env python3
"""Times the full_deprel traversal contraction (reference model/gcn.py:400-415) in both precisions: the hand-written kernels
(gcnpt_bilinear_fwd / _bwd_e / _bwd_w + packs; exact fp32 MFMA and bf16 MFMA) against the library path they replace
(materialised outer product e (x) x + one hipBLASLt GEMM, forward; autograd of it, backward).  Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_over_pruned_trees_amd import _lib  # noqa: E402
from gcn_over_pruned_trees_amd.model import gcn  # noqa: E402


def timed(fn, n=30):
    for _ in range(3):
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
        x = torch.randn(M, Tin, device=dev, requires_grad=True)
        e = torch.randn(M, D, device=dev, requires_grad=True)
        W = (torch.randn(D * H, Tin, device=dev) / 30).requires_grad_()
        b = torch.zeros(D * H, device=dev, requires_grad=True)
        gy = torch.randn(M, H, device=dev)
        rec = {}
        for name, cd in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            code = _lib.dtype_code(cd)
            k = 32 if cd == torch.bfloat16 else 16
            img = torch.empty((lib.gcnpt_bilinear_packed_bytes(D, Tin, H, code),), dtype=torch.uint8, device=dev)
            xb = torch.zeros((M, (Tin + k - 1) // k * k), dtype=cd, device=dev)
            xb[:, :Tin] = x.detach()
            y = torch.empty((lib.gcnpt_bilinear_planes(M, D, Tin, H, code), M, H), device=dev)
            st = _lib.stream()
            Wd, ed = W.detach(), e.detach()
            rec[name + "_pack_us"] = timed(lambda: _lib.check(lib.gcnpt_bilinear_pack(st, _lib.ptr(Wd), D, Tin, H, _lib.ptr(img), 0, code)))
            rec[name + "_fwd_kernel_us"] = timed(lambda: _lib.check(lib.gcnpt_bilinear_fwd(st, _lib.ptr(xb), _lib.ptr(ed), _lib.ptr(img), M, D, Tin, H, _lib.ptr(y), code)))

            def op_fwd_bwd():
                for t in (x, e, W, b):
                    t.grad = None
                gcn.bilinear_traverse(x, e, W, b, cd).backward(gy)
            rec[name + "_op_fwd_bwd_us"] = timed(op_fwd_bwd, 10)

            def lib_fwd():
                xx, ee, Wk = x.detach().to(cd), e.detach().to(cd), W.detach().reshape(D * Tin, H).to(cd)
                return torch.mm((ee.unsqueeze(2) * xx.unsqueeze(1)).reshape(M, D * Tin), Wk)
            rec[name + "_library_fwd_us"] = timed(lib_fwd, 10)
        # the library path's forward + backward in fp32 (what model/gcn.py ran in its default precision before round 3)
        def lib_fwd_bwd():
            for t in (x, e, W, b):
                t.grad = None
            yy = torch.mm((e.unsqueeze(2) * x.unsqueeze(1)).reshape(M, D * Tin), W.reshape(D * Tin, H)) + torch.mm(e, b.reshape(D, H))
            yy.backward(gy)
        rec["fp32_library_fwd_bwd_us"] = timed(lib_fwd_bwd, 10)
        flops = 2.0 * M * D * Tin * H
        rec["fp32_fwd_kernel_tflops"] = flops / rec["fp32_fwd_kernel_us"] / 1e6
        rec["bf16_fwd_kernel_tflops"] = flops / rec["bf16_fwd_kernel_us"] / 1e6
        out["M%d_D%d_T%d_H%d" % (M, D, Tin, H)] = {k: round(v, 2) for k, v in rec.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
