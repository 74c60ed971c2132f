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
        pack = lambda: _lib.check(lib.gcnpt_bilinear_pack(st, _lib.ptr(W), D, Tin, H, _lib.ptr(img), 0))  # noqa: E731
        fwd = lambda: _lib.check(lib.gcnpt_bilinear_fwd(st, _lib.ptr(xb), _lib.ptr(e), _lib.ptr(img), M, D, Tin, H, _lib.ptr(y)))  # noqa: E731
        lib32 = lambda: torch.mm((e.unsqueeze(2) * x.unsqueeze(1)).reshape(M, D * Tin), Wk)  # noqa: E731
        Wk16, e16, x16 = Wk.bfloat16(), e.bfloat16(), x.bfloat16()
        lib16 = lambda: torch.mm((e16.unsqueeze(2) * x16.unsqueeze(1)).reshape(M, D * Tin), Wk16)  # noqa: E731
        # backward pieces: dx (same kernel, transposed image), de (dot mode), dW (token contraction), their packs
        gy = torch.randn(M, H, device=dev)
        imgT = torch.empty((lib.gcnpt_bilinear_packed_bytes(D, H, Tin),), dtype=torch.uint8, device=dev)
        gyb = torch.zeros((M, (H + 31) // 32 * 32), dtype=torch.bfloat16, device=dev)
        gyb[:, :H] = gy
        dxp = torch.empty((lib.gcnpt_bilinear_planes(M, D, H, Tin), M, Tin), device=dev)
        dep = torch.empty((lib.gcnpt_bilinear_de_planes(M, D, Tin, H), M, D), device=dev)
        xI = torch.empty((lib.gcnpt_rows_image_bytes(M, Tin),), dtype=torch.uint8, device=dev)
        gI = torch.empty((lib.gcnpt_rows_image_bytes(M, H),), dtype=torch.uint8, device=dev)
        eT = torch.zeros((D, (M + 31) // 32 * 32), device=dev)
        eT[:, :M] = e.t()
        dW = torch.empty_like(W)
        packT = lambda: _lib.check(lib.gcnpt_bilinear_pack(st, _lib.ptr(W), D, Tin, H, _lib.ptr(imgT), 1))  # noqa: E731
        k_dx = lambda: _lib.check(lib.gcnpt_bilinear_fwd(st, _lib.ptr(gyb), _lib.ptr(e), _lib.ptr(imgT), M, D, H, Tin, _lib.ptr(dxp)))  # noqa: E731
        k_de = lambda: _lib.check(lib.gcnpt_bilinear_bwd_e(st, _lib.ptr(xb), _lib.ptr(gy), _lib.ptr(img), M, D, Tin, H, _lib.ptr(dep)))  # noqa: E731
        rows = lambda: (_lib.check(lib.gcnpt_rows_pack(st, _lib.ptr(x), M, Tin, _lib.ptr(xI))),  # noqa: E731
                        _lib.check(lib.gcnpt_rows_pack(st, _lib.ptr(gy), M, H, _lib.ptr(gI))))
        k_dw = lambda: _lib.check(lib.gcnpt_bilinear_bwd_w(st, _lib.ptr(xI), _lib.ptr(gI), _lib.ptr(eT), M, D, Tin, H, _lib.ptr(dW)))  # noqa: E731

        def lib_bwd():
            G = torch.mm(gy, Wk.t()).view(M, D, Tin)
            dx_ = (G * e.unsqueeze(2)).sum(1)
            de_ = (G * x.unsqueeze(1)).sum(2)
            dW_ = torch.mm((e.unsqueeze(2) * x.unsqueeze(1)).reshape(M, D * Tin).t(), gy)
            return dx_, de_, dW_
        pack(); packT(); rows()
        t_packT, t_dx, t_de, t_rows, t_dw, t_libb = timed(packT), timed(k_dx), timed(k_de), timed(rows), timed(k_dw), timed(lib_bwd, 10)
        flops = 2.0 * M * D * Tin * H
        t_pack, t_fwd, t32, t16 = timed(pack), timed(fwd), timed(lib32), timed(lib16)
        t_sum = timed(lambda: y.sum(0))
        out["M%d_D%d" % (M, D)] = dict(pack_us=round(t_pack, 1), kernel_us=round(t_fwd, 1), kernel_TFLOPs=round(flops / t_fwd / 1e6, 1), planes=int(y.shape[0]), plane_sum_us=round(t_sum, 1),
                                       library_fp32_us=round(t32, 1), library_bf16_us=round(t16, 1),
                                       bwd=dict(packT_us=round(t_packT, 1), dx_us=round(t_dx, 1), de_us=round(t_de, 1), rows_pack_us=round(t_rows, 1),
                                                dW_us=round(t_dw, 1), dW_TFLOPs=round(flops / t_dw / 1e6, 1), library_fp32_us=round(t_libb, 1)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
