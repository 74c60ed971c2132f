#!/usr/bin/env python3
"""
Developer diagnostic: per-phase cycle stamps of the hot kernels (needs `make -C gcn-over-pruned-trees_amd/csrc stamps`).
Runs the bench workload once per kernel with libgcnpt_stamps.so and prints, per stamp interval, the median /
max cycles over workgroups, the dispatch skew (100 MHz real time) and the first-start -> last-end span.
Not part of the product; timings of this build are NOT quoted anywhere (the stamps forbid overlaps).
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GCNPT_LIB"] = os.path.join(ROOT, "gcn-over-pruned-trees_amd", "csrc", "libgcnpt_stamps.so")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    args = bench.parse()
    dev = torch.device("cuda:0")
    stack = bench.Stack(args, dev, seed=1234)
    L = stack.L
    L.gcnpt_debug_set_stamps.argtypes = [ctypes.c_void_p]
    L.gcnpt_debug_set_stamps.restype = None
    L.gcnpt_debug_set_knob.argtypes = [ctypes.c_int]
    L.gcnpt_debug_set_knob.restype = None
    knob = int(os.environ.get("GCNPT_KNOB", "0"))
    L.gcnpt_debug_set_knob(knob)
    print("knob =", knob)
    buf = torch.zeros((4096 * 16,), dtype=torch.int64, device=dev)
    calls = [("prune", stack.prune)] + stack.calls(0)[1:]
    for _ in range(5):
        stack.step()
    torch.cuda.synchronize()
    for name, call in calls:
        L.gcnpt_debug_set_stamps(None)
        for _ in range(3):
            stack.step()
        torch.cuda.synchronize()
        buf.zero_()
        L.gcnpt_debug_set_stamps(buf.data_ptr())
        call()
        torch.cuda.synchronize()
        L.gcnpt_debug_set_stamps(None)
        s = buf.cpu().numpy().reshape(-1, 16).astype(np.int64)
        s = s[s[:, 0] != 0]
        slots = [k for k in range(15) if (s[:, k] != 0).all()]
        real = s[:, 15]
        print("== %s: %d workgroups, stamps %s" % (name, len(s), slots))
        print("   dispatch skew (real time, 10 ns ticks): start max-min = %d ticks" % (real.max() - real.min()))
        span = (s[:, slots[-1]].max() - s[:, slots[0]].min())
        print("   first start -> last end: %d cycles; per-WG total median %d max %d" %
              (span, np.median(s[:, slots[-1]] - s[:, slots[0]]), (s[:, slots[-1]] - s[:, slots[0]]).max()))
        for a, b in zip(slots[:-1], slots[1:]):
            d = s[:, b] - s[:, a]
            print("   %2d -> %2d : median %7d  p90 %7d  max %7d cycles" % (a, b, np.median(d), np.percentile(d, 90), d.max()))


if __name__ == "__main__":
    main()
