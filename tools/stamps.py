#!/usr/bin/env python3
"""
Developer diagnostic: per-phase cycle stamps of the hot kernels (needs `make -C gcn-over-pruned-trees_amd/csrc stamps`).
Runs the bench workload with libgcnpt_stamps.so and prints, for every launch of the step, per stamp interval the median /
max cycles over workgroups, the dispatch skew (100 MHz real time) and the first-start -> last-end span.  Row-tile workgroups stamp
slots 0-10 (the column-split form 0-8 and 9-11 inside its gather), a weight-gradient unit slots 11-14 (the LAST unit a workgroup ran), a
passenger workgroup 0 (entry) and 10 (exit).
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


def report(name, s):
    if os.environ.get("GCNPT_STAMPS_DUMP"):                 # raw stamps per launch, for offline analysis
        np.save(os.path.join(os.environ["GCNPT_STAMPS_DUMP"], "stamps_%s.npy" % name.replace("+", "_")), s)
    s = s[(s != 0).any(1)]
    print("== %s: %d workgroups stamped" % (name, len(s)))
    groups = {}
    for row in s:
        key = tuple(k for k in range(15) if row[k] != 0)
        groups.setdefault(key, []).append(row)
    for slots, rows in sorted(groups.items(), key=lambda kv: -len(kv[1])):
        r = np.array(rows)
        print("  -- %d workgroups with stamps %s" % (len(r), list(slots)))
        real = r[:, 15]
        if (real != 0).all():
            print("     dispatch skew (real time, 10 ns ticks): start max-min = %d" % (real.max() - real.min()))
        if len(slots) < 2:
            continue
        tot = r[:, slots[-1]] - r[:, slots[0]]
        print("     first start -> last end: %d cycles; per-WG total median %d max %d" %
              (r[:, slots[-1]].max() - r[:, slots[0]].min(), np.median(tot), tot.max()))
        order = sorted(slots, key=lambda k: np.median(r[:, k]))
        for a, b in zip(order[:-1], order[1:]):
            d = r[:, b] - r[:, a]
            print("     %2d -> %2d : median %7d  p90 %7d  max %7d cycles" % (a, b, np.median(d), np.percentile(d, 90), d.max()))


def main():
    args = bench.parse()
    dev = torch.device("cuda:0")
    stack = bench.Stack(args, dev, seed=1234, packed=args.layout == "packed")
    L = stack.L
    L.gcnpt_debug_set_stamps.argtypes = [ctypes.c_void_p]
    L.gcnpt_debug_set_stamps.restype = None
    L.gcnpt_debug_set_knob.argtypes = [ctypes.c_int]
    L.gcnpt_debug_set_knob.restype = None
    knob = int(os.environ.get("GCNPT_KNOB", "0"))
    L.gcnpt_debug_set_knob(knob)
    print("knob =", knob)
    buf = torch.zeros((4096 * 16,), dtype=torch.int64, device=dev)
    names = stack.launch_names()
    for _ in range(5):
        stack.step_native()
    torch.cuda.synchronize()
    # the tree build (slots: 0 entry, 1 parse staged, 2 entity chains, 3 LCA, 4 distances, 5 degrees + scans, 6 row info, 8 -> 7 rows emitted)
    # (padded layout: gcnpt_prune_to_csr; token-packed: gcnpt_prune_to_csr_packed)
    for _ in range(3):
        stack.prune()
    torch.cuda.synchronize()
    buf.zero_()
    L.gcnpt_debug_set_stamps(buf.data_ptr())
    stack.prune()
    torch.cuda.synchronize()
    L.gcnpt_debug_set_stamps(None)
    report("prune", buf.cpu().numpy().reshape(-1, 16).astype(np.int64))
    if os.environ.get("GCNPT_STAMPS_PRUNE_ONLY"):
        return
    for k in range(2, len(names) + 1):              # (the pack kernel has no stamps)
        L.gcnpt_debug_set_stamps(None)
        for _ in range(3):
            stack.step_native()
        stack.step_prefix(k - 1)
        torch.cuda.synchronize()
        buf.zero_()
        L.gcnpt_debug_set_stamps(buf.data_ptr())
        # launch k alone, behind the real prefix: its predecessors ran unstamped
        pack, fwd, bwd = stack._native_args(0)
        st, nl = stack._lib.stream(), len(stack.W)
        if k - 1 <= nl:
            l = k - 2
            stack._lib.check(L.gcnpt_layers_fwd(st, l + 1, *fwd[1:]) if l == 0 else
                             L.gcnpt_layer_fwd(st, stack._lib.ptr(stack.h1), stack.act, stack._lib.ptr(stack.wf[1]), stack._lib.ptr(stack.b[1]),
                                               stack._lib.ptr(stack.trees.row_ptr), stack._lib.ptr(stack.trees.col_idx), stack._lib.ptr(stack.trees.ell), None,
                                               stack.B, stack.T, stack.H, stack.H, stack._lib.ptr(stack.h2), stack.act, stack.compute, 0.0, 0,
                                               stack._lib.ptr(stack.sf[1]), None))
        else:
            stack._lib.check(L.gcnpt_layers_bwd_range(st, *(bwd + (0, k - 2 - nl, 1))))
        torch.cuda.synchronize()
        L.gcnpt_debug_set_stamps(None)
        report(names[k - 1], buf.cpu().numpy().reshape(-1, 16).astype(np.int64))


if __name__ == "__main__":
    main()
