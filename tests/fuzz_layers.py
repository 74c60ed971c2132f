#!/usr/bin/env python3
"""
Developer aid: random small shapes of the layer stack (1-3 layers, any widths, any K, ragged lengths, dropout off) through
model.gcn.gcn_layers in fp32 against the oracle (backward driven by the device's own activations), with the row-tile kernel's 4-wave form forced on a random half of the cases.
    python tests/fuzz_layers.py [seconds]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gcn_over_pruned_trees_amd import _lib  # noqa: E402
from gcn_over_pruned_trees_amd.model import gcn, tree  # noqa: E402
from gcn_over_pruned_trees_amd.utils import synthetic  # noqa: E402
from oracle import gcn_ref, prune_ref  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def fro(a, b):
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def bf16_case(rng, dev):
    """bf16 storage: the 4-wave, the 8-wave and the column-split form must agree bit for bit (forward, input gradient), with dropout on, padded and
    token-packed; and stay within bf16 distance of the fp32 result."""
    B, T, K, L = int(rng.randint(1, 24)), int(rng.randint(4, 90)), int(rng.randint(0, 3)), int(rng.randint(1, 4))
    dims = [int(rng.choice([8, 16, 40, 72, 104, 200, 300, 360])) for _ in range(L + 1)]
    packed = bool(rng.randint(0, 2))
    tb = synthetic.random_tree_batch(int(rng.randint(1 << 30)), B, T, "tacred")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    trees = tree.prune_to_csr(t(tb["head"]), t(tb["subj_pos"]), t(tb["obj_pos"]), t(tb["deprel"]), K, masks=t(tb["masks"]))
    Wn, bn = synthetic.layer_params(int(rng.randint(1 << 30)), dims)
    xn, gyn = synthetic.normal(int(rng.randint(1 << 30)), (B, T, dims[0])), synthetic.normal(int(rng.randint(1 << 30)), (B, T, dims[-1]))
    x0, g0 = t(xn).to(torch.bfloat16), t(gyn)
    if packed:
        keep = ~t(tb["masks"])
        trees = trees.pack(tb["lens"].tolist())
        x0, g0 = x0[keep].contiguous(), g0[keep].contiguous()
    drop = [0.3] * (L - 1) + [0.0]
    outs = {}
    split = int(rng.randint(1, 9))
    for mode in ("0", "1", "cs", "fp32"):
        _lib.set_option(_lib.OPT_FOUR_WAVES, -1 if mode == "cs" else (0 if mode == "fp32" else int(mode)))
        _lib.set_option(_lib.OPT_COL_SPLIT, split if mode == "cs" else 0)       # the column-split form, forced with a random split
        x = (x0.float() if mode == "fp32" else x0.clone()).requires_grad_()
        Ws = [t(w).requires_grad_() for w in Wn]
        bs = [t(b).requires_grad_() for b in bn]
        h = gcn.gcn_layers(x, Ws, bs, trees, drop, list(range(11, 11 + L)), torch.float32 if mode == "fp32" else torch.bfloat16, torch.float32)
        h.backward(g0)
        outs[mode] = (h.detach().float(), x.grad.float(), [w.grad for w in Ws])
    _lib.set_option(_lib.OPT_FOUR_WAVES, -1)
    _lib.set_option(_lib.OPT_COL_SPLIT, -1)
    a, b, c, f = outs["0"], outs["1"], outs["cs"], outs["fp32"]
    ok = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    e = fro(a[0].cpu().numpy(), f[0].cpu().numpy())
    if not ok or e > 5e-2:
        print("BF16 MISMATCH B=%d T=%d K=%d dims=%s packed=%s equal=%s fro_vs_fp32=%.2e" % (B, T, K, dims, packed, ok, e))
        sys.exit(1)
    return e


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    if os.environ.get("FUZZ_BF16"):
        dev = torch.device("cuda:0")
        rng = np.random.RandomState(int(os.environ.get("FUZZ_SEED", "7")))
        t0, n, worst = time.time(), 0, 0.0
        while time.time() - t0 < budget:
            worst = max(worst, bf16_case(rng, dev))
            n += 1
        print("bf16 fuzz ok: %d cases in %.0f s, worst Frobenius distance to fp32 %.2e" % (n, time.time() - t0, worst))
        return
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(int(os.environ.get("FUZZ_SEED", "7")))
    t0, n, worst = time.time(), 0, 0.0
    while time.time() - t0 < budget:
        B, T, K, L = int(rng.randint(1, 24)), int(rng.randint(4, 90)), int(rng.randint(0, 3)), int(rng.randint(1, 4))
        dims = [int(rng.choice([4, 8, 12, 20, 36, 52, 100, 200, 300, 360]) + rng.choice([0, 1, 2, 3, 4])) for _ in range(L + 1)]
        four = bool(rng.randint(0, 2))
        _lib.set_option(_lib.OPT_FOUR_WAVES, 1 if four else 0)
        tb = synthetic.random_tree_batch(int(rng.randint(1 << 30)), B, T, "tacred")
        adj = prune_ref.batch_adj(tb["head"], tb["subj_pos"], tb["obj_pos"], tb["deprel"], tb["lens"], K)["adj"]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        trees = tree.prune_to_csr(t(tb["head"]), t(tb["subj_pos"]), t(tb["obj_pos"]), t(tb["deprel"]), K, masks=t(tb["masks"]))
        trees.check(expect_maxlen=T)
        Wn, bn = synthetic.layer_params(int(rng.randint(1 << 30)), dims)
        xn, gyn = synthetic.normal(int(rng.randint(1 << 30)), (B, T, dims[0])), synthetic.normal(int(rng.randint(1 << 30)), (B, T, dims[-1]))
        x = t(xn).requires_grad_()
        Ws = [t(w).requires_grad_() for w in Wn]
        bs = [t(b).requires_grad_() for b in bn]
        h = gcn.gcn_layers(x, Ws, bs, trees, compute_dtype=torch.float32)
        h.backward(t(gyn))
        href, _ = gcn_ref.gcn_forward(adj, xn, Wn, bn)
        # the oracle's backward runs on the DEVICE's activations (the forward is deterministic, so the prefixes of the stack give
        # them): a pre-activation within rounding of zero may be cut the other way by either side, which is not an error
        with torch.no_grad():
            acts = [gcn.gcn_layers(t(xn), [w.detach() for w in Ws[:l + 1]], [b.detach() for b in bs[:l + 1]], trees,
                                   compute_dtype=torch.float32).cpu().numpy() for l in range(L)]
        dx, dWs, dbs = gcn_ref.gcn_backward(adj, xn, Wn, bn, gyn, acts=acts)
        errs = [rel(h.detach().cpu().numpy(), href), rel(x.grad.cpu().numpy(), dx)] + \
               [rel(Ws[l].grad.cpu().numpy(), dWs[l]) for l in range(L)] + [rel(bs[l].grad.cpu().numpy(), dbs[l]) for l in range(L)]
        worst = max(worst, max(errs))
        if errs[0] > 1e-5 or max(errs[1:]) > 1e-4:
            print("MISMATCH B=%d T=%d K=%d dims=%s four_waves=%s errs=%s" % (B, T, K, dims, four, ["%.2e" % e for e in errs]))
            sys.exit(1)
        n += 1
    print("fuzz ok: %d cases in %.0f s, worst relative error %.2e" % (n, time.time() - t0, worst))


if __name__ == "__main__":
    main()
