#!/usr/bin/env python3
"""
Full-model step times through the drop-in modules ("plus full-model step numbers for C3"), and the measured
device copy bandwidth (the denominator SURVEY asks for beside the nominal 8 TB/s).  Synthetic TACRED-shaped batch, random
weights; eager launches (ctypes + torch), one optimizer-free fwd+bwd per step.  Prints one JSON object.
  C2: GCNClassifier, 2-layer GCN, no LSTM, B=50 T=100 hidden=200 (emb 300 + pos 30 + ner 30)
  C3: the same with the BiLSTM in front (rnn_hidden 200), i.e. C-GCN
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_over_pruned_trees_amd.model import gcn, tree  # noqa: E402
from gcn_over_pruned_trees_amd.utils import synthetic  # noqa: E402


def opt_for(rnn, dtype):
    return dict(vocab_size=5000, emb_dim=300, pos_dim=30, ner_dim=30, hidden_dim=200, num_layers=2, input_dropout=0.5, gcn_dropout=0.5,
                word_dropout=0.04, emb_dropout=0.0, topn=1e10, prune_k=1, pooling="max", pooling_l2=0.003, mlp_layers=2, no_adj=False,
                rnn=rnn, rnn_hidden=200, rnn_layers=1, rnn_dropout=0.5, cuda=True, dataset="tacred", num_class=42, adj_type="regular",
                gcn_dtype=dtype, gcn_check_trees=False)


def main():
    dev = torch.device("cuda:0")
    B, T = 50, 100
    tb = synthetic.random_tree_batch(1236, B, T, "tacred")
    rng = np.random.RandomState(7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    words = rng.randint(2, 5000, size=(B, T)).astype(np.int64)
    words[tb["masks"]] = 0
    inputs = (t(words), t(tb["masks"]), t(rng.randint(0, 47, size=(B, T)).astype(np.int64)), t(rng.randint(0, 15, size=(B, T)).astype(np.int64)),
              t(tb["deprel"]), t(tb["head"]), t(tb["subj_pos"]), t(tb["obj_pos"]))
    labels = t(rng.randint(0, 42, size=(B,)).astype(np.int64))
    out = {}
    for name, rnn in (("C2_gcn", False), ("C3_cgcn", True)):
        for dtype in ("fp32", "bf16"):
            torch.manual_seed(1234)
            model = gcn.GCNClassifier(opt_for(rnn, dtype)).to(dev).train()
            cache = tree.TreeCache.build(inputs[5], inputs[6], inputs[7], inputs[4], 1, masks=inputs[1], want_label=False)
            idx = torch.arange(B, device=dev)

            def step(cached):
                model.zero_grad(set_to_none=True)
                logits, pooled = model(inputs, trees=cache.batch(idx, T) if cached else None)
                loss = torch.nn.functional.cross_entropy(logits, labels) + 0.003 * (pooled ** 2).sum(1).mean()
                loss.backward()
                return loss.detach()        # no reference to the autograd graph survives the step (its AccumulateGrad nodes would pin a stream)
            for cached in (False, True):
                for _ in range(10):
                    step(cached)
                torch.cuda.synchronize()
                n = 100
                t0 = time.perf_counter()
                for _ in range(n):
                    loss = step(cached)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                out["%s_%s_%s" % (name, dtype, "cached_trees" if cached else "pruned_in_step")] = dict(
                    ms_per_step=round(dt * 1e3, 3), sentences_per_s=round(B / dt), loss=float(loss))
    # device copy bandwidth: 1 GiB read + 1 GiB write per copy
    a = torch.empty((1 << 28,), dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    out["copy_bandwidth_GBps"] = round(20 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    out["note"] = "eager launches through the modules (no hipGraph): embeddings, optional BiLSTM (MIOpen), pruner or cached trees, GCN layers, " \
                  "fused pooling, MLP, loss, backward; B=50 T=100"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
