#!/usr/bin/env python3
"""
The no-LSTM classifier's whole training step (embeddings, pruned trees from a TreeCache, GCN layers, fused pooling, MLP, loss,
backward) on the tokens of the pruned trees only (CompactTrees) and on all rows, launched eagerly and replayed as
ONE hipGraph.
The mirror looks embeddings up through its own autograd function (index_add_ backward): PyTorch-ROCm 2.10's embedding backward
takes ~140 us per table and, above 3072 indices, a rocPRIM sort path that cannot be replayed in a hipGraph.
Prints one JSON object.
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
from model_step import opt_for  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, T = 50, 100
    out = {}
    for lengths, compact in (("tacred", True), ("full", True), ("tacred", False)):
        tb = synthetic.random_tree_batch(1236, B, T, lengths)
        rng = np.random.RandomState(7)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        words = rng.randint(2, 5000, size=(B, T)).astype(np.int64)
        words[tb["masks"]] = 0
        inputs = (t(words), t(tb["masks"]), t(rng.randint(0, 47, size=(B, T)).astype(np.int64)), t(rng.randint(0, 15, size=(B, T)).astype(np.int64)),
                  t(tb["deprel"]), t(tb["head"]), t(tb["subj_pos"]), t(tb["obj_pos"]))
        labels = t(rng.randint(0, 42, size=(B,)).astype(np.int64))
        torch.manual_seed(1234)
        opt = dict(opt_for(False, "bf16"), gcn_graph_rng=True, word_dropout=0.0)
        model = gcn.GCNClassifier(opt).to(dev).train()
        cache = tree.TreeCache.build(inputs[5], inputs[6], inputs[7], inputs[4], 1, masks=inputs[1], want_label=False, compact=True)
        idx = torch.arange(B, device=dev)
        Tc = cache.compact.Tc if compact else T

        def step():
            model.zero_grad(set_to_none=True)
            logits, pooled = model(inputs, trees=cache.batch(idx, T, compact=compact))
            loss = torch.nn.functional.cross_entropy(logits, labels) + 0.003 * (pooled ** 2).sum(1).mean()
            loss.backward()
            return loss.detach()

        def timeit(fn, n=200):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        key = "C2_gcn_bf16_%s_%s_len" % ("pooled_only" if compact else "all_rows", lengths)
        dt = timeit(step)
        out[key] = dict(rows=B * Tc, rows_full=B * T, eager_ms_per_step=round(dt * 1e3, 3), eager_sentences_per_s=round(B / dt))
        if True:        # (the mirror's own embedding backward, model/gcn.py::_EmbedFn, has no index-count limit)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                loss = step()
            torch.cuda.synchronize()
            dt = timeit(graph.replay)
            out[key].update(hipgraph_ms_per_step=round(dt * 1e3, 3), hipgraph_sentences_per_s=round(B / dt), loss=float(loss))
        del model
    out["note"] = "GCNClassifier fwd+bwd (no optimizer), B=50 T=100, bf16 layer stack, CompactTrees from a TreeCache; eager = ctypes + torch " \
                  "launches from Python, hipgraph = the same step captured once with torch.cuda.graph and replayed"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
