#!/usr/bin/env python3
"""
Generates tests/golden/*.npz by IMPORTING the reference (read-only, /root/reference) in the build
container and recording its outputs on fixed inputs.  Only data is written (integer arrays, float
arrays, seeds) -- never reference source.  Re-run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Reference entry points exercised:
  model/tree.py:58   head_to_tree        model/tree.py:167  tree_to_adj
  model/gcn.py:128   GCN (regular)       model/gcn.py:15    GCNClassifier (end-to-end logits)
  model/gcn.py:473   pool

Container versions are recorded inside every file (`meta`).
"""
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("GCNPT_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from model.tree import head_to_tree, tree_to_adj  # noqa: E402  (reference)
from model.gcn import GCN, GCNClassifier  # noqa: E402  (reference)
from utils import constant as ref_constant  # noqa: E402  (reference)

from gcn_over_pruned_trees_amd.utils import synthetic  # noqa: E402  (this repo)

META = json.dumps(dict(torch=torch.__version__, numpy=np.__version__, python=sys.version.split()[0],
                       reference="gstoica27/gcn-over-pruned-trees @ /root/reference"))

EXC_CODE = {"AttributeError": -3, "TypeError": -3, "UnboundLocalError": -4, "IndexError": -6, "AssertionError": -7}


def ref_adj(head, subj_pos, obj_pos, deprel, lens, T, prune):
    """model/gcn.py:105-107 on numpy inputs.  Returns dense adj [B,T,T], root [B], status [B]."""
    B = head.shape[0]
    adj = np.zeros((B, T, T), dtype=np.float32)
    root = np.full((B,), -1, dtype=np.int32)
    status = np.zeros((B,), dtype=np.int32)
    words = np.zeros_like(head)
    for b in range(B):
        try:
            tree = head_to_tree(head[b], words[b], int(lens[b]), prune, subj_pos[b], obj_pos[b], deprel[b])
            adj[b] = tree_to_adj(T, tree, directed=False, self_loop=True)
            root[b] = tree.idx
        except (AttributeError, TypeError, UnboundLocalError, IndexError, AssertionError) as e:
            status[b] = EXC_CODE[type(e).__name__]
    return adj, root, status


def coo(adj):
    b, r, c = np.nonzero(adj)
    return np.stack([b, r, c, adj[b, r, c].astype(np.int64)], 1).astype(np.int32)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, meta=np.array(META), **arrays)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------------------------
def tacred_sample_arrays():
    """The 60 sample sentences as integer arrays only (no token text)."""
    sents = []
    for split in ("train", "dev", "test"):
        with open(os.path.join(REF, "dataset", "tacred", split + ".json")) as f:
            sents += json.load(f)
    B, T = len(sents), max(len(s["token"]) for s in sents)
    head = np.zeros((B, T), np.int64)
    deprel = np.zeros((B, T), np.int64)
    subj_pos = np.full((B, T), synthetic.POS_PAD, np.int64)
    obj_pos = np.full((B, T), synthetic.POS_PAD, np.int64)
    lens = np.zeros((B,), np.int32)
    for b, s in enumerate(sents):
        n = len(s["token"])
        lens[b] = n
        head[b, :n] = [int(x) for x in s["stanford_head"]]
        deprel[b, :n] = [ref_constant.DEPREL_TO_ID.get(t, ref_constant.UNK_ID) for t in s["stanford_deprel"]]
        subj_pos[b, :n] = synthetic.positions(s["subj_start"], s["subj_end"], n)
        obj_pos[b, :n] = synthetic.positions(s["obj_start"], s["obj_end"], n)
    return dict(head=head, deprel=deprel, subj_pos=subj_pos, obj_pos=obj_pos, lens=lens), sents


def gen_trees():
    arr, _ = tacred_sample_arrays()
    T = arr["head"].shape[1]
    out = dict(arr)
    for K in (0, 1, 2):
        adj, root, status = ref_adj(arr["head"], arr["subj_pos"], arr["obj_pos"], arr["deprel"], arr["lens"], T, K)
        assert (status == 0).all()
        out["coo_k%d" % K], out["root_k%d" % K] = coo(adj), root
    # the known answers quoted in SURVEY.md 8c must come out of this very run
    adj0, _, _ = ref_adj(arr["head"][:1], arr["subj_pos"][:1], arr["obj_pos"][:1], arr["deprel"][:1], arr["lens"][:1], T, 0)
    assert int((adj0 != 0).sum()) == 10 and int(adj0.sum()) == 546
    save("trees_tacred_samples.npz", **out)

    # seeded random trees, incl. multi-token and overlapping (nested) entity spans
    batch = synthetic.random_tree_batch(20240, 1000, 48, "tacred", overlap_frac=0.15)
    out = {k: batch[k] for k in ("head", "deprel", "subj_pos", "obj_pos", "lens")}
    for K in (0, 1, 2, 3):
        adj, root, status = ref_adj(batch["head"], batch["subj_pos"], batch["obj_pos"], batch["deprel"], batch["lens"], 48, K)
        assert (status == 0).all()
        out["coo_k%d" % K], out["root_k%d" % K] = coo(adj), root
    save("trees_random.npz", **out)

    # edge cases the reference handles or raises on
    T = 12
    e = synthetic.random_tree_batch(7, 8, T, "full")
    head, deprel, subj_pos, obj_pos, lens = (e[k].copy() for k in ("head", "deprel", "subj_pos", "obj_pos", "lens"))
    # 0: subject and object are the same single token -> one-node tree -> empty adjacency
    subj_pos[0] = synthetic.positions(5, 5, T)
    obj_pos[0] = synthetic.positions(5, 5, T)
    # 1: deprel id 0 on every token: forward edges vanish under adj != 0, reverse edges (42) stay
    deprel[1] = 0
    # 2: forest, entities under different roots -> UnboundLocalError in the reference
    head[2] = [0, 1, 2, 0, 4, 5, 4, 1, 1, 5, 6, 6]
    subj_pos[2] = synthetic.positions(1, 1, T)
    obj_pos[2] = synthetic.positions(5, 5, T)
    # 3: forest, both entities under the same root -> fine, the other tree is dropped
    head[3] = [0, 1, 2, 0, 4, 5, 4, 1, 1, 5, 6, 6]
    subj_pos[3] = synthetic.positions(1, 2, T)
    obj_pos[3] = synthetic.positions(8, 8, T)
    # 4: no subject token at all -> AttributeError / TypeError in the reference
    subj_pos[4] = np.arange(1, T + 1)
    # 5: no object token: allowed, the tree is the subject's own subtree path
    obj_pos[5] = np.arange(1, T + 1)
    # 6: a head that points past the sentence length -> IndexError (sentence shortened to 9 tokens)
    lens[6] = 9
    head[6] = [2, 3, 0, 3, 4, 12, 6, 7, 8, 0, 0, 0]
    subj_pos[6, :9] = synthetic.positions(5, 5, 9)
    obj_pos[6, :9] = synthetic.positions(8, 8, 9)
    subj_pos[6, 9:] = obj_pos[6, 9:] = synthetic.POS_PAD
    deprel[6, 9:] = 0
    # 7: a chain (every token's parent is the previous one), entities at both ends
    head[7] = np.arange(0, T)
    subj_pos[7] = synthetic.positions(0, 0, T)
    obj_pos[7] = synthetic.positions(T - 1, T - 1, T)
    out = dict(head=head, deprel=deprel, subj_pos=subj_pos, obj_pos=obj_pos, lens=lens)
    for K in (0, 1, 2):
        adj, root, status = ref_adj(head, subj_pos, obj_pos, deprel, lens, T, K)
        out["coo_k%d" % K], out["root_k%d" % K], out["status_k%d" % K] = coo(adj), root, status
    print("edge-case status (K=1):", out["status_k1"].tolist())
    save("trees_edge_cases.npz", **out)


# ------------------------------------------------------------------------------------------------
def ref_opt(din, hidden, layers, prune_k):
    """isolates the layer loop behind the reference's own GCN.forward."""
    return dict(emb_dim=din, pos_dim=0, ner_dim=0, input_dropout=0.0, gcn_dropout=0.0, emb_dropout=0.0,
                rnn=False, adj_type="regular", cuda=False, dataset="tacred", prune_k=prune_k,
                hidden_dim=hidden, num_layers=layers)


def gen_layer_case(name, seed, B, T, din, hidden, layers, prune_k, lengths, store_inputs):
    batch = synthetic.random_tree_batch(seed, B, T, lengths)
    adj, _, status = ref_adj(batch["head"], batch["subj_pos"], batch["obj_pos"], batch["deprel"], batch["lens"], T, prune_k)
    assert (status == 0).all()
    dims = [din] + [hidden] * layers
    Ws, bs = synthetic.layer_params(seed + 1, dims)
    x = synthetic.normal(seed + 2, (B, T, din))
    gy = synthetic.normal(seed + 3, (B, T, hidden))
    gcn = GCN(ref_opt(din, hidden, layers, prune_k), (None, None, None, None), hidden, layers)
    with torch.no_grad():
        for l in range(layers):
            gcn.W[l].weight.copy_(torch.from_numpy(Ws[l]))
            gcn.W[l].bias.copy_(torch.from_numpy(bs[l]))
    gcn.eval()
    xt = torch.from_numpy(x).requires_grad_()
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    inputs = (xt, t(batch["masks"]), None, None, t(batch["deprel"]), t(batch["head"]), t(batch["subj_pos"]), t(batch["obj_pos"]))
    h, mask = gcn(torch.from_numpy(adj), inputs)
    h.backward(torch.from_numpy(gy))
    out = dict(seed=np.int64(seed), B=np.int64(B), T=np.int64(T), din=np.int64(din), hidden=np.int64(hidden),
               layers=np.int64(layers), prune_k=np.int64(prune_k), lens=batch["lens"],
               head=batch["head"], deprel=batch["deprel"], subj_pos=batch["subj_pos"], obj_pos=batch["obj_pos"],
               coo=coo(adj), h=h.detach().numpy(), mask=mask.numpy(), dx=xt.grad.numpy(),
               x_sum=np.float64(x.astype(np.float64).sum()), gy_sum=np.float64(gy.astype(np.float64).sum()))
    for l in range(layers):
        out["dW%d" % l] = gcn.W[l].weight.grad.numpy()
        out["db%d" % l] = gcn.W[l].bias.grad.numpy()
        out["W%d_sum" % l] = np.float64(Ws[l].astype(np.float64).sum())
        if store_inputs:
            out["W%d" % l], out["b%d" % l] = Ws[l], bs[l]
    if store_inputs:
        out["x"], out["gy"] = x, gy
    save(name, **out)


def gen_layers():
    # C1 exactly (BASELINE.json configs[0]): B=4, T=20, Din=H=200, 1 and 2 layers, K=1; inputs stored
    gen_layer_case("layers_c1_l1.npz", 100, 4, 20, 200, 200, 1, 1, np.array([20, 17, 11, 8], np.int32), True)
    gen_layer_case("layers_c1_l2.npz", 110, 4, 20, 200, 200, 2, 1, np.array([20, 17, 11, 8], np.int32), True)
    # down-scaled C2 / C3 / C5: inputs are regenerated from the seeds (checksums stored)
    gen_layer_case("layers_c2s.npz", 120, 4, 100, 360, 200, 2, 1, "tacred", False)
    gen_layer_case("layers_c3s.npz", 130, 2, 100, 400, 200, 2, 1, "full", False)
    gen_layer_case("layers_c5s.npz", 140, 1, 300, 600, 300, 2, 2, "full", False)


# ------------------------------------------------------------------------------------------------
def gen_diag():
    """adj_type='diagonal_deprel' layer stack (gcn.py:272-294) isolated behind the reference's GCN.forward."""
    seed, B, T, din, hidden, layers, K = 150, 4, 40, 56, 48, 2, 1
    batch = synthetic.random_tree_batch(seed, B, T, "tacred")
    adj, _, status = ref_adj(batch["head"], batch["subj_pos"], batch["obj_pos"], batch["deprel"], batch["lens"], T, K)
    assert (status == 0).all()
    rng = np.random.RandomState(seed + 1)
    E = rng.uniform(-1, 1, size=(85, hidden)).astype(np.float32)
    E[0] = 0.0
    (Wp,), (bp,) = synthetic.layer_params(seed + 2, [din, hidden])
    x = synthetic.normal(seed + 3, (B, T, din))
    gy = synthetic.normal(seed + 4, (B, T, hidden))
    opt = dict(ref_opt(din, hidden, layers, K), adj_type="diagonal_deprel", deprel_emb_dim=hidden)
    demb = torch.nn.Embedding(85, hidden, padding_idx=0)
    gcn = GCN(opt, (None, None, None, demb), hidden, layers)
    with torch.no_grad():
        demb.weight.copy_(torch.from_numpy(E))
        gcn.preprocessor.weight.copy_(torch.from_numpy(Wp))
        gcn.preprocessor.bias.copy_(torch.from_numpy(bp))
    gcn.eval()
    xt = torch.from_numpy(x).requires_grad_()
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    inputs = (xt, t(batch["masks"]), None, None, t(batch["deprel"]), t(batch["head"]), t(batch["subj_pos"]), t(batch["obj_pos"]))
    h, mask = gcn(torch.from_numpy(adj), inputs)
    h.backward(torch.from_numpy(gy))
    save("layers_diag_deprel.npz", B=np.int64(B), T=np.int64(T), din=np.int64(din), hidden=np.int64(hidden), layers=np.int64(layers),
         prune_k=np.int64(K), lens=batch["lens"], head=batch["head"], deprel=batch["deprel"], subj_pos=batch["subj_pos"],
         obj_pos=batch["obj_pos"], coo=coo(adj), x=x, gy=gy, E=E, Wp=Wp, bp=bp, h=h.detach().numpy(), mask=mask.numpy(),
         dx=xt.grad.numpy(), dWp=gcn.preprocessor.weight.grad.numpy(), dbp=gcn.preprocessor.bias.grad.numpy(), dE=demb.weight.grad.numpy())


# ------------------------------------------------------------------------------------------------
def gen_full():
    """adj_type='full_deprel' (gcn.py:156-167, 296-388, 400-434) isolated behind the reference's GCN.forward, eval mode
    (edge dropout and relation forgetting are training-time RNG).  in_dim == mem_dim, as the variant needs for > 1 layer."""
    seed, B, T, hidden, D, K = 170, 4, 36, 24, 6, 1
    batch = synthetic.random_tree_batch(seed, B, T, "tacred")
    adj, _, status = ref_adj(batch["head"], batch["subj_pos"], batch["obj_pos"], batch["deprel"], batch["lens"], T, K)
    assert (status == 0).all()
    rng = np.random.RandomState(seed + 1)
    E = rng.uniform(-1, 1, size=(85, D)).astype(np.float32)
    E[0] = 0.0
    W = (rng.uniform(-1, 1, size=(D * hidden, hidden)) / np.sqrt(hidden)).astype(np.float32)
    b = (rng.uniform(-1, 1, size=(D * hidden,)) / np.sqrt(hidden)).astype(np.float32)
    x = synthetic.normal(seed + 3, (B, T, hidden))
    gy = synthetic.normal(seed + 4, (B, T, hidden))
    out = dict(B=np.int64(B), T=np.int64(T), hidden=np.int64(hidden), D=np.int64(D), prune_k=np.int64(K), lens=batch["lens"],
               head=batch["head"], deprel=batch["deprel"], subj_pos=batch["subj_pos"], obj_pos=batch["obj_pos"], coo=coo(adj),
               x=x, gy=gy, E=E, W=W, b=b)
    cases = [dict(layers=2, deprel_max_depth=2, deprel_directed=False, deprel_self_loop=True),
             dict(layers=2, deprel_max_depth=1, deprel_directed=False, deprel_self_loop=True),
             dict(layers=1, deprel_max_depth=2, deprel_directed=True, deprel_self_loop=True),
             dict(layers=3, deprel_max_depth=0, deprel_directed=False, deprel_self_loop=False)]
    out["cases"] = np.array(json.dumps(cases))
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    for ci, c in enumerate(cases):
        opt = dict(ref_opt(hidden, hidden, c["layers"], K), adj_type="full_deprel", deprel_emb_dim=D, deprel_max_depth=c["deprel_max_depth"],
                   deprel_directed=c["deprel_directed"], deprel_self_loop=c["deprel_self_loop"])
        demb = torch.nn.Embedding(85, D, padding_idx=0)
        gcn = GCN(opt, (None, None, None, demb), hidden, c["layers"])
        with torch.no_grad():
            demb.weight.copy_(t(E))
            gcn.W.weight.copy_(t(W))
            gcn.W.bias.copy_(t(b))
        gcn.eval()
        xt = t(x.copy()).requires_grad_()
        inputs = (xt, t(batch["masks"]), None, None, t(batch["deprel"]), t(batch["head"]), t(batch["subj_pos"]), t(batch["obj_pos"]))
        h, mask = gcn(t(adj), inputs)
        h.backward(t(gy))
        out["h%d" % ci], out["mask%d" % ci] = h.detach().numpy(), mask.numpy()
        out["dx%d" % ci], out["dW%d" % ci], out["db%d" % ci] = xt.grad.numpy(), gcn.W.weight.grad.numpy(), gcn.W.bias.grad.numpy()
        out["dE%d" % ci] = demb.weight.grad.numpy() if demb.weight.grad is not None else np.zeros_like(E)
    save("layers_full_deprel.npz", **out)


# ------------------------------------------------------------------------------------------------
def gen_end_to_end():
    """GCNClassifier logits for 10 sample sentences, eval mode, seeded weights (GCN and C-GCN)."""
    arr, sents = tacred_sample_arrays()
    idx = np.argsort(-arr["lens"][:10], kind="stable")           # loader.py:93-94 sorts by length
    sel = idx
    T = int(arr["lens"][sel].max())
    vocab = {}
    words = np.zeros((10, T), np.int64)
    pos = np.zeros((10, T), np.int64)
    ner = np.zeros((10, T), np.int64)
    for r, b in enumerate(sel):
        s = sents[b]
        n = len(s["token"])
        toks = [w.lower() for w in s["token"]]
        toks[s["subj_start"]:s["subj_end"] + 1] = ["SUBJ-" + s["subj_type"]] * (s["subj_end"] - s["subj_start"] + 1)
        toks[s["obj_start"]:s["obj_end"] + 1] = ["OBJ-" + s["obj_type"]] * (s["obj_end"] - s["obj_start"] + 1)
        words[r, :n] = [vocab.setdefault(w, len(vocab) + 2) for w in toks]
        pos[r, :n] = [ref_constant.POS_TO_ID.get(p, 1) for p in s["stanford_pos"]]
        ner[r, :n] = [ref_constant.NER_TO_ID.get(p, 1) for p in s["stanford_ner"]]
    fields = {k: arr[k][sel][:, :T] for k in ("head", "deprel", "subj_pos", "obj_pos")}
    masks = words == 0
    for tag, rnn in (("gcn", False), ("cgcn", True), ("diag", False), ("full", False), ("semeval", False), ("avgpool", True)):
        opt = dict(vocab_size=len(vocab) + 2, emb_dim=24, pos_dim=6, ner_dim=6, hidden_dim=32, num_layers=2,
                   input_dropout=0.5, gcn_dropout=0.5, emb_dropout=0.0, word_dropout=0.04, topn=10 ** 10,
                   deprel_emb_dim=32, adj_type="regular", prune_k=1, pooling="max", mlp_layers=2,
                   pooling_l2=0.003, conv_l2=0.0, rnn=rnn, rnn_hidden=16, rnn_layers=1, rnn_dropout=0.5,
                   cuda=False, dataset="tacred", num_class=42, no_adj=False)
        if tag == "diag":
            opt["adj_type"] = "diagonal_deprel"
        if tag == "full":           # in_dim == mem_dim, as the variant needs for two layers
            opt.update(adj_type="full_deprel", deprel_emb_dim=5, emb_dim=20, deprel_max_depth=1, deprel_directed=False, deprel_self_loop=True)
        if tag == "semeval":        # 7-tuple inputs, no NER embedding (gcn.py:94, 136-139)
            opt.update(dataset="semeval", num_class=19)
        if tag == "avgpool":        # C-GCN with average pooling and one MLP layer
            opt.update(pooling="avg", mlp_layers=1)
        torch.manual_seed(4321)
        model = GCNClassifier(opt)
        model.eval()
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
        inputs = (t(words), t(masks), t(pos), t(ner), t(fields["deprel"]), t(fields["head"]),
                  t(fields["subj_pos"]), t(fields["obj_pos"]))
        if tag == "semeval":
            inputs = inputs[:3] + inputs[4:]
        with torch.no_grad():
            logits, pooled = model(inputs)
        out = dict(opt=np.array(json.dumps(opt)), words=words, masks=masks, pos=pos, ner=ner, **fields,
                   logits=logits.numpy(), pooling_output=pooled.numpy())
        for k, v in model.state_dict().items():
            out["sd:" + k] = v.numpy()
        save("e2e_%s.npz" % tag, **out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["trees", "layers", "diag", "full", "e2e"]
    if "trees" in which:
        gen_trees()
    if "layers" in which:
        gen_layers()
    if "diag" in which:
        gen_diag()
    if "full" in which:
        gen_full()
    if "e2e" in which:
        gen_end_to_end()
