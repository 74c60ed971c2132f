"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/gcnpt.h declares, argument validation works without a GPU, and the host-side mirror keeps the
reference's names.  No compute is launched here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "gcnpt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gcnpt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from gcn_over_pruned_trees_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 38
    for n in names:
        assert hasattr(handle, n), "libgcnpt.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names                      # the ctypes binding covers the header, nothing else
    assert _lib.lib().gcnpt_abi_version() == _lib.ABI_VERSION == 7


def test_argument_validation_needs_no_gpu():
    from gcn_over_pruned_trees_amd import _lib
    L = _lib.lib()
    assert L.gcnpt_prune_to_csr(None, None, None, None, None, None, None, 1, 1, 1, 3, None, None, None, None, None, None, None, None, None) == _lib.E_INVALID
    assert b"null" in L.gcnpt_last_error()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.addressof(buf)
    assert L.gcnpt_prune_to_csr(None, p, p, p, p, p, None, 1, 4, -1, 12, p, p, None, None, None, p, None, None, p) == _lib.E_PRUNE_NEGATIVE
    assert L.gcnpt_layer_fwd(None, p, 0, p, p, p, p, p, None, 1, 1, 8, 8, p, 0, 7, 0.0, 0, None, None) == _lib.E_INVALID
    assert L.gcnpt_layer_fwd(None, p, 0, p, p, p, p, p, None, 1, 1, 8, 8, p, 0, 0, 1.5, 0, None, None) == _lib.E_INVALID
    assert L.gcnpt_layer_bwd_data(None, p, p, 0, p, p, p, p, p, 1, 1, 8, 8, None, 0, 0, 1.0, None, None, None, None, 1.0, 0) == _lib.E_INVALID
    assert L.gcnpt_frag_bytes(5000, 360, _lib.BF16) == 23 * 157 * 1024 and L.gcnpt_frag_bytes(5000, 360, _lib.F32) == 23 * 314 * 1024
    assert L.gcnpt_packed_bytes(200, 360, _lib.BF16) == 13 * 12 * 64 * 16
    assert L.gcnpt_packed_bytes(200, 360, _lib.F32) == 13 * 23 * 64 * 16
    assert L.gcnpt_packed_bytes(0, 360, _lib.BF16) == 0
    # the whole-loop entry points check that the layers chain before anything is launched
    two = lambda *v: (ctypes.c_int * 2)(*v)  # noqa: E731
    ptrs = (ctypes.c_void_p * 2)(p, p)
    f2, u2 = (ctypes.c_float * 2)(0.0, 0.0), (ctypes.c_uint64 * 2)(0, 0)
    assert L.gcnpt_layers_fwd(None, 2, p, 0, ptrs, ptrs, p, p, p, None, 1, 1, two(8, 9), two(8, 8), ptrs, two(0, 0), 0, f2, u2, None, None) == _lib.E_INVALID
    assert b"layer 1 reads 9 columns" in L.gcnpt_last_error()
    assert L.gcnpt_layers_fwd(None, 9, p, 0, ptrs, ptrs, p, p, p, None, 1, 1, two(8, 8), two(8, 8), ptrs, two(0, 0), 0, f2, u2, None, None) == _lib.E_INVALID
    none2 = (ctypes.c_void_p * 2)(None, None)
    assert L.gcnpt_layers_bwd(None, 2, p, ptrs, two(0, 0), ptrs, p, p, p, p, 1, 1, two(8, 8), two(8, 8), none2, two(0, 0), 0, f2, None, None, None, None) == _lib.E_INVALID
    assert b"dh[1] must exist" in L.gcnpt_last_error()
    assert L.gcnpt_layers_bwd(None, 2, p, ptrs, two(0, 0), ptrs, p, p, p, p, 1, 1, two(8, 8), two(8, 8), ptrs, two(0, 0), 0, f2, ptrs, None, None, None) == _lib.E_INVALID
    assert b"weight gradients need" in L.gcnpt_last_error()
    # the whole step from one call: the struct is validated like the three calls it stands for
    st = _lib.Step()
    st.n_layers, st.B, st.T, st.parts = 9, 1, 1, 7
    assert L.gcnpt_layers_step(None, ctypes.byref(st)) == _lib.E_INVALID and L.gcnpt_layers_step(None, None) == _lib.E_INVALID
    st.n_layers = 2
    assert L.gcnpt_layers_step(None, ctypes.byref(st)) == _lib.E_INVALID and b"pack_weights" in L.gcnpt_last_error()
    # the backward-data call that carries the layer above's weight gradient needs that gradient's images and accumulators
    assert L.gcnpt_layer_bwd_data_wgrad(None, p, p, 0, p, p, p, p, p, 1, 1, 8, 8, p, 0, 0, 1.0, p, p, p, None, 1.0, 1, None, p, 8, 8, p, p) == _lib.E_INVALID
    assert b"two fragment images" in L.gcnpt_last_error()
    # the option table: the library's one piece of process state, no environment access after load
    assert L.gcnpt_get_option(_lib.OPT_SIDE_TILES) == 192 and L.gcnpt_get_option(_lib.OPT_FOUR_WAVES) == -1 and L.gcnpt_get_option(_lib.OPT_COL_SPLIT) == -1
    assert L.gcnpt_set_option(_lib.OPT_COL_SPLIT, 9) == _lib.E_INVALID
    assert L.gcnpt_set_option(99, 1) == _lib.E_INVALID and L.gcnpt_set_option(_lib.OPT_DETERMINISTIC, 2) == _lib.E_INVALID
    old = _lib.set_option(_lib.OPT_DETERMINISTIC, 1)
    os.environ["GCNPT_DETERMINISTIC"] = "0"                       # ignored: defaults were read once, at load
    try:
        assert L.gcnpt_get_option(_lib.OPT_DETERMINISTIC) == 1
    finally:
        os.environ.pop("GCNPT_DETERMINISTIC", None)
        _lib.set_option(_lib.OPT_DETERMINISTIC, old)
    assert L.gcnpt_launch_empty(None, 0, 64, 0, 64) == _lib.E_INVALID
    assert L.gcnpt_compact_trees(None, p, p, None, None, None, p, None, p, p, 1, 4, 12, 0, 12, p, p, None, None, None, p, None, p, p, p, p) == _lib.E_INVALID
    assert L.gcnpt_compact_trees(None, p, p, None, None, None, p, None, p, p, 1, 4, 12, 4, 12, p, p, p, None, None, p, None, p, p, p, p) == _lib.E_INVALID
    assert b"labels wanted" in L.gcnpt_last_error()


def test_no_cpu_fallback():
    import torch
    from gcn_over_pruned_trees_amd.model import gcn, tree
    head = torch.zeros((1, 4), dtype=torch.int64)
    with pytest.raises(RuntimeError, match="GPU only"):
        tree.prune_to_csr(head, head, head, head, 1, lens=torch.tensor([4], dtype=torch.int32))
    with pytest.raises(TypeError):
        gcn.gcn_layer(torch.zeros(1, 4, 8), torch.zeros(8, 8), torch.zeros(8), trees=None)


def test_host_mirror_keeps_reference_surface():
    import inspect
    from gcn_over_pruned_trees_amd.model import gcn
    assert list(inspect.signature(gcn.GCNClassifier.__init__).parameters) == ["self", "opt", "emb_matrix"]
    assert list(inspect.signature(gcn.GCNRelationModel.__init__).parameters) == ["self", "opt", "emb_matrix"]
    assert list(inspect.signature(gcn.GCN.__init__).parameters) == ["self", "opt", "embeddings", "mem_dim", "num_layers"]
    assert list(inspect.signature(gcn.GCN.forward).parameters) == ["self", "adj", "inputs"]
    assert list(inspect.signature(gcn.pool).parameters) == ["h", "mask", "type"]
    opt = dict(vocab_size=50, emb_dim=8, pos_dim=2, ner_dim=2, hidden_dim=16, num_layers=2, input_dropout=0.0, gcn_dropout=0.5,
               prune_k=1, pooling="max", mlp_layers=2, rnn=True, rnn_hidden=4, rnn_layers=1, rnn_dropout=0.0, dataset="tacred",
               num_class=42, topn=10, cuda=False)
    m = gcn.GCNClassifier(opt)
    keys = set(m.state_dict().keys())
    for k in ("gcn_model.emb.weight", "gcn_model.gcn.emb.weight", "gcn_model.deprel_emb.weight", "gcn_model.gcn.W.0.weight",
              "gcn_model.gcn.W.1.bias", "gcn_model.gcn.rnn.weight_ih_l0_reverse", "gcn_model.out_mlp.2.weight", "classifier.bias"):
        assert k in keys, k
    assert m.gcn_model.gcn.W[0].weight.shape == (16, 8) and m.gcn_model.deprel_emb.weight.shape == (85, 1)
    assert float(m.conv_l2()) > 0
    with pytest.raises(ValueError):               # gcn.py:388: 'Adjacency aggregation type not supported.'
        gcn.GCNClassifier(dict(opt, adj_type="concat_deprel"))
    full = gcn.GCNClassifier(dict(opt, adj_type="full_deprel", deprel_emb_dim=5))
    assert tuple(full.state_dict()["gcn_model.gcn.W.weight"].shape) == (5 * opt["hidden_dim"], full.gcn_model.gcn.in_dim)
    assert tuple(full.state_dict()["gcn_model.deprel_emb.weight"].shape) == (85, 5)
