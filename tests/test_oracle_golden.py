"""
Pins the CPU oracle (oracle/) to the golden vectors recorded from the reference.
CPU only; the whole file runs in a few seconds.
"""
import numpy as np
import pytest

from conftest import dense_from_coo, load_golden
from helpers import FWD_RTOL, GRAD_RTOL, LAYER_CASES, layer_case, max_rel
from oracle import gcn_ref, prune_ref


def _check_trees(name, Ks, with_status=False):
    g = load_golden(name)
    B, T = g["head"].shape
    for K in Ks:
        r = prune_ref.batch_adj(g["head"], g["subj_pos"], g["obj_pos"], g["deprel"], g["lens"], K)
        want = dense_from_coo(g["coo_k%d" % K], B, T)
        if with_status:
            np.testing.assert_array_equal(r["status"], g["status_k%d" % K])
            ok = g["status_k%d" % K] == 0
        else:
            assert r["rc"] == 0
            ok = np.ones(B, bool)
        np.testing.assert_array_equal(r["adj"][ok], want[ok])          # labels are integers: exact
        np.testing.assert_array_equal(r["root"][ok], g["root_k%d" % K][ok])


def test_pruner_tacred_samples():
    _check_trees("trees_tacred_samples.npz", (0, 1, 2))


def test_pruner_known_answers():
    """Known answers quoted in SURVEY.md 8c for train.json[0] and [1]."""
    g = load_golden("trees_tacred_samples.npz")
    got = {}
    for K in (0, 1, 2):
        r = prune_ref.batch_adj(g["head"][:2], g["subj_pos"][:2], g["obj_pos"][:2], g["deprel"][:2], g["lens"][:2], K)
        got[K] = r
    a0 = got[0]["adj"][0]
    assert got[0]["root"][0] == 14 and int((a0 != 0).sum()) == 10 and int(a0.sum()) == 546
    assert sorted(np.nonzero(got[0]["kept"][0])[0].tolist()) == [12, 14, 16, 20]
    a1 = got[1]["adj"][0]
    assert int((a1 != 0).sum()) == 28 and int(a1.sum()) == 1466
    assert sorted(np.nonzero(got[1]["kept"][0])[0].tolist()) == list(range(11, 21))
    assert got[0]["root"][1] == 7 and int((got[0]["adj"][1] != 0).sum()) == 13
    assert int((got[1]["adj"][1] != 0).sum()) == 28 and int((got[2]["adj"][1] != 0).sum()) == 34


def test_pruner_random_trees():
    _check_trees("trees_random.npz", (0, 1, 2, 3))


def test_pruner_edge_cases():
    _check_trees("trees_edge_cases.npz", (0, 1, 2), with_status=True)


def test_pruner_errors_not_in_reference_fixtures():
    """prune_k < 0 crashes the fork (tree.py:194); a head cycle hangs it (tree.py:91-94): codes only."""
    g = load_golden("trees_edge_cases.npz")
    r = prune_ref.batch_adj(g["head"][:1], g["subj_pos"][:1], g["obj_pos"][:1], g["deprel"][:1], g["lens"][:1], -1)
    assert r["status"][0] == prune_ref.E_PRUNE_NEGATIVE
    head = g["head"][:1].copy()
    head[0, :3] = [2, 3, 1]
    r = prune_ref.batch_adj(head, g["subj_pos"][:1], g["obj_pos"][:1], g["deprel"][:1], g["lens"][:1], 1)
    assert r["status"][0] == prune_ref.E_CYCLE


def test_python_closed_form_agrees_with_c():
    g = load_golden("trees_random.npz")
    T = g["head"].shape[1]
    for K in (0, 2):
        r = prune_ref.batch_adj(g["head"][:64], g["subj_pos"][:64], g["obj_pos"][:64], g["deprel"][:64], g["lens"][:64], K)
        for b in range(64):
            adj, keep, lca = prune_ref.head_to_adj_py(g["head"][b], g["subj_pos"][b], g["obj_pos"][b], g["deprel"][b],
                                                      int(g["lens"][b]), T, K)
            np.testing.assert_array_equal(adj, r["adj"][b])
            assert lca == r["root"][b]


@pytest.mark.parametrize("name", LAYER_CASES)
def test_layers_forward_backward(name):
    g = layer_case(name)
    h, mask = gcn_ref.gcn_forward(g["adj"], g["x"], g["Ws"], g["bs"])
    np.testing.assert_array_equal(mask, g["mask"])
    assert max_rel(h, g["h"]) <= FWD_RTOL
    dx, dWs, dbs = gcn_ref.gcn_backward(g["adj"], g["x"], g["Ws"], g["bs"], g["gy"])
    assert max_rel(dx, g["dx"]) <= GRAD_RTOL
    for l in range(int(g["layers"])):
        assert max_rel(dWs[l], g["dW%d" % l]) <= GRAD_RTOL
        assert max_rel(dbs[l], g["db%d" % l]) <= GRAD_RTOL


@pytest.mark.parametrize("name", ["layers_c1_l2.npz", "layers_c2s.npz"])
def test_torch_cpu_restatement_pinned_to_reference(name):
    """oracle/gcn_ref_torch.py (bench.py's cpu_baseline: the reference's own library ops -- bmm, 2 x linear, autograd) against the
    outputs and gradients recorded from the live reference."""
    import torch
    from oracle import gcn_ref_torch
    g = layer_case(name)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    h, dx, dWs, dbs = gcn_ref_torch.forward_backward(t(g["adj"]), t(g["x"]), [t(w) for w in g["Ws"]], [t(b) for b in g["bs"]], t(g["gy"]))
    _, _, mask = gcn_ref_torch.prep(t(g["adj"]))
    np.testing.assert_array_equal(mask.numpy(), g["mask"])
    assert max_rel(h.numpy(), g["h"]) <= FWD_RTOL
    assert max_rel(dx.numpy(), g["dx"]) <= GRAD_RTOL
    for l in range(int(g["layers"])):
        assert max_rel(dWs[l].numpy(), g["dW%d" % l]) <= GRAD_RTOL
        assert max_rel(dbs[l].numpy(), g["db%d" % l]) <= GRAD_RTOL


def test_bf16_variant_close_to_fp32():
    g = layer_case("layers_c1_l2.npz")
    h, _, _ = gcn_ref.gcn_forward_bf16(g["adj"], g["x"], g["Ws"], g["bs"])
    assert max_rel(h, g["h"]) <= 2e-2
    x = np.float32([1.0, 1.00390625, -3.14159, 1e-30, 65504.0])
    r = gcn_ref.round_bf16(x)
    assert r[0] == 1.0 and r[1] == 1.0 and abs(r[2] + 3.140625) < 1e-6


def test_dropout_mask_semantics():
    g = layer_case("layers_c1_l2.npz")
    rng = np.random.RandomState(5)
    m = [(rng.random_sample(g["h"].shape) < 0.5).astype(np.float32)]
    h, _ = gcn_ref.gcn_forward(g["adj"], g["x"], g["Ws"], g["bs"], drop_masks=m, drop_p=0.5)
    dx, dWs, dbs = gcn_ref.gcn_backward(g["adj"], g["x"], g["Ws"], g["bs"], g["gy"], drop_masks=m, drop_p=0.5)
    # the explicit backward must equal torch autograd of the same arithmetic with the same masks
    import torch
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    A = t((g["adj"] != 0).astype(np.float32))
    denom = A.sum(2, keepdim=True) + 1
    x = t(g["x"]).requires_grad_()
    Ws = [t(w).requires_grad_() for w in g["Ws"]]
    bs = [t(b).requires_grad_() for b in g["bs"]]
    hh = x
    for l in range(2):
        z = torch.nn.functional.linear(A.bmm(hh), Ws[l], bs[l]) + torch.nn.functional.linear(hh, Ws[l], bs[l])
        hh = torch.relu(z / denom)
        if l == 0:
            hh = hh * t(m[0]) * 2.0
    hh.backward(t(g["gy"]))
    assert max_rel(h, hh.detach().numpy()) <= FWD_RTOL
    assert max_rel(dx, x.grad.numpy()) <= GRAD_RTOL
    for l in range(2):
        assert max_rel(dWs[l], Ws[l].grad.numpy()) <= GRAD_RTOL
        assert max_rel(dbs[l], bs[l].grad.numpy()) <= GRAD_RTOL


def test_diag_deprel_oracle_matches_reference_golden():
    """N2: numpy restatement of the diagonal_deprel layers vs outputs/gradients recorded from the reference."""
    g = load_golden("layers_diag_deprel.npz")
    B, T, L = int(g["B"]), int(g["T"]), int(g["layers"])
    adj = dense_from_coo(g["coo"], B, T)
    h, mask = gcn_ref.diag_forward(adj, g["x"], g["deprel"], g["Wp"], g["bp"], g["E"], L)
    np.testing.assert_allclose(h, g["h"], rtol=1e-5, atol=1e-6)
    assert (mask == g["mask"]).all()
    dx, dWp, dbp, dE = gcn_ref.diag_backward(adj, g["x"], g["deprel"], g["Wp"], g["bp"], g["E"], L, g["gy"])
    for got, key in ((dx, "dx"), (dWp, "dWp"), (dbp, "dbp"), (dE, "dE")):
        np.testing.assert_allclose(got, g[key], rtol=2e-4, atol=2e-5, err_msg=key)
    assert (g["dE"][0] == 0).all() and np.abs(g["dE"]).max() > 0


def test_full_deprel_oracle_matches_reference_golden():
    """N3: numpy restatement of the full_deprel layers vs outputs/gradients recorded from the reference (4 option sets)."""
    import json
    g = load_golden("layers_full_deprel.npz")
    B, T = int(g["B"]), int(g["T"])
    adj = dense_from_coo(g["coo"], B, T)
    for ci, c in enumerate(json.loads(str(g["cases"]))):
        kw = dict(max_depth=c["deprel_max_depth"], directed=c["deprel_directed"], self_loop=c["deprel_self_loop"])
        h, mask = gcn_ref.full_forward(adj, g["x"], g["deprel"], g["W"], g["b"], g["E"], c["layers"], **kw)
        np.testing.assert_allclose(h, g["h%d" % ci], rtol=1e-4, atol=1e-5, err_msg="h, case %d" % ci)
        assert (mask == g["mask%d" % ci]).all()
        got = gcn_ref.full_backward(adj, g["x"], g["deprel"], g["W"], g["b"], g["E"], c["layers"], g["gy"], **kw)
        for a, key in zip(got, ("dx", "dW", "db", "dE")):
            ref = g["%s%d" % (key, ci)]
            assert np.abs(a - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), (key, ci)
