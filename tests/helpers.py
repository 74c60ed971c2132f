"""Shared by the CPU and GPU test modules: rebuild fixture inputs, tolerances."""
import numpy as np

from conftest import dense_from_coo, load_golden
from gcn_over_pruned_trees_amd.utils import synthetic

# fp32 tolerances stated by SURVEY.md 8c: fwd max-abs <= 1e-5 * max|h|, grads rel <= 1e-4
FWD_RTOL = 1e-5
GRAD_RTOL = 1e-4


def max_rel(a, b):
    """max |a-b| / max |b|  (normwise, the form SURVEY.md 8c states its tolerances in)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def fro_rel(a, b):
    """||a-b||_F / ||b||_F."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def layer_case(name):
    """Load a layers_*.npz fixture; regenerate seed-defined inputs and verify their checksums."""
    g = load_golden(name)
    B, T, din, hidden, layers, seed = (int(g[k]) for k in ("B", "T", "din", "hidden", "layers", "seed"))
    if "x" in g:
        x, gy = g["x"], g["gy"]
        Ws = [g["W%d" % l] for l in range(layers)]
        bs = [g["b%d" % l] for l in range(layers)]
    else:
        Ws, bs = synthetic.layer_params(seed + 1, [din] + [hidden] * layers)
        x = synthetic.normal(seed + 2, (B, T, din))
        gy = synthetic.normal(seed + 3, (B, T, hidden))
    assert abs(float(x.astype(np.float64).sum()) - float(g["x_sum"])) < 1e-6
    assert abs(float(gy.astype(np.float64).sum()) - float(g["gy_sum"])) < 1e-6
    for l in range(layers):
        assert abs(float(Ws[l].astype(np.float64).sum()) - float(g["W%d_sum" % l])) < 1e-6
    g.update(x=x, gy=gy, Ws=Ws, bs=bs, adj=dense_from_coo(g["coo"], B, T),
             masks=np.arange(T)[None, :] >= g["lens"][:, None])
    return g


LAYER_CASES = ["layers_c1_l1.npz", "layers_c1_l2.npz", "layers_c2s.npz", "layers_c3s.npz", "layers_c5s.npz"]
