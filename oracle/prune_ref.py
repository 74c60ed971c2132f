"""
oracle/prune_ref.py  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/prune_ref.c (the C restatement of model/tree.py:58-204) plus a tiny
pure-Python restatement used to cross-check the C one on small cases.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgcnpt_oracle.so")

OK, E_PRUNE_NEGATIVE, E_NO_SUBJECT, E_NO_LCA, E_CYCLE, E_BAD_HEAD, E_ASSERT = 0, -2, -3, -4, -5, -6, -7


def build(force=False):
    src = os.path.join(_HERE, "prune_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libgcnpt_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        p = ctypes.c_void_p
        _lib.gcnpt_oracle_batch_adj.argtypes = [p, p, p, p, p, ctypes.c_int, ctypes.c_int, ctypes.c_int, p, p, p, p]
        _lib.gcnpt_oracle_batch_adj.restype = ctypes.c_int
    return _lib


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def batch_adj(head, subj_pos, obj_pos, deprel, lens, prune):
    """
    head/subj_pos/obj_pos/deprel: int [B,T]; lens: int [B].
    Returns dict(adj float32 [B,T,T], kept uint8 [B,T], root int32 [B], status int32 [B], rc int).
    Mirrors model/gcn.py:102-108 (head_to_tree + tree_to_adj over the batch).
    """
    head, subj_pos, obj_pos, deprel = map(_i64, (head, subj_pos, obj_pos, deprel))
    B, T = head.shape
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    adj = np.empty((B, T, T), dtype=np.float32)
    kept = np.zeros((B, T), dtype=np.uint8)
    root = np.zeros((B,), dtype=np.int32)
    status = np.zeros((B,), dtype=np.int32)
    rc = lib().gcnpt_oracle_batch_adj(head.ctypes.data, subj_pos.ctypes.data, obj_pos.ctypes.data,
                                      deprel.ctypes.data, lens.ctypes.data, B, T, int(prune),
                                      adj.ctypes.data, kept.ctypes.data, root.ctypes.data, status.ctypes.data)
    return dict(adj=adj, kept=kept, root=root, status=status, rc=rc)


def head_to_adj_py(head, subj_pos, obj_pos, deprel, length, T, prune):
    """Pure-Python closed form of SURVEY.md 3c for ONE sentence (small cases only)."""
    par = [int(head[i]) - 1 for i in range(length)]
    ents = [i for i in range(length) if subj_pos[i] == 0] + [i for i in range(length) if obj_pos[i] == 0]

    def chain(i):
        out = [i]
        while par[out[-1]] >= 0:
            out.append(par[out[-1]])
        return out

    chains = [chain(t) for t in ents]
    ca = set(chains[0]).intersection(*map(set, chains[1:])) if len(chains) > 1 else set(chains[0])
    lca = [c for c in ca if not any(par[d] == c for d in ca)][0]
    on_path = (set().union(*map(set, chains)) - ca) | {lca}
    adj = np.zeros((T, T), dtype=np.float32)
    keep = []
    for i in range(length):
        d, j = 0, i
        while j >= 0 and j not in on_path:
            j, d = par[j], d + 1
        keep.append(j >= 0 and d <= prune)
    for c in range(length):
        if keep[c] and c != lca and par[c] >= 0:
            p = par[c]
            adj[p, c] = deprel[c]
            adj[c, p] = deprel[c] + 42
            adj[p, p] = adj[c, c] = 84
    return adj, np.array(keep, dtype=np.uint8), lca
