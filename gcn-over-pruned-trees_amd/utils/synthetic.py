"""
Synthetic TACRED-shaped batches for parity tests and bench.py (no dataset ships with the build).

Generator specification: SURVEY.md 8(d) "Synthetic inputs".  The tensors have exactly the layout
the reference loader hands to the model (data/loader.py:93-141): every field int64 [B,T],
sentences sorted by decreasing length with lens[0] == T, pads = 0 for head/deprel and 150 for the
two position fields (loader.py:120-121), `masks` True on pads (loader.py:110).

Only numpy's frozen legacy RandomState is used, so a seed reproduces the same batch everywhere.
"""
import numpy as np

POS_PAD = 150          # data/loader.py:120-121 fill value for subj/obj positions
DEPREL_LO, DEPREL_HI = 2, 41   # real dependency labels (utils/constant.py:29): never PAD(0)/UNK(1)


def positions(start, end, length):
    """Relative offsets to an entity span, 0 inside it (data/loader.py:162-165)."""
    return list(range(-start, 0)) + [0] * (end - start + 1) + list(range(1, length - end))


def tacred_lengths(rng, B, T, lo=8, mu=3.45, sigma=0.45):
    """len ~ clip(round(lognormal), lo, T), sorted descending, lens[0] = T."""
    lens = np.clip(np.rint(rng.lognormal(mu, sigma, size=B)), min(lo, T), T).astype(np.int32)
    lens = np.sort(lens)[::-1].copy()
    lens[0] = T
    return lens


def random_tree_batch(seed, B, T, lengths="full", overlap_frac=0.0):
    """
    Uniform random recursive dependency trees with one subject and one object span per sentence.

    lengths: "full" (every sentence has T tokens), "tacred" (log-normal, mean ~35) or an int array.
    overlap_frac: fraction of sentences whose object span is drawn without the disjointness
                  constraint (nested / overlapping entities).
    Returns dict of numpy arrays: head, deprel, subj_pos, obj_pos int64 [B,T]; lens int32 [B];
    masks bool [B,T] (True = pad); subj_span, obj_span int32 [B,2] (inclusive).
    """
    rng = np.random.RandomState(seed)
    if isinstance(lengths, str):
        lens = np.full((B,), T, dtype=np.int32) if lengths == "full" else tacred_lengths(rng, B, T)
    else:
        lens = np.asarray(lengths, dtype=np.int32)
        assert lens.shape == (B,) and lens.max() == T
    head = np.zeros((B, T), dtype=np.int64)
    deprel = np.zeros((B, T), dtype=np.int64)
    subj_pos = np.full((B, T), POS_PAD, dtype=np.int64)
    obj_pos = np.full((B, T), POS_PAD, dtype=np.int64)
    subj_span = np.zeros((B, 2), dtype=np.int32)
    obj_span = np.zeros((B, 2), dtype=np.int32)
    for b in range(B):
        n = int(lens[b])
        order = rng.permutation(n)
        head[b, order[0]] = 0                               # root
        for k in range(1, n):
            head[b, order[k]] = order[rng.randint(0, k)] + 1   # 1-based parent among placed tokens
        deprel[b, :n] = rng.randint(DEPREL_LO, DEPREL_HI + 1, size=n)
        sl = min(int(rng.randint(1, 4)), n)
        ol = min(int(rng.randint(1, 3)), n)
        ss = int(rng.randint(0, n - sl + 1))
        overlap = rng.random_sample() < overlap_frac
        for _try in range(64):                              # bounded: tiny sentences may not fit both
            os_ = int(rng.randint(0, n - ol + 1))
            disjoint = os_ + ol - 1 < ss or os_ > ss + sl - 1
            if overlap or disjoint:
                break
        subj_pos[b, :n] = positions(ss, ss + sl - 1, n)
        obj_pos[b, :n] = positions(os_, os_ + ol - 1, n)
        subj_span[b] = (ss, ss + sl - 1)
        obj_span[b] = (os_, os_ + ol - 1)
    masks = np.arange(T)[None, :] >= lens[:, None]
    return dict(head=head, deprel=deprel, subj_pos=subj_pos, obj_pos=obj_pos, lens=lens, masks=masks,
                subj_span=subj_span, obj_span=obj_span)


def layer_params(seed, dims):
    """nn.Linear-style init U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for W [H,Din] and b [H] per layer."""
    rng = np.random.RandomState(seed)
    Ws, bs = [], []
    for l in range(len(dims) - 1):
        din, h = dims[l], dims[l + 1]
        k = 1.0 / np.sqrt(din)
        Ws.append(rng.uniform(-k, k, size=(h, din)).astype(np.float32))
        bs.append(rng.uniform(-k, k, size=(h,)).astype(np.float32))
    return Ws, bs


def normal(seed, shape):
    return np.random.RandomState(seed).standard_normal(shape).astype(np.float32)
