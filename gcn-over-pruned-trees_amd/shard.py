"""
Data-parallel plumbing for the hot path: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) / xGMI.  The reference has no distributed code at all; this is
the sentence-sharded scheme of SURVEY.md 8(e):

  * sentences never interact inside the path (block-diagonal adjacency, per-sentence degrees), so a
    global batch is split BY SENTENCE and every rank prunes + runs the layer stack on its shard with
    no data-path collective;
  * the only exchange is the parameter gradient: ONE flat fp32 bucket per step, all-reduced
    (messages are 0.45-4.8 MB, i.e. latency-bound on xGMI, so one bucket), followed by the optimizer step -- synchronous SGD, as the
    reference updates every step (train.py:224-227);
  * the word-embedding table is fine-tuned as a whole by default (model/gcn.py:45,84-88; train.py:61) but a batch touches at most
    B*T of its V rows: its gradient is exchanged as (row ids, rows) by all-gather (SparseRowExchange) instead of all-reducing the
    dense [V,300] tensor (~60 MB per step at TACRED's vocabulary against < 1 MB);
  * pooled sentence vectors are all-gathered only when a consumer needs the global batch
    (predict()'s concatenate/unsort, model/trainer.py:121-123).

Everything here is backend-agnostic torch.distributed, so the same code is covered by world_size-2
`gloo` tests on CPU (tests/test_shard_gloo.py).
"""
import time

import numpy as np
import torch


def shard_sentences(lens, world_size, equal_count=True):
    """
    Split sentence indices over ranks, balancing the token count sum(len) (greedy, longest first).
    equal_count keeps |shard| within 1 of each other (the kernels pad every shard to its own max length,
    and the loss is a mean over sentences).  Returns a list of int64 index arrays, each sorted by
    decreasing length (the order the reference's loader uses, data/loader.py:93-94).
    """
    lens = np.asarray(lens, dtype=np.int64)
    n = len(lens)
    order = np.argsort(-lens, kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    count = np.zeros(world_size, dtype=np.int64)
    quota = np.full(world_size, n // world_size, dtype=np.int64)
    quota[: n % world_size] += 1
    shards = [[] for _ in range(world_size)]
    for i in order:
        free = np.nonzero(count < quota)[0] if equal_count else np.arange(world_size)
        r = free[np.argmin(load[free])]
        shards[r].append(int(i))
        load[r] += lens[i]
        count[r] += 1
    return [np.asarray(sorted(s, key=lambda j: (-lens[j], j)), dtype=np.int64) for s in shards]


def take_shard(batch, index):
    """Select the sentences `index` of a loader batch (tuple of [B,...] tensors) and trim the padding."""
    index = torch.as_tensor(index, dtype=torch.int64)
    out = [t.index_select(0, index.to(t.device)) for t in batch]
    masks = out[1]                                   # batch[1] is the pad mask (data/loader.py:140)
    keep = int((~masks.bool()).sum(1).max()) if masks.numel() else 0
    return tuple(t[:, :keep].contiguous() if t.dim() >= 2 and t.shape[1] == masks.shape[1] else t for t in out)


class FlatGradBucket(object):
    """
    All parameter gradients of a module as views into ONE flat fp32 buffer, so that a step needs one
    all-reduce.  weight = this rank's share of the global batch (shard size / global size) when shards
    are unequal; with equal shards use the default 1/world.
    """

    def __init__(self, params, exclude=()):
        """exclude: parameters exchanged some other way (the embedding table of SparseRowExchange)."""
        skip = {id(p) for p in exclude}
        self.params = [p for p in params if p.requires_grad and id(p) not in skip]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros((n,), dtype=torch.float32, device=dev)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        self.attach()

    def flatten_parameters(self):
        """Re-home the parameters' storage into ONE flat fp32 buffer, in the bucket's order (every p.data becomes a view of it; values
        kept).  With it the whole update -- global-norm clip + SGD on all dense parameters -- is one native call on two flat buffers
        (gcnpt_sgd_clip_update, used by sync_sgd_step when the tensors are on a GPU).  Returns the flat parameter tensor."""
        if getattr(self, "flat_params", None) is None:
            flat = torch.empty_like(self.flat)
            o = 0
            with torch.no_grad():
                for p in self.params:
                    if p.dtype != torch.float32:
                        raise TypeError("flatten_parameters: fp32 parameters only")
                    v = flat[o:o + p.numel()].view_as(p)
                    v.copy_(p.data)
                    p.data = v
                    o += p.numel()
            self.flat_params = flat
        return self.flat_params

    def attach(self):
        """(Re-)point every p.grad at its slice of the flat buffer."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero(self):
        """Clear the gradients IN PLACE.  Use this (or zero_grad(set_to_none=False)) between steps: torch's default
        optimizer.zero_grad() / model.zero_grad() sets p.grad = None, which detaches the parameters from the bucket."""
        self.flat.zero_()
        self.attach()

    def all_reduce(self, dist, weight=None, async_op=False):
        """One all-reduce of the flat buffer.  A parameter whose .grad no longer aliases its slice (zero_grad(set_to_none=True)
        followed by backward() gives it a fresh tensor) would be left out silently, so that is checked here: its gradient is
        copied into the slice and the alias restored."""
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()                                   # no gradient this step: contributes zeros
            elif g.data_ptr() != v.data_ptr():
                v.copy_(g)                                  # detached by set_to_none: fold it back in
            p.grad = v
        world = dist.get_world_size()
        self.flat.mul_(weight if weight is not None else 1.0 / world)
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)


class OverlappedAllReduce(object):
    """
    Multi-buffered gradient exchange: the kernels of step i write bucket i % n while the all-reduces of
    the previous steps (other buckets) are still in flight on the communication stream.

        before_write(k)   make sure the pending all-reduce of bucket k is over before the bucket is written again
        after_write(k)    enqueue the all-reduce of bucket k (asynchronous)
        finish()          wait for everything; afterwards every bucket holds the rank-sum / rank-average

    A completion the HOST has observed (Work.is_completed) already orders every later launch after it, so with poll=True
    the host spins until the all-reduce the bucket was last part of is over (n-1 steps ago: the host may run at most that
    far ahead of the device, which keeps the queue fed) and the compute stream gets no wait-for-event marker -- on ROCm such
    a marker breaks the back-to-back dispatch of the step's kernels (measured with RCCL: 6.6 us of device time per step).
    If the completion does not show within `spin_s` seconds the stream wait is inserted after all; poll=False always does.
    """

    def __init__(self, buckets, dist, average=True, poll=True, spin_s=0.05):
        self.buckets, self.dist, self.average, self.poll, self.spin_s = buckets, dist, average, poll, spin_s
        self.pending = [None] * len(buckets)
        self.world = dist.get_world_size()
        self.native_avg = average and dist.get_backend() == "nccl"
        self.stream_waits = 0                # how often the stream had to wait after all (diagnostics)

    def before_write(self, k):
        w = self.pending[k]
        if w is not None:
            done = False
            if self.poll:
                t_end = time.perf_counter() + self.spin_s
                done = w.is_completed()
                while not done and time.perf_counter() < t_end:
                    done = w.is_completed()
            if not done:
                w.wait()
                self.stream_waits += 1
            self.pending[k] = None
            if self.average and not self.native_avg:
                self.buckets[k].div_(self.world)

    def after_write(self, k):
        op = self.dist.ReduceOp.AVG if self.native_avg else self.dist.ReduceOp.SUM
        self.pending[k] = self.dist.all_reduce(self.buckets[k], op=op, async_op=True)

    def finish(self):
        for k in range(len(self.buckets)):
            self.before_write(k)


class SparseRowExchange(object):
    """
    Data-parallel exchange of a ROW-SPARSE gradient -- the word-embedding table's (model/gcn.py:45; fine-tuned as a whole unless
    `topn` says otherwise, gcn.py:84-88): every rank contributes the rows its shard touched, as (row ids, rows), through two
    all-gathers (padded to the largest contribution), and every rank ends up with the same coalesced rank-sum.  Volume per step:
    world x touched x (E + 1) words instead of V x E.

        ids, rows = SparseRowExchange(dist).exchange(idx, g)        # idx int64 [n], g [n, E]: gradient of table[idx]
        table.grad = ex.as_sparse(ids, rows, table.shape)             # for an optimizer that takes sparse gradients (SGD, Adagrad)
        ex.add_into(dense_grad, ids, rows)                            # or: the dense gradient a dense all-reduce would have given

    topn (gcn.py:84-88, torch_utils.keep_partial_grad): rows >= topn get no gradient; padding_idx rows neither (nn.Embedding).
    weight: this rank's share of the global batch (shard size / global size); default: plain sum.
    """

    def __init__(self, dist, topn=None, padding_idx=None):
        self.dist, self.topn, self.padding_idx = dist, topn, padding_idx
        self.last_volume_bytes = 0

    def local_rows(self, idx, g):
        """Coalesce this rank's contribution: unique ids (ascending) and the sum of their gradient rows."""
        idx = idx.reshape(-1)
        g = g.reshape(idx.numel(), -1)
        keep = torch.ones_like(idx, dtype=torch.bool)
        if self.topn is not None:
            keep &= idx < self.topn
        if self.padding_idx is not None:
            keep &= idx != self.padding_idx
        idx, g = idx[keep], g[keep]
        ids, inv = torch.unique(idx, return_inverse=True)
        rows = torch.zeros((ids.numel(), g.shape[1]), dtype=g.dtype, device=g.device)
        rows.index_add_(0, inv, g)
        return ids, rows

    def exchange(self, idx, g, weight=None):
        dist = self.dist
        ids, rows = self.local_rows(idx, g)
        if weight is not None:
            rows = rows * weight
        world = dist.get_world_size()
        n = torch.tensor([ids.numel()], dtype=torch.int64, device=ids.device)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n)
        counts = [int(c.item()) for c in counts]
        width = max(max(counts), 1)
        ids_pad = torch.zeros((width,), dtype=torch.int64, device=ids.device)
        rows_pad = torch.zeros((width, rows.shape[1]), dtype=rows.dtype, device=rows.device)
        ids_pad[: ids.numel()] = ids
        rows_pad[: ids.numel()] = rows
        all_ids = [torch.empty_like(ids_pad) for _ in range(world)]
        all_rows = [torch.empty_like(rows_pad) for _ in range(world)]
        dist.all_gather(all_ids, ids_pad)
        dist.all_gather(all_rows, rows_pad)
        out_ids = torch.unique(torch.cat([a[:c] for a, c in zip(all_ids, counts)]))
        out_rows = torch.zeros((out_ids.numel(), rows.shape[1]), dtype=rows.dtype, device=rows.device)
        # one rank's contribution at a time: its ids are unique, so every add below is collision-free and the ranks are taken in rank
        # order -- every replica computes bit-identical sums (one index_add_ over the concatenation would use float atomics in whatever
        # order the device schedules them, and the replicas' tables would drift apart in the last bit)
        for a_ids, a_rows, c in zip(all_ids, all_rows, counts):
            if c:
                out_rows.index_add_(0, torch.searchsorted(out_ids, a_ids[:c]), a_rows[:c])
        self.last_volume_bytes = world * width * (rows.shape[1] * rows.element_size() + 8)
        return out_ids, out_rows

    @staticmethod
    def as_sparse(ids, rows, shape):
        return torch.sparse_coo_tensor(ids.unsqueeze(0), rows, size=tuple(shape)).coalesce()

    @staticmethod
    def add_into(dense, ids, rows):
        dense.index_add_(0, ids, rows.to(dense.dtype))
        return dense


def sync_sgd_step(dist, bucket, lr, sparse=(), weight=None, max_grad_norm=None, accumulate=1, state=None):
    """One micro-batch of synchronous data-parallel SGD with the reference's update rule (train.py:209, 224-227; plain SGD is its
    default, train.py:82): gradients of `accumulate` micro-batches are SUMMED (update_gap = int(50 / batch_size); the reference does not
    divide), the global L2 norm of the summed gradient -- every parameter, the embedding table included -- is clipped to `max_grad_norm`
    exactly as torch.nn.utils.clip_grad_norm_ does (coefficient max_norm / (norm + 1e-6), capped at 1), then p -= lr * grad.

    bucket: FlatGradBucket holding this micro-batch's gradients of the dense parameters; weight = this rank's share of the micro-batch
    (shard size / global size; default 1 / world).  sparse: [(param, idx, g, SparseRowExchange)] row-sparse gradients (the embedding
    table: g = gradient of param[idx]).  state: a dict the caller keeps across calls (needed when accumulate > 1: it holds the running
    sums).  Returns True when this call applied the update (every `accumulate`-th call); the bucket is cleared for the next
    micro-batch either way.  One all-reduce of the flat bucket and one exchange per sparse parameter per UPDATE, not per micro-batch.
    After an update every rank holds the weights a single process would hold after the same micro-batches on the concatenated
    batches (tests/test_shard_gloo.py).  torch.nn.utils.clip_grad_norm_ itself cannot be used on a model whose embedding gradient is a
    sparse tensor (linalg_vector_norm has no sparse kernel), which is one more reason the clip lives here."""
    world = dist.get_world_size()
    w = weight if weight is not None else 1.0 / world
    if accumulate > 1 and state is None:
        raise ValueError("sync_sgd_step(accumulate > 1) needs a `state` dict kept across calls")
    state = state if state is not None else {}
    for p, v in zip(bucket.params, bucket.views):               # (as FlatGradBucket.all_reduce: fold detached gradients back in)
        g = p.grad
        if g is None:
            v.zero_()
        elif g.data_ptr() != v.data_ptr():
            v.copy_(g)
        p.grad = v
    with torch.no_grad():
        acc = state.get("flat")
        if acc is None:
            acc = state["flat"] = torch.zeros_like(bucket.flat)
        acc.add_(bucket.flat, alpha=w)
        rows_acc = state.setdefault("rows", [[] for _ in sparse])
        for j, (p, idx, g, ex) in enumerate(sparse):
            rows_acc[j].append((idx.reshape(-1), g.reshape(idx.numel(), -1) * w))
        state["n"] = state.get("n", 0) + 1
        bucket.zero()
        if state["n"] < accumulate:
            return False
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        exchanged = []
        for j, (p, idx, g, ex) in enumerate(sparse):
            ids, rows = ex.exchange(torch.cat([a for a, _ in rows_acc[j]]), torch.cat([b for _, b in rows_acc[j]]))
            exchanged.append((p, ids, rows))
        coef = None
        flat_w = getattr(bucket, "flat_params", None)
        if acc.is_cuda and flat_w is not None:
            # the dense parameters' whole update in one native call (two launches): sum of squares, clip coefficient, w -= lr * coef * g
            from . import _lib
            scratch = state.get("partials")
            if scratch is None:
                scratch = state["partials"] = torch.empty((65,), dtype=torch.float32, device=acc.device)
            extra = None
            if max_grad_norm is not None and exchanged:
                extra = sum(rows.float().pow(2).sum() for _, _, rows in exchanged).reshape(1).contiguous()
            _lib.check(_lib.lib().gcnpt_sgd_clip_update(_lib.stream(), _lib.ptr(flat_w), _lib.ptr(acc), acc.numel(), 1.0,
                                                        float(max_grad_norm) if max_grad_norm is not None else 0.0, float(lr),
                                                        _lib.ptr(scratch), _lib.ptr(extra) if extra is not None else None))
            coef = scratch[64] if max_grad_norm is not None else None
        else:
            if max_grad_norm is not None:
                sq = acc.double().pow(2).sum()
                for _, _, rows in exchanged:
                    sq = sq + rows.double().pow(2).sum()
                coef = (float(max_grad_norm) / (sq.sqrt() + 1e-6)).clamp(max=1.0).to(acc.dtype)     # clip_grad_norm_'s coefficient, no host sync
            flat = acc if coef is None else acc * coef
            o = 0
            for p in bucket.params:
                p.add_(flat[o:o + p.numel()].view_as(p), alpha=-lr)
                o += p.numel()
        for p, ids, rows in exchanged:
            p.index_add_(0, ids, (rows if coef is None else rows * coef).to(p.dtype), alpha=-lr)
        acc.zero_()
        state["rows"] = [[] for _ in sparse]
        state["n"] = 0
    return True


def all_gather_pooled(dist, pooled, sizes=None):
    """
    Concatenate per-rank pooled sentence vectors [B_local, H] into [B_global, H] on every rank
    (predict()'s unsort needs the whole batch).  sizes: per-rank B_local when shards are unequal.
    """
    world = dist.get_world_size()
    if sizes is None:
        parts = [torch.empty_like(pooled) for _ in range(world)]
        dist.all_gather(parts, pooled.contiguous())
        return torch.cat(parts, 0)
    width = max(sizes)
    pad = pooled.new_zeros((width,) + tuple(pooled.shape[1:]))
    pad[: pooled.shape[0]] = pooled
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], 0)
