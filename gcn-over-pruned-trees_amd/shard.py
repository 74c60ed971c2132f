"""
Data-parallel plumbing for the hot path: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) / xGMI.  The reference has no distributed code at all; this is
the sentence-sharded scheme of SURVEY.md 8(e):

  * sentences never interact inside the path (block-diagonal adjacency, per-sentence degrees), so a
    global batch is split BY SENTENCE and every rank prunes + runs the layer stack on its shard with
    no data-path collective;
  * the only exchange is the parameter gradient: ONE flat fp32 bucket per step, all-reduced
    (messages are 0.45-4.8 MB, i.e. latency-bound on xGMI, so one bucket and overlap with compute);
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

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros((n,), dtype=torch.float32, device=dev)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        self.attach()

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
