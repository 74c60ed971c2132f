"""
Host half of loader-side pre-pruning (reference data/loader.py:81-141, model/trainer.py:52-73).

The reference's DataLoader pads every batch on the host, sorts it by length and the trainer uploads eight tensors with
blocking `.cuda()` calls.  Here:
  * `length_buckets`  groups sentence numbers of similar length into batches (less padding: T of a batch is its longest
                      sentence) and keeps the reference's in-batch order (longest first, loader.py:93-94);
  * `PinnedStager`    keeps two sets of pinned host buffers and uploads a batch with non-blocking copies on a side stream,
                      so the upload of batch i+1 overlaps the step of batch i; the device pruner / TreeCache (model.tree)
                      then works on tensors that are already in HBM.
Nothing here touches the dataset format: inputs are the integer arrays the reference's loader already produces.
"""
import numpy as np
import torch


def length_buckets(lens, batch_size, shuffle_seed=None):
    """
    lens: int array [S].  Returns a list of int64 index arrays, each a batch of <= batch_size sentences of neighbouring
    lengths, longest sentence first inside the batch.  shuffle_seed: shuffle the ORDER of the batches (not their content).
    """
    lens = np.asarray(lens)
    order = np.argsort(-lens, kind="stable")
    batches = [order[i:i + batch_size].astype(np.int64) for i in range(0, len(order), batch_size)]
    if shuffle_seed is not None:
        np.random.RandomState(shuffle_seed).shuffle(batches)
    return batches


def padding_waste(lens, batches):
    """Fraction of the padded token slots that are padding, for a list of batches (index arrays)."""
    lens = np.asarray(lens)
    slots = sum(int(lens[b].max()) * len(b) for b in batches)
    return 1.0 - float(lens.sum()) / max(slots, 1)


class PinnedStager(object):
    """
    Double-buffered pinned staging of a batch's loader tensors.  fields: {name: (dtype, pad_value)} of the per-token
    arrays (e.g. words, pos, ner, deprel, head, subj_pos, obj_pos); capacity: (max batch, max T).
    """

    def __init__(self, fields, max_batch, max_T, device):
        self.fields, self.device = dict(fields), torch.device(device)
        # FLAT pinned buffers: a batch's [B, T] staging area is the first B*T elements, i.e. contiguous whatever T is -- a
        # [:B, :T] corner of a [max_batch, max_T] buffer is not, and a non-contiguous pinned source makes `.to(non_blocking=True)`
        # go through a pageable temporary (the copy then blocks the host and nothing overlaps)
        n = int(max_batch) * int(max_T)
        self.host = [{k: torch.empty((n,), dtype=dt).pin_memory() for k, (dt, _) in self.fields.items()} for _ in range(2)]
        self.host_mask = [torch.empty((n,), dtype=torch.bool).pin_memory() for _ in range(2)]
        self.stream = torch.cuda.Stream(device=self.device)
        self.done = [torch.cuda.Event(), torch.cuda.Event()]
        self.turn = 0

    def upload(self, dataset, idx):
        """dataset: {name: int array [S, Ts]} padded to the dataset's longest sentence, plus 'lens' [S]; idx: the batch.
        Returns ({name: CUDA tensor [B, T]}, masks bool [B, T] CUDA, event); wait for `event` (or call .ready()) before use.
        The tensors are allocated on the upload stream; they are marked as used by the stream that is current when upload() is
        called (Tensor.record_stream), so the caching allocator does not hand their memory to a later upload while kernels of that
        stream may still read them.  A consumer on yet another stream passes the tensors to ready()."""
        idx = np.asarray(idx)
        lens = np.asarray(dataset["lens"])[idx]
        B, T = len(idx), int(lens.max())
        slot = self.host[self.turn]
        self.done[self.turn].synchronize()                           # the copy that last used these buffers has finished
        consumer = torch.cuda.current_stream(self.device)
        pad_mask = torch.from_numpy(np.arange(T)[None, :] >= lens[:, None])
        out = {}
        with torch.cuda.stream(self.stream):
            for k, (dt, pad) in self.fields.items():
                h = slot[k][:B * T].view(B, T)
                h.copy_(torch.from_numpy(np.ascontiguousarray(dataset[k][idx, :T])))
                if pad != 0:
                    h[pad_mask] = pad
                out[k] = h.to(self.device, non_blocking=True)
                out[k].record_stream(consumer)
            m = self.host_mask[self.turn][:B * T].view(B, T)
            m.copy_(pad_mask)
            masks = m.to(self.device, non_blocking=True)
            masks.record_stream(consumer)
            self.done[self.turn].record(self.stream)
        ev = self.done[self.turn]
        self.turn ^= 1
        return out, masks, ev

    @staticmethod
    def ready(event, tensors=()):
        """Make the current stream wait for an upload (no host sync).  tensors: the uploaded tensors, when the current stream is
        not the one that was current at upload() -- they are then marked as used by this stream too."""
        cur = torch.cuda.current_stream()
        cur.wait_event(event)
        for t in (tensors.values() if isinstance(tensors, dict) else tensors):
            t.record_stream(cur)
