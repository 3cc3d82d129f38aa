"""Data-parallel training of VqaNet over RCCL (one process per GPU, torch.distributed 'nccl').

The reference is single-GPU (main.py:23 pins CUDA_VISIBLE_DEVICES=0); SURVEY.md §8e defines the
sharding: the global minibatch is split by sample, every rank divides its loss by the GLOBAL batch,
and parameter gradients are SUM-all-reduced.  Gradients live in one flat buffer ordered as backward
produces them (classifier, attention, text, image), so each group is one contiguous bucket whose
all-reduce is launched as soon as its kernels are enqueued and overlaps the rest of backward (the
convolution gradients, the longest tail).  xGMI is point-to-point, so few large buckets (4) are
used rather than many small ones.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

GROUPS = ("classifier", "attention", "text", "image")


class DataParallel:
    """Gradient synchroniser attached to a model that exposes flat_buffers() / group_range()."""

    def __init__(self, model, process_group: Optional[dist.ProcessGroup] = None, broadcast: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model = model
        self.pg = process_group
        self.world_size = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self._handles: List = []
        self.issued: List[str] = []             # groups whose collective has been issued, in order (tests, bench)
        # overlap evidence (bench.py --gpus N): with record_events set, every bucket's issue point and the end of
        # backward are marked with HIP events on the streams they happen on; timings() reads them after a sync
        self.record_events = False
        self._marks: List = []
        model._grad_sync = self
        model._seed_rank = self.rank            # different dropout masks per rank
        if broadcast:
            flat_p, _, _ = model.flat_buffers()
            dist.broadcast(flat_p, src=0, group=process_group)

    # called by the backward schedule after the kernels of `group` have been enqueued
    def bucket_ready(self, model, group: str, flat: Optional[torch.Tensor] = None) -> None:
        """Start the SUM all-reduce of one group's contiguous range of `flat` (default: the model's own flat
        gradient buffer).  Every backward of every rank issues exactly these len(GROUPS) collectives, in GROUPS
        order, over the same ranges -- whichever buffer this rank's backward wrote into (its own flat buffer, or a
        fresh one under gradient accumulation / a second pending forward): the collective sequence does not depend
        on rank-local state, so ranks that disagree about that state cannot deadlock or mix up buckets."""
        if flat is None:
            _, flat, _ = model.flat_buffers()
        lo, hi = model.group_range(group)
        if self.record_events and flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(flat.device))        # the stream whose kernels produced the bucket
            self._marks.append((group, ev))
        self._handles.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self.issued.append(group)

    def finish(self, model) -> None:
        mark = bool(self.record_events and self._marks and self._marks[-1][0] not in ("_backward_end", "_reduced"))
        if mark:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()                         # current stream: the last kernel of backward has been enqueued
            self._marks.append(("_backward_end", ev))
        for h in self._handles:
            h.wait()                            # stream-ordered on NCCL: no host block
        if mark:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._marks.append(("_reduced", ev))
        self._handles = []

    def timings(self):
        """After torch.cuda.synchronize(): per backward recorded, ms from each bucket's issue point to the end of that
        backward's kernels (= how much backward work its all-reduce can hide under), and ms the stream then still
        waits for the collectives ('_exposed_ms').  Clears the marks."""
        out, cur = [], {}
        for name, ev in self._marks:
            if name == "_backward_end":
                cur = {"_end": ev, **cur}
            elif name == "_reduced":
                end = cur.pop("_end")
                row = {g: round(e.elapsed_time(end), 3) for g, e in cur.items()}
                row["_exposed_ms"] = round(end.elapsed_time(ev), 3)
                out.append(row)
                cur = {}
            else:
                cur[name] = ev
        self._marks = []
        return out

    def reduce_flat(self, flat: torch.Tensor) -> None:
        """SUM all-reduce of a whole gradient buffer laid out like the model's flat buffer, as the same
        len(GROUPS) bucket collectives in the same order that bucket_ready issues from inside backward (for callers
        that produced gradients outside the backward schedule)."""
        for group in GROUPS:
            self.bucket_ready(self.model, group, flat)
        self.finish(self.model)

    def __call__(self, *args, **kwargs):
        return self.model(*args, **kwargs)

    def __getattr__(self, name):
        return getattr(self.model, name)


def shard_batch(batch_data, rank: int, world_size: int):
    """Split a global 7-tuple (data_preprocessing.py:74-87 layout) evenly by sample."""
    B = batch_data[0].shape[0]
    assert B % world_size == 0, "global batch must divide evenly over ranks"
    per = B // world_size
    sl = slice(rank * per, (rank + 1) * per)
    return tuple(t[sl] for t in batch_data)
