"""Data parallelism for the encoder: one process per GPU, replicated parameters, bucketed gradient all-reduce on
RCCL (torch.distributed backend "nccl" on ROCm) launched from post-accumulate-grad hooks so that communication of
the last layers' gradients overlaps the backward pass of the earlier layers.

The reference has no distributed code; this is the batch-sharded training path of the hot loop.  Every op of the
encoder is independent across the batch (attention is per sample, norms per token, drop-path permutes within the
local batch), so the only exchange per step is the gradient average.

xGMI is point-to-point (7 links x ~153 GB/s per GPU); a ring all-reduce is bound by one link, so buckets are sized
large enough (default 64 MiB) to be bandwidth- rather than latency-bound while still leaving several buckets per
step to overlap with compute.

Weight gradients can be produced straight into the bucket: while a reducer is attached, ``grad_sink(param)`` hands
the backward kernels the bucket's slice for that parameter, autograd adopts it as ``param.grad`` without a copy, and
the hook finds the gradient already in place."""
from __future__ import annotations

import contextlib
import time
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

# parameter storage address -> (reducer, bucket, index); consulted by the backward kernels' host code (vit_amd/ops.py)
_SINKS: Dict[int, tuple] = {}


def grad_sink(param_ptr: int, shape, dtype: torch.dtype) -> Optional[torch.Tensor]:
    """The bucket slice that the gradient of the parameter stored at ``param_ptr`` is reduced from, if a reducer is
    attached to it, the slice has not been handed out yet in this step and gradients are being synchronised; else None
    (the caller allocates)."""
    entry = _SINKS.get(param_ptr)
    if entry is None:
        return None
    red, b, i = entry
    if not red._sync or b.issued or b.given[i]:
        return None
    p = b.params[i]
    if tuple(p.shape) != tuple(shape) or p.dtype != dtype:
        return None
    # A gradient that is already there (zero_grad(set_to_none=False), or micro-batches accumulated under no_sync())
    # makes autograd ADD the new one to it.  If the kernel wrote into the slice, that would either add the slice to
    # itself (p.grad is the slice from the last step: the gradient doubles) or overwrite what was accumulated, so the
    # slice is only handed out when autograd will adopt it as a fresh .grad.
    if p.grad is not None:
        return None
    b.given[i] = True
    return red._view(b, i)


class _Bucket:
    __slots__ = ("params", "offsets", "flat", "arrived", "given", "work", "issued", "host")

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64  # keep every slice 256-byte aligned
        self.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        self.arrived = [False] * len(params)
        self.given = [False] * len(params)
        self.work = None
        self.issued = False
        self.host = None   # CPU copy of `flat` while a host-side backend reduces it


class BucketedGradAllReduce:
    """Averages ``p.grad`` of every trainable parameter over the process group, bucket by bucket.

    Buckets are filled in reverse registration order (the order gradients become ready in backward).  A bucket is
    all-reduced asynchronously once all its gradients have arrived AND every earlier bucket has been issued, so every
    rank issues the collectives in the same order (bucket 0, 1, 2, ...) whatever order its gradients arrive in;
    ``finish()`` issues what is left (parameters without a gradient contribute zeros), waits for all buckets and
    leaves ``p.grad`` pointing at the averaged values (views into the flat buffers).

    One backward pass per ``finish()``: a second gradient for a parameter before ``finish()`` raises, because it would
    be accumulated in place into a buffer whose all-reduce may be in flight.  For gradient accumulation run the
    earlier micro-batches under ``no_sync()`` (hooks do nothing, gradients accumulate in ``p.grad`` as usual) and the
    last one outside it."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group: Optional[dist.ProcessGroup] = None,
                 bucket_bytes: int = 64 << 20, always_reduce: bool = False):
        """``always_reduce``: issue the collectives even in a group of one rank (a rehearsal of the multi-rank path on a
        single GPU: the same launches, side stream and tile scheduling, averaging over one rank)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._reduce = self.world > 1 or (always_reduce and dist.is_initialized())
        plist = [p for p in params if p.requires_grad and p.numel() > 0]
        self.buckets: List[_Bucket] = []
        cur: List[torch.nn.Parameter] = []
        size = 0
        # The gradients that arrive last (the first layers) have the least backward left to hide their all-reduce behind:
        # the last `bucket_bytes` of the walk go out in buckets a quarter the size, so that only the final small bucket
        # is queued at the very end of backward instead of a full one.
        left = sum(p.numel() * p.element_size() for p in plist)
        for p in reversed(plist):
            nbytes = p.numel() * p.element_size()
            cap = bucket_bytes if left > bucket_bytes else max(bucket_bytes // 4, min(bucket_bytes, 1 << 20))
            if cur and (size + nbytes > cap or p.dtype != cur[0].dtype):
                self.buckets.append(_Bucket(cur))
                cur, size = [], 0
            cur.append(p)
            size += nbytes
            left -= nbytes
        if cur:
            self.buckets.append(_Bucket(cur))
        self._where = {}
        self._handles = []
        self._sync = True
        self._next = 0   # index of the first bucket not yet issued
        self.timeline = None   # a list while tracing: (bucket index, event on the compute stream at issue, host time)
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
                _SINKS[p.data_ptr()] = (self, b, i)
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        # the all-reduce kernels hold CUs beside the backward GEMMs: those then draw their tiles dynamically
        self.dynamic_tiles = self._reduce and backend == "nccl"
        if self.dynamic_tiles:
            from . import kernels as K

            self._old_dynamic = K.set_dynamic_tiles(True)
        self._avg = dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM
        self._div = backend != "nccl"
        # A backend that reduces in host memory (gloo: tests and debugging, never the RCCL path) gets a host copy of the
        # bucket made here, on the calling thread and the current stream, instead of the copies c10d would queue on
        # side streams from its own threads.
        self._host_staged = backend == "gloo" and any(b.flat.is_cuda for b in self.buckets)

    def _view(self, b: _Bucket, i: int) -> torch.Tensor:
        p = b.params[i]
        return b.flat[b.offsets[i]: b.offsets[i] + p.numel()].view_as(p)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside this context leave ``p.grad`` alone (no copy, no all-reduce)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if not self._sync:
            return
        b, i = self._where[p]
        if b.issued or b.arrived[i]:
            raise RuntimeError("BucketedGradAllReduce: a parameter received a second gradient before finish(); run "
                               "accumulation micro-batches under no_sync() and call finish() after every synchronised "
                               "backward pass")
        v = self._view(b, i)
        if p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)
            p.grad = v
        b.arrived[i] = True
        self._issue_ready()

    def _issue_ready(self) -> None:
        while self._next < len(self.buckets) and all(self.buckets[self._next].arrived):
            self._launch(self.buckets[self._next])
            self._next += 1

    def _launch(self, b: _Bucket) -> None:
        b.issued = True
        if self.timeline is not None and b.flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()   # where the backward pass stands when this bucket's collective is queued
            self.timeline.append((self.buckets.index(b), ev, time.perf_counter()))
        if self._reduce:
            if self._host_staged:
                b.host = b.flat.cpu()
                b.work = dist.all_reduce(b.host, op=self._avg, group=self.group, async_op=True)
            else:
                b.work = dist.all_reduce(b.flat, op=self._avg, group=self.group, async_op=True)

    def finish(self) -> None:
        """Call after backward: issue the remaining buckets in order, wait for every bucket, re-arm for the next step."""
        for b in self.buckets[self._next:]:
            for i, p in enumerate(b.params):
                if b.arrived[i]:
                    continue
                v = self._view(b, i)
                if p.grad is None:
                    v.zero_()    # no gradient this step: contribute zeros
                elif p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                p.grad = v
                b.arrived[i] = True
            self._launch(b)
        self._next = len(self.buckets)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
                if b.host is not None:
                    b.flat.copy_(b.host)
                    b.host = None
                if self._div:
                    b.flat.div_(self.world)
            b.arrived = [False] * len(b.params)
            b.given = [False] * len(b.params)
            b.issued = False
        self._next = 0

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []
        for b in self.buckets:
            for p in b.params:
                if _SINKS.get(p.data_ptr(), (None,))[0] is self:
                    del _SINKS[p.data_ptr()]
        if self.dynamic_tiles:
            from . import kernels as K

            K.set_dynamic_tiles(self._old_dynamic)
            self.dynamic_tiles = False
