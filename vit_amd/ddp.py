"""Data parallelism for the encoder: one process per GPU, replicated parameters, bucketed gradient all-reduce on
RCCL (torch.distributed backend "nccl" on ROCm) launched from post-accumulate-grad hooks so that communication of
the last layers' gradients overlaps the backward pass of the earlier layers.

The reference has no distributed code; this is the batch-sharded training path of the hot loop.  Every op of the
encoder is independent across the batch (attention is per sample, norms per token, drop-path permutes within the
local batch), so the only exchange per step is the gradient average.

xGMI is point-to-point (7 links x ~153 GB/s per GPU); a ring all-reduce is bound by one link, so buckets are sized
large enough (default 64 MiB) to be bandwidth- rather than latency-bound while still leaving several buckets per
step to overlap with compute."""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("params", "offsets", "flat", "pending", "work")

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64  # keep every slice 256-byte aligned
        self.flat = torch.empty(n, dtype=params[0].dtype, device=params[0].device)
        self.pending = len(params)
        self.work = None


class BucketedGradAllReduce:
    """Averages ``p.grad`` of every trainable parameter over the process group, bucket by bucket.

    Buckets are filled in reverse registration order (the order gradients become ready in backward).  When the last
    gradient of a bucket has been accumulated the bucket's flat buffer is all-reduced asynchronously; ``finish()``
    waits for all buckets and leaves ``p.grad`` pointing at the averaged values (views into the flat buffers).
    Parameters that did not receive a gradient in a step are reduced as zeros in ``finish()``."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group: Optional[dist.ProcessGroup] = None,
                 bucket_bytes: int = 64 << 20):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        plist = [p for p in params if p.requires_grad and p.numel() > 0]
        self.buckets: List[_Bucket] = []
        cur: List[torch.nn.Parameter] = []
        size = 0
        for p in reversed(plist):
            if cur and (size + p.numel() * p.element_size() > bucket_bytes or p.dtype != cur[0].dtype):
                self.buckets.append(_Bucket(cur))
                cur, size = [], 0
            cur.append(p)
            size += p.numel() * p.element_size()
        if cur:
            self.buckets.append(_Bucket(cur))
        self._where = {}
        self._handles = []
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        if self.world > 1 and backend == "nccl":
            # the all-reduce kernels will hold CUs beside the backward GEMMs: let those draw their tiles dynamically
            from . import _lib as L

            L.check(L.load().vitk_set_option(b"dynamic_tiles", 1))
        self._avg = dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM
        self._div = backend != "nccl"

    def _view(self, b: _Bucket, i: int) -> torch.Tensor:
        p = b.params[i]
        return b.flat[b.offsets[i]: b.offsets[i] + p.numel()].view_as(p)

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        b, i = self._where[p]
        v = self._view(b, i)
        if p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)
            p.grad = v
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: _Bucket) -> None:
        if self.world > 1:
            b.work = dist.all_reduce(b.flat, op=self._avg, group=self.group, async_op=True)

    def finish(self) -> None:
        """Call after backward: reduce stragglers, wait for every bucket, re-arm for the next step."""
        for b in self.buckets:
            if b.pending > 0:  # some parameter got no gradient this step: contribute zeros
                for i, p in enumerate(b.params):
                    if p.grad is None or p.grad.data_ptr() != self._view(b, i).data_ptr():
                        v = self._view(b, i)
                        if p.grad is None:
                            v.zero_()
                        else:
                            v.copy_(p.grad)
                        p.grad = v
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
                if self._div:
                    b.flat.div_(self.world)
            b.pending = len(b.params)

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []
