"""Data-parallel glue: whole queries shard across ranks (one process per GPU), gradients are
summed with ONE RCCL all-reduce of a flat fp32 bucket per step (torch.distributed backend "nccl"
is RCCL on ROCm; xGMI underneath).  The reference has no distributed code at all (SURVEY.md
section 2.1) — this is the north star's scaling path, not a port.

Normalisation: ListMLE / evidential_ranking average over queries, ListNet
and MSE over candidates, RankNet over pairs.  Each rank weights its local gradient by
local_count / global_count so the reduced gradient equals the single-process one.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def shard_queries(n_queries: int, rank: int, world: int):
    """Contiguous block of whole queries for `rank` (never splits a list)."""
    base, rem = divmod(n_queries, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradBucket:
    """Flat fp32 gradient bucket (3.16 MB at H=300: latency-bound, so exactly one collective)."""

    profile = False          # bench.py: record a HIP-event pair around every collective (class-wide switch)
    events: List = []        # [(start, end)] on the compute stream, filled while `profile` is on

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        dtype = self.params[0].dtype if self.params else torch.float32      # fp32 in production (fp64 in the CPU identity test)
        self.flat = torch.zeros(self.numel, dtype=dtype, device=dev)
        self._views, self._where, off = [], {}, 0
        for p in self.params:
            self._views.append(self.flat[off:off + p.numel()].view_as(p))
            self._where[p.data_ptr()] = (off, p.numel(), tuple(p.shape), p)
            off += p.numel()

    def sink(self, t: torch.Tensor):
        """A FRESH view of the flat buffer for the gradient of the parameter whose storage `t` is (None: not ours)."""
        w = self._where.get(t.data_ptr())
        if w is None or w[1] != t.numel() or t.dtype != self.flat.dtype:
            return None
        if w[3].grad is not None:          # a second backward before the optimizer step accumulates: it needs its own buffer
            return None
        return self.flat[w[0]:w[0] + w[1]].view(w[2])

    _attached: List["GradBucket"] = []

    @staticmethod
    def _lookup(t: torch.Tensor):
        for b in GradBucket._attached:
            v = b.sink(t)
            if v is not None:
                return v
        return None

    def attach(self) -> "GradBucket":
        """Let the explicit backward of reactranker_amd.functions write gradients straight into this bucket."""
        from . import functions as Fn
        if self not in GradBucket._attached:
            GradBucket._attached.append(self)
        Fn.GradSink.lookup = GradBucket._lookup
        return self

    def detach(self) -> None:
        from . import functions as Fn
        if self in GradBucket._attached:
            GradBucket._attached.remove(self)
        if not GradBucket._attached:
            Fn.GradSink.lookup = None

    def allreduce(self, local_weight: float = 1.0, group=None) -> None:
        """grad <- sum_ranks(local_weight_r * grad_r).  With equal shards pass 1/world."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            if local_weight != 1.0:
                for p in self.params:
                    if p.grad is not None:
                        p.grad.mul_(local_weight)
            return
        # gradients born in the bucket (attach()): nothing to pack or unpack
        born = [p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(self.params, self._views)]
        inplace = self in GradBucket._attached and any(born)
        if inplace:
            # Mixed case: a gradient autograd produced outside the explicit backward (an extra head, a cloned or
            # non-contiguous gradient) is copied into ITS view only - a `cat(..., out=flat)` over all of them would read
            # and write the same memory for the ones already living in `flat`.
            for p, v, b in zip(self.params, self._views, born):
                if p.grad is None:                         # a parameter the step did not touch contributes zeros
                    v.zero_()
                    p.grad = v
                elif not b:
                    v.copy_(p.grad)
                    p.grad = v
        else:
            # pack: one batched cat kernel (missing grads contribute zeros), one collective, one batched copy back
            missing = [p for p in self.params if p.grad is None]
            for p in missing:
                p.grad = torch.zeros_like(p)
            torch.cat([p.grad.reshape(-1) for p in self.params], out=self.flat)
        if local_weight != 1.0:
            self.flat.mul_(local_weight)
        if GradBucket.profile and self.flat.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            e1.record()
            GradBucket.events.append((e0, e1))
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if not inplace:
            torch._foreach_copy_([p.grad for p in self.params], list(self._views))


def loss_weight(kind: str, local_queries: int, global_queries: int, local_cands: int, global_cands: int,
                local_pairs: Optional[int] = None, global_pairs: Optional[int] = None) -> float:
    """Weight of this rank's gradient so the reduced gradient equals the single-process one."""
    if kind in ("mle", "evidential_ranking"):
        return local_queries / max(1, global_queries)
    if kind in ("listnet", "mse", "regression", "gauss_regression"):
        return local_cands / max(1, global_cands)
    if kind == "ranknet":
        return (local_pairs or 0) / max(1, global_pairs or 1)
    raise ValueError(kind)
