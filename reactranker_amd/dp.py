"""Data-parallel glue: whole queries shard across ranks (one process per GPU), gradients are
summed with ONE RCCL all-reduce of a flat fp32 bucket per step (torch.distributed backend "nccl"
is RCCL on ROCm; xGMI underneath).  The reference has no distributed code at all (SURVEY.md
section 2.1) — this is the north star's scaling path, not a port.

Normalisation: ListMLE / evidential_ranking average over queries, ListNet
and MSE over candidates, RankNet over pairs.  Each rank weights its local gradient by
local_count / global_count so the reduced gradient equals the single-process one.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Tuple

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
    algo = None              # None -> $RR_ALLREDUCE or "allreduce"; "rsag" = reduce-scatter + all-gather (see _collective)
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

    def _collective(self, group) -> None:
        """flat <- sum over ranks.  Default: ONE all-reduce (latency-bound at 3-12 MB: SURVEY.md section 8e).  GradBucket.algo =
        "rsag" (or RR_ALLREDUCE=rsag in the environment) issues the same sum as reduce-scatter + all-gather over the largest
        prefix that divides by the world size (+ a tiny all-reduce for the < world remaining elements): on xGMI's point-to-point
        links every rank then owns 1/R of the reduction (rr_allreduce_rsag_f32 is the C-ABI twin).  Which
        is faster is a measurement for an 8-GPU node; with 2 ranks the two give the same bits.  Backends without
        reduce_scatter_tensor (gloo) keep the all-reduce."""
        algo = GradBucket.algo or os.environ.get("RR_ALLREDUCE", "allreduce")
        world = dist.get_world_size(group)
        if algo == "rsag" and self.flat.is_cuda and dist.get_backend(group) == "nccl" and self.flat.numel() >= world:
            chunk = self.flat.numel() // world
            body = self.flat[:chunk * world]
            mine = body[dist.get_rank(group) * chunk:(dist.get_rank(group) + 1) * chunk]
            dist.reduce_scatter_tensor(mine, body, op=dist.ReduceOp.SUM, group=group)
            dist.all_gather_into_tensor(body, mine, group=group)
            if chunk * world < self.flat.numel():
                dist.all_reduce(self.flat[chunk * world:], op=dist.ReduceOp.SUM, group=group)
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)

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
            self._collective(group)
            e1.record()
            GradBucket.events.append((e0, e1))
        else:
            self._collective(group)
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


# ------------------------------------------------------------------------------------------------ trainer-side glue
def count_pairs(scope, targets) -> int:
    """Ordered pairs (i, j), i != j, of one window with different targets inside the same query - `num_pairs = 2 *
    num_pos_pairs` of the reference (train_pairwise.py:99-106: positive pairs `rel_diff > 0` and as many negative ones),
    the normaliser of the RankNet loss (:147) - counted on the host from the targets the batch carries, so the trainer
    never has to read the device-side count back.  64 distinct targets give 64 * 63 = 4032."""
    import numpy as np
    t = np.asarray(targets.detach().cpu() if torch.is_tensor(targets) else targets, dtype=np.float64).reshape(-1)
    n, off = 0, 0
    for c in scope:
        q = t[off:off + c]
        n += 2 * int((q[:, None] > q[None, :]).sum())
        off += c
    return n


def step_counts(scope, targets) -> dict:
    """The three normalisers a step can be averaged over: queries, candidates, ordered pairs."""
    scope = [int(c) for c in scope]
    return dict(queries=len(scope), cands=int(sum(scope)), pairs=count_pairs(scope, targets))


def shard_query_batch(qb, rank: int, world: int):
    """This rank's contiguous block of whole queries of one global step (a synth.QueryBatch or anything with the same
    fields) + the counts of the WHOLE step, which weight the rank's gradient.  A rank can come out empty (fewer queries
    than ranks): it still joins the step's all-reduce with a zero gradient."""
    import copy
    lo, hi = shard_queries(len(qb.scope), rank, world)
    m0, m1 = int(sum(qb.scope[:lo])), int(sum(qb.scope[:hi]))
    local = copy.copy(qb)
    local.r_specs, local.p_specs = qb.r_specs[m0:m1], qb.p_specs[m0:m1]
    local.scope, local.targets = list(qb.scope[lo:hi]), qb.targets[m0:m1]
    local.add_features = None if qb.add_features is None else qb.add_features[m0:m1]
    return local, step_counts(qb.scope, qb.targets)


_KIND = {"mle": "mle", "evidential_ranking": "mle", "listnet": "listnet", "regression": "listnet", "gauss_regression": "listnet",
         "mse": "listnet", "ranknet": "ranknet"}


class Exchange:
    """What a trainer needs from the process group, and nothing when there is none: the per-step gradient all-reduce
    with the loss's weight, sums of validation statistics, and who writes checkpoints.  `group=None` with an initialised
    torch.distributed means the default group; without torch.distributed every method is the identity, so the trainers
    run ONE code path for 1 and N processes.

    A batch handed to a data-parallel trainer is this rank's shard of a global step.  It carries `global` = the counts
    of the whole step (dp.step_counts / dp.shard_query_batch) so that no collective is needed to weight the gradient;
    without it the counts are all-reduced per step (one extra tiny collective and a host read)."""

    def __init__(self, model: Optional[torch.nn.Module] = None, group=None):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.bucket = None
        if self.on and model is not None:
            self.bucket = GradBucket(model.parameters()).attach()

    @property
    def is_writer(self) -> bool:
        return self.rank == 0

    def close(self) -> None:
        if self.bucket is not None:
            self.bucket.detach()
            self.bucket = None

    def counts(self, batch, device) -> Tuple[dict, dict]:
        """(local counts, global counts) of a step."""
        local = step_counts(batch["scope"], batch["targets"]) if len(batch["scope"]) else dict(queries=0, cands=0, pairs=0)
        if not self.on:
            return local, local
        g = batch.get("global")
        if g is None:
            v = torch.tensor([local["queries"], local["cands"], local["pairs"]], dtype=torch.float64, device=device)
            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
            q, m, p = (int(x) for x in v.tolist())
            g = dict(queries=q, cands=m, pairs=p)
        return local, g

    def weight(self, kind: str, local: dict, glob: dict) -> float:
        return loss_weight(_KIND[kind], local["queries"], glob["queries"], local["cands"], glob["cands"],
                           local["pairs"], glob["pairs"])

    def reduce_grads(self, weight: float) -> None:
        """p.grad <- sum over ranks of weight_r * p.grad_r, for every parameter of the model (missing gradients - an
        empty shard, a window without ordered pairs - count as zeros)."""
        if self.on:
            if self.bucket is None:
                raise RuntimeError("dp.Exchange.reduce_grads: this Exchange was built without a model (Exchange(None, group)), so it "
                                   "owns no gradient bucket; construct it as Exchange(model, group) in a data-parallel job")
            self.bucket.allreduce(weight, group=self.group)

    def sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.on:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def check_same_steps(self, n_steps: int, device) -> None:
        """Every rank must take the same number of optimizer steps (each is one all-reduce): a rank that is handed fewer
        batches would leave the others waiting in the collective for ever.  Checked up front where the count is known."""
        if not self.on:
            return
        v = torch.tensor([float(n_steps), -float(n_steps)], dtype=torch.float64, device=device)
        dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self.group)
        lo, hi = -float(v[1]), float(v[0])
        if lo != hi:
            raise RuntimeError(f"data-parallel training: ranks hold different numbers of steps ({int(lo)} .. {int(hi)}); shard every "
                               "global step over all ranks (reactranker_amd.dp.shard_query_batch), empty shards included")

    def broadcast_model(self, model: torch.nn.Module) -> None:
        """Identical replicas: rank 0's parameters and buffers everywhere (what DistributedDataParallel does at wrap time)."""
        if self.on:
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
