"""Ranking / pointwise losses — mirror of reference `reactranker/train/loss.py` for the losses on
the hot path (MLEloss :64-99, ListnetLoss :317-352, evidential_ranking :477-556, GaussDisLoss
:144-162, LogCumsumExp :9-61) plus RankNet's inline loss (train/train_pairwise.py:99-137).

Same call signatures `loss(score, scope, targets, gpu)` and return shapes ([1] for ListMLE and
evidential_ranking, 0-d for ListNet).  Each loss is one fused HIP kernel per direction (one
wavefront per query) instead of a Python loop of ~10 ATen ops per query.
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, lib, ptr, stream


@lru_cache(maxsize=256)
def _segments(scope: tuple, device_str: str):
    off = np.zeros(len(scope) + 1, np.int32)
    np.cumsum(np.asarray(scope, np.int64), out=off[1:])
    t = torch.from_numpy(off).to(device_str)
    return t, int(off[-1]), (max(scope) if scope else 0)


def _prep(score: torch.Tensor, scope, targets, gpu):
    if gpu is not None:
        torch.cuda.set_device(gpu)                      # reference loss.py:83
    _lib.require_cuda(score, "score")
    scope = tuple(int(c) for c in (scope.tolist() if hasattr(scope, "tolist") else scope))
    seg, total, max_len = _segments(scope, str(score.device))
    if total != score.shape[0]:
        raise RuntimeError(f"sum(scope) = {total} but score has {score.shape[0]} rows")
    t = torch.as_tensor(targets, dtype=torch.float32)
    if t.device != score.device:
        t = t.to(score.device)                          # reference loss.py:85
    t = t.reshape(-1).contiguous()
    if t.numel() != total:
        raise RuntimeError("targets and score lengths differ")
    return scope, seg, total, max_len, t


def _vec(x: torch.Tensor) -> torch.Tensor:
    """1-D float32 view with arbitrary element stride (a column of the [M,2] model output is fine)."""
    if x.dtype != torch.float32:
        x = x.float()
    if x.dim() != 1:
        x = x.reshape(-1)
    return x


def _f1(dev):
    return torch.empty(1, dtype=torch.float32, device=dev)


class FusedStep:
    """Loss and d loss / d score in ONE launch (rr_listmle_step_f32 and its ListNet / evidential twins, csrc/loss.hip).

    The reference's trainer step is `loss = loss_func(score, ...); loss.backward()` (train/train_listwise.py:287-288): the loss
    is the root of the graph and its upstream gradient is the constant one.  When the score requires a gradient the forward
    therefore also writes d loss / d score for that case - same operations in the same order as the backward kernel with an
    upstream gradient of 1.0f, the bits are the same - and `reactranker_amd.loss.backward(loss)` starts the backward with the
    library's cached constant-one tensor: the loss's backward then recognises it (by address) and hands out the gradient
    that already exists.  What a step no longer launches: the separate reduction of the per-query partials, autograd's
    ones_like(loss) fill, a `.sum()` over one element, the loss's backward kernel.  Any other upstream gradient (a plain
    `loss.backward()`, a scaled loss) takes the backward kernel as before - nothing depends on the fast path being hit.
    hits counts the backward calls it served (tests)."""
    enabled = True
    hits = 0
    _unit = {}
    _counter = {}


def _unit_like(t: torch.Tensor) -> torch.Tensor:
    key = (str(t.device), tuple(t.shape))
    u = FusedStep._unit.get(key)
    if u is None:
        u = FusedStep._unit[key] = torch.ones(tuple(t.shape), dtype=torch.float32, device=t.device)
    return u


def _is_unit(g: torch.Tensor) -> bool:
    u = FusedStep._unit.get((str(g.device), tuple(g.shape)))
    return u is not None and g.dtype == torch.float32 and g.data_ptr() == u.data_ptr()


def _counter(dev) -> torch.Tensor:
    """the zero-initialised ticket word of the step kernels: one per (device, stream) - launches of one stream are ordered,
    and every launch leaves it at zero"""
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    c = FusedStep._counter.get(key)
    if c is None:
        c = FusedStep._counter[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return c


def backward(loss: torch.Tensor) -> None:
    """`loss.backward()` for a loss that is the root of the graph (reference train/train_listwise.py:288), started with the
    library's constant-one gradient instead of a freshly filled one: a loss of this module then returns the gradient its
    forward launch already wrote (FusedStep).  Works for any other one-element float32 loss as well (plain autograd)."""
    if loss.numel() == 1 and loss.dtype == torch.float32 and loss.is_cuda:
        loss.backward(gradient=_unit_like(loss))
    else:
        loss.backward()


class _ListMLEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score, targets, seg, Q, max_len):
        s = _vec(score.detach())
        loss, part = _f1(s.device), torch.empty(max(Q, 1), dtype=torch.float32, device=s.device)
        ctx.ds_unit = None
        if FusedStep.enabled and ctx.needs_input_grad[0]:
            ds = torch.empty(s.shape[0], dtype=torch.float32, device=s.device)
            check(lib().rr_listmle_step_f32(ptr(s), s.stride(0), ptr(targets), ptr(seg), Q, max_len, ptr(loss), ptr(part),
                                            ptr(_counter(s.device)), ptr(ds), 1, stream()), "rr_listmle_step_f32")
            ctx.ds_unit = ds
        else:
            check(lib().rr_listmle_fwd_f32(ptr(s), s.stride(0), ptr(targets), ptr(seg), Q, max_len, ptr(loss), ptr(part),
                                           stream()), "rr_listmle_fwd_f32")
        ctx.save_for_backward(s, targets, seg)
        ctx.meta = (Q, max_len)
        return loss

    @staticmethod
    def backward(ctx, g):
        if ctx.ds_unit is not None and _is_unit(g):      # the gradient the forward launch already wrote (handed out once)
            ds, ctx.ds_unit = ctx.ds_unit, None
            FusedStep.hits += 1
            return ds, None, None, None, None
        s, targets, seg = ctx.saved_tensors
        Q, max_len = ctx.meta
        g = g.reshape(-1).contiguous().float()
        ds = torch.empty(s.shape[0], dtype=torch.float32, device=s.device)
        check(lib().rr_listmle_bwd_f32(ptr(s), s.stride(0), ptr(targets), ptr(seg), Q, max_len, ptr(g), ptr(ds), 1,
                                       stream()), "rr_listmle_bwd_f32")
        return ds, None, None, None, None


class _ListNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score, targets, seg, Q, max_len, total):
        s = _vec(score.detach())
        loss, part = _f1(s.device), torch.empty(max(Q, 1), dtype=torch.float32, device=s.device)
        ctx.ds_unit = None
        if FusedStep.enabled and ctx.needs_input_grad[0]:
            ds = torch.empty(s.shape[0], dtype=torch.float32, device=s.device)
            check(lib().rr_listnet_step_f32(ptr(s), s.stride(0), ptr(targets), ptr(seg), Q, max_len, total, ptr(loss), ptr(part),
                                            ptr(_counter(s.device)), ptr(ds), 1, stream()), "rr_listnet_step_f32")
            ctx.ds_unit = ds
        else:
            check(lib().rr_listnet_fwd_f32(ptr(s), s.stride(0), ptr(targets), ptr(seg), Q, max_len, total, ptr(loss),
                                           ptr(part), stream()), "rr_listnet_fwd_f32")
        ctx.save_for_backward(s, targets, seg)
        ctx.meta = (Q, max_len, total)
        return loss.reshape(())                         # torch.mean -> 0-d (reference loss.py:347)

    @staticmethod
    def backward(ctx, g):
        if ctx.ds_unit is not None and _is_unit(g):
            ds, ctx.ds_unit = ctx.ds_unit, None
            FusedStep.hits += 1
            return ds, None, None, None, None, None
        s, targets, seg = ctx.saved_tensors
        Q, max_len, total = ctx.meta
        g = g.reshape(-1).contiguous().float()
        ds = torch.empty(s.shape[0], dtype=torch.float32, device=s.device)
        check(lib().rr_listnet_bwd_f32(ptr(s), s.stride(0), ptr(targets), ptr(seg), Q, max_len, total, ptr(g), ptr(ds),
                                       1, stream()), "rr_listnet_bwd_f32")
        return ds, None, None, None, None, None


class _EvidentialFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, poss, targets, seg, Q, max_len):
        p = poss.detach()
        if p.dtype != torch.float32 or p.dim() != 2 or p.shape[1] != 2 or p.stride(1) != 1:
            p = p.float().reshape(-1, 2).contiguous()
        loss, part = _f1(p.device), torch.empty(max(Q, 1), dtype=torch.float32, device=p.device)
        mu, var = p[:, 0], p[:, 1]
        ctx.ds_unit = None
        if FusedStep.enabled and ctx.needs_input_grad[0]:
            d = torch.empty(p.shape[0], 2, dtype=torch.float32, device=p.device)
            check(lib().rr_evidential_ranking_step_f32(ptr(mu), ptr(var), p.stride(0), ptr(targets), ptr(seg), Q, max_len,
                                                       ptr(loss), ptr(part), ptr(_counter(p.device)), ptr(d[:, 0]), ptr(d[:, 1]), 2,
                                                       stream()), "rr_evidential_ranking_step_f32")
            ctx.ds_unit = d
        else:
            check(lib().rr_evidential_ranking_fwd_f32(ptr(mu), ptr(var), p.stride(0), ptr(targets), ptr(seg), Q, max_len,
                                                      ptr(loss), ptr(part), stream()), "rr_evidential_ranking_fwd_f32")
        ctx.save_for_backward(p, targets, seg)
        ctx.meta = (Q, max_len)
        return loss

    @staticmethod
    def backward(ctx, g):
        if ctx.ds_unit is not None and _is_unit(g):
            d, ctx.ds_unit = ctx.ds_unit, None
            FusedStep.hits += 1
            return d, None, None, None, None
        p, targets, seg = ctx.saved_tensors
        Q, max_len = ctx.meta
        g = g.reshape(-1).contiguous().float()
        d = torch.empty(p.shape[0], 2, dtype=torch.float32, device=p.device)
        check(lib().rr_evidential_ranking_bwd_f32(ptr(p[:, 0]), ptr(p[:, 1]), p.stride(0), ptr(targets), ptr(seg), Q,
                                                  max_len, ptr(g), ptr(d[:, 0]), ptr(d[:, 1]), 2, stream()),
              "rr_evidential_ranking_bwd_f32")
        return d, None, None, None, None


class _RankNetFn(torch.autograd.Function):
    """loss_sum of 'sum_session' (train_pairwise.py:118-122); backward = its true gradient."""

    @staticmethod
    def forward(ctx, score, targets, seg, Q, max_len, sigma):
        s = _vec(score.detach())
        loss = _f1(s.device)
        pairs = torch.empty(1, dtype=torch.int64, device=s.device)
        part = torch.empty(max(2 * Q, 2), dtype=torch.float32, device=s.device)
        check(lib().rr_ranknet_fwd_f32(ptr(s), s.stride(0), ptr(targets), ptr(seg), Q, max_len, float(sigma), ptr(loss),
                                       ptr(pairs), ptr(part), stream()), "rr_ranknet_fwd_f32")
        ctx.save_for_backward(s, targets, seg)
        ctx.meta = (Q, max_len, float(sigma))
        ctx.mark_non_differentiable(pairs)
        return loss.reshape(()), pairs

    @staticmethod
    def backward(ctx, g, _gp):
        s, targets, seg = ctx.saved_tensors
        Q, max_len, sigma = ctx.meta
        g = g.reshape(-1).contiguous().float()
        ds = torch.empty(s.shape[0], dtype=torch.float32, device=s.device)
        check(lib().rr_ranknet_bwd_f32(ptr(s), s.stride(0), ptr(targets), ptr(seg), Q, max_len, sigma, 0, ptr(g),
                                       ptr(ds), 1, stream()), "rr_ranknet_bwd_f32")
        return ds, None, None, None, None, None


class _PointwiseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, var, targets):
        m = _vec(mean.detach())
        v = None if var is None else _vec(var.detach())
        if v is not None and v.stride(0) != m.stride(0):
            v = v.contiguous()
            m = m.contiguous()
        n = m.shape[0]
        loss = _f1(m.device)
        part = torch.empty(int(lib().rr_pointwise_partial_count(n)), dtype=torch.float32, device=m.device)
        if v is None:
            check(lib().rr_mse_fwd_f32(ptr(m), m.stride(0), ptr(targets), n, ptr(loss), ptr(part), stream()),
                  "rr_mse_fwd_f32")
        else:
            check(lib().rr_gauss_nll_fwd_f32(ptr(m), ptr(v), m.stride(0), ptr(targets), n, ptr(loss), ptr(part),
                                             stream()), "rr_gauss_nll_fwd_f32")
        ctx.save_for_backward(m, targets, *([] if v is None else [v]))
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        m, targets = saved[0], saved[1]
        v = saved[2] if len(saved) > 2 else None
        n = m.shape[0]
        g = g.reshape(-1).contiguous().float()
        dm = torch.empty(n, dtype=torch.float32, device=m.device)
        if v is None:
            check(lib().rr_mse_bwd_f32(ptr(m), m.stride(0), ptr(targets), n, ptr(g), ptr(dm), 1, stream()),
                  "rr_mse_bwd_f32")
            return dm, None, None
        dv = torch.empty(n, dtype=torch.float32, device=m.device)
        check(lib().rr_gauss_nll_bwd_f32(ptr(m), ptr(v), m.stride(0), ptr(targets), n, ptr(g), ptr(dm), ptr(dv), 1,
                                         stream()), "rr_gauss_nll_bwd_f32")
        return dm, dv, None


class LogCumsumExp(torch.autograd.Function):
    """Reference train/loss.py:9-61 for a 1-D input (dim 0), forward and backward in HIP."""

    @staticmethod
    def forward(ctx, input_data):
        x = input_data.detach().float().contiguous()
        _lib.require_cuda(x, "input_data")
        if x.dim() != 1:
            raise RuntimeError("LogCumsumExp: 1-D input expected (the reference applies it per sorted list)")
        y = torch.empty_like(x)
        check(lib().rr_logcumsumexp_fwd_f32(ptr(x), x.shape[0], ptr(y), stream()), "rr_logcumsumexp_fwd_f32")
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, grad_output):
        x, y = ctx.saved_tensors
        g = grad_output.float().contiguous()
        gx = torch.empty_like(x)
        check(lib().rr_logcumsumexp_bwd_f32(ptr(x), ptr(y), ptr(g), x.shape[0], ptr(gx), stream()),
              "rr_logcumsumexp_bwd_f32")
        return gx


class MLEloss(nn.Module):
    """ListMLE (reference train/loss.py:64-99)."""

    def forward(self, score, scope, targets_train, gpu: int = None):
        scope, seg, total, max_len, t = _prep(score, scope, targets_train, gpu)
        return _ListMLEFn.apply(score, t, seg, len(scope), max_len)


class ListnetLoss(nn.Module):
    """ListNet top-1 (reference train/loss.py:317-352)."""

    def forward(self, score, scope, targets, gpu: int = None):
        scope, seg, total, max_len, t = _prep(score, scope, targets, gpu)
        return _ListNetFn.apply(score, t, seg, len(scope), max_len, total)


class evidential_ranking(nn.Module):
    """UC-Listwise (reference train/loss.py:477-556); max_coeff/epoch/epochs are accepted and unused, as there."""

    def forward(self, possibilities, scope, targets, max_coeff=None, epoch=None, epochs=None, gpu: int = None):
        scope, seg, total, max_len, t = _prep(possibilities, scope, targets, gpu)
        return _EvidentialFn.apply(possibilities, t, seg, len(scope), max_len)


class GaussDisLoss(nn.Module):
    """Reference train/loss.py:144-162."""

    def forward(self, mean_scores, std_scores, targets, gpu: int = None):
        if gpu is not None:
            torch.cuda.set_device(gpu)
        _lib.require_cuda(mean_scores, "mean_scores")
        t = torch.as_tensor(targets, dtype=torch.float32).to(mean_scores.device).reshape(-1).contiguous()
        return _PointwiseFn.apply(mean_scores, std_scores, t)


class MSELoss(nn.Module):
    """nn.MSELoss() of the default 'regression' branch (reference train/train_listwise.py:166-167,282-285)."""

    def forward(self, output, targets):
        _lib.require_cuda(output, "output")
        t = torch.as_tensor(targets, dtype=torch.float32).to(output.device).reshape(-1).contiguous()
        return _PointwiseFn.apply(output, None, t)


def ranknet_loss(y_pred, scope, targets, sigma: float = 1.0, gpu: int = None):
    """RankNet 'sum_session' over a window of queries (reference train/train_pairwise.py:99-122,141).

    Returns (loss_sum, pairs): loss_sum is differentiable (divide by pairs and call backward,
    as the trainer does at :147-150); pairs is an int64 device scalar.  `y_pred` may be [M] or
    [M, k] (first column used, :115-116).
    """
    if y_pred.dim() > 1:
        y_pred = y_pred[:, 0]
    scope, seg, total, max_len, t = _prep(y_pred, scope, targets, gpu)
    loss, pairs = _RankNetFn.apply(y_pred, t, seg, len(scope), max_len, sigma)
    return loss, pairs


def ranknet_lambda(y_pred, scope, targets, sigma: float = 1.0, gpu: int = None):
    """'accelerate_grad' closed-form lambdas `back` (reference train/train_pairwise.py:125-133)."""
    if y_pred.dim() > 1:
        y_pred = y_pred[:, 0]
    scope, seg, total, max_len, t = _prep(y_pred, scope, targets, gpu)
    s = _vec(y_pred.detach())
    one = torch.ones(1, dtype=torch.float32, device=s.device)
    out = torch.empty(s.shape[0], dtype=torch.float32, device=s.device)
    check(lib().rr_ranknet_bwd_f32(ptr(s), s.stride(0), ptr(t), ptr(seg), len(scope), max_len, float(sigma), 1,
                                   ptr(one), ptr(out), 1, stream()), "rr_ranknet_bwd_f32")
    return out
