"""On-device ranking evaluation — mirror of the metric part of reference `reactranker/train/eval.py`
(`ranking_metrics` :475-555, `evaluate_top_scores` :76-177, `calculate_ndcg` :329-457, `compute_NDCG` :460-472) and
`reactranker/metrics.py` (NDCG@k).

The reference scores ONE query per forward and does the ranking in Python lists; here all queries of a batch are
scored in one forward and ranked by one kernel (one wavefront per query).  The data-loading half of the
reference's function (DataProcessor / SMILES parsing) is outside the hot path: callers hand in batches.
"""
from __future__ import annotations

from typing import Iterable, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream
from .loss import _prep, _vec


NSTATS = 12          # RR_RANKING_NSTATS (include/reactranker_hip.h)


def ranking_stats(scores: torch.Tensor, scope, targets, gpu: int = None, ratio: float = 0.25,
                  ndcg_cut: float = 0.5) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-query statistics and predicted orders for lists described by `scope`.

    Returns (stats [Q, 12] float64 = top1 hit, predicted top-1 in target top-25%, recall@25%, NDCG1, NDCG2, NDCG25%,
             NDCG_all, NDCG@10(exp2), target top-1 in predicted top-`ratio`, calculate_ndcg's NDCG@`ndcg_cut`, KL,
             recall@`ratio`;
             order [M] int32 = within-list index of the candidate at each predicted rank)."""
    if scores.dim() > 1:
        scores = scores[:, 0]                              # eval.py:125-126, :390-394, :508-509
    scope, seg, total, max_len, t = _prep(scores, scope, targets, gpu)
    s = _vec(scores.detach())
    order = torch.empty(total, dtype=torch.int32, device=s.device)
    stats = torch.empty(len(scope), NSTATS, dtype=torch.float64, device=s.device)
    check(lib().rr_ranking_metrics_f32(ptr(s), s.stride(0), ptr(t), ptr(seg), len(scope), max_len, float(ratio),
                                       float(ndcg_cut), ptr(order), ptr(stats), stream()), "rr_ranking_metrics_f32")
    return stats, order


def _mean_stats(stats: torch.Tensor, exchange=None) -> np.ndarray:
    """Mean of the per-query statistics over every query - of every rank when `exchange` (reactranker_amd.dp.Exchange)
    spans a process group: each rank evaluated its own queries, the float64 column sums and the count are all-reduced."""
    if exchange is None or not exchange.on:
        return stats.mean(dim=0).cpu().numpy()
    v = torch.cat([stats.sum(dim=0), torch.tensor([float(stats.shape[0])], dtype=torch.float64, device=stats.device)])
    v = exchange.sum(v).cpu().numpy()
    return v[:-1] / max(v[-1], 1.0)


def ranking_metrics_from_scores(scores, scope, targets, gpu: int = None):
    """(top1, recall25, top25, NDCG_[NDCG1, NDCG2, NDCG25, NDCG_all]) as the reference's ranking_metrics returns
    them (eval.py:548-555)."""
    stats, _ = ranking_stats(scores, scope, targets, gpu)
    m = stats.mean(dim=0).cpu().numpy()
    return float(m[0]), float(m[2]), float(m[1]), m[3:7].copy()


def ranking_metrics(model, gpu, batches: Iterable, show_info: bool = False, exchange=None):
    """Evaluate `model` over an iterable of (r_batch, p_batch, scope, targets, add_features) batches of whole
    queries; same return value as the reference's ranking_metrics (which switches the model to eval mode, :487).
    exchange: a reactranker_amd.dp.Exchange when every rank evaluates its own shard of the validation queries."""
    was_training = model.training
    model.eval()
    stats, _ = _eval_stats(model, gpu, batches)
    model.train(was_training)
    m = _mean_stats(stats, exchange)
    return float(m[0]), float(m[2]), float(m[1]), m[3:7].copy()


def _eval_stats(model, gpu, batches, ratio=0.25, ndcg_cut=0.5, transform=None, keep_mode=False):
    """Score every batch of whole queries with `model` and rank it on the device: (stats [Q_total, 12], list of
    (scores, order, scope, targets) per batch).  The mode of `model` is left as the caller set it (the reference's
    evaluate_top_scores / calculate_ndcg do not call model.eval(): eval.py:83-84, :337-338)."""
    rows, per_batch = [], []
    with torch.no_grad():
        for r_batch, p_batch, scope, targets, add_features in batches:
            if len(scope) == 0:                              # an empty shard of a data-parallel evaluation
                continue
            out = model(r_batch, p_batch, gpu=gpu, add_features=add_features)
            if transform is not None:
                out = transform(out)
            stats, order = ranking_stats(out, scope, targets, gpu, ratio, ndcg_cut)
            rows.append(stats)
            per_batch.append((out, order, list(scope), targets))
    if not rows:
        dev = torch.device("cuda", torch.cuda.current_device() if gpu is None else gpu)
        return torch.zeros(0, NSTATS, dtype=torch.float64, device=dev), per_batch
    return torch.cat(rows, 0), per_batch


def top_scores_from_scores(scores, scope, targets, gpu: int = None, ratio: float = 0.25):
    """(average_score, average_pred_in_targ, average_top1_in_pred) of evaluate_top_scores (eval.py:172-177) for one
    batch of scored lists."""
    stats, _ = ranking_stats(scores, scope, targets, gpu, ratio)
    m = stats.mean(dim=0).cpu().numpy()
    return float(m[0]), float(m[11]), float(m[8])


def evaluate_top_scores(model, gpu, batches: Iterable, ratio: float = 0.25, show_info=False, exchange=None):
    """Reference evaluate_top_scores (train/eval.py:76-177) over an iterable of (r_batch, p_batch, scope, targets,
    add_features) batches of whole queries: returns (average_score = top-1 by first maximum, average_pred_in_targ =
    share of the predicted top-`ratio` inside the target top-`ratio`, average_top1_in_pred = the TARGET's top-1 inside
    the predicted top-`ratio`).  This - not ranking_metrics - is what the RankNet epoch driver validates with
    (run_train_pairwise.py:91-96) and what both test() functions report (test_listwise.py:51-54, test_ranknet.py:59)."""
    stats, _ = _eval_stats(model, gpu, batches, ratio=ratio)
    m = _mean_stats(stats, exchange)
    return float(m[0]), float(m[11]), float(m[8])


def calculate_ndcg(model, gpu, batches: Iterable, NDCG_cut: float = 0.5, is_order: bool = True, means=None, stds=None,
                   show_info=False, exchange=None):
    """Reference calculate_ndcg (train/eval.py:329-457): (NDCG_mean, KL_mean, total_order, None).

    `means` / `stds` de-standardise the model output first (:379-388: mean column * std + mean, variance column *
    std^2).  With is_order, `total_order` holds one row per candidate, queries in input order and candidates in TARGET
    order: [target, prediction, (uncertainty,) true rank, predicted rank] (:406-420); without it the raw
    [target, prediction...] rows (:444-449) and NDCG / KL are None.  The reference's fourth value (the SMILES of every
    row, :427-437) belongs to the DataFrame side and is returned as None."""
    def destd(out):
        if means is None:
            return out
        if out.dim() > 1:                                   # :380-386
            scale = torch.ones(out.size(1), dtype=out.dtype, device=out.device)
            shift = torch.zeros_like(scale)
            scale[0], shift[0] = float(stds), float(means)
            if out.size(1) > 1:
                scale[1] = float(stds) ** 2
            return out * scale + shift
        return out * float(stds) + float(means)

    stats, per_batch = _eval_stats(model, gpu, batches, ndcg_cut=NDCG_cut, transform=destd)
    total_order = []
    for out, order, scope, targets in per_batch:
        o = out.detach().cpu()
        t = torch.as_tensor(np.asarray(targets), dtype=torch.float32).reshape(-1)
        if not is_order:
            cols = o.reshape(o.size(0), -1)
            total_order.extend([[float(a)] + [float(x) for x in b] if o.dim() > 1 else [float(a), float(b[0])]
                                for a, b in zip(t.tolist(), cols.tolist())])
            continue
        pred = o[:, 0] if o.dim() > 1 else o
        off = 0
        for c in scope:                                      # the listing only (host bookkeeping, :406-420)
            bt, bp = t[off:off + c], pred[off:off + c]
            idx = torch.sort(bt, descending=True, stable=True)[1]
            sp = bp[idx]
            pred_order = torch.argsort(torch.argsort(sp, descending=True, stable=True), stable=True) + 1
            cols = [bt[idx], sp]
            if o.dim() > 1:
                cols.append(o[off:off + c, 1][idx])
            cols += [torch.arange(1, c + 1, dtype=torch.float32), pred_order.float()]
            total_order.extend(torch.stack(cols, dim=1).tolist())
            off += c
    if not is_order:
        return None, None, total_order, None
    m = _mean_stats(stats, exchange)         # (the listing stays this rank's own queries)
    return float(m[9]), float(m[10]), total_order, None


def calculate_mse(model, gpu, batches: Iterable, exchange=None) -> float:
    """Reference calculate_mse (train/eval.py:558-609; the `save_metric='mse'` criterion of train_listwise.py:345-351):
    switches the model to eval mode and returns the mean squared error between the first output column and the targets
    of the LAST batch - the reference overwrites its `MSE` in every iteration of the loop and returns the final one (:607-609),
    which is kept.  Data parallel: the last global batch's squared errors and count are summed over the ranks."""
    from .loss import MSELoss
    model.eval()
    last = None
    with torch.no_grad():
        for r_batch, p_batch, scope, targets, add_features in batches:
            last = (r_batch, p_batch, scope, targets, add_features)
        dev = torch.device("cuda", torch.cuda.current_device() if gpu is None else gpu)
        v = torch.zeros(2, dtype=torch.float64, device=dev)                 # [sum of squared errors, count]
        if last is not None and len(last[2]) > 0:
            out = model(last[0], last[1], gpu=gpu, add_features=last[4])
            pred = out[:, 0] if out.dim() > 1 else out
            t = torch.as_tensor(last[3], dtype=torch.float32).reshape(-1)
            n = int(pred.shape[0])
            v[0] = MSELoss()(pred.contiguous(), t).double().sum() * n      # rr_mse_fwd_f32: the mean over this shard
            v[1] = n
        if exchange is not None and exchange.on:
            v = exchange.sum(v)
    return float(v[0] / v[1].clamp(min=1.0))


def ndcg_at_k(scores, scope, relevance, gpu: int = None) -> np.ndarray:
    """metrics.NDCG(k=10, 'exp2') of every query: `relevance` are the grades, ranked by `scores`."""
    stats, _ = ranking_stats(scores, scope, relevance, gpu)
    return stats[:, 7].cpu().numpy()
