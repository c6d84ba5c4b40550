"""On-device ranking evaluation — mirror of the metric part of reference `reactranker/train/eval.py`
(`ranking_metrics` :475-555, `compute_NDCG` :460-472) and `reactranker/metrics.py` (NDCG@k).

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


def ranking_stats(scores: torch.Tensor, scope, targets, gpu: int = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-query statistics and predicted orders for lists described by `scope`.

    Returns (stats [Q, 8] float64 = top1 hit, top25 hit, recall@25%, NDCG1, NDCG2, NDCG25%, NDCG_all, NDCG@10(exp2),
             order [M] int32 = within-list index of the candidate at each predicted rank)."""
    if scores.dim() > 1:
        scores = scores[:, 0]                              # eval.py:508-509
    scope, seg, total, max_len, t = _prep(scores, scope, targets, gpu)
    s = _vec(scores.detach())
    order = torch.empty(total, dtype=torch.int32, device=s.device)
    stats = torch.empty(len(scope), 8, dtype=torch.float64, device=s.device)
    check(lib().rr_ranking_metrics_f32(ptr(s), s.stride(0), ptr(t), ptr(seg), len(scope), max_len, ptr(order),
                                       ptr(stats), stream()), "rr_ranking_metrics_f32")
    return stats, order


def ranking_metrics_from_scores(scores, scope, targets, gpu: int = None):
    """(top1, recall25, top25, NDCG_[NDCG1, NDCG2, NDCG25, NDCG_all]) as the reference's ranking_metrics returns
    them (eval.py:548-555)."""
    stats, _ = ranking_stats(scores, scope, targets, gpu)
    m = stats.mean(dim=0).cpu().numpy()
    return float(m[0]), float(m[2]), float(m[1]), m[3:7].copy()


def ranking_metrics(model, gpu, batches: Iterable, show_info: bool = False):
    """Evaluate `model` over an iterable of (r_batch, p_batch, scope, targets, add_features) batches of whole
    queries; same return value as the reference's ranking_metrics."""
    was_training = model.training
    model.eval()
    rows = []
    with torch.no_grad():
        for r_batch, p_batch, scope, targets, add_features in batches:
            out = model(r_batch, p_batch, gpu=gpu, add_features=add_features)
            stats, _ = ranking_stats(out, scope, targets, gpu)
            rows.append(stats)
    model.train(was_training)
    m = torch.cat(rows, 0).mean(dim=0).cpu().numpy()
    return float(m[0]), float(m[2]), float(m[1]), m[3:7].copy()


def ndcg_at_k(scores, scope, relevance, gpu: int = None) -> np.ndarray:
    """metrics.NDCG(k=10, 'exp2') of every query: `relevance` are the grades, ranked by `scores`."""
    stats, _ = ranking_stats(scores, scope, relevance, gpu)
    return stats[:, 7].cpu().numpy()
