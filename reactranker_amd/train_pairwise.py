"""RankNet training loop on pre-packed windows of whole queries: `factorized_training_loop` of the reference
(reactranker/train/train_pairwise.py:81-173).  The reference runs one forward per query and accumulates loss /
lambdas until `batch_size` candidates have been seen (:141-160); here a batch already IS such a window (its queries
are scored in one forward, reactranker_amd.loss.ranknet_loss / ranknet_lambda handle every query of the window), so
one batch = one optimizer step with the same normalisation (loss / ordered pairs of the window).

A batch is a mapping with keys  r, p (BatchMolGraph), scope (list[int]), targets (float32 [M]), add (ndarray or None).
"""
from __future__ import annotations

from typing import Iterable

import torch

from .loss import backward as loss_backward, ranknet_lambda, ranknet_loss


def factorized_training_loop(epoch: int, model, optimizer, scheduler, batches: Iterable, sigma: float = 1.0,
                             training_algo: str = "sum_session", gpu: int = 0, exchange=None) -> float:
    """One epoch; returns the mean of the per-step losses like the reference (:173).
    training_algo: 'sum_session' (autograd through the pair losses, :117-122,147-148) or 'accelerate_grad'
    (closed-form lambdas pushed through y_pred.backward, :123-137,149-151).

    The window's ordered-pair count - the loss's normaliser (:106,147) and the reason a window is skipped (:101-103) - is
    counted on the host from the targets the batch carries (dp.count_pairs, cached on the batch), and the per-step losses
    stay on the device until the epoch ends: no host synchronisation per step (the reference reads `.item()` per query).
    exchange: a reactranker_amd.dp.Exchange; a batch is then this rank's shard of the window and the normaliser is the
    whole window's pair count."""
    from .dp import Exchange, step_counts
    if training_algo not in ("sum_session", "accelerate_grad"):
        raise ValueError("training algo {} not implemented".format(training_algo))
    own_exchange = exchange is None
    ex = exchange if exchange is not None else Exchange(model)   # (under torch.distributed it owns the gradient bucket)
    dev = next(model.parameters()).device
    minibatch_loss = []
    for b in batches:
        if "_counts" not in b:
            b["_counts"] = step_counts(b["scope"], b["targets"]) if len(b["scope"]) else dict(queries=0, cands=0, pairs=0)
        local = b["_counts"]
        glob = ex.counts(b, dev)[1] if ex.on else local
        pairs = glob["pairs"]
        if pairs == 0:                                   # windows without any ordered pair carry no information (:101-103)
            continue
        model.zero_grad()
        if local["pairs"] > 0:
            y_pred = model(b["r"], b["p"], gpu=gpu, add_features=b.get("add"))
            if y_pred.dim() > 1:
                y_pred = y_pred[:, 0]
            loss_sum, _ = ranknet_loss(y_pred if training_algo == "sum_session" else y_pred.detach(), b["scope"],
                                       b["targets"], sigma, gpu)
            loss = loss_sum / pairs
            if training_algo == "sum_session":
                loss_backward(loss)
            else:
                back = ranknet_lambda(y_pred, b["scope"], b["targets"], sigma, gpu)
                y_pred.backward(back / pairs)
            minibatch_loss.append(loss.detach().sum().reshape(1))
        else:                                            # this rank's shard has no ordered pair: zero gradient, zero loss
            minibatch_loss.append(torch.zeros(1, device=dev))
        ex.reduce_grads(1.0)                             # already normalised by the WINDOW's pair count: a plain sum
        optimizer.step()
        scheduler.step()
    model.zero_grad()
    if own_exchange:
        ex.close()
    if not minibatch_loss:
        return float("nan")
    per_step = ex.sum(torch.cat(minibatch_loss).double())
    return float(per_step.mean())
