"""Listwise training loop on pre-packed batches: the loop body, loss dispatch, validation and checkpoint selection
of the reference trainer (reactranker/train/train_listwise.py:21-372) for the task types whose losses exist in
reactranker_amd.loss.  The reference's DataFrame / SMILES plumbing (DataProcessor, Parsing_features: out of scope,
SURVEY.md section 2 row 17) is replaced by an iterable of batches; everything downstream of
`model(r_inputs, p_inputs, gpu=gpu, add_features=...)` keeps the reference's call shapes.

A batch is a mapping with keys  r, p (BatchMolGraph), scope (list[int]), targets (float32 [M]), add (ndarray [M,F] or
None) - what train_listwise.py:176-189 gets from generate_batch_reactions + parsing_reactions.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence, Union

import numpy as np
import torch

from . import loss as RL
from .eval import calculate_mse, ranking_metrics
from .utils import save_checkpoint

NDCG_METRICS = ["NDCG@1", "NDCG@2", "NDCG@25%", "NDCG@all"]
SUPPORTED_TASKS = ("mle", "listnet", "evidential_ranking", "gauss_regression", "mle_gaussian", "listnet_gauss",
                   "mle_regression", "listnet_regression", "regression")


def standardize_targets(train_targets, val_targets, target_name: str = "ea", normalize_target=True, save_metric=None):
    """Target standardisation and sign flip of the reference trainer (train/train_listwise.py:66-122), on arrays.

    Activation energies ('ea' and every name other than 'lgk' / 'lgk_bi') rank the LOWEST value first, so their sign
    is flipped; 'lgk' keeps its sign, 'lgk_bi' is passed through.  `normalize_target`: True -> z-score with the
    training set's mean / population std (ddof=0); False -> sign only; a float f -> x * f / (max - min) of the
    training set; a string 'lo,hi' -> min-max onto [lo, hi].  With an NDCG save metric the validation targets stay
    raw (:117-120).  Returns (train_std, val_std, mean, std) - mean / std are what the checkpoint stores."""
    tr = np.asarray(train_targets, np.float64)
    va = np.asarray(val_targets, np.float64)
    mean, std = float(tr.mean()), float(tr.std())                 # pandas .std(ddof=0)
    if target_name == "lgk_bi":
        f = lambda x: x                                           # noqa: E731
    else:
        sign = 1.0 if target_name == "lgk" else -1.0
        if isinstance(normalize_target, float):
            mx, mn = tr.max(), tr.min()
            f = lambda x: sign * (x * normalize_target) / (mx - mn)                       # noqa: E731
        elif isinstance(normalize_target, str):
            mx, mn = tr.max(), tr.min()
            lo, hi = (int(v) for v in normalize_target.split(","))
            f = lambda x: sign * (x - mn) * (hi - lo) / (mx - mn) + lo                   # noqa: E731
        elif normalize_target:
            f = lambda x: sign * (x - mean) / std                                        # noqa: E731
        else:
            f = lambda x: sign * x                                                       # noqa: E731
    val_std = va if save_metric in NDCG_METRICS else f(va)
    return f(tr), val_std, mean, std


def standardize_batches(train_batches, val_batches, target_name="ea", normalize_target=True, save_metric=None, exchange=None):
    """standardize_targets over lists of packed batches: statistics from ALL training targets (the reference computes
    them on the training DataFrame), new batch dicts with standardised float32 `targets`.  With a data-parallel
    `exchange` the statistics are those of every rank's training targets (gathered in rank order), so each rank applies
    the transformation the single-process run applies."""
    def cat(bs):
        return np.concatenate([np.asarray(torch.as_tensor(b["targets"]).cpu(), np.float64).reshape(-1) for b in bs]) \
            if len(bs) else np.zeros(0)
    tr_local, va_local = cat(train_batches), cat(val_batches)
    if exchange is not None and exchange.on:
        import torch.distributed as dist
        parts = [None] * exchange.world
        dist.all_gather_object(parts, tr_local, group=exchange.group)
        tr_all, _, mean, std = standardize_targets(np.concatenate(parts), np.zeros(0), target_name, normalize_target, save_metric)
        lo = sum(len(q) for q in parts[:exchange.rank])
        tr = tr_all[lo:lo + len(tr_local)]
        # the validation transform uses the training statistics only: apply it to this rank's validation targets
        _, va, _, _ = standardize_targets(np.concatenate(parts), va_local, target_name, normalize_target, save_metric)
    else:
        tr, va, mean, std = standardize_targets(tr_local, va_local, target_name, normalize_target, save_metric)

    def split(bs, flat):
        out, off = [], 0
        for b in bs:
            n = int(torch.as_tensor(b["targets"]).numel())
            nb = dict(b)
            nb["targets"] = torch.tensor(flat[off:off + n], dtype=torch.float32)
            off += n
            out.append(nb)
        return out
    return split(train_batches, tr), split(val_batches, va), mean, std


def batch_loss(task_type: str, output, scope, targets, gpu, epoch: int = 0, epochs: int = 1, max_coeff: float = 1e-4):
    """The loss the reference trainer forms for one batch (train_listwise.py:196-285)."""
    mle, listnet, evid, gauss, mse = RL.MLEloss(), RL.ListnetLoss(), RL.evidential_ranking(), RL.GaussDisLoss(), RL.MSELoss()
    if task_type == "mle":
        return mle(output, scope, targets, gpu)
    if task_type == "listnet":
        return listnet(output, scope, targets, gpu)
    if task_type == "evidential_ranking":
        return evid(output, scope, targets, max_coeff, epoch, epochs, gpu)
    if task_type == "gauss_regression":
        return gauss(output[:, 0], output[:, 1], targets, gpu)
    if task_type == "mle_gaussian":
        return mle(output[:, 0], scope, targets, gpu) + gauss(output[:, 0], output[:, 1], targets, gpu)
    if task_type == "listnet_gauss":
        return listnet(output[:, 0], scope, targets, gpu) + gauss(output[:, 0], output[:, 1], targets, gpu)
    if task_type == "mle_regression":
        return mse(output, targets) + mle(output, scope, targets, gpu)
    if task_type == "listnet_regression":
        return listnet(output, scope, targets, gpu) + mse(output, targets)
    if task_type == "regression":                       # the reference's default branch: nn.MSELoss
        return mse(output, targets)
    raise ValueError(f"task_type {task_type!r} is not covered by reactranker_amd (supported: {SUPPORTED_TASKS})")


def train(model: torch.nn.Module, scheduler, train_batches: Union[Sequence, Callable[[int], Iterable]],
          val_batches: Sequence, path_checkpoints: Union[str, List[str], None], optimizer, epochs: int, seed: int, gpu: int,
          task_type: str = "mle", logger=None, save_metric: Optional[str] = None, max_coeff: float = 1e-4,
          mean: float = 0.0, std: float = 1.0, target_name: Optional[str] = None, normalize_target=True,
          epoch_hook: Optional[Callable] = None, group=None):
    """Same control flow as the reference train(): fixed seed, per batch forward / loss / zero_grad / backward /
    optimizer.step / scheduler.step (train_listwise.py:287-290), validation with ranking_metrics after every epoch,
    checkpoint whenever the selected metric does not get worse (:310-350).  `train_batches` is a sequence, or a
    callable epoch -> iterable (the reference reshuffles with seed=epoch, :178).  Returns the per-epoch history.
    epoch_hook(epoch, model, record): optional observer called after every epoch's validation (not in the reference; the
    trajectory tests read the validation scores through it).

    Data parallel (the reference has no distributed code): under an initialised torch.distributed
    (`group` = a process group, None = the default one) every rank passes ITS shard of every global step - whole queries,
    see reactranker_amd.dp.shard_query_batch - and of the validation queries.  Per step the rank's gradient is weighted
    by its share of the loss's normaliser and summed over the ranks in one all-reduce of the flat bucket the explicit
    backward writes into; validation statistics are summed over the ranks; rank 0 writes the checkpoints; every rank
    returns the same history.  One process (no torch.distributed) runs the same code with the exchange switched off."""
    from .dp import Exchange
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    model = model.cuda(gpu)
    ex = Exchange(model, group)
    if ex.on and task_type not in ("mle", "listnet", "evidential_ranking", "regression", "gauss_regression"):
        raise ValueError(f"data-parallel training covers losses with ONE normaliser; {task_type!r} sums two")
    try:
        return _train(model, scheduler, train_batches, val_batches, path_checkpoints, optimizer, epochs, gpu, task_type, logger,
                      save_metric, max_coeff, mean, std, target_name, normalize_target, epoch_hook, ex)
    finally:
        ex.close()


def _train(model, scheduler, train_batches, val_batches, path_checkpoints, optimizer, epochs, gpu, task_type, logger, save_metric,
           max_coeff, mean, std, target_name, normalize_target, epoch_hook, ex):
    if target_name is not None:                       # raw targets: standardise / flip like the reference (:66-122)
        if callable(train_batches):
            raise ValueError("target standardisation needs the training batches as a sequence (statistics over all of them)")
        train_batches, val_batches, mean, std = standardize_batches(list(train_batches), list(val_batches), target_name,
                                                                     normalize_target, save_metric, ex)
    ex.broadcast_model(model)                         # identical replicas (a no-op for one process)
    if not callable(train_batches):
        ex.check_same_steps(len(train_batches), next(model.parameters()).device)
    # dropout streams: every forward draws a fresh stream seed from torch's generator (mpn._fresh_seed), which the
    # manual_seed above makes reproducible - like the reference, whose nn.Dropout advances that generator per call.
    # model.dropout_seed (a test knob that pins ONE stream for every step) must stay unset here.
    if getattr(model, "dropout_seed", None) is not None:
        model.dropout_seed = None
    score_old = [0.0, 0.0, 0.0] if save_metric == "all" else (float("inf") if save_metric == "mse" else 0.0)   # :54-59
    history = []
    say = logger.info if (logger is not None and ex.is_writer) else (lambda *_: None)
    dev = next(model.parameters()).device
    for epoch in range(epochs):
        say("learning rate is: {}".format(optimizer.param_groups[0]["lr"]))
        model.train()
        loss = torch.zeros(1, device=dev)
        for b in (train_batches(epoch) if callable(train_batches) else train_batches):
            optimizer.zero_grad()
            if len(b["scope"]) > 0:
                output = model(b["r"], b["p"], gpu=gpu, add_features=b.get("add"))
                loss = batch_loss(task_type, output, b["scope"], b["targets"], gpu, epoch, epochs, max_coeff)
                RL.backward(loss)                       # loss.backward() (:288) seeded with the library's constant one (loss.FusedStep)
            else:                                       # an empty shard: this rank adds nothing to the step
                loss = torch.zeros(1, device=dev)
            if ex.on:
                local, glob = ex.counts(b, dev)
                w = ex.weight(task_type, local, glob)
                ex.reduce_grads(w)
                loss = loss.detach().reshape(-1)[:1] * w    # summed over the ranks below: the whole step's loss
            optimizer.step()
            scheduler.step()
        model.eval()
        top1, recall25, top25, ndcg = ranking_metrics(
            model, gpu, [(b["r"], b["p"], b["scope"], b["targets"], b.get("add")) for b in val_batches], exchange=ex)
        saved = False

        def keep(path):
            nonlocal saved
            saved = True
            if path is not None and ex.is_writer:
                save_checkpoint(path, model, mean, std)
        if save_metric is None or save_metric == "average_score":
            if top1 >= score_old:
                score_old = top1
                keep(path_checkpoints)
        elif save_metric == "all":
            for i, v in enumerate((top1, recall25, top25)):
                if v >= score_old[i]:
                    score_old[i] = v
                    keep(path_checkpoints[i] if path_checkpoints is not None else None)
        elif save_metric == "average_pred_in_targ":
            if recall25 >= score_old:
                score_old = recall25
                keep(path_checkpoints)
        elif save_metric == "average_top1_in_pred":
            if top25 >= score_old:
                score_old = top25
                keep(path_checkpoints)
        elif save_metric in NDCG_METRICS:
            v = ndcg[NDCG_METRICS.index(save_metric)]
            if v >= score_old:
                score_old = v
                keep(path_checkpoints)
        elif save_metric == "mse":                          # :345-351 (calculate_mse: the LAST validation batch's error)
            mse_val = calculate_mse(model, gpu, [(b["r"], b["p"], b["scope"], b["targets"], b.get("add")) for b in val_batches],
                                    exchange=ex)
            if mse_val <= score_old:
                score_old = mse_val
                keep(path_checkpoints)
        else:
            raise Exception("Unknown save metric")
        saved = saved and path_checkpoints is not None
        last_loss = float(ex.sum(loss.detach().sum().reshape(1).clone()))     # the reference logs the last step's loss (:353)
        rec = dict(epoch=epoch + 1, train_loss=last_loss, top1=float(top1), top1_in_pred_top25=float(top25),
                   pred_top25_in_targ_top25=float(recall25), ndcg=[float(x) for x in ndcg], checkpoint=saved)
        if save_metric == "mse":
            rec["mse"] = float(mse_val)
        history.append(rec)
        if epoch_hook is not None:
            epoch_hook(epoch, model, rec)
        say("Epoch [{}/{}], train_loss,{:.4f}, top1,{:.4f}, top1_in_pred_top25%,{:.4f}, pred_top25%_in_targ_top25%,{:.4f}"
            .format(epoch + 1, epochs, rec["train_loss"], top1, top25, recall25))
    return history
