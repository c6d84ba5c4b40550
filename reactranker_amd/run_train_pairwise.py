"""RankNet epoch driver on pre-packed windows of whole queries: `run_train` of the reference
(reactranker/train/run_train_pairwise.py:18-140) for the strategies main_ranknet.py uses ('sum_session' and
'accelerate_grad', task_type 'baseline').  Standardises the targets like the reference (:36-45: z-score with the training
set's statistics, sign flipped unless the target is 'lgk'), runs `factorized_training_loop` per epoch, evaluates on the
validation windows with `evaluate_top_scores` (:91-96) and checkpoints on the selected metric (:97-117).  The DataFrame / SMILES side stays the reference's."""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from .eval import evaluate_top_scores
from .train_listwise import standardize_batches
from .train_pairwise import factorized_training_loop
from .utils import save_checkpoint


def standardize_pairwise(train_targets, val_targets, target_name: str = "ea"):
    """run_train_pairwise.py:36-45: z-score with the training mean / population std; every target except 'lgk' flips sign."""
    tr, va = np.asarray(train_targets, np.float64), np.asarray(val_targets, np.float64)
    mean, std = float(tr.mean()), float(tr.std())
    sign = 1.0 if target_name == "lgk" else -1.0
    return sign * (tr - mean) / std, sign * (va - mean) / std, mean, std


def run_train(model: torch.nn.Module, scheduler, train_batches: Sequence, val_batches: Sequence,
              path_checkpoints: Union[str, List[str], None], optimizer, epochs: int, seed: int, gpu: int,
              train_strategy: str = "sum_session", task_type: str = "baseline", logger=None,
              target_name: Optional[str] = "ea", save_metric: Optional[str] = None, sigma: float = 1.0, epoch_hook=None,
              group=None):
    """Returns the per-epoch history [{epoch, train_loss, top1, pred_top25_in_targ_top25, top1_in_pred_top25 (the TARGET's
    top-1 inside the predicted top-25 %), checkpoint, checkpoint_all (which of the three 'all' metrics improved)}].
    epoch_hook(epoch, model, record): optional observer called after every epoch's validation (not in the reference).
    group: data-parallel training under torch.distributed exactly as in reactranker_amd.train_listwise.train - every rank
    passes its shard of every window (whole queries, each batch carrying the window's `global` counts) and of the
    validation queries; the loss is normalised by the WINDOW's ordered pairs, gradients are summed over the ranks,
    validation statistics likewise, rank 0 writes the checkpoints (main_ranknet.py:143-160 is the single-process driver)."""
    from .dp import Exchange
    if train_strategy not in ("sum_session", "accelerate_grad") or task_type != "baseline":
        raise ValueError("reactranker_amd covers the RankNet strategies main_ranknet.py selects: train_strategy "
                         "'sum_session' / 'accelerate_grad' with task_type 'baseline'")
    if gpu is not None:
        torch.cuda.set_device(gpu)
    model = model.cuda(gpu)
    ex = Exchange(model, group)
    try:
        return _run_train(model, scheduler, train_batches, val_batches, path_checkpoints, optimizer, epochs, gpu, train_strategy,
                          logger, target_name, save_metric, sigma, epoch_hook, ex)
    finally:
        ex.close()


def _run_train(model, scheduler, train_batches, val_batches, path_checkpoints, optimizer, epochs, gpu, train_strategy, logger,
               target_name, save_metric, sigma, epoch_hook, ex):
    mean, std = 0.0, 1.0
    if target_name is not None:
        # same statistics as the listwise trainer's default branch: z-score, sign flipped unless 'lgk' (:39-44)
        tn = "lgk" if target_name == "lgk" else "ea"
        train_batches, val_batches, mean, std = standardize_batches(list(train_batches), list(val_batches), tn, True, None, ex)
    ex.broadcast_model(model)
    ex.check_same_steps(len(train_batches), next(model.parameters()).device)
    score_old = [0.0, 0.0, 0.0] if save_metric == "all" else 0.0
    say = logger.info if (logger is not None and ex.is_writer) else (lambda *_: None)
    history = []
    for epoch in range(epochs):
        say("learning rate is: {}".format(optimizer.param_groups[0]["lr"]))
        model.zero_grad()
        model.train()
        epoch_loss = factorized_training_loop(epoch, model, optimizer, scheduler, train_batches, sigma=sigma,
                                              training_algo=train_strategy, gpu=gpu, exchange=ex)
        model.eval()
        with torch.no_grad():
            # evaluate_top_scores, not ranking_metrics (:91-96): its third value is the TARGET's top-1 inside the
            # predicted top-25 % (eval.py:156-159)
            top1, recall25, top25 = evaluate_top_scores(
                model, gpu, [(b["r"], b["p"], b["scope"], b["targets"], b.get("add")) for b in val_batches], ratio=0.25,
                exchange=ex)
        saved = False
        saved_which = [False, False, False]                  # save_metric='all': T1 / T25_in_T25 / T25 (main_ranknet.py:68-74)

        def keep(path, which=0):
            nonlocal saved
            saved_which[which] = True
            if path is not None:
                saved = True
                if ex.is_writer:
                    save_checkpoint(path, model, mean, std)
        if save_metric is None or save_metric == "average_score":
            if top1 >= score_old:
                score_old = top1
                keep(path_checkpoints)
        elif save_metric == "all":
            for i, v in enumerate((top1, recall25, top25)):
                if v >= score_old[i]:
                    score_old[i] = v
                    keep(path_checkpoints[i] if path_checkpoints is not None else None, i)
        else:
            raise Exception("Unknown save metric")
        history.append(dict(epoch=epoch + 1, train_loss=float(epoch_loss), top1=float(top1),
                            pred_top25_in_targ_top25=float(recall25), top1_in_pred_top25=float(top25), checkpoint=saved,
                            checkpoint_all=list(saved_which)))
        if epoch_hook is not None:
            epoch_hook(epoch, model, history[-1])
        say("Epoch [{}/{}],train_loss,{:.4f}, average_score_top1,{:.4f}, average_pred_in_targ_top25%,{:.4f}"
            .format(epoch + 1, epochs, epoch_loss, top1, top25))
    return history
