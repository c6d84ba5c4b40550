"""k-fold driver with a config object: what the reference's template scripts `main.py` (:51-173, listwise / pointwise
losses) and `main_ranknet.py` (pairwise) do after their data has been read, on pre-packed batches.

The reference scripts are templates with placeholders (`user_defined`, `your_gpu`, Windows path stubs) and read CSVs
through pandas + RDKit (out of scope here).  `run(cfg, folds)` keeps their control
flow: per fold - seeds (:106-110), build_model (:111-120), build_optimizer / build_lr_scheduler (:135-143), train
(:145-162) or run_train (main_ranknet.py:143-155), then test() on the best checkpoint (:169-172) - and their checkpoint
layout (one file per fold, or T1/ T25_in_T25/ T25/ sub-directories for save_metric='all', :68-88).

`folds(i)` supplies fold i as (train_batches, val_batches, test_batches); a batch is the mapping the trainers take
(r, p, scope, targets, add) - e.g. steps of a shard file (reactranker_amd.shards) or in-memory BatchMolGraphs.
"""
from __future__ import annotations

import logging
import os
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple, Union

import torch

from .base_model import build_model
from .eval import calculate_ndcg, evaluate_top_scores
from .run_train_pairwise import run_train
from .train_listwise import train
from .train_utils import build_lr_scheduler, build_optimizer
from .utils import load_checkpoint


@dataclass
class Config:
    """The module-level constants of main.py:15-49 / main_ranknet.py:17-51 as one object."""
    path: str                                   # checkpoint / log directory
    k_fold: int = 1
    total_epochs: int = 1
    batch_size: int = 64                        # queries per optimizer step (only enters the LR schedule here)
    task_type: str = "listnet"                  # loss: mle / listnet / evidential_ranking / regression / ... ; 'ranknet'
    train_strategy: str = "sum_session"         # RankNet only (main_ranknet.py:38): 'sum_session' | 'accelerate_grad'
    target_name: Optional[str] = "lgk"          # None: targets are already standardised
    normalize_target: Union[bool, float, str] = True
    init_lr: float = 1e-4
    max_lr: float = 1e-3
    final_lr: float = 1e-4
    warmup_epochs: float = 2.0
    save_metric: Optional[str] = "all"
    add_features_dim: int = 1
    gpu: int = 0
    model: dict = field(default_factory=lambda: dict(hidden_size=300, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3,
                                                     use_bias=True, dropout=0.1, task_num=1,
                                                     ffn_last_layer="with_softplus"))


def test(model, test_batches: Sequence, path_checkpoints: str, gpu: int, logger=None, target_name: Optional[str] = "ea",
         cal_ngcd: bool = False, is_order: bool = True, return_order: bool = True, task_type: Optional[str] = None,
         exchange=None):
    """Reference train/test_listwise.py:10-86 (and test_ranknet.py:30-80, which reports the same triple): load the
    fold's checkpoint, flip the sign of raw targets unless 'lgk' (:30-35), evaluate with `evaluate_top_scores` at
    ratio 0.25 (:51-54) -> (average_score [top-1], average_pred_in_targ [predicted top-25 % inside the target top-25 %],
    average_top1_in_pred [the target's top-1 inside the predicted top-25 %]).  With cal_ngcd also `calculate_ndcg` at
    NDCG_cut = 0.25 on the de-standardised outputs (:58-63), logged, and - with return_order - its per-candidate
    listing appended to the return value (:83-84; the SMILES column stays with the DataFrame side: None).
    task_type 'MC_dropout' keeps the model in train mode (:42-45).  exchange: a reactranker_amd.dp.Exchange when every
    rank evaluates its own shard of the test queries (the means are then over all ranks' queries)."""
    scaler = load_checkpoint(path_checkpoints, model, map_location="cpu")
    model = model.cuda(gpu)
    model.train() if task_type == "MC_dropout" else model.eval()
    sign = 1.0
    means = stds = None
    if scaler is not None and scaler.get("means") is not None:
        means, stds = scaler["means"], scaler["stds"]
        if target_name is not None and target_name != "lgk":
            sign = -1.0
    batches = [(b["r"], b["p"], b["scope"], sign * torch.as_tensor(b["targets"], dtype=torch.float32), b.get("add"))
               for b in test_batches]
    with torch.no_grad():
        top1, recall25, top25 = evaluate_top_scores(model, gpu, batches, ratio=0.25, exchange=exchange)
        if cal_ngcd:
            ndcg, kl_div, order, smiles_and_index = calculate_ndcg(model, gpu, batches, NDCG_cut=0.25,
                                                                   is_order=is_order, means=means, stds=stds, exchange=exchange)
    if logger is not None:
        if cal_ngcd:
            logger.info("test: NDCG0.25 {}, KL divergence {}".format(ndcg, kl_div))
        logger.info("test: average score {:.4f}, pred top25% in targ top25% {:.4f}, targ top1 in pred top25% {:.4f}"
                    .format(top1, recall25, top25))
    if cal_ngcd and return_order:
        return float(top1), float(recall25), float(top25), order, smiles_and_index
    return float(top1), float(recall25), float(top25)


def run(cfg: Config, folds: Callable[[int], Tuple[Sequence, Sequence, Sequence]], logger=None, group=None) -> List[List[float]]:
    """Returns the per-fold test scores [[top1, pred_top25_in_targ_top25, top1_in_pred_top25], ...] (main.py:173).
    Under an initialised torch.distributed (`group`: a process group, None = the default one) `folds(i)` supplies THIS
    rank's shards of fold i (reactranker_amd.dp.shard_query_batch); training, validation and the test evaluation are
    data-parallel (see train_listwise.train), rank 0 writes checkpoints and logs, every rank returns the same scores."""
    from .dp import Exchange
    ex = Exchange(None, group)
    os.makedirs(cfg.path, exist_ok=True)
    if logger is None:
        logger = logging.getLogger("reactranker_amd.main")
    if not ex.is_writer:
        logger = logging.getLogger("reactranker_amd.main.silent")
        logger.disabled = True
    paths = cfg.path
    if cfg.save_metric == "all":                                 # main.py:68-74
        paths = [os.path.join(cfg.path, m) for m in ("T1", "T25_in_T25", "T25")]
        for p in paths:
            os.makedirs(p, exist_ok=True)
    logger.info("Task type is: {}, and target name is: {}".format(cfg.task_type, cfg.target_name))
    logger.info("{} fold train with {} epochs every fold. The batch size is: {}".format(cfg.k_fold, cfg.total_epochs,
                                                                                     cfg.batch_size))
    test_score = []
    for ii in range(cfg.k_fold):
        logger.info("This is the fold [{}/{}]".format(ii + 1, cfg.k_fold))
        seed = ii                                                # main.py:83
        ck = [os.path.join(p, f"{ii}.pt") for p in paths] if cfg.save_metric == "all" else os.path.join(paths, f"{ii}.pt")
        train_b, val_b, test_b = folds(ii)
        torch.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
        mk = dict(cfg.model)
        if cfg.task_type == "evidential_ranking":                # main.py:123 (commented-out kwarg the user enables)
            mk.update(task_num=2, task_type="evidential_ranking")
        model = build_model(add_features_dim=cfg.add_features_dim, **mk).cuda(cfg.gpu)
        optimizer = build_optimizer(model)
        # queries of the WHOLE training set (a data-parallel batch is a shard and carries its step's global counts): every
        # rank must build the same learning-rate schedule
        n_train = sum((b.get("global") or {}).get("queries", len(b["scope"])) for b in train_b)
        scheduler = build_lr_scheduler(optimizer, warmup_epochs=cfg.warmup_epochs, total_epochs=cfg.total_epochs,
                                       train_data_size=max(n_train, cfg.batch_size), batch_size=cfg.batch_size,
                                       init_lr=cfg.init_lr, max_lr=cfg.max_lr, final_lr=cfg.final_lr)
        if cfg.task_type == "ranknet":
            run_train(model, scheduler, train_b, val_b, ck, optimizer, cfg.total_epochs, seed, cfg.gpu,
                      train_strategy=cfg.train_strategy, task_type="baseline", logger=logger,
                      target_name=cfg.target_name, save_metric=cfg.save_metric, group=group)
        else:
            train(model, scheduler, train_b, val_b, ck, optimizer, cfg.total_epochs, seed, cfg.gpu,
                  task_type=cfg.task_type, logger=logger, save_metric=cfg.save_metric, target_name=cfg.target_name,
                  normalize_target=cfg.normalize_target, group=group)
        if ex.on:                                                # rank 0's checkpoint must be on disk before anyone loads it
            import torch.distributed as dist
            dist.barrier(group=group)
        test_path = ck[0] if cfg.save_metric == "all" else ck    # main.py:164-168
        test_score.append(list(test(model, test_b, test_path, cfg.gpu, logger, cfg.target_name, exchange=ex))[:3])
    logger.info("test score for k_fold vailidation is: {}".format(test_score))
    return test_score
