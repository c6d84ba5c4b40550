#!/usr/bin/env python3
"""The same short training (examples/train_synthetic.py's flow: build_model -> build_optimizer -> NoamLR -> train_listwise.train)
in the three GEMM arithmetics of the step plans - two f16 terms (default), three bf16 terms, exact-f32 MFMA - from the same
initial weights, batches, shuffles and dropout streams: per-epoch training loss and validation metrics side by side.
What differs between the columns is rounding (and what Adam makes of it), not the training.
Usage (GPU box): python tools/train_curves.py [--task-type mle] [--epochs 12] [--queries 256] [--cands 32] [--hidden 300]"""
import argparse
import copy
import logging
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from reactranker_amd import functions as Fn                            # noqa: E402
from reactranker_amd.base_model import build_model                    # noqa: E402
from reactranker_amd.train_listwise import train                      # noqa: E402
from reactranker_amd.train_utils import build_lr_scheduler, build_optimizer   # noqa: E402
from train_synthetic import make_batches                              # noqa: E402


def run_modes(a, seed):
    task_num = 2 if a.task_type == "evidential_ranking" else 1
    torch.manual_seed(seed)
    proto = build_model(hidden_size=a.hidden, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, dropout=0.1,
                        task_num=task_num, ffn_last_layer="with_softplus" if task_num == 1 else "no_softplus",
                        task_type=a.task_type if a.task_type == "evidential_ranking" else None, add_features_dim=1)
    train_b = make_batches(0, a.queries, a.cands, a.batch_queries)
    val_b = make_batches(10 ** 6, max(a.batch_queries, a.queries // 8), a.cands, a.batch_queries)
    log = logging.getLogger("train_curves")
    hists = {}
    for name, split, f16 in MODES:
        Fn.SplitGemm.enabled, Fn.SplitGemm.f16 = split, f16
        model = copy.deepcopy(proto).cuda(0)
        opt = build_optimizer(model)
        sch = build_lr_scheduler(opt, warmup_epochs=2, total_epochs=a.epochs, train_data_size=a.queries,
                                 batch_size=a.batch_queries, init_lr=1e-4, max_lr=1e-3, final_lr=1e-4)
        rng = np.random.default_rng(seed)
        hists[name] = train(model, sch, lambda ep: [train_b[i] for i in rng.permutation(len(train_b))], val_b,
                            f"/tmp/rr_train_curves_{name}.pt", opt, a.epochs, seed=seed, gpu=0, task_type=a.task_type, logger=log,
                            save_metric="NDCG@all")
    Fn.SplitGemm.enabled, Fn.SplitGemm.f16 = True, True
    return hists


MODES = [("f16x2", True, True), ("bf16x3", True, False), ("f32", False, False)]


def over_seeds(a):
    logging.basicConfig(level=logging.WARNING)
    rows = []
    for seed in range(a.seeds):
        h = run_modes(a, seed)
        rows.append([(h[n][-1]["train_loss"], h[n][-1]["ndcg"][3], h[n][-1]["top1"], max(x["ndcg"][3] for x in h[n])) for n, _, _ in MODES])
    print(f"{a.task_type}, {a.queries} queries x {a.cands} candidates, H = {a.hidden}, {a.epochs} epochs, {a.seeds} seeds (initialisation, shuffles, "
          f"dropout streams); last epoch per seed, columns: f16x2 | bf16x3 | f32")
    print(f"{'seed':>4s}  {'last training loss':^32s}  {'validation NDCG@all':^29s}  {'best NDCG@all of the run':^29s}")
    for i, r in enumerate(rows):
        print(f"{i:4d}  " + " | ".join(f"{x[0]:9.5f}" for x in r) + "  " + " | ".join(f"{x[1]:.6f}" for x in r) + "  " + " | ".join(f"{x[3]:.6f}" for x in r))
    arr = np.array(rows)                                  # [seed, mode, stat]
    print("mean  " + " | ".join(f"{v:9.5f}" for v in arr[:, :, 0].mean(0)) + "  " + " | ".join(f"{v:.6f}" for v in arr[:, :, 1].mean(0)) +
          "  " + " | ".join(f"{v:.6f}" for v in arr[:, :, 3].mean(0)))
    print("std   " + " | ".join(f"{v:9.5f}" for v in arr[:, :, 0].std(0)) + "  " + " | ".join(f"{v:.6f}" for v in arr[:, :, 1].std(0)) +
          "  " + " | ".join(f"{v:.6f}" for v in arr[:, :, 3].std(0)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task-type", default="mle")
    ap.add_argument("--epochs", type=int, default=12)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--cands", type=int, default=32)
    ap.add_argument("--batch-queries", type=int, default=32)
    ap.add_argument("--hidden", type=int, default=300)
    ap.add_argument("--seeds", type=int, default=0, help="> 0: repeat over that many initialisations / dropout streams and print each "
                    "arithmetic's last-epoch numbers per seed with mean and spread (is a difference between columns more than seed noise?)")
    a = ap.parse_args()
    if a.seeds > 0:
        return over_seeds(a)
    logging.basicConfig(level=logging.WARNING)
    log = logging.getLogger("train_curves")
    task_num = 2 if a.task_type == "evidential_ranking" else 1
    torch.manual_seed(0)
    proto = build_model(hidden_size=a.hidden, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, dropout=0.1,
                        task_num=task_num, ffn_last_layer="with_softplus" if task_num == 1 else "no_softplus",
                        task_type=a.task_type if a.task_type == "evidential_ranking" else None, add_features_dim=1)
    train_b = make_batches(0, a.queries, a.cands, a.batch_queries)
    val_b = make_batches(10 ** 6, max(a.batch_queries, a.queries // 8), a.cands, a.batch_queries)
    modes = [("f16x2", True, True), ("bf16x3", True, False), ("f32", False, False)]
    hists = {}
    for name, split, f16 in modes:
        Fn.SplitGemm.enabled, Fn.SplitGemm.f16 = split, f16
        model = copy.deepcopy(proto).cuda(0)
        opt = build_optimizer(model)
        sch = build_lr_scheduler(opt, warmup_epochs=2, total_epochs=a.epochs, train_data_size=a.queries,
                                 batch_size=a.batch_queries, init_lr=1e-4, max_lr=1e-3, final_lr=1e-4)
        rng = np.random.default_rng(0)
        hists[name] = train(model, sch, lambda ep: [train_b[i] for i in rng.permutation(len(train_b))], val_b,
                            f"/tmp/rr_train_curves_{name}.pt", opt, a.epochs, seed=0, gpu=0, task_type=a.task_type, logger=log,
                            save_metric="NDCG@all")
    Fn.SplitGemm.enabled, Fn.SplitGemm.f16 = True, True
    print(f"{a.task_type}, {a.queries} queries x {a.cands} candidates, {a.batch_queries} queries per step, H = {a.hidden}, dropout 0.1, "
          f"{a.epochs} epochs; columns: f16x2 | bf16x3 | f32")
    print(f"{'epoch':>5s}  {'training loss':^38s}  {'validation NDCG@all':^29s}  {'validation top-1':^23s}")
    for e in range(a.epochs):
        h = [hists[n][e] for n, _, _ in modes]
        print(f"{e + 1:5d}  " + " | ".join(f"{x['train_loss']:11.7f}" for x in h) + "  " +
              " | ".join(f"{x['ndcg'][3]:.6f}" for x in h) + "  " + " | ".join(f"{x['top1']:.4f}" for x in h))
    ref = hists["f32"]
    for n in ("f16x2", "bf16x3"):
        dl = max(abs(x["train_loss"] - y["train_loss"]) / max(1e-9, abs(y["train_loss"])) for x, y in zip(hists[n], ref))
        dn = max(abs(x["ndcg"][3] - y["ndcg"][3]) for x, y in zip(hists[n], ref))
        print(f"{n} against f32: largest relative difference of an epoch's training loss {dl:.2e}, largest |NDCG@all difference| {dn:.2e}")


if __name__ == "__main__":
    main()
