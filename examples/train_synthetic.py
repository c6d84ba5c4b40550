#!/usr/bin/env python3
"""main.py-shaped driver on synthetic reactions (the reference's main.py reads a CSV and needs RDKit; its flow is
build_model -> build_optimizer -> build_lr_scheduler -> train -> checkpoint, main.py:90-140).  Needs an MI355X.

    python examples/train_synthetic.py --task-type mle --epochs 5 --queries 256 --cands 32
"""
import argparse
import logging
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reactranker_amd import featurization, synth                      # noqa: E402
from reactranker_amd.base_model import build_model                    # noqa: E402
from reactranker_amd.train_listwise import train                      # noqa: E402
from reactranker_amd.train_utils import build_lr_scheduler, build_optimizer, param_count   # noqa: E402
from reactranker_amd.utils import load_checkpoint                     # noqa: E402


def make_batches(seed, n_queries, cands, per_batch):
    out = []
    for b0 in range(0, n_queries, per_batch):
        qb = synth.make_queries(seed + b0, min(per_batch, n_queries - b0), cands)
        # a learnable target: a fixed function of the product graph and the extra feature, distinct inside a query
        tg = np.array([s.edges.shape[0] for s in qb.p_specs], np.float32) * 0.3 + qb.add_features[:, 0]
        tg = (tg - tg.mean()) / (tg.std() + 1e-6) + 1e-3 * np.arange(len(tg), dtype=np.float32)
        out.append(dict(r=featurization.BatchMolGraph(qb.r_specs, K=4), p=featurization.BatchMolGraph(qb.p_specs, K=4),
                        scope=qb.scope, targets=torch.tensor(tg.astype(np.float32)), add=qb.add_features))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task-type", default="mle")
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--cands", type=int, default=32)
    ap.add_argument("--batch-queries", type=int, default=32)
    ap.add_argument("--hidden", type=int, default=300)
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--checkpoint", default="/tmp/reactranker_amd_synthetic/model.pt")
    args = ap.parse_args()
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    log = logging.getLogger("train_synthetic")
    task_num = 2 if args.task_type in ("evidential_ranking", "gauss_regression", "mle_gaussian", "listnet_gauss") else 1
    model = build_model(hidden_size=args.hidden, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, dropout=0.1,
                        task_num=task_num, ffn_last_layer="with_softplus" if task_num == 1 else "no_softplus",
                        task_type=args.task_type if args.task_type == "evidential_ranking" else None, add_features_dim=1)
    log.info("parameters: %d", param_count(model))
    train_b = make_batches(0, args.queries, args.cands, args.batch_queries)
    val_b = make_batches(10 ** 6, max(args.batch_queries, args.queries // 8), args.cands, args.batch_queries)
    opt = build_optimizer(model.cuda(args.gpu))
    sch = build_lr_scheduler(opt, warmup_epochs=2, total_epochs=args.epochs, train_data_size=args.queries,
                             batch_size=args.batch_queries, init_lr=1e-4, max_lr=1e-3, final_lr=1e-4)
    rng = np.random.default_rng(0)
    hist = train(model, sch, lambda ep: [train_b[i] for i in rng.permutation(len(train_b))], val_b, args.checkpoint, opt,
                 args.epochs, seed=0, gpu=args.gpu, task_type=args.task_type, logger=log, save_metric="NDCG@all")
    best = max(hist, key=lambda h: h["ndcg"][3])
    log.info("best epoch %d: NDCG@all %.4f top1 %.4f", best["epoch"], best["ndcg"][3], best["top1"])
    load_checkpoint(args.checkpoint, model)
    log.info("checkpoint %s restored", args.checkpoint)


if __name__ == "__main__":
    main()
