"""A short training job through the trainer mirrors, for 1 or N processes (not a test module: tests/test_gpu_dp_trainers.py
starts it - once as a plain process, once under `python -m torch.distributed.run --nproc-per-node 2` - and compares the
histories rank 0 writes).  Every rank builds the SAME global steps (seeded), keeps its contiguous block of whole queries
of each (reactranker_amd.dp.shard_query_batch: ragged lists, so the shards are ragged too) and hands the trainer its
shards.  Backend: $RR_DIST_BACKEND (default "nccl" = RCCL); RR_SINGLE_DEVICE=1 puts every rank on GPU 0 (a one-GPU box)."""
import argparse
import json
import logging
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

SCOPES = [[5, 3, 7, 2, 6], [4, 8, 3, 6, 5, 2], [9, 2, 4], [6, 6, 1, 3, 7]]       # global steps: ragged lists
VAL_SCOPES = [[5, 4, 7], [3, 8, 2, 6]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kinds", default="mle,listnet,ranknet,evidential_ranking")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--out", required=True)
    ap.add_argument("--ckdir", required=True)
    args = ap.parse_args()
    import torch.distributed as dist
    from reactranker_amd import dp, featurization, synth
    from reactranker_amd import main as RM
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("RR_SINGLE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group(os.environ.get("RR_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    def batches(seed0, scopes):
        out = []
        for i, scope in enumerate(scopes):
            qb = synth.make_queries(seed0 + i, len(scope), scope, atoms_lo=6, atoms_hi=12)
            # learnable, well separated raw targets ('ea': lower is better, the trainers flip and z-score them)
            tg = np.array([s.edges.shape[0] for s in qb.p_specs], np.float32) * 0.7 + 3.0 * qb.add_features[:, 0]
            qb.targets = (tg + 0.05 * np.arange(len(tg), dtype=np.float32)).astype(np.float32)
            mine, glob = dp.shard_query_batch(qb, rank, world)
            b = dict(scope=mine.scope, targets=torch.tensor(mine.targets), add=mine.add_features, **{"global": glob})
            if len(mine.scope):
                b["r"] = featurization.BatchMolGraph(mine.r_specs, K=4)        # global pad width on every rank (hazard H1)
                b["p"] = featurization.BatchMolGraph(mine.p_specs, K=4)
            else:
                b["r"] = b["p"] = None
            out.append(b)
        return out

    result = {}
    for kind in args.kinds.split(","):
        cfg = RM.Config(path=os.path.join(args.ckdir, kind), k_fold=1, total_epochs=args.epochs, batch_size=5,
                        task_type=kind, train_strategy="sum_session", target_name="ea", normalize_target=True,
                        init_lr=1e-4, max_lr=4e-4, final_lr=1e-4, warmup_epochs=1.0, save_metric="all", add_features_dim=1,
                        gpu=local, model=dict(hidden_size=64, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True,
                                              dropout=0.0, task_num=1,
                                              ffn_last_layer="no_softplus" if kind == "ranknet" else "with_softplus"))
        hist = []
        from reactranker_amd import run_train_pairwise as RP, train_listwise as TL
        orig_train, orig_run = TL.train, RP.run_train

        def tap(fn):
            def wrapped(*a, **k):
                h = fn(*a, **k)
                hist.extend(h)
                return h
            return wrapped
        RM.train, RM.run_train = tap(orig_train), tap(orig_run)
        try:
            scores = RM.run(cfg, lambda i: (batches(7000, SCOPES), batches(7100, VAL_SCOPES), batches(7200, VAL_SCOPES)),
                            logger=logging.getLogger("dp_job"), group=None)
        finally:
            RM.train, RM.run_train = orig_train, orig_run
        result[kind] = dict(history=hist, test=scores)
    if rank == 0:
        with open(args.out, "w") as f:
            json.dump(dict(world=world, backend=os.environ.get("RR_DIST_BACKEND", "nccl") if world > 1 else None,
                           result=result), f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
