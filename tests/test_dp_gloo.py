"""Data-parallel path on CPU: 2 processes over gloo (the GPU job uses the same code with the
"nccl" = RCCL backend).  Proves SURVEY.md section 8e: with the per-loss weights of
reactranker_amd.dp.loss_weight, the all-reduced gradient equals the single-process gradient for
every loss normalisation (per-query mean, per-candidate mean, per-pair mean), with ragged shards.
The math runs on the CPU oracle here (the HIP kernels need a GPU); what is under test is the
sharding + bucket + weighting logic that bench.py and a trainer use."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as O
from reactranker_amd import synth
from reactranker_amd.dp import GradBucket, loss_weight, shard_queries

CFG = dict(depth=2, diff_depth=2, ffn_depth=2, task_type="no_softplus")
SCOPE = [5, 3, 7, 2, 6]          # ragged lists; shards get 3 and 2 queries
H = 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _loss(kind, out, scope, targets):
    if kind == "mle":
        return O.listmle_loss(out, scope, targets).sum(), None
    if kind == "listnet":
        return O.listnet_loss(out, scope, targets), None
    if kind == "mse":
        return O.mse_loss(out, targets), None
    if kind == "evidential_ranking":                             # loss.py:554: sum over queries / number of queries
        return O.evidential_ranking_loss(out, scope, targets).sum(), None
    ls, pairs = O.ranknet_sum_session(out, scope, targets, 1.0)
    return ls, pairs


def _grads(kind, w, qb, lo, hi, K):
    m0, m1 = sum(qb.scope[:lo]), sum(qb.scope[:hi])
    # fp64 so the identity under test (weighting / sharding) is not blurred by fp32 summation order
    P = {k: v.double().requires_grad_(v.requires_grad) for k, v in O.params_from_numpy(w, requires_grad=True).items()}
    names = [k for k in P if P[k].requires_grad]

    def g64(specs):
        g = O.graph_tensors(O.pack_batch(specs, K=K))
        g["f_atoms"], g["f_bonds"] = g["f_atoms"].double(), g["f_bonds"].double()
        return g
    cfg = dict(CFG, task_type="evidential_ranking") if kind == "evidential_ranking" else CFG
    out = O.reaction_forward(P, cfg, g64(qb.r_specs[m0:m1]), g64(qb.p_specs[m0:m1]),
                             torch.tensor(qb.add_features[m0:m1]).double())
    loss, pairs = _loss(kind, out, qb.scope[lo:hi], torch.tensor(qb.targets[m0:m1]).double())
    if kind == "ranknet":
        loss = loss / max(pairs, 1.0)            # train_pairwise.py:147 (per accumulation window)
    g = torch.autograd.grad(loss, [P[k] for k in names], allow_unused=True)
    return names, [torch.zeros_like(P[k]) if x is None else x for k, x in zip(names, g)], pairs, (m1 - m0)


def _worker(rank, world, port, kind, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        shapes = O.model_shapes(H, 2, 2, 2, 2 if kind == "evidential_ranking" else 1, 1, True)
        w = synth.seeded_weights(shapes, 3)
        qb = synth.make_queries(99, len(SCOPE), SCOPE, atoms_lo=4, atoms_hi=8)
        K = 4                                                    # global pad width on every rank (hazard H1)
        names, full, pairs_all, m_all = _grads(kind, w, qb, 0, len(SCOPE), K)
        lo, hi = shard_queries(len(SCOPE), rank, world)
        _, local, pairs_loc, m_loc = _grads(kind, w, qb, lo, hi, K)
        params = [torch.nn.Parameter(torch.zeros_like(g, dtype=torch.float32)) for g in local]
        for p, g in zip(params, local):
            p.grad = g.float()                                   # the bucket itself is fp32, like the GPU job's
        wgt = loss_weight(kind, hi - lo, len(SCOPE), m_loc, m_all, pairs_loc, pairs_all)
        GradBucket(params).allreduce(wgt)
        # relative to each tensor's largest entry, with an absolute floor for gradients that are analytically
        # zero (ListMLE's output bias: sum_j dL/ds_j = 0 per list -> pure rounding noise)
        err = max(float((p.grad.double() - f).abs().max() / (f.abs().max() + 1e-6)) for p, f in zip(params, full))
        ret[rank] = err
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["mle", "listnet", "mse", "ranknet", "evidential_ranking"])
def test_weighted_allreduce_equals_single_process_gradient(kind):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), kind, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert ret[r] < 1e-5, (kind, dict(ret))      # only the bucket's fp32 rounding remains


def test_shard_queries_covers_everything_without_splitting():
    for n in (1, 5, 8, 13, 64):
        for world in (1, 2, 3, 8):
            spans = [shard_queries(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bucket_single_process_is_a_scale():
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    GradBucket([p]).allreduce(0.5)
    assert torch.equal(p.grad, torch.ones(3))


def _mixed_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # an attached bucket with three kinds of parameter: one whose gradient was BORN in the bucket (the explicit
        # backward's sink), one whose gradient autograd produced elsewhere (an extra head), one the step did not touch
        a, b, c = (torch.nn.Parameter(torch.zeros(n)) for n in (5, 3, 4))
        bucket = GradBucket([a, b, c]).attach()
        try:
            va = bucket.sink(a)
            assert va is not None and va.data_ptr() == bucket.flat.data_ptr()
            va.copy_(torch.arange(5.0) + rank)
            a.grad = va
            b.grad = torch.full((3,), 10.0 + rank)               # not a view of the flat buffer
            bucket.allreduce(0.5)
        finally:
            bucket.detach()
        ok = (torch.equal(a.grad, torch.arange(5.0) + 0.5) and torch.equal(b.grad, torch.full((3,), 10.5)) and
              torch.equal(c.grad, torch.zeros(4)) and b.grad.data_ptr() == bucket._views[1].data_ptr())
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_attached_bucket_with_a_gradient_from_outside_the_explicit_backward():
    """the pack fallback ran `cat(..., out=flat)` over gradients some of which already alias `flat`
    (torch refuses overlapping input/output); the mixed case now copies only the outsiders into their views."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_mixed_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


# ---------------------------------------------------------------------------------------------- trainer-side glue
def test_count_pairs_is_the_ranknet_normaliser():
    from reactranker_amd.dp import count_pairs, step_counts
    rng = np.random.default_rng(3)
    scope = [1, 2, 7, 5]
    t = np.round(rng.standard_normal(sum(scope)), 1).astype(np.float32)       # rounded: ties do not count
    s = torch.tensor(rng.standard_normal(sum(scope)).astype(np.float32))
    _, pairs = O.ranknet_sum_session(s, scope, torch.tensor(t), 1.0)          # train_pairwise.py:99-106
    assert count_pairs(scope, t) == int(pairs) == count_pairs(scope, torch.tensor(t))
    assert step_counts(scope, t) == dict(queries=4, cands=15, pairs=int(pairs))


def test_shard_query_batch_partitions_whole_queries_and_keeps_the_global_counts():
    from reactranker_amd.dp import shard_query_batch, step_counts
    qb = synth.make_queries(5, len(SCOPE), SCOPE, atoms_lo=4, atoms_hi=8)
    for world in (1, 2, 3, 8):
        parts = [shard_query_batch(qb, r, world) for r in range(world)]
        assert all(g == step_counts(qb.scope, qb.targets) for _, g in parts)
        assert sum((p.scope for p, _ in parts), []) == list(qb.scope)
        assert np.array_equal(np.concatenate([p.targets for p, _ in parts]), qb.targets)
        assert sum(len(p.p_specs) for p, _ in parts) == len(qb.p_specs)
        for p, _ in parts:
            assert len(p.r_specs) == len(p.p_specs) == sum(p.scope) == len(p.add_features)
    assert [len(p.scope) for p, _ in [shard_query_batch(qb, r, 8) for r in range(8)]].count(0) == 3      # empty shards exist


def _exchange_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from reactranker_amd.dp import Exchange, shard_query_batch
        from reactranker_amd.eval import _mean_stats
        ex = Exchange(None)
        assert ex.on and ex.world == world and ex.rank == rank and ex.is_writer == (rank == 0)
        rng = np.random.default_rng(1)
        stats = torch.tensor(rng.standard_normal((7, 12)))                    # 7 queries, ragged over the ranks: 4 + 3
        lo, hi = shard_queries(7, rank, world)
        m = _mean_stats(stats[lo:hi], ex)
        ok = np.allclose(m, stats.mean(dim=0).numpy(), rtol=0, atol=1e-14)
        qb = synth.make_queries(5, len(SCOPE), SCOPE, atoms_lo=4, atoms_hi=8)
        mine, glob = shard_query_batch(qb, rank, world)
        b = dict(scope=mine.scope, targets=mine.targets)                      # no `global`: the counts are all-reduced
        local, g2 = ex.counts(b, "cpu")
        ok = ok and g2 == glob and local["queries"] == len(mine.scope)
        w = [ex.weight(k, local, g2) for k in ("mle", "listnet", "ranknet", "evidential_ranking", "regression")]
        tot = ex.sum(torch.tensor(w, dtype=torch.float64))
        ok = ok and bool(torch.allclose(tot, torch.ones(5, dtype=torch.float64), atol=1e-12))     # the weights partition 1
        ex.check_same_steps(7, "cpu")                                          # equal everywhere: passes
        try:
            ex.check_same_steps(7 + rank, "cpu")                               # one rank holds a step more: every rank raises
            ok = False
        except RuntimeError:
            pass
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_exchange_sums_validation_statistics_and_partitions_the_loss_weights():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_exchange_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_exchange_without_a_process_group_is_the_identity():
    from reactranker_amd.dp import Exchange
    ex = Exchange(None)
    assert not ex.on and ex.world == 1 and ex.is_writer
    t = torch.arange(3.0)
    assert ex.sum(t) is t
    local, glob = ex.counts(dict(scope=[2, 3], targets=np.array([1., 2., 3., 1., 2.])), "cpu")
    assert local == glob == dict(queries=2, cands=5, pairs=2 * (1 + 3))
    ex.reduce_grads(0.5)
    ex.close()
