"""GPU side of the packed-step pipeline: a step streamed from a shard file through the
pinned / copy-stream prefetcher, or a batch reloaded with load_packed(), drives the model to bit-identical scores,
loss and gradients as the in-memory batch it was packed from - in eval mode (reactant de-duplication) and in train
mode (shared reactant prefix + per-copy dropout streams)."""
import numpy as np
import pytest
import torch

from reactranker_amd import featurization, shards, synth
from reactranker_amd import loss as RL
from oracle import ref_cpu as O
from tests.test_gpu_model import make_model

pytestmark = pytest.mark.gpu

CFG = dict(hidden_size=64, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, task_num=1,
           ffn_last_layer="with_softplus", task_type=None, add_features_dim=1)


def _model(dropout, train):
    w = synth.seeded_weights(O.model_shapes(64, 3, 3, 3, 1, 1, True), 9)
    m = make_model(CFG, w, dropout=dropout)
    return m.train() if train else m.eval()


def _run(model, r, p, scope, targets, add, seed):
    model.zero_grad()
    model.dropout_seed = seed
    out = model(r, p, gpu=0, add_features=add)
    l = RL.MLEloss()(out, scope, targets, 0)
    l.sum().backward()
    return out.detach().clone(), l.detach().clone(), {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None}


def _same(a, b):
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert a[2].keys() == b[2].keys()
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k


def _make_steps(n, scope):
    steps = []
    for i in range(n):
        qb = synth.make_queries(40 + i, len(scope), list(scope), atoms_lo=5, atoms_hi=12)
        steps.append((qb, featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)))
    return steps


def test_fbonds_rebuilt_on_device_equals_packed_fbonds():
    qb = synth.make_queries(1, 3, [5, 3, 6], atoms_lo=5, atoms_hi=12)
    pb = featurization.BatchMolGraph(qb.p_specs, K=4)
    h = pb._host
    fb = np.zeros((h["nB"], 24), np.float32)
    fb[:, :22] = h["f_bonds"][:, 61:83]
    t = {k: torch.from_numpy(h[k]).cuda() for k in featurization._GRAPH_KEYS}
    t["fbond"] = torch.from_numpy(fb).cuda()
    g = featurization.DeviceGraph.from_device(t, h["nA"], h["nB"], h["K"], h["M"], "cuda:0")
    assert torch.equal(g.f_bonds.cpu(), torch.from_numpy(h["f_bonds"]))
    assert torch.equal(g.fb_sum().cpu(), pb.device_graph(0).fb_sum().cpu())


@pytest.mark.parametrize("train", [False, True])
def test_load_packed_batch_drives_the_model_like_the_in_memory_batch(tmp_path, train):
    qb, rb, pb = _make_steps(1, [6, 3, 8, 2])[0]
    featurization.save_packed(rb, str(tmp_path / "r.npz"))
    featurization.save_packed(pb, str(tmp_path / "p.npz"))
    rl, pl = featurization.load_packed(str(tmp_path / "r.npz")), featurization.load_packed(str(tmp_path / "p.npz"))
    model = _model(0.2 if train else 0.0, train)
    tg = torch.tensor(qb.targets)
    a = _run(model, rb, pb, qb.scope, tg, qb.add_features, 77)
    b = _run(model, rl, pl, qb.scope, tg, qb.add_features, 77)
    _same(a, b)


@pytest.mark.parametrize("train", [False, True])
def test_streamed_steps_equal_in_memory_steps_bit_for_bit(tmp_path, train):
    steps = _make_steps(7, [6, 3, 8, 2])                          # more steps than prefetch slots: every slot is recycled
    path = str(tmp_path / "epoch.rrshard")
    with shards.ShardWriter(path) as w:
        for qb, rb, pb in steps:
            w.add_step(rb, pb, qb.scope, qb.targets, qb.add_features)
    reader = shards.ShardReader(path)
    model = _model(0.15 if train else 0.0, train)
    want = [_run(model, rb, pb, qb.scope, torch.tensor(qb.targets), qb.add_features, 1000 + i)
            for i, (qb, rb, pb) in enumerate(steps)]
    order = [3, 0, 6, 1, 5, 2, 4, 3, 3, 0]
    pf = shards.StepPrefetcher(reader, "cuda:0", order, depth=3)
    seen = []
    for st in pf:
        i = st["index"]
        assert st["scope"] == steps[i][0].scope and st["meta"]["M"] == steps[i][2].n_mols
        got = _run(model, st["r"], st["p"], st["scope"], st["targets"], st["add"], 1000 + i)
        _same(want[i], got)
        seen.append(i)
    assert seen == order and pf.bytes_copied == sum(int(reader.index[i, 1]) for i in order)
    # the synchronous loader yields the same objects
    st = shards.load_step(reader, 4, "cuda:0")
    _same(want[4], _run(model, st["r"], st["p"], st["scope"], st["targets"], st["add"], 1004))


def test_streamed_step_without_dedup_builds_the_full_reactant_features(tmp_path):
    """dedup_reactants=False reads the FULL reactant graph's f_bonds, which a streamed step rebuilds lazily from the
    distinct reactants' rows (gathers through amap / bmap) and rr_build_fbonds_f32."""
    qb, rb, pb = _make_steps(1, [4, 5])[0]
    path = str(tmp_path / "s.rrshard")
    with shards.ShardWriter(path) as w:
        w.add_step(rb, pb, qb.scope, qb.targets, qb.add_features)
    st = shards.load_step(shards.ShardReader(path), 0, "cuda:0")
    model = _model(0.0, False)
    model.dedup_reactants = False
    tg = torch.tensor(qb.targets)
    _same(_run(model, rb, pb, qb.scope, tg, qb.add_features, 0), _run(model, st["r"], st["p"], st["scope"], st["targets"], st["add"], 0))
    assert torch.equal(st["r"].graph.f_bonds.cpu(), torch.from_numpy(rb._host["f_bonds"]))
