"""SURVEY.md section 8 f-4 on the GPU against the oracle: the trainer mirrors (reactranker_amd.train_listwise.train,
reactranker_amd.run_train_pairwise.run_train) run a short training on the HIP path, and the SAME loop - forward, loss,
zero_grad / backward / Adam step / NoamLR step, per-epoch validation metrics, checkpoint decision - runs on
oracle/ref_cpu.py with torch's Adam on the CPU.  Compared per epoch: the training loss (1e-4 relative), the validation
SCORES of the two models (1e-3), the trainer's metrics against the reference's metric code on those scores, the
checkpoint decisions against the reference's rule - and against the oracle loop's metrics / decisions wherever the
ranking is well conditioned (see _compare_epochs).  Dropout is 0 so both sides see the same arithmetic (train-mode masks
are pinned separately, tests/test_gpu_model.py / test_gpu_headline_kernels.py).
Reference control flow: train/train_listwise.py:176-354, train/run_train_pairwise.py:59-117, train/train_pairwise.py:81-173."""
import os

import numpy as np
import pytest
import torch

from reactranker_amd import featurization, synth
from reactranker_amd import train_listwise as TL, run_train_pairwise as RP, train_utils as TU
from reactranker_amd.base_model import build_model
from reactranker_amd.utils import load_checkpoint
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu

SCHED = dict(warmup_epochs=1, total_epochs=3, train_data_size=4 * 6, batch_size=6, init_lr=5e-4, max_lr=2e-3, final_lr=5e-4)


def _data(seed0, n_batches, nq, nc):
    """Packed batches for the HIP path + the same queries as oracle graphs; learnable targets (a fixed function of the
    product graph and the extra feature) so three epochs move the loss."""
    hip, ora = [], []
    for i in range(n_batches):
        qb = synth.make_queries(seed0 + i, nq, nc, atoms_lo=6, atoms_hi=12)
        tg = np.array([s.edges.shape[0] for s in qb.p_specs], np.float32) * 0.3 + qb.add_features[:, 0]
        tg = ((tg - tg.mean()) / (tg.std() + 1e-6) + 1e-3 * np.arange(len(tg), dtype=np.float32)).astype(np.float32)
        hip.append(dict(r=featurization.BatchMolGraph(qb.r_specs, K=4), p=featurization.BatchMolGraph(qb.p_specs, K=4),
                        scope=qb.scope, targets=torch.tensor(tg), add=qb.add_features))
        ora.append(dict(r=O.graph_tensors(O.pack_batch(qb.r_specs, K=4)), p=O.graph_tensors(O.pack_batch(qb.p_specs, K=4)),
                        scope=qb.scope, targets=torch.tensor(tg), add=qb.add_features))
    return hip, ora


def _oracle_val_scores(P, mc, batches):
    """First score column of the oracle model on the validation batches, one float64 array per batch."""
    out = []
    with torch.no_grad():
        for b in batches:
            o = O.reaction_forward(P, mc, b["r"], b["p"], b["add"])
            out.append((o[:, 0] if o.dim() > 1 else o).double().numpy().copy())
    return out


def _hip_val_scores(model, batches):
    was = model.training
    model.eval()
    out = []
    with torch.no_grad():
        for b in batches:
            o = model(b["r"], b["p"], gpu=0, add_features=b["add"])
            out.append((o[:, 0] if o.dim() > 1 else o).double().cpu().numpy().copy())
    model.train(was)
    return out


def _metrics(scores, batches):
    """ranking_metrics (train/eval.py:475-555) of given validation scores, by the oracle's restatement of it."""
    sq, tq = [], []
    for s, b in zip(scores, batches):
        off = 0
        for n in b["scope"]:
            sq.append(s[off:off + n].tolist())
            tq.append(b["targets"][off:off + n].tolist())
            off += n
    top1, recall25, top25, nd, _ = O.ranking_metrics_from_scores(sq, tq)
    return dict(top1=top1, recall25=recall25, top25=top25, nd=np.asarray(nd))


def _min_gap(scores, batches):
    """Smallest gap between two candidates of one validation query: a score perturbation below half of it cannot change
    any ranking metric."""
    g = np.inf
    for s, b in zip(scores, batches):
        off = 0
        for n in b["scope"]:
            if n > 1:
                g = min(g, float(np.min(np.diff(np.sort(s[off:off + n])))))
            off += n
    return g


def _centered(s, scope):
    out, off = s.copy(), 0
    for n in scope:
        out[off:off + n] -= out[off:off + n].mean()
        off += n
    return out


def _decisions(top1_seq):
    """save_metric None: checkpoint whenever top-1 does not get worse (train_listwise.py:310-316)."""
    old, out = 0.0, []
    for t in top1_seq:
        out.append(t >= old)
        if t >= old:
            old = t
    return out


def _compare_epochs(hist, hip_scores, ora_scores, ora_losses, val_batches, has_ndcg):
    """Per epoch: training loss 1e-4 relative; validation scores of the two trajectories within 1e-3; the trainer's metrics
    == the reference's metric code applied to ITS scores (exact) and its checkpoint decisions == the reference's rule on
    those metrics; and wherever the oracle's validation ranking is robust against the score difference (every score gap
    inside a query above twice that difference) the metrics and decisions equal the oracle loop's as well.  (With lists
    of 12 near-identical products an untrained model leaves score gaps of 1e-5..1e-4 between candidates - below what two
    fp32 training trajectories can agree on - so an unconditional comparison of rank metrics would test luck.)"""
    robust_all, n_robust = True, 0
    m_hip_seq = []
    for e, (h, sh, so, lo) in enumerate(zip(hist, hip_scores, ora_scores, ora_losses)):
        rel = abs(h["train_loss"] - lo) / max(1e-6, abs(lo))
        assert rel <= 1e-4, (e, h["train_loss"], lo)
        # scores are compared per query up to a common offset: every loss here sees score DIFFERENCES inside a list only,
        # so the output bias has an analytically zero gradient - pure rounding noise, on which Adam still moves by +-lr per
        # step - and the two trajectories' absolute scores drift apart by a per-model constant that no ranking, loss or
        # metric can see
        d = max(float(np.abs(_centered(a, b["scope"]) - _centered(o, b["scope"])).max()) for a, o, b in zip(sh, so, val_batches))
        scale = max(float(np.abs(_centered(o, b["scope"])).max()) for o, b in zip(so, val_batches))
        assert d <= 1e-3 * (1.0 + scale), (e, d)
        mh, mo = _metrics(sh, val_batches), _metrics(so, val_batches)
        m_hip_seq.append(mh)
        assert abs(h["top1"] - mh["top1"]) < 1e-9 and abs(h["top1_in_pred_top25"] - mh["top25"]) < 1e-9
        assert abs(h["pred_top25_in_targ_top25"] - mh["recall25"]) < 1e-9
        if has_ndcg:
            assert np.allclose(h["ndcg"], mh["nd"], rtol=0, atol=1e-6)
        robust = _min_gap(so, val_batches) > 2.0 * d
        robust_all = robust_all and robust
        n_robust += int(robust)
        print(f"[trajectory] epoch {e + 1}: loss rel err {rel:.1e}, max |score diff| {d:.1e}, min oracle score gap "
              f"{_min_gap(so, val_batches):.1e} -> ranking {'robust' if robust else 'ill-conditioned'}")
        if robust:
            assert abs(mh["top1"] - mo["top1"]) < 1e-9 and abs(mh["top25"] - mo["top25"]) < 1e-9 and abs(mh["recall25"] - mo["recall25"]) < 1e-9
    assert [h["checkpoint"] for h in hist] == _decisions([m["top1"] for m in m_hip_seq])
    if robust_all:
        assert [h["checkpoint"] for h in hist] == _decisions([_metrics(so, val_batches)["top1"] for so in ora_scores])
    return n_robust


def _cfg(task_num, task_type):
    return dict(hidden_size=64, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, task_num=task_num,
                ffn_last_layer="with_softplus" if task_num == 1 else "no_softplus", task_type=task_type, add_features_dim=1)


def _oracle_side(cfg, w):
    P = O.params_from_numpy(w, requires_grad=True)
    opt = torch.optim.Adam([{"params": [p for p in P.values() if p.requires_grad], "lr": 1e-4, "weight_decay": 0}])
    sch = TU.build_lr_scheduler(opt, **SCHED)
    mc = dict(depth=cfg["mpnn_depth"], diff_depth=cfg["mpnn_diff_depth"], ffn_depth=cfg["ffn_depth"],
              task_type=O.resolve_task_type(cfg["task_num"], cfg["ffn_last_layer"], cfg["task_type"]))
    return P, opt, sch, mc


def _hip_side(cfg, w):
    model = build_model(dropout=0.0, **cfg)
    model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    model = model.cuda()
    opt = TU.build_optimizer(model)
    return model, opt, TU.build_lr_scheduler(opt, **SCHED)


@pytest.mark.parametrize("task_type,task_num", [("mle", 1), ("evidential_ranking", 2)])
def test_listwise_trainer_trajectory_matches_the_oracle_loop(tmp_path, task_type, task_num):
    cfg = _cfg(task_num, "evidential_ranking" if task_type == "evidential_ranking" else None)
    shapes = O.model_shapes(64, 3, 3, 3, task_num, 1, True)
    w = synth.seeded_weights(shapes, 21)
    hip_tr, ora_tr = _data(4000, 4, 6, 12)
    hip_va, ora_va = _data(4100, 2, 6, 12)
    epochs = 3
    model, opt, sch = _hip_side(cfg, w)
    path = str(tmp_path / "ck" / "model.pt")
    hip_scores = []
    hist = TL.train(model, sch, hip_tr, hip_va, path, opt, epochs, seed=5, gpu=0, task_type=task_type, save_metric=None,
                    epoch_hook=lambda e, m, rec: hip_scores.append(_hip_val_scores(m, hip_va)))

    P, o_opt, o_sch, mc = _oracle_side(cfg, w)
    ora_scores, ora_losses = [], []
    for epoch in range(epochs):
        loss = None
        for b in ora_tr:
            out = O.reaction_forward(P, mc, b["r"], b["p"], b["add"])
            if task_type == "mle":
                loss = O.listmle_loss(out, b["scope"], b["targets"])
            else:
                loss = O.evidential_ranking_loss(out, b["scope"], b["targets"])
            o_opt.zero_grad()
            loss.sum().backward()
            o_opt.step()
            o_sch.step()
        ora_losses.append(float(loss.detach().sum()))
        ora_scores.append(_oracle_val_scores(P, mc, ora_va))
    assert sch.current_step == o_sch.current_step and abs(opt.param_groups[0]["lr"] - o_opt.param_groups[0]["lr"]) < 1e-12
    assert ora_losses[-1] < ora_losses[0]                                   # the epochs really trained
    _compare_epochs(hist, hip_scores, ora_scores, ora_losses, ora_va, True)
    # the checkpoint on disk is the HIP model after its last saving epoch; with a save at the last epoch it equals the
    # oracle's parameters to training accuracy
    assert os.path.exists(path)
    if hist[-1]["checkpoint"]:
        m2 = build_model(dropout=0.0, **cfg).cuda().eval()
        load_checkpoint(path, m2)
        for k, v in m2.state_dict().items():
            if k.endswith("cached_zero_vector"):
                continue
            ref = P[k].detach()
            if k == "ffn.ffn.7.bias" and task_type == "mle":
                continue        # ListMLE is invariant to a score offset: this gradient is pure rounding noise, and Adam moves by +-lr on noise
            assert float((v.cpu() - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 2e-5, k


def test_ranknet_trainer_trajectory_matches_the_oracle_loop(tmp_path):
    cfg = dict(_cfg(1, None), ffn_last_layer="no_softplus")
    shapes = O.model_shapes(64, 3, 3, 3, 1, 1, True)
    w = synth.seeded_weights(shapes, 22)
    hip_tr, ora_tr = _data(5000, 4, 6, 12)
    hip_va, ora_va = _data(5100, 2, 6, 12)
    epochs = 3
    model, opt, sch = _hip_side(cfg, w)
    path = str(tmp_path / "ck" / "rank.pt")
    hip_scores = []
    hist = RP.run_train(model, sch, hip_tr, hip_va, path, opt, epochs, seed=5, gpu=0, train_strategy="sum_session",
                        target_name=None, save_metric=None,
                        epoch_hook=lambda e, m, rec: hip_scores.append(_hip_val_scores(m, hip_va)))
    P, o_opt, o_sch, mc = _oracle_side(cfg, w)
    ora_scores, ora_losses = [], []
    for epoch in range(epochs):
        losses = []
        for b in ora_tr:
            y = O.reaction_forward(P, mc, b["r"], b["p"], b["add"])
            ls, pairs = O.ranknet_sum_session(y, b["scope"], b["targets"], 1.0)      # train_pairwise.py:99-122
            if int(pairs) == 0:
                continue
            loss = ls / pairs                                                          # :147
            losses.append(float(loss.detach()))
            loss.backward()
            o_opt.step()
            o_opt.zero_grad()
            o_sch.step()
        ora_losses.append(float(np.mean(losses)))
        ora_scores.append(_oracle_val_scores(P, mc, ora_va))
    assert ora_losses[-1] < ora_losses[0]
    _compare_epochs(hist, hip_scores, ora_scores, ora_losses, ora_va, False)
    assert os.path.exists(path)
