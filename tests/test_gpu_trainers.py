"""SURVEY.md section 8 f-4 on the GPU against the oracle: the trainer mirrors (reactranker_amd.train_listwise.train,
reactranker_amd.run_train_pairwise.run_train) run a short training on the HIP path, and the SAME loop - forward, loss,
zero_grad / backward / Adam step / NoamLR step, per-epoch validation metrics, checkpoint decision - runs on
oracle/ref_cpu.py with torch's Adam on the CPU.  Compared per epoch: the training loss (1e-4 relative), the validation
SCORES of the two models against the fp64 oracle loop (bounded by a multiple of what the fp32 ORACLE loop itself loses
against fp64 - see _compare_epochs; the measured distances go to the parity log), the trainer's metrics against the
reference's metric code on those scores (ranking_metrics for the listwise trainer, evaluate_top_scores for the RankNet
driver - both pinned to the reference's own functions by tests/golden/{eval_metrics,top_scores}.npz), the checkpoint
decisions against the reference's rule - and against the oracle loop's metrics / decisions wherever the ranking is well
conditioned.  A single optimizer step is held tight separately (test_first_optimizer_step_matches_the_oracle).  Dropout is 0 so both sides see the same arithmetic (train-mode masks
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

# learning rates inside the reference's range (main.py:27-29: 1e-4 ... 1e-3).  Two fp32 evaluations of this training diverge
# under Adam: at step 1 every parameter moves by ~lr whatever the size of its gradient, so an entry whose gradient is
# below the arithmetic's noise moves in a noise-determined direction, and with near-tied scores inside a list the ListMLE
# / RankNet gradients react to 1e-4 of score change by per cent (measured: after ONE step with max |dw| 1.4e-4 the two
# gradients differ by 2-12 % of their largest entry while the losses still agree to 1e-7).  The fp32 and fp64 runs of the
# oracle itself drift apart by 5e-7 / 5e-6 / 3e-6 of the loss over these three epochs; the HIP path, whose gradients
# carry ~1e-5 of rounding in the encoder's cancelling product / reactant sums, by 2e-5 ... 1e-4 at max_lr 1e-3.
SCHED = dict(warmup_epochs=1, total_epochs=3, train_data_size=4 * 6, batch_size=6, init_lr=1e-4, max_lr=5e-4, final_lr=1e-4)


def _data(seed0, n_batches, nq, nc, reverse=False):
    """Packed batches for the HIP path + the same queries as oracle graphs; learnable targets (a fixed function of the
    product graph and the extra feature) so three epochs move the loss.  reverse=True: the oracle batches hold the same
    molecules in REVERSED order (queries and candidates) - the same losses and gradients mathematically, every row sum
    in another order: a second, equally valid fp32 evaluation (see _compare_epochs)."""
    hip, ora = [], []
    for i in range(n_batches):
        qb = synth.make_queries(seed0 + i, nq, nc, atoms_lo=6, atoms_hi=12)
        tg = np.array([s.edges.shape[0] for s in qb.p_specs], np.float32) * 0.3 + qb.add_features[:, 0]
        tg = ((tg - tg.mean()) / (tg.std() + 1e-6) + 1e-3 * np.arange(len(tg), dtype=np.float32)).astype(np.float32)
        hip.append(dict(r=featurization.BatchMolGraph(qb.r_specs, K=4), p=featurization.BatchMolGraph(qb.p_specs, K=4),
                        scope=qb.scope, targets=torch.tensor(tg), add=qb.add_features))
        rs, ps, sc, tt, ad = qb.r_specs, qb.p_specs, list(qb.scope), tg, qb.add_features
        if reverse:
            rs, ps, sc, tt, ad = rs[::-1], ps[::-1], sc[::-1], tg[::-1].copy(), ad[::-1].copy()
        ora.append(dict(r=O.graph_tensors(O.pack_batch(rs, K=4)), p=O.graph_tensors(O.pack_batch(ps, K=4)),
                        scope=sc, targets=torch.tensor(tt), add=ad))
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


def _metrics(scores, batches, kind="ranking_metrics"):
    """The trainer's validation metrics of given scores by the oracle's restatement of the reference's metric code:
    ranking_metrics (train/eval.py:475-555; train_listwise.py:305-308) or evaluate_top_scores (:76-177;
    run_train_pairwise.py:91-96) - whose third value is a different quantity."""
    sq, tq = [], []
    for s, b in zip(scores, batches):
        off = 0
        for n in b["scope"]:
            sq.append(s[off:off + n].tolist())
            tq.append(b["targets"][off:off + n].tolist())
            off += n
    if kind == "top_scores":
        top1, recall25, top25, _ = O.top_scores_from_scores(sq, tq, 0.25)
        return dict(top1=top1, recall25=recall25, top25=top25, nd=None)
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


def _decisions_all(metric_seq):
    """save_metric 'all' (run_train_pairwise.py:103-113, train_listwise.py:317-328): three running maxima, one per metric."""
    old, out = [0.0, 0.0, 0.0], []
    for m in metric_seq:
        row = []
        for i, k in enumerate(("top1", "recall25", "top25")):
            row.append(m[k] >= old[i])
            if m[k] >= old[i]:
                old[i] = m[k]
        out.append(row)
    return out


def _compare_epochs(hist, hip_scores, ora_scores, ora_losses, ora64_scores, val_batches, has_ndcg, log, kind="ranking_metrics",
                    ora_rev_scores=None, yard_factor=10.0):
    """Per epoch: training loss 1e-4 relative to the fp32 oracle loop; validation scores (per query, up to the common
    offset no loss, ranking or metric can see) against the fp64 oracle loop's, bounded by 10 x what an fp32 ORACLE loop
    itself loses against fp64 on the same epoch (floor 1e-5).  Two fp32 oracle loops are the yardstick - the plain one,
    which shares its summation orders with the fp64 loop, and one fed the same molecules in reversed order
    (`ora_rev_scores`, un-reversed by the caller), i.e. the same mathematics with every sum over rows in another order -
    and the larger of their distances counts: what separates fp32 trajectories under Adam is not the size of the gradient
    error but the parameters whose gradient is ANALYTICALLY zero (the ranking losses cannot see a per-list shift, so the
    bias of an always-active last-hidden unit gets sum_j dL/ds_j = 0, and the output bias likewise): their computed
    gradient is rounding noise of ~1e-9, Adam's first-moment / second-moment ratio turns it into a step of
    lr * noise / (|noise| + 1e-8), and which way each of them goes depends on the summation order alone
    (tools/adam_divergence_probe.py: one-tensor-at-a-time attribution of the step-2 gradient difference puts ALL of it on
    ffn.ffn.4.bias, 24 entries with |g64| ~ 1e-18 and a computed gradient of 1e-10..1e-8 = W[n] * (sum_j dL/ds_j as rounded);
    profiles/r04_adam_divergence_probe.txt).  A model WITHOUT biases has no such parameter: there the factor is 10
    (`yard_factor`), with biases it is 100 - the same trainer, optimizer and data, so the pair is the evidence that the
    extra distance is this mechanism and not the gradients.  The trainer's metrics == the reference's metric code applied to ITS scores (exact) and its checkpoint
    decisions == the reference's rule on those metrics; and wherever the oracle's validation ranking is robust against
    the score difference (every score gap inside a query above twice that difference) the metrics and decisions equal
    the oracle loop's as well.  (With lists of 12 near-identical products a barely trained model leaves score gaps of
    1e-5..1e-4 between candidates - below what two fp32 training trajectories can agree on - so an unconditional
    comparison of rank metrics would test luck.)"""
    robust_all, n_robust = True, 0
    m_hip_seq, far = [], []

    def dist(xs, ys):
        return max(float(np.abs(_centered(a, b["scope"]) - _centered(o, b["scope"])).max()) for a, o, b in zip(xs, ys, val_batches))
    for e, (h, sh, so, s64, lo) in enumerate(zip(hist, hip_scores, ora_scores, ora64_scores, ora_losses)):
        rel = abs(h["train_loss"] - lo) / max(1e-6, abs(lo))
        d_hip64, d_ref = dist(sh, s64), dist(so, s64)
        d_rev = dist(ora_rev_scores[e], s64) if ora_rev_scores is not None else 0.0
        spread = max(float(np.abs(_centered(o, b["scope"])).max()) for o, b in zip(s64, val_batches))
        d = dist(sh, so)
        robust = _min_gap(so, val_batches) > 2.0 * d
        log(f"epoch {e + 1}: loss rel err {rel:.1e}; centred validation scores vs the fp64 oracle loop: HIP {d_hip64:.1e}, fp32 oracle "
            f"{d_ref:.1e}, fp32 oracle on the reversed batch {d_rev:.1e} (HIP / larger of the two: {d_hip64 / max(d_ref, d_rev, 1e-30):.1f}); "
            f"spread {spread:.1e}; HIP vs fp32 oracle {d:.1e}; min oracle score "
            f"gap {_min_gap(so, val_batches):.1e} -> ranking {'robust' if robust else 'ill-conditioned'}")
        assert rel <= 1e-4, (e, h["train_loss"], lo)
        far.append((e, d_hip64, max(d_ref, d_rev), spread))
        mh, mo = _metrics(sh, val_batches, kind), _metrics(so, val_batches, kind)
        m_hip_seq.append(mh)
        assert abs(h["top1"] - mh["top1"]) < 1e-9 and abs(h["top1_in_pred_top25"] - mh["top25"]) < 1e-9
        assert abs(h["pred_top25_in_targ_top25"] - mh["recall25"]) < 1e-9
        if has_ndcg:
            assert np.allclose(h["ndcg"], mh["nd"], rtol=0, atol=1e-6)
        robust_all = robust_all and robust
        n_robust += int(robust)
        if robust:
            assert abs(mh["top1"] - mo["top1"]) < 1e-9 and abs(mh["top25"] - mo["top25"]) < 1e-9 and abs(mh["recall25"] - mo["recall25"]) < 1e-9
    for e, d_hip64, d_ref, spread in far:                 # the same model, not a look-alike
        assert d_hip64 <= max(yard_factor * d_ref, 1e-5), (e, d_hip64, d_ref, spread, yard_factor)
    if "checkpoint_all" in hist[0]:
        assert [h["checkpoint_all"] for h in hist] == _decisions_all(m_hip_seq)
        if robust_all:
            assert [h["checkpoint_all"] for h in hist] == _decisions_all([_metrics(so, val_batches, kind) for so in ora_scores])
    else:
        assert [h["checkpoint"] for h in hist] == _decisions([m["top1"] for m in m_hip_seq])
        if robust_all:
            assert [h["checkpoint"] for h in hist] == _decisions([_metrics(so, val_batches, kind)["top1"] for so in ora_scores])
    return n_robust


def _cfg(task_num, task_type, use_bias=True):
    return dict(hidden_size=64, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=use_bias, task_num=task_num,
                ffn_last_layer="with_softplus" if task_num == 1 else "no_softplus", task_type=task_type, add_features_dim=1)


def _oracle_side(cfg, w, dtype=torch.float32):
    P = {k: v.detach().to(dtype).requires_grad_(v.requires_grad) for k, v in O.params_from_numpy(w, requires_grad=True).items()}
    opt = torch.optim.Adam([{"params": [p for p in P.values() if p.requires_grad], "lr": 1e-4, "weight_decay": 0}])
    sch = TU.build_lr_scheduler(opt, **SCHED)
    mc = dict(depth=cfg["mpnn_depth"], diff_depth=cfg["mpnn_diff_depth"], ffn_depth=cfg["ffn_depth"],
              task_type=O.resolve_task_type(cfg["task_num"], cfg["ffn_last_layer"], cfg["task_type"]))
    return P, opt, sch, mc


def _cast(b, dtype):
    """An oracle batch with its floating-point tensors in `dtype`."""
    out = dict(b)
    out["r"] = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in b["r"].items()}
    out["p"] = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in b["p"].items()}
    out["targets"] = b["targets"].to(dtype)
    out["add"] = torch.tensor(b["add"]).to(dtype)
    return out


def _oracle_loop(kind, cfg, w, train, val, epochs, dtype):
    """The reference trainer's loop (train_listwise.py:176-290 / train_pairwise.py:99-160) on the oracle with torch's Adam
    and the NoamLR mirror, in `dtype`: per-epoch loss as the trainers report it, validation scores, final parameters."""
    P, opt, sch, mc = _oracle_side(cfg, w, dtype)
    train, val = [_cast(b, dtype) for b in train], [_cast(b, dtype) for b in val]
    losses, scores = [], []
    for epoch in range(epochs):
        per_step = []
        for b in train:
            out = O.reaction_forward(P, mc, b["r"], b["p"], b["add"])
            if kind == "mle":
                loss = O.listmle_loss(out, b["scope"], b["targets"])
            elif kind == "evidential_ranking":
                loss = O.evidential_ranking_loss(out, b["scope"], b["targets"])
            else:                                                                      # ranknet sum_session
                ls, pairs = O.ranknet_sum_session(out, b["scope"], b["targets"], 1.0)   # train_pairwise.py:99-122
                if int(pairs) == 0:
                    continue
                loss = ls / pairs                                                      # :147
            per_step.append(float(loss.detach().sum()))
            opt.zero_grad()
            loss.sum().backward()
            opt.step()
            sch.step()
        losses.append(float(np.mean(per_step)) if kind == "ranknet" else per_step[-1])
        scores.append(_oracle_val_scores(P, mc, val))
    return losses, scores, P, opt, sch


def _reversed_oracle_scores(kind, cfg, w, train_spec, val_spec, epochs):
    """Per-epoch validation scores of the fp32 oracle loop run on the same batches with their molecules in reversed order,
    put back into the original order."""
    _, tr = _data(train_spec[0], train_spec[1], 6, 12, reverse=True)
    _, va = _data(val_spec[0], val_spec[1], 6, 12, reverse=True)
    _, scores, _, _, _ = _oracle_loop(kind, cfg, w, tr, va, epochs, torch.float32)
    return [[s[::-1].copy() for s in per_epoch] for per_epoch in scores]


def _hip_side(cfg, w):
    model = build_model(dropout=0.0, **cfg)
    model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    model = model.cuda()
    opt = TU.build_optimizer(model)
    return model, opt, TU.build_lr_scheduler(opt, **SCHED)


@pytest.mark.parametrize("task_type,task_num,use_bias", [("mle", 1, True), ("mle", 1, False), ("evidential_ranking", 2, True),
                                                         ("evidential_ranking", 2, False)])
def test_listwise_trainer_trajectory_matches_the_oracle_loop(tmp_path, task_type, task_num, use_bias, parity_log):
    cfg = _cfg(task_num, "evidential_ranking" if task_type == "evidential_ranking" else None, use_bias)
    shapes = O.model_shapes(64, 3, 3, 3, task_num, 1, use_bias)
    w = synth.seeded_weights(shapes, 21)
    hip_tr, ora_tr = _data(4000, 4, 6, 12)
    hip_va, ora_va = _data(4100, 2, 6, 12)
    epochs = 3
    model, opt, sch = _hip_side(cfg, w)
    path = str(tmp_path / "ck" / "model.pt")
    hip_scores = []
    hist = TL.train(model, sch, hip_tr, hip_va, path, opt, epochs, seed=5, gpu=0, task_type=task_type, save_metric=None,
                    epoch_hook=lambda e, m, rec: hip_scores.append(_hip_val_scores(m, hip_va)))

    ora_losses, ora_scores, P, o_opt, o_sch = _oracle_loop(task_type, cfg, w, ora_tr, ora_va, epochs, torch.float32)
    _, ora64_scores, _, _, _ = _oracle_loop(task_type, cfg, w, ora_tr, ora_va, epochs, torch.float64)
    rev_scores = _reversed_oracle_scores(task_type, cfg, w, (4000, 4), (4100, 2), epochs)
    assert sch.current_step == o_sch.current_step and abs(opt.param_groups[0]["lr"] - o_opt.param_groups[0]["lr"]) < 1e-12
    assert ora_losses[-1] < ora_losses[0]                                   # the epochs really trained
    _compare_epochs(hist, hip_scores, ora_scores, ora_losses, ora64_scores, ora_va, True, parity_log, ora_rev_scores=rev_scores,
                    yard_factor=100.0 if use_bias else 10.0)
    # the checkpoint on disk (reference layout, utils.py:152-173) is the model as it stood after its last saving epoch:
    # reloaded into a fresh model it reproduces that epoch's validation scores bit for bit
    assert os.path.exists(path)
    last = max(i for i, h in enumerate(hist) if h["checkpoint"])
    m2 = build_model(dropout=0.0, **cfg).cuda().eval()
    load_checkpoint(path, m2)
    for a, b in zip(_hip_val_scores(m2, hip_va), hip_scores[last]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("use_bias", [True, False])
def test_ranknet_trainer_trajectory_matches_the_oracle_loop(tmp_path, use_bias, parity_log):
    cfg = dict(_cfg(1, None, use_bias), ffn_last_layer="no_softplus")
    shapes = O.model_shapes(64, 3, 3, 3, 1, 1, use_bias)
    w = synth.seeded_weights(shapes, 22)
    hip_tr, ora_tr = _data(5000, 4, 6, 12)
    hip_va, ora_va = _data(5100, 2, 6, 12)
    epochs = 3
    model, opt, sch = _hip_side(cfg, w)
    # save_metric 'all' as main_ranknet.py:49 sets it: three checkpoints, each on its own evaluate_top_scores value
    paths = [str(tmp_path / "ck" / sub / "0.pt") for sub in ("T1", "T25_in_T25", "T25")]
    hip_scores = []
    hist = RP.run_train(model, sch, hip_tr, hip_va, paths, opt, epochs, seed=5, gpu=0, train_strategy="sum_session",
                        target_name=None, save_metric="all",
                        epoch_hook=lambda e, m, rec: hip_scores.append(_hip_val_scores(m, hip_va)))
    ora_losses, ora_scores, _, _, _ = _oracle_loop("ranknet", cfg, w, ora_tr, ora_va, epochs, torch.float32)
    _, ora64_scores, _, _, _ = _oracle_loop("ranknet", cfg, w, ora_tr, ora_va, epochs, torch.float64)
    rev_scores = _reversed_oracle_scores("ranknet", cfg, w, (5000, 4), (5100, 2), epochs)
    assert ora_losses[-1] < ora_losses[0]
    _compare_epochs(hist, hip_scores, ora_scores, ora_losses, ora64_scores, ora_va, False, parity_log, kind="top_scores",
                    ora_rev_scores=rev_scores, yard_factor=100.0 if use_bias else 10.0)
    assert all(os.path.exists(p) for p in paths)
    # the T25 checkpoint is the model of the last epoch whose TARGET-top-1-in-predicted-top-25% did not get worse
    last = max(i for i, h in enumerate(hist) if h["checkpoint_all"][2])
    m2 = build_model(dropout=0.0, **cfg).cuda().eval()
    load_checkpoint(paths[2], m2)
    for a, b in zip(_hip_val_scores(m2, hip_va), hip_scores[last]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kind", ["mle", "ranknet"])
def test_first_optimizer_step_matches_the_oracle(kind, parity_log):
    """One step, tight: the gradients of the first training batch against the fp64 oracle, bounded by 3 x the fp32
    oracle's own distance to fp64 (floor 5e-5 of the tensor's largest entry), and Adam's first update.  That update is
    lr * g / (|g| + 1e-8): every entry moves by the full lr in the direction of its gradient's SIGN, however small the
    gradient - so two arithmetics part ways on exactly the entries whose gradient is smaller than their error, by 2 lr
    each.  The test counts those entries for the HIP path and for the fp32 oracle (both against fp64) and requires every
    HIP sign mismatch to sit on an entry whose |g64| is below 3 x the HIP gradient's error on that tensor."""
    cfg = dict(_cfg(1, None), ffn_last_layer="no_softplus" if kind == "ranknet" else "with_softplus")
    shapes = O.model_shapes(64, 3, 3, 3, 1, 1, True)
    w = synth.seeded_weights(shapes, 23)
    hip_tr, ora_tr = _data(6000, 1, 6, 12)
    model, opt, sch = _hip_side(cfg, w)
    model.train()
    b = hip_tr[0]
    before = {k: v.detach().double().cpu().clone() for k, v in model.named_parameters()}
    out = model(b["r"], b["p"], gpu=0, add_features=b["add"])
    if kind == "mle":
        loss = TL.batch_loss("mle", out, b["scope"], b["targets"], 0)
    else:
        from reactranker_amd.loss import ranknet_loss
        ls, pairs = ranknet_loss(out, b["scope"], b["targets"], 1.0, 0)
        loss = ls / pairs
    opt.zero_grad()
    loss.sum().backward()
    grads = {k: v.grad.detach().double().cpu() for k, v in model.named_parameters() if v.grad is not None}
    lr = opt.param_groups[0]["lr"]
    opt.step()
    after = {k: v.detach().double().cpu() for k, v in model.named_parameters()}

    def oracle(dtype):
        P, o_opt, _, mc = _oracle_side(cfg, w, dtype)
        ob = _cast(ora_tr[0], dtype)
        o = O.reaction_forward(P, mc, ob["r"], ob["p"], ob["add"])
        if kind == "mle":
            l = O.listmle_loss(o, ob["scope"], ob["targets"])
        else:
            ls, pairs = O.ranknet_sum_session(o, ob["scope"], ob["targets"], 1.0)
            l = ls / pairs
        o_opt.zero_grad()
        l.sum().backward()
        g = {k: v.grad.detach().double().clone() for k, v in P.items() if v.grad is not None}
        p0 = {k: v.detach().double().clone() for k, v in P.items()}
        assert abs(o_opt.param_groups[0]["lr"] - lr) < 1e-15
        o_opt.step()
        return float(l.sum()), g, {k: P[k].detach().double() - p0[k] for k in g}
    l64, g64, s64 = oracle(torch.float64)
    l32, g32, s32 = oracle(torch.float32)
    assert abs(float(loss.sum()) - l64) <= 1e-5 * (1 + abs(l64))
    worst = (0.0, "", 0.0)
    n_mis_hip = n_mis_32 = n_entries = 0
    for k, gd in g64.items():
        scale = float(gd.abs().max())
        if scale < 1e-12:                       # analytically zero (ListMLE / RankNet cannot see the output bias): noise everywhere
            continue
        err = float((grads[k] - gd).abs().max())
        noise = float((g32[k] - gd).abs().max())
        if err / scale > worst[0]:
            worst = (err / scale, k, noise / scale)
        assert err <= max(5e-5 * scale + 1e-7, 3.0 * noise), (k, err, noise, scale)
        step = after[k] - before[k]
        assert float(step.abs().max()) <= 1.001 * lr + 1e-9                     # Adam's first update is at most lr
        moved = s64[k].abs() > 0.5 * lr                                         # |g64| well above Adam's eps
        mis_hip = moved & (torch.sign(step) != torch.sign(s64[k]))
        mis_32 = moved & (torch.sign(s32[k]) != torch.sign(s64[k]))
        n_mis_hip += int(mis_hip.sum())
        n_mis_32 += int(mis_32.sum())
        n_entries += int(moved.sum())
        if bool(mis_hip.any()):
            assert float(gd[mis_hip].abs().max()) <= 3.0 * err + 1e-12, (k, float(gd[mis_hip].abs().max()), err)
    parity_log(f"{kind}: loss err {abs(float(loss.sum()) - l64):.1e}; worst gradient tensor {worst[1]}: {worst[0]:.1e} of its max "
               f"(fp32 oracle on the same tensor: {worst[2]:.1e}); Adam step 1 (lr {lr:.1e}): update sign differs from the fp64 "
               f"oracle's on {n_mis_hip} of {n_entries} entries (fp32 oracle: {n_mis_32})")


def test_save_metric_mse_uses_the_last_validation_batch_like_the_reference(tmp_path):
    """train_listwise.py:345-351 + eval.py:558-609: `save_metric='mse'` checkpoints whenever calculate_mse does not get
    worse, and calculate_mse returns the squared error of the LAST validation batch only (the reference overwrites it in
    every loop iteration)."""
    from reactranker_amd import eval as RE
    cfg = _cfg(1, None)
    shapes = O.model_shapes(64, 3, 3, 3, 1, 1, True)
    w = synth.seeded_weights(shapes, 31)
    hip_tr, _ = _data(7000, 2, 4, 6)
    hip_va, _ = _data(7100, 2, 4, 6)
    model, opt, sch = _hip_side(cfg, w)
    path = str(tmp_path / "ck" / "mse.pt")
    seen = []

    def hook(e, m, rec):
        was = m.training
        m.eval()
        with torch.no_grad():
            b = hip_va[-1]
            o = m(b["r"], b["p"], gpu=0, add_features=b["add"]).double().cpu()
        m.train(was)
        seen.append(float(((o - b["targets"].double()) ** 2).mean()))
    hist = TL.train(model, sch, hip_tr, hip_va, path, opt, 3, seed=5, gpu=0, task_type="regression", save_metric="mse", epoch_hook=hook)
    assert all(abs(h["mse"] - s) <= 1e-6 * (1 + s) for h, s in zip(hist, seen))
    best, want = float("inf"), []
    for h in hist:
        want.append(h["mse"] <= best)
        best = min(best, h["mse"])
    assert [h["checkpoint"] for h in hist] == want and os.path.exists(path)
    assert abs(RE.calculate_mse(model, 0, [(b["r"], b["p"], b["scope"], b["targets"], b["add"]) for b in hip_va]) - seen[-1]) <= 1e-6 * (1 + seen[-1])


def test_test_function_reports_evaluate_top_scores_and_calculate_ndcg(tmp_path):
    """reactranker_amd.main.test (train/test_listwise.py:10-86): the checkpoint's scaler flips the sign of raw targets unless
    'lgk', the triple is evaluate_top_scores at ratio 0.25, and with cal_ngcd the NDCG@0.25 / KL of calculate_ndcg on the
    DE-STANDARDISED outputs plus its per-candidate listing - all against the oracle's restatements on the model's own scores."""
    import logging
    from reactranker_amd import main as RM
    from reactranker_amd.utils import save_checkpoint
    cfg = _cfg(1, None)
    shapes = O.model_shapes(64, 3, 3, 3, 1, 1, True)
    w = synth.seeded_weights(shapes, 41)
    model = build_model(dropout=0.0, **cfg)
    model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    model = model.cuda()
    path = str(tmp_path / "ck" / "0.pt")
    mean, std = 1.7, 0.6
    save_checkpoint(path, model, mean, std)
    hip_te, _ = _data(8000, 2, 5, 9)
    scores = _hip_val_scores(model, hip_te)
    sq, tq = [], []
    for s, b in zip(scores, hip_te):
        off = 0
        for n in b["scope"]:
            sq.append(s[off:off + n])
            tq.append(-b["targets"][off:off + n].double().numpy())          # target_name 'ea': sign flipped (:30-35)
            off += n
    a, bb, c, _ = O.top_scores_from_scores(sq, tq, 0.25)
    m2 = build_model(dropout=0.0, **cfg)
    got = RM.test(m2, hip_te, path, 0, logging.getLogger("t"), "ea", cal_ngcd=True, is_order=True, return_order=True)
    assert abs(got[0] - a) < 1e-12 and abs(got[1] - bb) < 1e-12 and abs(got[2] - c) < 1e-12      # (a mean over 10 queries)
    nd, kl, _ = O.calculate_ndcg_from_scores(sq, tq, 0.25, mean, std)
    rows = np.asarray(got[3])
    assert rows.shape == (sum(len(x) for x in sq), 4) and got[4] is None
    # the listing: targets in descending order per query, predictions de-standardised (:379-388)
    off = 0
    for s, t in zip(sq, tq):
        blk = rows[off:off + len(s)]
        assert np.all(np.diff(blk[:, 0]) <= 0) and np.allclose(np.sort(blk[:, 1]), np.sort(s * std + mean), atol=1e-5)
        off += len(s)
    got2 = RM.test(m2, hip_te, path, 0, None, "lgk")                        # 'lgk': no sign flip (:33-34)
    a2, b2, c2, _ = O.top_scores_from_scores(sq, [-t for t in tq], 0.25)
    assert abs(got2[0] - a2) < 1e-12 and abs(got2[1] - b2) < 1e-12 and abs(got2[2] - c2) < 1e-12
    # NDCG / KL through the logger-free path
    from reactranker_amd.eval import calculate_ndcg
    g_nd, g_kl, _, _ = calculate_ndcg(m2.cuda().eval(), 0, [(b["r"], b["p"], b["scope"], -b["targets"], b["add"]) for b in hip_te],
                                      NDCG_cut=0.25, means=mean, stds=std)
    assert abs(g_nd - nd) < 1e-6 and abs(g_kl - kl) < 1e-5 * max(1.0, abs(kl))
