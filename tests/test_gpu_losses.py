"""GPU parity of the fused loss kernels: against vectors produced by the reference itself
(tests/golden/losses.npz) and against the CPU oracle on fresh random lists (ragged, C in
{1,2,3,32,64,65,129,300}).  Tolerance 1e-5 * (1 + |ref|)."""
import numpy as np
import pytest
import torch

from tests import helpers as Hh

from reactranker_amd import loss as RL
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
CASES = ["single", "tiny", "c32", "c64", "ragged", "long"]


def close(got, ref, tol=1e-5, what=""):
    got = got.detach().cpu().double().numpy().reshape(-1) if torch.is_tensor(got) else np.asarray(got, np.float64).reshape(-1)
    ref = ref.detach().cpu().double().numpy().reshape(-1) if torch.is_tensor(ref) else np.asarray(ref, np.float64).reshape(-1)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = np.max(np.abs(got - ref) / (1 + np.abs(ref))) if got.size else 0
    Hh.record(what, err, tol)
    assert err <= tol, f"{what}: err {err:.3e}"


@pytest.mark.parametrize("name", CASES)
def test_losses_against_reference_vectors(name, golden_dir):
    L = np.load(golden_dir + "/losses.npz")
    P = name + "."
    scope = L[P + "scope"].tolist()
    targets = torch.tensor(L[P + "targets"])

    def fresh():
        return torch.tensor(L[P + "score"]).cuda().requires_grad_(True)

    s = fresh()
    l = RL.MLEloss()(s, scope, targets, 0)
    assert l.shape == (1,)
    l.sum().backward()
    close(l, L[P + "mle"], what="mle"); close(s.grad, L[P + "mle_g"], what="mle_g")

    s = fresh()
    l = RL.ListnetLoss()(s, scope, targets, 0)
    assert l.dim() == 0
    l.backward()
    close(l, L[P + "listnet"], what="listnet"); close(s.grad, L[P + "listnet_g"], what="listnet_g")

    poss = torch.stack([torch.tensor(L[P + "score"]), torch.tensor(L[P + "var"])], 1).cuda().requires_grad_(True)
    l = RL.evidential_ranking()(poss, scope, targets, 0.01, 0, 10, 0)
    assert l.shape == (1,)
    l.sum().backward()
    close(l, L[P + "evid"], what="evid")
    close(poss.grad[:, 0], L[P + "evid_gs"], tol=1e-5, what="evid_gs")
    close(poss.grad[:, 1], L[P + "evid_gv"], tol=1e-5, what="evid_gv")

    s = fresh()
    l = RL.MSELoss()(s, targets)
    l.backward()
    close(l, L[P + "mse"]); close(s.grad, L[P + "mse_g"])

    s = fresh()
    v = torch.tensor(L[P + "var"]).cuda().requires_grad_(True)
    l = RL.GaussDisLoss()(s, v, targets, 0)
    l.backward()
    close(l, L[P + "gauss"]); close(s.grad, L[P + "gauss_gs"]); close(v.grad, L[P + "gauss_gv"], tol=1e-5)

    for sigma in (1.0, 0.5):
        s = fresh()
        ls, pairs = RL.ranknet_loss(s, scope, targets, sigma, 0)
        assert int(pairs.item()) == int(L[P + "rank_pairs"])
        if int(pairs.item()) == 0:
            assert float(ls) == 0.0
            continue
        (ls / pairs).backward()
        close(ls / pairs, L[P + f"rank_ss_{sigma}"], what="rank_ss")
        close(s.grad, L[P + f"rank_ss_g_{sigma}"], what="rank_ss_g")
        lam = RL.ranknet_lambda(s.detach(), scope, targets, sigma, 0) / pairs
        close(lam, L[P + f"rank_ag_g_{sigma}"], what="rank_ag_g")


def test_logcumsumexp_op_and_ranknet_overflow(golden_dir):
    L = np.load(golden_dir + "/losses.npz")
    for nm in ("lce_small", "lce_large"):
        x = torch.tensor(L[nm + ".x"]).cuda().requires_grad_(True)
        y = RL.LogCumsumExp.apply(x)
        y.backward(torch.tensor(L[nm + ".go"]).cuda())
        close(y, L[nm + ".y"], what=nm)
        ref = L[nm + ".g"]
        sc = max(1.0, float(np.abs(ref).max()))
        close(x.grad / sc, ref / sc, what=nm + ".g")
    ls, pairs = RL.ranknet_loss(torch.tensor(L["rank_overflow.score"]).cuda(), [3],
                                torch.tensor(L["rank_overflow.targets"]), 1.0, 0)
    assert torch.isinf(ls) and int(pairs) == 6          # naive log(1+exp(x)) overflows like the reference (H4)


@pytest.mark.parametrize("seed,scope", [(0, [1, 2, 3, 32, 64, 65, 129, 300]), (1, [64] * 64), (2, [5, 1, 1, 7]),
                                        (3, [1000])])
def test_losses_against_oracle_random(seed, scope):
    rng = np.random.default_rng(seed)
    m = sum(scope)
    score = (rng.standard_normal(m) * 2).astype(np.float32)
    targets = np.concatenate([rng.permutation(c) for c in scope]).astype(np.float32)
    targets = ((targets - targets.mean()) / (targets.std() + 1e-6)).astype(np.float32)
    var = (np.log1p(np.exp(rng.standard_normal(m))) + 1e-6).astype(np.float32)
    ts, tt, tv = torch.tensor(score, requires_grad=True), torch.tensor(targets), torch.tensor(var, requires_grad=True)

    ref = O.listmle_loss(ts, scope, tt); g_ref, = torch.autograd.grad(ref.sum(), ts)
    s = torch.tensor(score).cuda().requires_grad_(True)
    l = RL.MLEloss()(s, scope, tt, 0); l.sum().backward()
    close(l, ref, what="mle"); close(s.grad, g_ref, what="mle_g")

    ref = O.listnet_loss(ts, scope, tt); g_ref, = torch.autograd.grad(ref, ts)
    s = torch.tensor(score).cuda().requires_grad_(True)
    l = RL.ListnetLoss()(s, scope, tt, 0); l.backward()
    close(l, ref, what="listnet"); close(s.grad, g_ref, what="listnet_g")

    ref = O.evidential_ranking_loss(torch.stack([ts, tv], 1), scope, tt)
    gs_ref, gv_ref = torch.autograd.grad(ref.sum(), [ts, tv])
    poss = torch.stack([torch.tensor(score), torch.tensor(var)], 1).cuda().requires_grad_(True)
    l = RL.evidential_ranking()(poss, scope, tt, None, None, None, 0); l.sum().backward()
    close(l, ref, tol=1e-5, what="evid")
    sc = max(1.0, float(gv_ref.abs().max()))
    close(poss.grad[:, 0], gs_ref, tol=5e-5, what="evid_gs"); close(poss.grad[:, 1] / sc, gv_ref / sc, tol=5e-5, what="evid_gv")

    ref, pairs_ref = O.ranknet_sum_session(ts, scope, tt, 1.0)
    s = torch.tensor(score).cuda().requires_grad_(True)
    ls, pairs = RL.ranknet_loss(s, scope, tt, 1.0, 0)
    assert int(pairs) == int(pairs_ref)
    if pairs_ref > 0:
        g_ref, = torch.autograd.grad(ref / pairs_ref, ts)
        (ls / pairs).backward()
        close(ls / pairs, ref / pairs_ref, what="ranknet"); close(s.grad, g_ref, what="ranknet_g")


def test_strided_score_column_and_query_permutation_invariance():
    rng = np.random.default_rng(9)
    scope = [64] * 16
    m = sum(scope)
    out = torch.tensor(rng.standard_normal((m, 2)).astype(np.float32)).cuda()
    targets = torch.tensor(np.concatenate([rng.permutation(64) for _ in scope]).astype(np.float32) / 64)
    a = RL.MLEloss()(out[:, 0], scope, targets, 0)
    b = RL.MLEloss()(out[:, 0].contiguous(), scope, targets, 0)
    assert torch.equal(a, b)
    perm = rng.permutation(16)
    idx = np.concatenate([np.arange(q * 64, (q + 1) * 64) for q in perm])
    c = RL.MLEloss()(out[:, 0].contiguous()[idx], scope, targets[idx], 0)
    close(c, a, tol=1e-6, what="query order")


def test_ranking_metrics_kernel_against_reference_loop(golden_dir):
    """On-device ranking_metrics vs the numbers the reference's own loop produced (tests/golden/eval_metrics.npz)
    and vs the oracle on random lists incl. ties and C > 64."""
    from reactranker_amd import eval as RE
    E = np.load(golden_dir + "/eval_metrics.npz")
    for name in ("plain", "ties"):
        scope = E[name + ".scope"].tolist()
        sc = torch.tensor(E[name + ".scores"]).cuda()
        tg = torch.tensor(E[name + ".targets"])
        top1, rec, top25, nd = RE.ranking_metrics_from_scores(sc, scope, tg, 0)
        assert top1 == float(E[name + ".top1"]) and top25 == float(E[name + ".top25"])
        assert abs(rec - float(E[name + ".recall25"])) < 1e-12
        assert np.allclose(nd, E[name + ".ndcg"], rtol=0, atol=1e-12)
        stats, order = RE.ranking_stats(sc, scope, tg, 0)
        assert np.array_equal(order.cpu().numpy(), E[name + ".order"])          # bit-exact candidate ordering
        rel = torch.tensor(np.maximum(np.round(E[name + ".targets"].astype(np.float64) * 2 + 3), 0).astype(np.float32))
        nd10 = RE.ndcg_at_k(sc, scope, rel, 0)
        ok = ~np.isnan(E[name + ".ndcg10_rel"])
        assert np.allclose(nd10[ok], E[name + ".ndcg10_rel"][ok], rtol=0, atol=1e-12)
    rng = np.random.default_rng(5)
    scope = [1, 2, 64, 65, 130, 7, 300]
    sc = [np.round(rng.standard_normal(c), 1).astype(np.float32) for c in scope]
    tg = [rng.standard_normal(c).astype(np.float32) for c in scope]
    r_top1, r_rec, r_top25, r_nd, r_orders = O.ranking_metrics_from_scores(sc, tg)
    top1, rec, top25, nd = RE.ranking_metrics_from_scores(torch.tensor(np.concatenate(sc)).cuda(), scope,
                                                          torch.tensor(np.concatenate(tg)), 0)
    assert (top1, top25) == (r_top1, r_top25) and abs(rec - r_rec) < 1e-12 and np.allclose(nd, r_nd, rtol=0, atol=1e-12)
    _, order = RE.ranking_stats(torch.tensor(np.concatenate(sc)).cuda(), scope, torch.tensor(np.concatenate(tg)), 0)
    assert np.array_equal(order.cpu().numpy(), np.concatenate(r_orders).astype(np.int32))


@pytest.mark.parametrize("name", ("plain", "pred_ties", "ties", "two_col", "scaled", "two_col_scaled"))
def test_top_scores_and_calculate_ndcg_kernel_against_reference_loops(golden_dir, name):
    """rr_ranking_metrics_f32's evaluate_top_scores / calculate_ndcg statistics vs the numbers the reference's own
    evaluate_top_scores (eval.py:76-177) and calculate_ndcg (:329-457) produced on preset scores
    (tests/golden/top_scores.npz), through the trainer-level mirrors that take a model and batches."""
    from reactranker_amd import eval as RE
    from tests.test_oracle_golden import top_scores_case
    sc, tg, scope, scaler, E = top_scores_case(golden_dir, name)
    scores = torch.tensor(np.concatenate(sc)).cuda()
    targets = np.concatenate(tg)

    class _Model:                                       # two batches of whole queries with preset outputs
        training = False

        def __init__(self):
            self.at = 0

        def __call__(self, r, p, gpu=None, add_features=None):
            out = scores[self.at:self.at + r]
            self.at += r
            return out
    half = len(scope) // 2
    n0 = int(sum(scope[:half]))

    def batches():
        return [(n0, None, scope[:half], targets[:n0], None),
                (int(sum(scope[half:])), None, scope[half:], targets[n0:], None)]
    for ratio in (0.25, 0.1, 0.5):
        got = RE.evaluate_top_scores(_Model(), 0, batches(), ratio=ratio)
        want = E[f"{name}.top_scores_r{ratio}"]
        assert got[0] == want[0] and abs(got[1] - want[1]) < 1e-12 and got[2] == want[2], (name, ratio, got, want)
        one = RE.top_scores_from_scores(scores, scope, targets, 0, ratio)
        assert one[0] == want[0] and abs(one[1] - want[1]) < 1e-12 and one[2] == want[2]
    means, stds = scaler if scaler is not None else (None, None)
    first = [s[:, 0] if s.ndim > 1 else s for s in sc]
    for cut in (0.5, 0.25):
        nd, kl, rows, _ = RE.calculate_ndcg(_Model(), 0, batches(), NDCG_cut=cut, means=means, stds=stds)
        o_nd, o_kl, _ = O.calculate_ndcg_from_scores(first, tg, cut, means, stds)
        assert abs(nd - o_nd) < 1e-6 and abs(kl - o_kl) < 1e-6 * max(1.0, abs(o_kl)), (name, cut)   # f64 vs torch f32
        assert abs(kl - float(E[f"{name}.kl"])) < 1e-6 * max(1.0, abs(kl))
        if "ties" not in name:          # with ties the reference's unstable torch.sort decides (see the oracle's test)
            assert abs(nd - float(E[f"{name}.ndcg_c{cut}"])) < 1e-6, (name, cut)
            want_rows = E[f"{name}.order_rows"]
            assert np.allclose(np.asarray(rows), want_rows, rtol=0, atol=2e-6)      # the per-candidate listing
    raw, _, rows, _ = RE.calculate_ndcg(_Model(), 0, batches(), is_order=False)
    assert raw is None and len(rows) == len(targets)


def test_top_scores_third_value_is_not_ranking_metrics_third_value():
    """The two 'top-25 %' hits differ in kind: ranking_metrics asks for the PREDICTED top-1 inside the target top-25 %
    (eval.py:526-530), evaluate_top_scores for the TARGET's top-1 inside the predicted top-25 % (:156-159)."""
    from reactranker_amd import eval as RE
    s = torch.tensor([9., 8., 7., 6., 5., 4., 3., 2.]).cuda()          # predicted order 0, 1, 2, ...
    t = torch.tensor([6., 1., 2., 3., 4., 5., 8., 7.])                 # target order 6, 7, 0, ... ; cut = 2
    _, _, top25, _ = RE.ranking_metrics_from_scores(s, [8], t, 0)
    a, b, c = RE.top_scores_from_scores(s, [8], t, 0)
    assert top25 == 0.0 and c == 0.0 and a == 0.0 and b == 0.0
    t2 = torch.tensor([7., 8., 2., 3., 4., 5., 1., 0.])                # target order 1, 0: predicted top-1 IS in it ...
    assert RE.ranking_metrics_from_scores(s, [8], t2, 0)[2] == 1.0
    assert RE.top_scores_from_scores(s, [8], t2, 0)[2] == 1.0          # ... and target top-1 (1) is in predicted {0, 1}
    t3 = torch.tensor([7., 1., 8., 3., 4., 5., 6., 0.])                # target order 2, 0: predicted top-1 (0) is in it,
    assert RE.ranking_metrics_from_scores(s, [8], t3, 0)[2] == 1.0     # but the target's top-1 (2) is predicted third
    assert RE.top_scores_from_scores(s, [8], t3, 0)[2] == 0.0
    rng = np.random.default_rng(11)
    scope = [1, 2, 64, 65, 130, 7, 300]
    sc = [np.round(rng.standard_normal(c), 1).astype(np.float32) for c in scope]
    tg = [rng.standard_normal(c).astype(np.float32) for c in scope]
    for ratio in (0.25, 0.03, 1.0):
        want = O.top_scores_from_scores(sc, tg, ratio)
        got = RE.top_scores_from_scores(torch.tensor(np.concatenate(sc)).cuda(), scope, np.concatenate(tg), 0, ratio)
        assert got[0] == want[0] and abs(got[1] - want[1]) < 1e-12 and got[2] == want[2]
    for cut in (0.5, 1.0, 0.01):
        want = O.calculate_ndcg_from_scores(sc, tg, cut)
        stats, _ = RE.ranking_stats(torch.tensor(np.concatenate(sc)).cuda(), scope, np.concatenate(tg), 0, 0.25, cut)
        assert np.allclose(stats[:, 9].cpu().numpy(), want[2][:, 0], rtol=0, atol=1e-6)
        assert np.allclose(stats[:, 10].cpu().numpy(), want[2][:, 1], rtol=1e-6, atol=1e-6)


def test_ranking_metrics_at_the_longest_supported_list():
    """One list of 8192 candidates (the kernels' kMaxLen: 128 KB of LDS for the metrics kernel's two f32 and four 16-bit
    arrays) next to short ones, against the oracle's three evaluation loops; 8193 is refused with a status, not a fault."""
    from reactranker_amd import eval as RE
    rng = np.random.default_rng(21)
    scope = [8192, 3, 64]
    sc = [np.round(rng.standard_normal(c), 2).astype(np.float32) for c in scope]          # rounded: thousands of ties
    tg = [rng.standard_normal(c).astype(np.float32) for c in scope]
    s, t = torch.tensor(np.concatenate(sc)).cuda(), np.concatenate(tg)
    r_top1, r_rec, r_top25, r_nd, r_orders = O.ranking_metrics_from_scores(sc, tg)
    top1, rec, top25, nd = RE.ranking_metrics_from_scores(s, scope, t, 0)
    assert (top1, top25) == (r_top1, r_top25) and abs(rec - r_rec) < 1e-12 and np.allclose(nd, r_nd, rtol=1e-10, atol=1e-12)
    stats, order = RE.ranking_stats(s, scope, t, 0, 0.25, 0.5)
    assert np.array_equal(order.cpu().numpy(), np.concatenate(r_orders).astype(np.int32))
    a, b, c, _ = O.top_scores_from_scores(sc, tg, 0.25)
    got = RE.top_scores_from_scores(s, scope, t, 0, 0.25)
    assert got[0] == a and abs(got[1] - b) < 1e-12 and got[2] == c
    o_nd, o_kl, rows = O.calculate_ndcg_from_scores(sc, tg, 0.5)
    assert np.allclose(stats[:, 9].cpu().numpy(), rows[:, 0], rtol=0, atol=1e-6)
    assert np.allclose(stats[:, 10].cpu().numpy(), rows[:, 1], rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        RE.ranking_stats(torch.zeros(8193, device="cuda"), [8193], np.zeros(8193, np.float32), 0)


@pytest.mark.parametrize("scope", [[64] * 64, [1, 2, 3, 32, 64, 65, 129, 300], [5, 0, 7, 0], [3], [40] * 300])
def test_fused_loss_step_has_the_bits_of_the_two_kernel_path(scope, parity_log):
    """rr_listmle_step_f32 / rr_listnet_step_f32 / rr_evidential_ranking_step_f32 (ABI revision 8): loss AND d loss / d score in
    one launch, the per-query partials summed by the last workgroup in the reduction kernel's order.  Loss and gradient must be
    torch.equal to the forward kernel + reduction + backward kernel sequence; `RL.backward(loss)` must be served by the
    gradient the forward already wrote (FusedStep.hits); a plain loss.backward() and a scaled loss take the backward kernel
    and give the same numbers; the ticket word re-arms itself (three launches in a row)."""
    rng = np.random.default_rng(11)
    m = sum(scope)
    score = torch.tensor((rng.standard_normal(m) * 2).astype(np.float32)).cuda()
    var = torch.tensor((np.log1p(np.exp(rng.standard_normal(m))) + 1e-3).astype(np.float32)).cuda()
    targets = torch.tensor(rng.standard_normal(m).astype(np.float32))

    def run(kind, fused, how):
        RL.FusedStep.enabled = fused
        try:
            if kind == "evid":
                x = torch.stack([score, var], 1).clone().requires_grad_(True)
                l = RL.evidential_ranking()(x, scope, targets, None, None, None, 0)
            else:
                x = score.clone().requires_grad_(True)
                l = (RL.MLEloss() if kind == "mle" else RL.ListnetLoss())(x, scope, targets, 0)
            if how == "unit":
                RL.backward(l)
            elif how == "plain":
                l.sum().backward()
            else:
                (l * 0.37).sum().backward()
            return l.detach().clone(), x.grad.clone()
        finally:
            RL.FusedStep.enabled = True

    for kind in ("mle", "listnet", "evid"):
        for how in ("unit", "plain", "scaled"):
            l0, g0 = run(kind, False, how)
            for rep in range(3):
                h0 = RL.FusedStep.hits
                l1, g1 = run(kind, True, how)
                assert torch.equal(l0, l1), (kind, how, float(l0.sum()), float(l1.sum()))
                assert torch.equal(g0, g1), (kind, how, float((g0 - g1).abs().max()))
                assert RL.FusedStep.hits - h0 == (1 if how == "unit" else 0), (kind, how)
    parity_log(f"scope of {len(scope)} lists / {m} candidates: fused step == two-kernel path bit for bit (mle, listnet, evidential; "
               "unit, plain and scaled upstream gradients; 3 launches each)")


def test_fused_loss_step_inside_a_training_step_reaches_the_model_gradients():
    """the gradient handed out by the fused step drives the model's explicit backward: parameter gradients equal those of the
    two-kernel path bit for bit"""
    from oracle import ref_cpu as O
    from reactranker_amd import featurization, synth
    from tests.test_gpu_model import make_model
    cfg = dict(hidden_size=32, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, task_num=1, ffn_last_layer="with_softplus",
               task_type=None, add_features_dim=1)
    w = synth.seeded_weights(O.model_shapes(32, 3, 3, 3, 1, 1, True), 5)
    model = make_model(cfg, w, dropout=0.1).train()
    qb = synth.make_queries(17, 4, [7, 3, 9, 5], atoms_lo=5, atoms_hi=14)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    res = []
    for fused in (False, True):
        RL.FusedStep.enabled = fused
        try:
            model.zero_grad()
            model.dropout_seed = 77
            out = model(rb, pb, gpu=0, add_features=qb.add_features)
            l = RL.MLEloss()(out, qb.scope, torch.tensor(qb.targets), 0)
            h0 = RL.FusedStep.hits
            RL.backward(l)
            assert RL.FusedStep.hits - h0 == (1 if fused else 0)
            res.append((l.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
        finally:
            RL.FusedStep.enabled = True
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
