"""Pins the CPU oracle (oracle/ref_cpu.py) to vectors produced by the reference itself
(tools/make_golden.py) and to the reference's only in-tree known-answer test
(reactranker/metrics.py:82-90).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from tests import helpers as Hh

TOL = 2e-6


def _close(a, b, tol=TOL, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.max(np.abs(a - b) / (1.0 + np.abs(b))) if a.size else 0.0
    assert err <= tol, f"{what}: max rel-abs err {err:.3e} > {tol}"


@pytest.mark.parametrize("path", Hh.model_case_files(), ids=lambda p: p.split("model_")[-1][:-4])
def test_pack_batch_matches_reference_batchmolgraph(path):
    d, cfg = Hh.load_case(path)
    qb = Hh.case_queries(cfg)
    for prefix, specs in (("r_", qb.r_specs), ("p_", qb.p_specs)):
        g = O.pack_batch(specs)
        for k in ("f_atoms", "f_bonds", "a2b", "b2a", "b2revb", "a2a", "a_scope"):
            assert np.array_equal(np.asarray(g[k]), d[prefix + k]), (prefix, k)
    assert int(O.pack_batch(qb.r_specs)["K"]) == int(d["K_r"])
    assert int(O.pack_batch(qb.p_specs)["K"]) == int(d["K_p"])
    assert np.array_equal(qb.targets, d["targets"])


@pytest.mark.parametrize("faithful", [False, True])
@pytest.mark.parametrize("path", Hh.model_case_files(), ids=lambda p: p.split("model_")[-1][:-4])
def test_model_forward_and_grads(path, faithful):
    d, cfg = Hh.load_case(path)
    H = cfg["hidden_size"]
    shapes = O.model_shapes(H, cfg["mpnn_depth"], cfg["mpnn_diff_depth"], cfg["ffn_depth"], cfg["task_num"],
                            cfg["add_features_dim"], cfg["use_bias"])
    w = Hh.case_weights(d, cfg, shapes)
    assert set(w) == set(shapes)
    for k in shapes:
        assert tuple(w[k].shape) == tuple(shapes[k]), k
    head = O.resolve_task_type(cfg["task_num"], cfg["ffn_last_layer"], cfg["task_type"])
    assert head == cfg["head"]
    mc = dict(depth=cfg["mpnn_depth"], diff_depth=cfg["mpnn_diff_depth"], ffn_depth=cfg["ffn_depth"],
              task_type=head)
    rg, pg = Hh.golden_graph(d, "r_"), Hh.golden_graph(d, "p_")
    add = d["add_features"] if "add_features" in d.files else None
    targets = torch.tensor(d["targets"])
    scope = cfg["scope"]

    P = O.params_from_numpy(w, requires_grad=True)
    out, parts = O.reaction_forward(P, mc, rg, pg, add, faithful=faithful, return_parts=True)
    n = d["r_h"].shape[0]
    _close(parts["r_h"].detach().numpy()[:n], d["r_h"], what="r_h")
    _close(parts["p_h"].detach().numpy()[:n], d["p_h"], what="p_h")
    _close(parts["vecs"].detach().numpy(), d["vecs"], what="vecs")
    _close(out.detach().numpy(), d["out"], what="out")

    def check(loss, lname, gprefix):
        _close(loss.detach().numpy().reshape(-1), np.asarray(d[lname]).reshape(-1), what=lname)
        names = [k for k in P if P[k].requires_grad]
        grads = torch.autograd.grad(loss.sum(), [P[k] for k in names], allow_unused=True, retain_graph=True)
        seen = 0
        for k, g in zip(names, grads):
            key = gprefix + "." + k
            if key not in d.files:
                continue
            seen += 1
            g = torch.zeros_like(P[k]) if g is None else g
            ref = d[key]
            got = Hh.sample_like(g.numpy(), H)
            scale = max(1.0, float(np.abs(ref).max()))
            _close(got / scale, ref / scale, tol=5e-6, what=key)
        assert seen > 0

    if "loss_mle" in d.files:
        s = out if out.dim() == 1 else out[:, 0]
        check(O.listmle_loss(s, scope, targets), "loss_mle", "gmle")
    if "loss_listnet" in d.files:
        check(O.listnet_loss(out, scope, targets), "loss_listnet", "glistnet")
    if "loss_mse" in d.files:
        l = O.mse_loss(out, targets)
        _close(l.detach().numpy(), d["loss_mse"], what="mse")
        if H < 300:
            check(l, "loss_mse", "gmse")
    if "loss_evidential" in d.files:
        check(O.evidential_ranking_loss(out, scope, targets), "loss_evidential", "gevidential")
    if "loss_gauss" in d.files:
        var = out[:, 1] if "with_softplus" in head else torch.exp(out[:, 1])
        check(O.gauss_nll_loss(out[:, 0], var, targets), "loss_gauss", "ggauss")
    if "loss_lin" in d.files:
        check((out * torch.linspace(0.5, 1.5, out.numel()).view_as(out)).sum(), "loss_lin", "glin")

    # ordering + NDCG@10 (eval.py:516-519, metrics.py)
    sc = out.detach()
    sc1 = (sc[:, 0] if sc.dim() > 1 else sc).numpy()
    order, nd, off = [], [], 0
    for cnt in scope:
        o = O.ranking_order(sc1[off:off + cnt].tolist())
        order.extend(o)
        ts = d["targets"][off:off + cnt]
        rel = np.argsort(np.argsort(ts)).astype(np.float64) / max(1, cnt - 1) * 4.0
        nd.append(O.ndcg(rel[o], 10))
        off += cnt
    assert np.array_equal(np.asarray(order, np.int32), d["order"])
    _close(np.asarray(nd), d["ndcg10"], tol=1e-12, what="ndcg10")


LOSS_CASES = ["single", "tiny", "c32", "c64", "ragged", "long"]


@pytest.mark.parametrize("name", LOSS_CASES)
def test_losses(name, golden_dir):
    L = np.load(golden_dir + "/losses.npz")
    P = name + "."
    scope = L[P + "scope"].tolist()
    score = torch.tensor(L[P + "score"], requires_grad=True)
    var = torch.tensor(L[P + "var"], requires_grad=True)
    targets = torch.tensor(L[P + "targets"])

    l = O.listmle_loss(score, scope, targets)
    g, = torch.autograd.grad(l.sum(), score)
    _close(l.detach(), L[P + "mle"], what="mle"); _close(g, L[P + "mle_g"], what="mle_g")
    l = O.listnet_loss(score, scope, targets)
    g, = torch.autograd.grad(l, score)
    _close(l.detach(), L[P + "listnet"], what="listnet"); _close(g, L[P + "listnet_g"], what="listnet_g")
    l = O.evidential_ranking_loss(torch.stack([score, var], 1), scope, targets)
    gs, gv = torch.autograd.grad(l.sum(), [score, var])
    _close(l.detach(), L[P + "evid"], what="evid")
    _close(gs, L[P + "evid_gs"], tol=1e-5, what="evid_gs"); _close(gv, L[P + "evid_gv"], tol=1e-5, what="evid_gv")
    l = O.mse_loss(score, targets)
    g, = torch.autograd.grad(l, score)
    _close(l.detach(), L[P + "mse"]); _close(g, L[P + "mse_g"])
    l = O.gauss_nll_loss(score, var, targets)
    gs, gv = torch.autograd.grad(l, [score, var])
    _close(l.detach(), L[P + "gauss"]); _close(gs, L[P + "gauss_gs"]); _close(gv, L[P + "gauss_gv"], tol=1e-5)

    for sigma in (1.0, 0.5):
        ls, pairs = O.ranknet_sum_session(score, scope, targets, sigma)
        assert pairs == float(L[P + "rank_pairs"])
        if pairs == 0:
            assert np.isnan(L[P + f"rank_ss_{sigma}"])
            continue
        g, = torch.autograd.grad(ls / pairs, score)
        _close((ls / pairs).detach(), L[P + f"rank_ss_{sigma}"], what="rank_ss")
        _close(g, L[P + f"rank_ss_g_{sigma}"], what="rank_ss_g")
        lam = O.ranknet_lambda(score.detach(), scope, targets, sigma) / pairs
        _close(lam, L[P + f"rank_ag_g_{sigma}"], what="rank_ag_g")
        _close((ls / pairs).detach(), L[P + f"rank_ag_{sigma}"], what="rank_ag")


def test_logcumsumexp_and_overflow(golden_dir):
    L = np.load(golden_dir + "/losses.npz")
    for nm in ("lce_small", "lce_large"):
        x = torch.tensor(L[nm + ".x"], requires_grad=True)
        y = O.LogCumsumExp.apply(x)
        g, = torch.autograd.grad(y, x, torch.tensor(L[nm + ".go"]))
        _close(y.detach(), L[nm + ".y"], what=nm)
        ref = L[nm + ".g"]
        s = max(1.0, float(np.abs(ref).max()))
        _close(g.numpy() / s, ref / s, what=nm + ".g")
    ls, pairs = O.ranknet_sum_session(torch.tensor(L["rank_overflow.score"]), [3],
                                      torch.tensor(L["rank_overflow.targets"]), 1.0)
    assert np.isinf(float(ls / pairs)) and np.isinf(float(L["rank_overflow.loss"]))


def test_metrics_known_answers(golden_dir):
    # the reference's own self-test (reactranker/metrics.py:82-90)
    t = [3, 2, 3, 0, 1, 2, 3, 2]
    assert 6.861 < O.dcg(t, 6, "identity") < 6.862
    assert 0.785 < O.ndcg(t, 6, "identity") < 0.786
    assert 0 < O.ndcg(t, 10) < 1.0
    assert 0 < O.ndcg([1, 2, 3], 10) < 1.0
    Mx = np.load(golden_dir + "/metrics.npz")
    assert abs(O.dcg(t, 6, "identity") - float(Mx["selftest.dcg6_identity"])) < 1e-12
    assert abs(O.ndcg(t, 10) - float(Mx["selftest.ndcg10_exp2"])) < 1e-12
    for i in range(4):
        rel = Mx[f"rand{i}.rel"]
        assert abs(O.ndcg(rel, 10) - float(Mx[f"rand{i}.ndcg10"])) < 1e-12
        assert abs(O.ndcg(rel, 5, "identity") - float(Mx[f"rand{i}.ndcg5_id"])) < 1e-12
        assert abs(O.compute_ndcg_eval(Mx[f"rand{i}.truth"], Mx[f"rand{i}.pred"]) - float(Mx[f"rand{i}.eval_ndcg"])) < 1e-12
        assert O.ranking_order(Mx[f"rand{i}.scores"].tolist()) == Mx[f"rand{i}.order"].tolist()


def test_ranking_metrics_against_reference_loop(golden_dir):
    """oracle.ranking_metrics_from_scores vs the reference's own ranking_metrics run on preset scores."""
    E = np.load(golden_dir + "/eval_metrics.npz")
    for name in ("plain", "ties"):
        scope = E[name + ".scope"].tolist()
        offs = np.cumsum([0] + scope)
        sc = [E[name + ".scores"][a:b] for a, b in zip(offs[:-1], offs[1:])]
        tg = [E[name + ".targets"][a:b] for a, b in zip(offs[:-1], offs[1:])]
        top1, rec, top25, nd, orders = O.ranking_metrics_from_scores(sc, tg)
        assert top1 == float(E[name + ".top1"]) and top25 == float(E[name + ".top25"])
        assert abs(rec - float(E[name + ".recall25"])) < 1e-15
        assert np.allclose(nd, E[name + ".ndcg"], rtol=0, atol=1e-14)
        assert np.array_equal(np.concatenate(orders).astype(np.int32), E[name + ".order"])


TOP_SCORES_CASES = ("plain", "pred_ties", "ties", "two_col", "scaled", "two_col_scaled")


def top_scores_case(golden_dir, name):
    """(per-query score rows, per-query targets, scope, scaler or None, npz) of one tests/golden/top_scores.npz case."""
    E = np.load(golden_dir + "/top_scores.npz")
    scope = E[name + ".scope"].tolist()
    offs = np.cumsum([0] + scope)
    sc = [E[name + ".scores"][a:b] for a, b in zip(offs[:-1], offs[1:])]
    tg = [E[name + ".targets"][a:b] for a, b in zip(offs[:-1], offs[1:])]
    scaler = tuple(E[name + ".scaler"].tolist()) if name + ".scaler" in E.files else None
    return sc, tg, scope, scaler, E


@pytest.mark.parametrize("name", TOP_SCORES_CASES)
def test_top_scores_and_calculate_ndcg_against_reference_loops(golden_dir, name):
    """oracle.top_scores_from_scores / calculate_ndcg_from_scores vs the reference's own evaluate_top_scores
    (eval.py:76-177) and calculate_ndcg (:329-457) driven with preset scores (tools/make_golden.py gen_top_scores)."""
    sc, tg, _, scaler, E = top_scores_case(golden_dir, name)
    first = [s[:, 0] if s.ndim > 1 else s for s in sc]                        # eval.py:125-126, :390-394
    for ratio in (0.25, 0.1, 0.5):
        a, b, c, _ = O.top_scores_from_scores(first, tg, ratio)
        want = E[f"{name}.top_scores_r{ratio}"]
        assert a == want[0] and abs(b - want[1]) < 1e-15 and c == want[2], (name, ratio)
    means, stds = scaler if scaler is not None else (None, None)
    for cut in (0.5, 0.25):
        nd, kl, _ = O.calculate_ndcg_from_scores(first, tg, cut, means, stds)
        if "ties" not in name:      # the reference ranks with torch.sort / argsort(stable=False): with tied keys and
            #                         more than 16 elements its order - hence its NDCG - is implementation-defined
            assert abs(nd - float(E[f"{name}.ndcg_c{cut}"])) < 1e-7, (name, cut)
        assert abs(kl - float(E[f"{name}.kl"])) < 1e-7 * max(1.0, abs(kl)), name
