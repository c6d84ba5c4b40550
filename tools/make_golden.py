#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (imported from /root/reference).

Runs only in the build container (the reference never travels to the GPU box).  The
reference's modules are imported unmodified; `rdkit` (absent here) is replaced by an
in-process stub that provides just the names featurization.py/scaffold.py touch at import
time.  Synthetic graphs (reactranker_amd.synth) are handed to the reference's own
BatchMolGraph as duck-typed MolGraph objects, weights come from a numpy formula
(synth.seeded_weights), and every stored output is produced by reference code:

  models:  build_model(...) -> ReactionModel.forward, MPN / MPNDiff intermediates
  losses:  MLEloss, ListnetLoss, evidential_ranking, GaussDisLoss, LogCumsumExp, nn.MSELoss
  ranknet: the real factorized_training_loop ('sum_session' and 'accelerate_grad') driven
           with stub data-processor / optimizer objects
  metrics: reactranker.metrics.NDCG / DCG, train.eval.compute_NDCG, eval's sorted() ordering; ranking_metrics,
           evaluate_top_scores and calculate_ndcg driven with preset scores
  featurizer: MolGraph(smiles) on molecule descriptions served through a stand-in Chem namespace (tests/fake_rdkit.py)

Usage: python tools/make_golden.py            (writes tests/golden/)
"""
import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"


def _install_rdkit_stub():
    rdkit = types.ModuleType("rdkit")
    chem = types.ModuleType("rdkit.Chem")
    rdchem = types.ModuleType("rdkit.Chem.rdchem")
    scaff = types.ModuleType("rdkit.Chem.Scaffolds")
    murcko = types.ModuleType("rdkit.Chem.Scaffolds.MurckoScaffold")

    class _Params:
        removeHs = False

    class _Hyb:
        SP, SP2, SP3, SP3D, SP3D2 = range(1, 6)

    class _BT:
        SINGLE, DOUBLE, TRIPLE, AROMATIC = range(1, 5)

    chem.SmilesParserParams = _Params
    chem.Mol = object
    chem.BondType = _BT
    rdchem.HybridizationType = _Hyb
    rdchem.Atom = object
    rdchem.Bond = object
    chem.rdchem = rdchem
    scaff.MurckoScaffold = murcko
    chem.Scaffolds = scaff
    rdkit.Chem = chem
    sys.modules.update({"rdkit": rdkit, "rdkit.Chem": chem, "rdkit.Chem.rdchem": rdchem,
                        "rdkit.Chem.Scaffolds": scaff,
                        "rdkit.Chem.Scaffolds.MurckoScaffold": murcko})


_install_rdkit_stub()
sys.path.insert(0, REF)

from reactranker.features.featurization import BatchMolGraph, ATOM_FDIM, BOND_FDIM  # noqa: E402
from reactranker.models.base_model import build_model  # noqa: E402
from reactranker.train.loss import (MLEloss, ListnetLoss, evidential_ranking,  # noqa: E402
                                    GaussDisLoss, LogCumsumExp)
from reactranker.train.train_pairwise import factorized_training_loop  # noqa: E402
from reactranker.train import eval as ref_eval  # noqa: E402
from reactranker import metrics as ref_metrics  # noqa: E402
from reactranker.utils import index_select_ND  # noqa: E402

from reactranker_amd import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
assert ATOM_FDIM == 61 and BOND_FDIM == 22


def ref_batch(specs):
    return BatchMolGraph([synth.ListMolGraph(s) for s in specs])


def batch_arrays(bg, prefix):
    f_atoms, f_bonds, a2b, b2a, b2revb, a_scope, b_scope = bg.get_components()
    return {
        prefix + "f_atoms": f_atoms.numpy(), prefix + "f_bonds": f_bonds.numpy(),
        prefix + "a2b": a2b.numpy().astype(np.int32), prefix + "b2a": b2a.numpy().astype(np.int32),
        prefix + "b2revb": b2revb.numpy().astype(np.int32),
        prefix + "a2a": bg.get_a2a().numpy().astype(np.int32),
        prefix + "a_scope": np.asarray(a_scope, np.int32).reshape(-1, 2),
        prefix + "b_scope": np.asarray(b_scope, np.int32).reshape(-1, 2),
    }


MODEL_CASES = [
    # name, build_model kwargs, queries spec, extras
    dict(name="A_h32_d3_mle", hidden_size=32, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True,
         task_num=1, ffn_last_layer="with_softplus", task_type=None, add_features_dim=1,
         scope=[3, 5, 8, 13], seed=11, wseed=101),
    dict(name="B_h32_d1_nobias_gauss", hidden_size=32, mpnn_depth=1, mpnn_diff_depth=1, ffn_depth=2,
         use_bias=False, task_num=2, ffn_last_layer="no_softplus", task_type=None, add_features_dim=0,
         scope=[4, 6], seed=12, wseed=102),
    dict(name="C_h32_d2_evidential_ranking", hidden_size=32, mpnn_depth=2, mpnn_diff_depth=2, ffn_depth=3,
         use_bias=True, task_num=2, ffn_last_layer="no_softplus", task_type="evidential_ranking",
         add_features_dim=1, scope=[7, 2, 9], seed=13, wseed=103),
    dict(name="D_h32_d6_gauss_softplus", hidden_size=32, mpnn_depth=6, mpnn_diff_depth=3, ffn_depth=3,
         use_bias=True, task_num=2, ffn_last_layer="with_softplus", task_type=None, add_features_dim=1,
         scope=[5, 5], seed=14, wseed=104),
    dict(name="E_h32_kmix", hidden_size=32, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True,
         task_num=1, ffn_last_layer="no_softplus", task_type=None, add_features_dim=1,
         scope=[4, 3], seed=15, wseed=105, kmix=True),
    dict(name="F_h300_d3_c64", hidden_size=300, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True,
         task_num=1, ffn_last_layer="with_softplus", task_type=None, add_features_dim=1,
         scope=[64, 64], seed=16, wseed=106),
    dict(name="G_h32_dd0_listnet_softplus", hidden_size=32, mpnn_depth=2, mpnn_diff_depth=0, ffn_depth=1,
         use_bias=True, task_num=1, ffn_last_layer="with_softplus", task_type="listnet",
         add_features_dim=0, scope=[6, 4], seed=17, wseed=107),
    dict(name="H_h64_evidential4", hidden_size=64, mpnn_depth=2, mpnn_diff_depth=2, ffn_depth=3,
         use_bias=True, task_num=4, ffn_last_layer="with_softplus", task_type=None, add_features_dim=1,
         scope=[5, 3], seed=18, wseed=108),
    # hidden size NOT a multiple of 4 (no 16-byte rows): the generic kernels and the non-fused ReLU-backward branch
    dict(name="I_h30_d4_mle", hidden_size=30, mpnn_depth=4, mpnn_diff_depth=4, ffn_depth=3, use_bias=True,
         task_num=1, ffn_last_layer="with_softplus", task_type=None, add_features_dim=1,
         scope=[6, 9, 4], seed=19, wseed=109),
]


def kmix_queries(seed, scope):
    """Reactants are chains (K_r = 2); products gain a third neighbour on some atom (K_p = 3)."""
    rng = np.random.default_rng(seed)
    r_specs, p_specs, targets = [], [], []
    for c in scope:
        n = int(rng.integers(5, 9))
        r = synth.random_reactant(rng, n, max_degree=2)
        for _ in range(c):
            r_specs.append(r)
            p_specs.append(synth.random_product(rng, r, max_degree=4))
        targets.append(rng.standard_normal(c).astype(np.float32))
    m = sum(scope)
    return synth.QueryBatch(r_specs, p_specs, list(scope), np.concatenate(targets),
                            rng.random((m, 1)).astype(np.float32))


def gen_model_case(c):
    kw = {k: c[k] for k in ("hidden_size", "mpnn_depth", "mpnn_diff_depth", "ffn_depth", "use_bias",
                            "task_num", "ffn_last_layer", "task_type", "add_features_dim")}
    model = build_model(dropout=0.0, **kw)
    model.eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    w = synth.seeded_weights(shapes, c["wseed"])
    model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    qb = kmix_queries(c["seed"], c["scope"]) if c.get("kmix") else \
        synth.make_queries(c["seed"], len(c["scope"]), c["scope"], atoms_lo=5, atoms_hi=12)
    rb, pb = ref_batch(qb.r_specs), ref_batch(qb.p_specs)
    add = qb.add_features if c["add_features_dim"] else None
    targets = torch.tensor(qb.targets)

    out = {"cfg": json.dumps({**kw, "scope": c["scope"], "seed": c["seed"], "wseed": c["wseed"],
                              "kmix": bool(c.get("kmix", False)),
                              "head": model.ffn.task_type})}
    out.update(batch_arrays(rb, "r_"))
    out.update(batch_arrays(pb, "p_"))
    out["K_r"], out["K_p"] = np.int32(rb.max_num_bonds), np.int32(pb.max_num_bonds)
    out["targets"] = qb.targets
    if add is not None:
        out["add_features"] = add
    big = c["hidden_size"] >= 300
    if not big:
        for k, v in w.items():
            out["w." + k] = v

    # forward with intermediates (reference modules, eval mode)
    r_h = model.encoder.forward(rb, gpu=None)
    p_h = model.encoder.forward(pb, gpu=None)
    diff = p_h - r_h
    vecs = model.diff_encoder(diff, pb, gpu=None, features_batch=add)
    score = model(rb, pb, gpu=None, add_features=add)
    assert torch.equal(score, model.ffn(vecs))
    out["r_h"] = r_h.detach().numpy() if not big else r_h.detach().numpy()[:64]
    out["p_h"] = p_h.detach().numpy() if not big else p_h.detach().numpy()[:64]
    out["vecs"] = vecs.detach().numpy()
    out["out"] = score.detach().numpy()

    # one gather primitive sample (utils.py:176-193)
    msg = torch.relu(model.encoder.W_i(rb.f_bonds))
    out["gather_sum_r"] = index_select_ND(msg, rb.a2b).sum(dim=1).detach().numpy()[:64]

    def grads_of(loss):
        model.zero_grad()
        loss.backward()
        g = {}
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            a = p.grad.detach().numpy()
            if big and a.ndim == 2:
                a = a[::7, ::5]          # strided sample keeps the fixture small
            g[k] = a.copy()
        return g

    scope = c["scope"]
    if c["task_num"] == 1:
        s = model(rb, pb, gpu=None, add_features=add)
        l_mle = MLEloss()(s, scope, targets, None)
        out["loss_mle"] = l_mle.detach().numpy()
        for k, v in grads_of(l_mle).items():
            out["gmle." + k] = v
        s = model(rb, pb, gpu=None, add_features=add)
        l_ln = ListnetLoss()(s, scope, targets, None)
        out["loss_listnet"] = l_ln.detach().numpy()
        for k, v in grads_of(l_ln).items():
            out["glistnet." + k] = v
        s = model(rb, pb, gpu=None, add_features=add)
        l_mse = torch.nn.MSELoss()(s, targets)
        out["loss_mse"] = l_mse.detach().numpy()
        if not big:
            for k, v in grads_of(l_mse).items():
                out["gmse." + k] = v
    elif c["task_num"] == 2:
        s = model(rb, pb, gpu=None, add_features=add)
        if model.ffn.task_type == "evidential_ranking":
            l_ev = evidential_ranking()(s, scope, targets, 0.01, 0, 10, None)
            out["loss_evidential"] = l_ev.detach().numpy()
            for k, v in grads_of(l_ev).items():
                out["gevidential." + k] = v
        else:
            var = s[:, 1] if "with_softplus" in model.ffn.task_type else torch.exp(s[:, 1])
            l_g = GaussDisLoss()(s[:, 0], var, targets, None)
            out["loss_gauss"] = l_g.detach().numpy()
            for k, v in grads_of(l_g).items():
                out["ggauss." + k] = v
            s = model(rb, pb, gpu=None, add_features=add)
            l_mle = MLEloss()(s[:, 0], scope, targets, None)
            out["loss_mle"] = l_mle.detach().numpy()
            for k, v in grads_of(l_mle).items():
                out["gmle." + k] = v
    else:
        s = model(rb, pb, gpu=None, add_features=add)
        l = (s * torch.linspace(0.5, 1.5, s.numel()).view_as(s)).sum()
        out["loss_lin"] = l.detach().numpy()
        for k, v in grads_of(l).items():
            out["glin." + k] = v

    # ordering + NDCG per query (eval.py:516-519; metrics.py NDCG(10))
    sc = score.detach()
    sc1 = sc[:, 0] if sc.dim() > 1 else sc
    orders, nd10 = [], []
    off = 0
    for cnt in scope:
        ps = sc1[off:off + cnt].tolist()
        ts = qb.targets[off:off + cnt].tolist()
        srt = sorted(enumerate(ps), key=lambda x: x[1], reverse=True)
        order = [i for i, _ in srt]
        orders.extend(order)
        # relevance = rank-derived non-negative grades so exp2 gains stay finite
        rel = np.argsort(np.argsort(ts)).astype(np.float64) / max(1, cnt - 1) * 4.0
        nd10.append(ref_metrics.NDCG(10, "exp2").evaluate(rel[order]))
        off += cnt
    out["order"] = np.asarray(orders, np.int32)
    out["ndcg10"] = np.asarray(nd10, np.float64)
    np.savez_compressed(os.path.join(OUT, f"model_{c['name']}.npz"), **out)
    print("wrote model case", c["name"], "out", tuple(score.shape),
          {k: float(np.ravel(v)[0]) for k, v in out.items() if k.startswith("loss_")})


# ------------------------------------------------------------------ loss-only cases
LOSS_SCOPES = {
    "single": [1],
    "tiny": [2, 3],
    "c32": [32, 32, 32, 32],
    "c64": [64, 64, 64],
    "ragged": [100, 7, 64, 1, 129, 33],
    "long": [300, 65],
}


class _StubProcessor:
    def __init__(self, scope, targets):
        self.scope, self.targets = scope, targets

    def generate_batch_per_query(self, **kw):
        off = 0
        for c in self.scope:
            X = np.array([["r", "p%d" % i] for i in range(c)])
            yield X, self.targets[off:off + c].astype(np.float64), None
            off += c


class _StubGraphs:
    def parsing_smiles(self, smi):
        return len(smi)


class _StubModel:
    """Hands the reference loop preset leaf tensors so its backward lands in their .grad."""

    def __init__(self, leaves_used):
        self._it = iter(leaves_used)

    def __call__(self, r_batch, p_batch, gpu=None, add_features=None):
        return next(self._it)

    def zero_grad(self):
        pass


class _Noop:
    def step(self):
        pass


def run_ref_ranknet(scores, scope, targets, sigma, algo):
    """Drive the real factorized_training_loop; returns (mean minibatch loss, grad, used)."""
    offs = np.cumsum([0] + list(scope[:-1]))
    used = [bool((targets[o:o + c][:, None] - targets[o:o + c][None, :] > 0).sum() > 0)
            for o, c in zip(offs, scope)]
    leaves = [s.clone().requires_grad_(True) for s in scores.split(list(scope))]
    # queries with no positive pair are skipped BEFORE the model call (train_pairwise.py:103-104)
    model = _StubModel([l for l, u in zip(leaves, used) if u])
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        val = factorized_training_loop(0, model, None, _Noop(), _Noop(), _StubGraphs(),
                                       _StubProcessor(scope, targets), batch_size=10 ** 9, sigma=sigma,
                                       training_algo=algo, gpu=None, smiles_list=None,
                                       target_name="ea", add_features_name=None)
    grad = torch.cat([l.grad if l.grad is not None else torch.zeros_like(l) for l in leaves])
    return float(val), grad.numpy(), np.asarray(used)


def gen_losses():
    out = {}
    rng = np.random.default_rng(2024)
    for name, scope in LOSS_SCOPES.items():
        m = sum(scope)
        score = torch.tensor(rng.standard_normal(m).astype(np.float32) * 1.5, requires_grad=True)
        targ_np = np.concatenate([rng.permutation(c).astype(np.float32) * 0.37 - 0.1 * c
                                  + rng.random(1).astype(np.float32) for c in scope])
        # standardise like the trainer does (train_listwise.py:66-122) so softmax(targets) is tame
        targ_np = ((targ_np - targ_np.mean()) / (targ_np.std() + 1e-6)).astype(np.float32)
        targets = torch.tensor(targ_np)
        var = torch.tensor((np.log1p(np.exp(rng.standard_normal(m))) + 1e-6).astype(np.float32),
                           requires_grad=True)
        P = f"{name}."
        out[P + "scope"] = np.asarray(scope, np.int32)
        out[P + "score"] = score.detach().numpy()
        out[P + "targets"] = targ_np
        out[P + "var"] = var.detach().numpy()

        l = MLEloss()(score, scope, targets, None)
        g, = torch.autograd.grad(l.sum(), score)
        out[P + "mle"], out[P + "mle_g"] = l.detach().numpy(), g.numpy()

        l = ListnetLoss()(score, scope, targets, None)
        g, = torch.autograd.grad(l, score)
        out[P + "listnet"], out[P + "listnet_g"] = l.detach().numpy(), g.numpy()

        poss = torch.stack([score, var], dim=1)
        l = evidential_ranking()(poss, scope, targets, 0.01, 0, 10, None)
        gs, gv = torch.autograd.grad(l.sum(), [score, var])
        out[P + "evid"], out[P + "evid_gs"], out[P + "evid_gv"] = l.detach().numpy(), gs.numpy(), gv.numpy()

        l = torch.nn.MSELoss()(score, targets)
        g, = torch.autograd.grad(l, score)
        out[P + "mse"], out[P + "mse_g"] = l.detach().numpy(), g.numpy()

        l = GaussDisLoss()(score, var, targets, None)
        gs, gv = torch.autograd.grad(l, [score, var])
        out[P + "gauss"], out[P + "gauss_gs"], out[P + "gauss_gv"] = l.detach().numpy(), gs.numpy(), gv.numpy()

        for sigma in (1.0, 0.5):
            v, g, used = run_ref_ranknet(score.detach(), scope, targ_np, sigma, "sum_session")
            out[P + f"rank_ss_{sigma}"], out[P + f"rank_ss_g_{sigma}"] = np.float64(v), g
            v2, g2, _ = run_ref_ranknet(score.detach(), scope, targ_np, sigma, "accelerate_grad")
            out[P + f"rank_ag_{sigma}"], out[P + f"rank_ag_g_{sigma}"] = np.float64(v2), g2
            out[P + "rank_used"] = used
        pairs = 0.0
        off = 0
        for c in scope:
            t = targ_np[off:off + c]
            pairs += 2.0 * float(((t[:, None] - t[None, :]) > 0).sum())
            off += c
        out[P + "rank_pairs"] = np.float64(pairs)

    # LogCumsumExp standalone (loss.py:9-61) incl. large inputs that exercise the un-shifted exp(x)
    for nm, x_np in (("lce_small", rng.standard_normal(17).astype(np.float32)),
                     ("lce_large", (rng.standard_normal(40) * 15 + 20).astype(np.float32))):
        x = torch.tensor(x_np, requires_grad=True)
        go = torch.tensor(rng.standard_normal(len(x_np)).astype(np.float32))
        y = LogCumsumExp.apply(x)
        g, = torch.autograd.grad(y, x, go)
        out[nm + ".x"], out[nm + ".go"], out[nm + ".y"], out[nm + ".g"] = x_np, go.numpy(), y.detach().numpy(), g.numpy()

    # RankNet overflow (hazard H4): naive log(1+exp(x)) gives inf for x > 88
    sc = torch.tensor(np.array([100.0, 0.0, -5.0], np.float32))
    tg = np.array([0.0, 1.0, 2.0], np.float32)
    v, g, _ = run_ref_ranknet(sc, [3], tg, 1.0, "sum_session")
    out["rank_overflow.score"], out["rank_overflow.targets"] = sc.numpy(), tg
    out["rank_overflow.loss"] = np.float64(v)
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)
    print("wrote losses.npz with", len(out), "arrays")


def gen_metrics():
    out = {}
    t = [3, 2, 3, 0, 1, 2, 3, 2]
    out["selftest.targets"] = np.asarray(t, np.float64)
    out["selftest.dcg6_identity"] = np.float64(ref_metrics.DCG(6, "identity").evaluate(t))
    out["selftest.ndcg6_identity"] = np.float64(ref_metrics.NDCG(6, "identity").evaluate(t))
    out["selftest.ndcg10_exp2"] = np.float64(ref_metrics.NDCG(10).evaluate(t))
    out["selftest.ndcg10_exp2_123"] = np.float64(ref_metrics.NDCG(10).evaluate([1, 2, 3]))
    rng = np.random.default_rng(7)
    for i, n in enumerate((5, 10, 37, 64)):
        rel = rng.integers(0, 5, size=n).astype(np.float64)
        out[f"rand{i}.rel"] = rel
        out[f"rand{i}.ndcg10"] = np.float64(ref_metrics.NDCG(10).evaluate(rel))
        out[f"rand{i}.ndcg5_id"] = np.float64(ref_metrics.NDCG(5, "identity").evaluate(rel))
        truth = np.sort(rng.standard_normal(n))[::-1]
        pred = rng.permutation(truth)
        out[f"rand{i}.truth"], out[f"rand{i}.pred"] = truth, pred
        out[f"rand{i}.eval_ndcg"] = np.float64(ref_eval.compute_NDCG(list(truth), list(pred)))
        sc = np.round(rng.standard_normal(n), 1)     # rounded -> ties exercise sorted() stability
        out[f"rand{i}.scores"] = sc
        out[f"rand{i}.order"] = np.asarray(
            [k for k, _ in sorted(enumerate(sc.tolist()), key=lambda x: x[1], reverse=True)], np.int32)
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    print("wrote metrics.npz")


def gen_eval_metrics():
    """Drive the reference's own ranking_metrics (train/eval.py:475-555) with preset scores."""
    rng = np.random.default_rng(77)
    out = {}
    for name, scope, decimals in (("plain", [1, 2, 3, 8, 10, 64, 5, 7], 6), ("ties", [4, 9, 16, 6, 2, 33], 1)):
        scores = [np.round(rng.standard_normal(c) * 1.3, decimals).astype(np.float32) for c in scope]
        targets = [np.round(rng.standard_normal(c), decimals).astype(np.float32) for c in scope]

        class _Model:
            def __init__(self):
                self.i = 0

            def eval(self):
                pass

            def __call__(self, r, p, gpu=None, add_features=None):
                t = torch.tensor(scores[self.i])
                self.i += 1
                return t

        class _DP:
            def generate_batch_per_query(self, **kw):
                for c, t in zip(scope, targets):
                    yield np.array([["r", "p%d" % i] for i in range(c)]), t, None
        top1, recall25, top25, nd = ref_eval.ranking_metrics(_Model(), None, _DP(), _StubGraphs(), show_info=False)
        out[name + ".scope"] = np.asarray(scope, np.int32)
        out[name + ".scores"] = np.concatenate(scores)
        out[name + ".targets"] = np.concatenate(targets)
        out[name + ".top1"], out[name + ".recall25"], out[name + ".top25"] = np.float64(top1), np.float64(recall25), np.float64(top25)
        out[name + ".ndcg"] = np.asarray(nd, np.float64)
        # per-query pieces, recomputed with the reference's own helpers for finer-grained checks
        orders, nd10 = [], []
        for sc, tg in zip(scores, targets):
            order = [k for k, _ in sorted(enumerate(sc.tolist()), key=lambda x: x[1], reverse=True)]
            orders.extend(order)
            rel = np.maximum(np.round(tg.astype(np.float64) * 2 + 3), 0)      # non-negative grades for exp2 gains
            nd10.append(ref_metrics.NDCG(10, "exp2").evaluate(rel[order]) if rel.max() > 0 else np.nan)
        out[name + ".order"] = np.asarray(orders, np.int32)
        out[name + ".ndcg10_rel"] = np.asarray(nd10, np.float64)
    np.savez_compressed(os.path.join(OUT, "eval_metrics.npz"), **out)
    print("wrote eval_metrics.npz", {k: v for k, v in out.items() if k.endswith((".top1", ".recall25", ".top25"))})


def gen_top_scores():
    """Drive the reference's own evaluate_top_scores (train/eval.py:76-177) and calculate_ndcg (:329-457) with preset
    scores: the validation metric of the RankNet epoch driver (run_train_pairwise.py:91-96) and what both test()
    functions report (test_listwise.py:51-66, test_ranknet.py:59-64)."""
    import contextlib
    import io
    rng = np.random.default_rng(404)
    out = {}
    cases = (   # name, scope, queries per model call, decimals of scores, decimals of targets, 2-column output, scaler
        ("plain", [3, 2, 8, 10, 64, 5, 7, 4, 13], 3, 6, 6, False, None),
        ("pred_ties", [4, 9, 16, 6, 2, 33, 12], 2, 1, 6, False, None),
        ("ties", [4, 9, 16, 6, 2, 33], 2, 1, 1, False, None),
        ("two_col", [6, 11, 2, 40, 9], 2, 6, 6, True, None),
        ("scaled", [5, 12, 3, 21], 4, 6, 6, False, (0.37, 1.9)),
        ("two_col_scaled", [7, 10, 30, 4], 3, 6, 6, True, (-1.2, 0.6)),
    )
    for name, scope, qpb, dec_s, dec_t, two_col, scaler in cases:
        M = int(sum(scope))
        scores = np.round(rng.standard_normal(M) * 1.3, dec_s).astype(np.float32)
        targets = np.round(rng.standard_normal(M), dec_t).astype(np.float32)
        if two_col:
            scores = np.stack([scores, np.abs(rng.standard_normal(M)).astype(np.float32) + 0.1], axis=1)
        offs = np.concatenate([[0], np.cumsum(scope)])

        class _DP:
            def generate_batch_querys(self, **kw):
                assert kw["shuffle_query"] is False and kw["shuffle_batch"] is False
                for q0 in range(0, len(scope), qpb):
                    sc = scope[q0:q0 + qpb]
                    lo, hi = offs[q0], offs[q0 + len(sc)]
                    X = np.array([["r", "p%d" % i] for i in range(hi - lo)])
                    yield X, targets[lo:hi].astype(np.float64)[:, None], list(sc), None

        class _Model:
            def __init__(self):
                self.at = 0

            def __call__(self, r, p, gpu=None, add_features=None):
                n = r                      # _StubGraphs.parsing_smiles returns the batch's row count
                t = torch.tensor(scores[self.at:self.at + n])
                self.at += n
                return t

        out[name + ".scope"] = np.asarray(scope, np.int32)
        out[name + ".scores"], out[name + ".targets"] = scores, targets
        for ratio in (0.25, 0.1, 0.5):
            with contextlib.redirect_stdout(io.StringIO()):
                a, b, c = ref_eval.evaluate_top_scores(_Model(), None, _DP(), _StubGraphs(), ratio=ratio, batch_size=qpb,
                                                       show_info=False, smiles_list=None, target_name="stdea")
            out[f"{name}.top_scores_r{ratio}"] = np.asarray([a, b, c], np.float64)
        for cut in (0.5, 0.25):
            kw = {}
            if scaler is not None:
                kw = dict(means=scaler[0], stds=scaler[1])
                out[name + ".scaler"] = np.asarray(scaler, np.float64)
            with contextlib.redirect_stdout(io.StringIO()):
                nd, kl, order, _ = ref_eval.calculate_ndcg(_Model(), None, _DP(), _StubGraphs(), batch_size=qpb,
                                                           NDCG_cut=cut, show_info=False, target_name="stdea", **kw)
            out[f"{name}.ndcg_c{cut}"] = np.float64(nd)
            out[f"{name}.kl"] = np.float64(kl)
            out[f"{name}.order_rows"] = np.asarray(order, np.float64)
    np.savez_compressed(os.path.join(OUT, "top_scores.npz"), **out)
    print("wrote top_scores.npz", {k: v for k, v in out.items() if ".top_scores_r0.25" in k or ".ndcg_c0.5" in k or k.endswith(".kl")})


def gen_train_utils():
    """Learning-rate schedule and optimizer defaults of reactranker/train/utils.py (NoamLR, build_optimizer,
    build_lr_scheduler), produced by the reference classes themselves."""
    from reactranker.train.utils import build_lr_scheduler, build_optimizer, param_count
    out = {}
    cfgs = [dict(warmup_epochs=2, total_epochs=10, train_data_size=1000, batch_size=50, init_lr=1e-4, max_lr=1e-3,
                 final_lr=1e-4),
            dict(warmup_epochs=2.0, total_epochs=25, train_data_size=100000, batch_size=64, init_lr=1e-4, max_lr=1e-3,
                 final_lr=1e-4),
            dict(warmup_epochs=1, total_epochs=3, train_data_size=37, batch_size=5, init_lr=5e-5, max_lr=2e-3,
                 final_lr=1e-5)]
    for i, c in enumerate(cfgs):
        net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 1))
        opt = build_optimizer(net)
        sch = build_lr_scheduler(opt, **c)
        total = int(c["total_epochs"] * (c["train_data_size"] // c["batch_size"]))
        n = min(total + 5, 400)                          # past the end as well (lr stays at final_lr)
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(n):
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        sch.step(current_step=7)
        out[f"c{i}.cfg"] = np.array(json.dumps(c))
        out[f"c{i}.lrs"] = np.asarray(lrs, np.float64)
        out[f"c{i}.jump7"] = np.float64(opt.param_groups[0]["lr"])
        out[f"c{i}.warmup_steps"] = np.int64(sch.warmup_steps)
        out[f"c{i}.total_steps"] = np.int64(sch.total_steps)
        g = opt.param_groups[0]
        out[f"c{i}.adam"] = np.asarray([g["lr"] if False else 1e-4, g["weight_decay"], g["betas"][0], g["betas"][1], g["eps"]],
                                       np.float64)
        out[f"c{i}.param_count"] = np.int64(param_count(net))
    np.savez_compressed(os.path.join(OUT, "train_utils.npz"), **out)
    print("wrote train_utils.npz", [float(out["c0.lrs"][k]) for k in (0, 1, 40, 41, 200)])


def gen_standardize():
    """Target standardisation / sign flip of the reference trainers, produced by running the reference's OWN
    train() (train/train_listwise.py:66-122) and run_train() (train/run_train_pairwise.py:36-45) with epochs=0:
    the functions standardise, build their loss objects and return before the first epoch.  They deep-copy their
    DataFrames, so the frames handed in answer __deepcopy__ with themselves and the 'std<target>' column the
    reference writes can be read back.  (torch.utils.tensorboard is absent here: a module with a dummy
    SummaryWriter stands in for the import only.)"""
    import logging
    import pandas as pd
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    from reactranker.train.train_listwise import train as ref_train
    from reactranker.train.run_train_pairwise import run_train as ref_run_train

    class KeepDF(pd.DataFrame):
        @property
        def _constructor(self):
            return KeepDF

        def __deepcopy__(self, memo):
            return self

    rng = np.random.default_rng(7)
    tr = rng.normal(12.0, 4.0, 23)
    va = rng.normal(11.0, 5.0, 9)
    log = logging.getLogger("golden")
    out = {"train_raw": tr, "val_raw": va}
    net = torch.nn.Linear(1, 1)
    i = 0
    for target_name in ("ea", "lgk", "lgk_bi"):
        for norm in (True, False, 0.5, "1,5"):
            for save_metric in (None, "NDCG@1"):
                dtr = KeepDF({"rsmi": [f"r{k % 5}" for k in range(len(tr))], target_name: tr.copy()})
                dva = KeepDF({"rsmi": [f"r{k % 3}" for k in range(len(va))], target_name: va.copy()})
                ref_train(net, None, dtr, dva, None, None, 0, None, 2, 0, None, task_type="mle", writer=None, logger=log,
                          target_name=target_name, save_metric=save_metric, normalize_target=norm)
                out[f"l{i}.cfg"] = np.array(json.dumps(dict(target_name=target_name, normalize_target=norm,
                                                            save_metric=save_metric)))
                out[f"l{i}.train"] = np.asarray(dtr["std" + target_name], np.float64)
                out[f"l{i}.val"] = np.asarray(dva["std" + target_name], np.float64)
                i += 1
    out["n_listwise"] = np.int64(i)
    for j, target_name in enumerate(("ea", "lgk")):
        dtr = KeepDF({"rsmi": [f"r{k % 5}" for k in range(len(tr))], target_name: tr.copy()})
        dva = KeepDF({"rsmi": [f"r{k % 3}" for k in range(len(va))], target_name: va.copy()})
        ref_run_train(net, None, dtr, dva, None, None, 0, None, 2, 0, None, "sum_session", "baseline", None, log, None,
                      target_name)
        out[f"p{j}.target_name"] = np.array(target_name)
        out[f"p{j}.train"] = np.asarray(dtr["std" + target_name], np.float64)
        out[f"p{j}.val"] = np.asarray(dva["std" + target_name], np.float64)
    out["mean"], out["std"] = np.float64(tr.mean()), np.float64(tr.std())
    np.savez_compressed(os.path.join(OUT, "standardize.npz"), **out)
    print("wrote standardize.npz:", i, "listwise variants, 2 pairwise;", out["l0.train"][:3], out["l2.train"][:3])


def gen_featurizer():
    """The reference's own MolGraph (featurization.py:135-210) on molecule descriptions served through a stand-in Chem
    namespace (tests/fake_rdkit.py): pins feature layout, atom order and bond numbering - not RDKit's chemistry."""
    from reactranker.features import featurization as ref_feat
    from tests import fake_rdkit
    descs = fake_rdkit.descriptions()
    fake = fake_rdkit.chem_namespace(descs)
    # the enumerations the reference froze at import (this script's stub) and the stand-in's must agree
    assert [fake.rdchem.HybridizationType.SP, fake.rdchem.HybridizationType.SP3D2] == [ref_feat.Chem.rdchem.HybridizationType.SP, ref_feat.Chem.rdchem.HybridizationType.SP3D2]
    assert [fake.BondType.SINGLE, fake.BondType.AROMATIC] == [ref_feat.Chem.BondType.SINGLE, ref_feat.Chem.BondType.AROMATIC]
    ref_feat.Chem.MolFromSmiles = fake.MolFromSmiles
    ref_feat.Chem.AddHs = fake.AddHs
    ref_feat.Chem.RemoveHs = fake.RemoveHs
    out = {"descriptions": np.array(json.dumps(descs))}
    for name in descs:
        for reaction in (True, False):
            g = ref_feat.MolGraph(name, reaction=reaction)
            k = f"{name}.{'rxn' if reaction else 'plain'}"
            out[k + ".f_atoms"] = np.asarray(g.f_atoms, np.float32).reshape(g.n_atoms, ATOM_FDIM)
            out[k + ".f_bonds"] = np.asarray(g.f_bonds, np.float32).reshape(g.n_bonds, ATOM_FDIM + BOND_FDIM)
            out[k + ".b2a"] = np.asarray(g.b2a, np.int32)
            out[k + ".b2revb"] = np.asarray(g.b2revb, np.int32)
            a2b = np.full((g.n_atoms, max([len(r) for r in g.a2b] + [1])), -1, np.int32)
            for i, r in enumerate(g.a2b):
                a2b[i, :len(r)] = r
            out[k + ".a2b"] = a2b
    np.savez_compressed(os.path.join(OUT, "featurizer.npz"), **out)
    print("wrote featurizer.npz:", len(descs), "molecule descriptions x 2 atom orders")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(4)
    if len(sys.argv) > 2 and sys.argv[1] == "--only":      # e.g. --only train_utils, or --only model:I_h30_d4_mle
        if sys.argv[2].startswith("model:"):
            gen_model_case(next(c for c in MODEL_CASES if c["name"] == sys.argv[2][6:]))
        else:
            globals()["gen_" + sys.argv[2]]()
        sys.exit(0)
    for case in MODEL_CASES:
        gen_model_case(case)
    gen_losses()
    gen_metrics()
    gen_eval_metrics()
    gen_top_scores()
    gen_train_utils()
    gen_standardize()
    gen_featurizer()
