"""CPU oracle for the ReactRanker hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A PyTorch-CPU fp32 restatement of the reference's D-MPNN reaction encoder and per-query
ranking losses.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product package (reactranker_amd) never does and fails loudly
when its HIP library is missing.

Parity status: PINNED.  tools/make_golden.py imports the reference itself from
/root/reference in the build container (with an in-process stub for the absent `rdkit`
package) and writes tests/golden/*.npz; tests/test_oracle_golden.py checks every function
here against those vectors, and against the reference's only in-tree known-answer test
(reactranker/metrics.py:82-90).

Each function cites the reference lines it restates (paths relative to /root/reference).
Two variants of the model forward exist: `faithful=True` keeps the reference's
per-molecule readout loop and per-query loss loops (its real cost profile);
`faithful=False` uses segment ops so the CPU baseline is not penalised by the reference's
quadratic-in-batch backward.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

ATOM_FDIM = 61     # reactranker/features/featurization.py:63
BOND_FDIM = 22     # reactranker/features/featurization.py:64


# --------------------------------------------------------------------------- batching
def pack_batch(specs: Sequence, K: Optional[int] = None) -> Dict[str, np.ndarray]:
    """BatchMolGraph.__init__ restated (features/featurization.py:246-288).

    `specs` are reactranker_amd.synth.MolSpec-like objects (n_atoms, f_atoms, directed()).
    Row 0 of every array is the padding row (:255-264); a2b is right-padded with 0 to
    K = max(1, max in-degree) (:281,286).  `K` overrides the pad width (must be >= that).
    """
    f_atoms = [np.zeros((1, ATOM_FDIM), np.float32)]
    f_bonds = [np.zeros((1, ATOM_FDIM + BOND_FDIM), np.float32)]
    a2b: List[List[int]] = [[]]
    b2a = [0]
    b2revb = [0]
    a_scope = []
    b_scope = []
    n_atoms, n_bonds = 1, 1
    for s in specs:
        fb, lb2a, lb2revb, la2b = s.directed()
        f_atoms.append(np.asarray(s.f_atoms, np.float32).reshape(s.n_atoms, ATOM_FDIM))
        f_bonds.append(fb)
        for a in range(s.n_atoms):
            a2b.append([b + n_bonds for b in la2b[a]])
        b2a.extend((n_atoms + lb2a).tolist())
        b2revb.extend((n_bonds + lb2revb).tolist())
        a_scope.append((n_atoms, s.n_atoms))
        b_scope.append((n_bonds, s.n_bonds))
        n_atoms += s.n_atoms
        n_bonds += s.n_bonds
    kmax = max(1, max(len(x) for x in a2b))
    if K is None:
        K = kmax
    assert K >= kmax
    a2b_arr = np.zeros((n_atoms, K), np.int64)
    for a, lst in enumerate(a2b):
        a2b_arr[a, :len(lst)] = lst
    b2a_arr = np.asarray(b2a, np.int64)
    return dict(
        f_atoms=np.concatenate(f_atoms, 0), f_bonds=np.concatenate(f_bonds, 0),
        a2b=a2b_arr, b2a=b2a_arr, b2revb=np.asarray(b2revb, np.int64),
        a2a=b2a_arr[a2b_arr],                                  # get_a2a, :326-327
        a_scope=np.asarray(a_scope, np.int64).reshape(-1, 2),
        b_scope=np.asarray(b_scope, np.int64).reshape(-1, 2),
        K=np.int64(K))


def _t(x, dtype=None):
    t = torch.as_tensor(x)
    return t if dtype is None else t.to(dtype)


def index_select_nd(source: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """utils.py:176-193."""
    return source.index_select(0, index.reshape(-1)).view(index.shape + source.shape[1:])


# --------------------------------------------------------------------------- model
def model_shapes(hidden=300, depth=3, diff_depth=3, ffn_depth=3, task_num=1, add_features_dim=0,
                 bias=True) -> Dict[str, tuple]:
    """state_dict names/shapes of build_model(...) (models/base_model.py:235-297)."""
    H, F_ = hidden, add_features_dim
    s = {"encoder.cached_zero_vector": (H,), "diff_encoder.cached_zero_vector": (H,)}

    def lin(name, out, inp, b=True):
        s[name + ".weight"] = (out, inp)
        if b:
            s[name + ".bias"] = (out,)
    lin("encoder.W_i", H, ATOM_FDIM + BOND_FDIM, bias)
    if depth > 1:
        lin("encoder.W_h", H, H, bias)
    lin("encoder.W_o", H, ATOM_FDIM + H, True)                 # models/mpn.py:59 (always biased)
    lin("diff_encoder.W_i", H, H, bias)
    if diff_depth > 1:
        lin("diff_encoder.W_h", H, H + ATOM_FDIM + BOND_FDIM, bias)
    if diff_depth > 0:
        lin("diff_encoder.W_o", H, 2 * H, True)                # models/mpn.py:168
    # FFN Sequential indices: [Dropout, Linear, (ReLU, Dropout, Linear)*]  (base_model.py:32-57)
    if ffn_depth == 1:
        lin("ffn.ffn.1", task_num, H + F_, bias)
    else:
        lin("ffn.ffn.1", H, H + F_, bias)
        idx = 4
        for _ in range(ffn_depth - 2):
            lin(f"ffn.ffn.{idx}", H, H, bias)
            idx += 3
        lin(f"ffn.ffn.{idx}", task_num, H, bias)
    return s


def resolve_task_type(task_num=2, ffn_last_layer="no_softplus", task_type=None) -> str:
    """build_model's head-string logic (models/base_model.py:252-264)."""
    if task_type is None:
        if task_num == 2:
            return "gaussian_" + ffn_last_layer
        if task_num == 4:
            return "evidential_" + ffn_last_layer
        return ffn_last_layer
    if task_type == "evidential_ranking":
        return task_type
    return task_type + "_" + ffn_last_layer


def _linear(P, name, x):
    b = P.get(name + ".bias")
    return F.linear(x, P[name + ".weight"], b)


def _drop(x, masks, key, p):
    """Dropout with an externally supplied keep-mask (hazard H2: RNG streams cannot match)."""
    if masks is None or p == 0.0:
        return x
    return x * masks[key] / (1.0 - p)


def _relu(x, key, gates=None, trace=None):
    """torch.relu - or, for the gate-attribution tests, the same layer with its gates dictated from outside:
    `gates[key]` (0/1, same shape) replaces the sign test, so a pre-activation within rounding of zero opens exactly
    where the arithmetic under test opened it.  `trace[key]` records the pre-activation."""
    if trace is not None:
        trace[key] = x.detach()
    if gates is not None and key in gates:
        return x * gates[key].to(x.dtype)
    return torch.relu(x)


def mpn_forward(P, g, depth, prefix="encoder", masks=None, p=0.0, tag="r", gates=None, trace=None):
    """MPN.forward with return_atom_hiddens=True (models/mpn.py:61-108)."""
    f_atoms, f_bonds = g["f_atoms"], g["f_bonds"]
    a2b, b2a, b2revb = g["a2b"], g["b2a"], g["b2revb"]
    inp = _linear(P, prefix + ".W_i", f_bonds)                              # :80
    message = _relu(inp, f"{tag}.enc.in", gates, trace)                    # :81
    for it in range(depth - 1):                                            # :84
        a_message = index_select_nd(message, a2b).sum(dim=1)               # :89-90
        rev_message = message[b2revb]                                      # :91
        message = a_message[b2a] - rev_message                             # :92
        message = _linear(P, prefix + ".W_h", message)                     # :94
        message = _relu(inp + message, f"{tag}.enc.{it}", gates, trace)    # :95
        message = _drop(message, masks, f"{tag}.enc.{it}", p)              # :97
    a_message = index_select_nd(message, a2b).sum(dim=1)                   # :101-102
    a_input = torch.cat([f_atoms, a_message], dim=1)                       # :103
    atom_hiddens = _relu(_linear(P, prefix + ".W_o", a_input), f"{tag}.enc.out", gates, trace)   # :104
    return _drop(atom_hiddens, masks, f"{tag}.enc.out", p)                 # :105


def readout_mean(atom_hiddens, a_scope, faithful):
    """Per-molecule mean readout (models/mpn.py:224-235)."""
    if faithful:
        vecs = []
        for a_start, a_size in a_scope.tolist():
            if a_size == 0:
                vecs.append(torch.zeros(atom_hiddens.shape[1]))
            else:
                vecs.append(atom_hiddens.narrow(0, a_start, a_size).sum(dim=0) / a_size)
        return torch.stack(vecs, dim=0)
    M = a_scope.shape[0]
    sizes = a_scope[:, 1]
    seg = torch.repeat_interleave(torch.arange(M), sizes)
    # molecules are packed back to back starting at row 1 (featurization.py:276-278)
    rows = atom_hiddens[1:1 + int(sizes.sum())]
    out = torch.zeros(M, atom_hiddens.shape[1], dtype=atom_hiddens.dtype).index_add_(0, seg, rows)
    return out / sizes.clamp(min=1).to(atom_hiddens.dtype).unsqueeze(1)


def mpn_diff_forward(P, atom_features, g, depth, features_batch=None, prefix="diff_encoder",
                     masks=None, p=0.0, faithful=False, gates=None, trace=None):
    """MPNDiff.forward (models/mpn.py:170-240); bond_fdim = 83 so the slice :206 keeps every column."""
    f_bonds, a2b, a2a = g["f_bonds"], g["a2b"], g["a2a"]
    inp = _linear(P, prefix + ".W_i", atom_features)                       # :194
    message = _relu(inp, "diff.in", gates, trace)                          # :195
    if depth > 0:
        for it in range(depth - 1):                                        # :199
            nei_a = index_select_nd(message, a2a)                          # :201
            nei_fb = index_select_nd(f_bonds, a2b)                         # :202,206
            message = torch.cat((nei_a, nei_fb), dim=2).sum(dim=1)         # :208-209
            message = _linear(P, prefix + ".W_h", message)                 # :211
            message = _relu(inp + message, f"diff.{it}", gates, trace)     # :212
            message = _drop(message, masks, f"diff.{it}", p)               # :213
        a_message = index_select_nd(message, a2a).sum(dim=1)               # :215-216
        a_input = torch.cat([atom_features, a_message], dim=1)             # :217
        atom_hiddens = _relu(_linear(P, prefix + ".W_o", a_input), "diff.out", gates, trace)   # :218
        atom_hiddens = _drop(atom_hiddens, masks, "diff.out", p)           # :219
    else:
        atom_hiddens = _drop(message, masks, "diff.out", p)                # :221
    vecs = readout_mean(atom_hiddens, g["a_scope"], faithful)              # :224-235
    if features_batch is not None:
        vecs = torch.cat([vecs, features_batch], dim=1)                    # :237-238
    return vecs


def softplus(x):
    return F.softplus(x)      # torch.nn.Softplus(): beta=1, threshold=20


def ffn_forward(P, x, ffn_depth, task_type, masks=None, p=0.0, gates=None, trace=None):
    """FFN.forward (models/base_model.py:59-108): Dropout before every Linear, ReLU between."""
    h = _drop(x, masks, "ffn.0", p)
    if ffn_depth == 1:
        out = _linear(P, "ffn.ffn.1", h)
    else:
        h = _linear(P, "ffn.ffn.1", h)
        idx = 4
        for li in range(ffn_depth - 2):
            h = _drop(_relu(h, f"ffn.{li + 1}", gates, trace), masks, f"ffn.{li + 1}", p)
            h = _linear(P, f"ffn.ffn.{idx}", h)
            idx += 3
        h = _drop(_relu(h, f"ffn.{ffn_depth - 1}", gates, trace), masks, f"ffn.{ffn_depth - 1}", p)
        out = _linear(P, f"ffn.ffn.{idx}", h)
    out = out.squeeze(-1)                                                  # :60
    mv = 1e-6
    if task_type == "evidential_with_softplus":                            # :61-70
        mu, ll, la, lb = torch.split(out, out.shape[1] // 4, dim=1)
        return torch.stack((mu, softplus(ll) + mv, softplus(la) + mv + 1, softplus(lb) + mv),
                           dim=2).view(out.size())
    if task_type in ("gauss_regression_with_softplus", "gaussian_with_softplus"):   # :71-82
        mu, lv = torch.split(out, out.shape[1] // 2, dim=1)
        return torch.stack((mu, softplus(lv)), dim=2).view(out.size())
    if task_type == "listnetdis_lognorm_with_softplus":                    # :83-90
        mu, lv = torch.split(out, out.shape[1] // 2, dim=1)
        return torch.stack((softplus(mu) + mv, softplus(lv) + mv), dim=2).view(out.size())
    if task_type == "evidential_ranking":                                  # :91-98
        sc, uf = torch.split(out, out.shape[1] // 2, dim=1)
        return torch.stack((sc, softplus(uf) + mv), dim=2).view(out.size())
    if task_type == "listnet_with_softplus":                               # :99-100
        return softplus(out)
    if task_type in ("listnet_with_uncertainty", "evidential"):            # :101-104
        return softplus(out) + 1
    return out                                                             # :105-106


def graph_tensors(g: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    out = {}
    for k in ("f_atoms", "f_bonds"):
        out[k] = _t(g[k], torch.float32)
    for k in ("a2b", "b2a", "b2revb", "a2a", "a_scope"):
        out[k] = _t(g[k], torch.int64)
    return out


def reaction_forward(P, cfg, r_graph, p_graph, add_features=None, masks=None, faithful=False,
                     return_parts=False, gates=None, trace=None):
    """ReactionModel.forward (models/base_model.py:150-171).

    cfg keys: depth, diff_depth, ffn_depth, task_type (already resolved), dropout.
    gates / trace: see _relu (keys `{r,p}.enc.{in,0..,out}`, `diff.{in,0..,out}`, `ffn.{1..}`).
    """
    p = float(cfg.get("dropout", 0.0)) if masks is not None else 0.0
    r = graph_tensors(r_graph) if not torch.is_tensor(r_graph["f_atoms"]) else r_graph
    pg = graph_tensors(p_graph) if not torch.is_tensor(p_graph["f_atoms"]) else p_graph
    r_h = mpn_forward(P, r, cfg["depth"], masks=masks, p=p, tag="r", gates=gates, trace=trace)       # :155
    p_h = mpn_forward(P, pg, cfg["depth"], masks=masks, p=p, tag="p", gates=gates, trace=trace)      # :156
    diff = p_h - r_h                                                       # :168
    if add_features is None:
        fb = None
    elif torch.is_tensor(add_features):
        fb = add_features                                    # caller chose the dtype (fp64 identity checks)
    else:
        fb = _t(np.asarray(add_features), torch.float32)
    vecs = mpn_diff_forward(P, diff, pg, cfg["diff_depth"], fb, masks=masks, p=p, faithful=faithful, gates=gates, trace=trace)
    out = ffn_forward(P, vecs, cfg["ffn_depth"], cfg["task_type"], masks=masks, p=p, gates=gates, trace=trace)   # :169
    if return_parts:
        return out, dict(r_h=r_h, p_h=p_h, diff=diff, vecs=vecs)
    return out


# --------------------------------------------------------------------------- losses
class LogCumsumExp(torch.autograd.Function):
    """train/loss.py:9-61 — including the un-shifted exp(x) in backward (:59, hazard H4)."""

    @staticmethod
    def forward(ctx, x):
        m, _ = torch.max(x, dim=0, keepdim=True)                           # :28
        y = torch.exp(x - m)                                               # :29-30
        rc = torch.flip(torch.cumsum(torch.flip(y, dims=[0]), dim=0), dims=[0])   # :32
        fd = torch.log(rc) + m                                             # :34
        ctx.save_for_backward(x, fd)
        return fd

    @staticmethod
    def backward(ctx, g):
        x, fd = ctx.saved_tensors
        return g * (torch.exp(x) * torch.cumsum(torch.exp(-fd), dim=0))    # :59


def listmle_loss(score, scope, targets):
    """MLEloss.forward (train/loss.py:69-99). Returns a [1] tensor like the reference."""
    losses = torch.zeros(1, dtype=score.dtype)
    for item, t in zip(score.split(scope, dim=0), targets.split(scope, dim=0)):
        idx = torch.argsort(t, descending=True, stable=True)               # :87 (ties: see DESIGN.md)
        s = torch.gather(item, 0, idx)                                     # :92
        losses = losses + torch.mean(LogCumsumExp.apply(s) - s)            # :93-95
    return losses / len(scope)                                             # :97


def listnet_loss(score, scope, targets):
    """ListnetLoss.forward (train/loss.py:327-352): ONE global mean over all candidates (:347)."""
    parts = []
    for item, t in zip(score.split(scope, dim=0), targets.split(scope, dim=0)):
        pred = torch.log(F.softmax(item, dim=0))                           # :339
        targ = F.softmax(t, dim=0)                                         # :341
        parts.append(-targ * pred)                                         # :343
    return torch.mean(torch.cat(parts, dim=0))                             # :344-347


def evidential_ranking_loss(poss, scope, targets):
    """evidential_ranking.forward live branch (train/loss.py:526-556); pi = 3.141592653 (:543)."""
    losses = torch.zeros(1, dtype=poss.dtype)
    for item, t in zip(poss.split(scope, dim=0), targets.split(scope, dim=0)):
        mu, var = item[:, 0], item[:, 1]                                   # :526-527
        pp = F.softmax(mu, dim=0)                                          # :531
        tp = F.softmax(t, dim=0)                                           # :532
        unc = 0.5 * (torch.log(tp) - torch.log(pp)) ** 2 / var + \
            0.5 * torch.log(2 * 3.141592653 * var)                         # :541-543
        pen = torch.abs(mu - t)                                            # :545
        losses = losses + torch.mean(-torch.log(tp) + unc + pen)           # :549-552
    return losses / len(scope)                                             # :554


def ranknet_pairs(targets_q: torch.Tensor):
    """pos/neg masks and pair count for one query (train/train_pairwise.py:98-106)."""
    rel = targets_q.reshape(-1, 1) - targets_q.reshape(1, -1)
    pos = (rel > 0).to(torch.float32)
    neg = (rel < 0).to(torch.float32)
    return pos, neg, 2.0 * pos.sum()


def ranknet_sum_session(score, scope, targets, sigma=1.0):
    """factorized_training_loop 'sum_session' (train/train_pairwise.py:117-122,141).

    Returns (loss_sum, pairs); queries without a positive pair are skipped (:103-104).
    The naive log(1+exp(x)) (overflows for x>88, hazard H4) is kept.
    """
    loss = torch.zeros((), dtype=score.dtype)
    pairs = 0.0
    for y, t in zip(score.split(scope, dim=0), targets.split(scope, dim=0)):
        pos, neg, npairs = ranknet_pairs(t)
        if float(npairs) == 0:
            continue
        y = y.unsqueeze(1)
        c_pos = torch.log(1 + torch.exp(-sigma * (y - y.t())))             # :119
        c_neg = torch.log(1 + torch.exp(sigma * (y - y.t())))              # :120
        loss = loss + torch.sum(pos * c_pos + neg * c_neg, (0, 1))         # :121-122
        pairs += float(npairs)
    return loss, pairs


def ranknet_lambda(score, scope, targets, sigma=1.0):
    """'accelerate_grad' closed-form lambdas (train/train_pairwise.py:125-133): row sums only."""
    outs = []
    for y, t in zip(score.split(scope, dim=0), targets.split(scope, dim=0)):
        pos, neg, npairs = ranknet_pairs(t)
        y = y.unsqueeze(1)
        l_pos = 1 + torch.exp(sigma * (y - y.t()))                         # :126
        l_neg = 1 + torch.exp(-sigma * (y - y.t()))                        # :127
        bl = -sigma * pos / l_pos + sigma * neg / l_neg                    # :128
        back = torch.sum(bl, dim=1)                                        # :133
        outs.append(back if float(npairs) > 0 else torch.zeros_like(back))
    return torch.cat(outs)


def mse_loss(out, targets):
    """nn.MSELoss() default branch (train/train_listwise.py:166-167,282-285)."""
    return torch.mean((out - targets) ** 2)


def gauss_nll_loss(mean, var, targets):
    """GaussDisLoss.forward (train/loss.py:154-162); pi = float32(np.pi) (:152)."""
    pi = torch.tensor([np.pi], dtype=torch.float32)
    mse = 0.5 * torch.log(2 * pi) + 0.5 * torch.log(var) + torch.pow(mean - targets, 2) / (2 * var)
    return torch.mean(mse)


# --------------------------------------------------------------------------- ranking / metrics
def ranking_order(scores: Sequence[float]) -> List[int]:
    """Predicted order of one query: python stable sort, descending (train/eval.py:516-519)."""
    return [i for i, _ in sorted(enumerate(list(scores)), key=lambda x: x[1], reverse=True)]


def dcg(targets, k=10, gain_type="exp2"):
    """metrics.py:29-57 (DCG.evaluate)."""
    t = np.asarray(targets, dtype=np.float64)[:k]
    gain = np.power(2.0, t) - 1.0 if gain_type == "exp2" else t
    disc = np.log2(np.arange(1, len(gain) + 1) + 1)
    return float(np.sum(gain / disc))


def ndcg(targets, k=10, gain_type="exp2"):
    """metrics.py:70-80 (NDCG.evaluate): targets listed in predicted rank order."""
    t = np.asarray(targets, dtype=np.float64)
    ideal = np.sort(t)[::-1]
    return dcg(t, k, gain_type) / dcg(ideal, k, gain_type)


def compute_ndcg_eval(truth, pred):
    """train/eval.py:460-472 (exp gain, all positions)."""
    n = len(truth)
    d = np.log2(np.arange(2, n + 2))
    return float(np.sum(np.exp(pred) / d) / np.sum(np.exp(truth) / d))


def ranking_metrics_from_scores(scores_per_query, targets_per_query):
    """ranking_metrics (train/eval.py:475-555) on given per-query scores: returns (top1, recall25, top25, NDCG_[4])
    plus the per-query orders.  Keeps the reference's quirks: python round() for the 25 % cut (:522), stable
    sorted(..., reverse=True) (:516-519), and NDCG2 computed on nested lists, i.e. without discount (:543)."""
    top1 = top25 = 0
    recall, nd, orders = [], [], []
    for pred, targ in zip(scores_per_query, targets_per_query):
        pred_scores, target_scores = [float(x) for x in pred], [float(x) for x in targ]
        n = len(target_scores)
        sp = sorted(enumerate(pred_scores), key=lambda x: x[1], reverse=True)
        st = sorted(enumerate(target_scores), key=lambda x: x[1], reverse=True)
        pidx, tidx = [i for i, _ in sp], [i for i, _ in st]
        tsorted = [v for _, v in st]
        orders.append(pidx)
        if pidx[0] == tidx[0]:
            top1 += 1
        len25 = round(n * 0.25)
        if len25 < 1:
            len25 = 1
        ptop, ttop = pidx[:len25], tidx[:len25]
        if ptop[0] in ttop:
            top25 += 1
        recall.append(sum(1 for i in ptop if i in ttop) / len25)
        prt = [target_scores[i] for i in pidx]
        nd.append([compute_ndcg_eval([tsorted[0]], [prt[0]]), compute_ndcg_eval([tsorted[:2]], [prt[:2]]),
                   compute_ndcg_eval(tsorted[:len25], prt[:len25]), compute_ndcg_eval(tsorted, prt)])
    q = len(orders)
    return top1 / q, float(np.mean(recall)), top25 / q, np.mean(nd, axis=0), orders


def top_scores_from_scores(scores_per_query, targets_per_query, ratio=0.25):
    """evaluate_top_scores (train/eval.py:76-177) on given per-query scores: returns (average_score,
    average_pred_in_targ, average_top1_in_pred) plus the per-query triples.  First-maximum top-1 (:133), python
    round() for the cut (:144-146), and - unlike ranking_metrics' third value - the TARGET's top-1 looked up in the
    PREDICTED top-`ratio` (:156-159)."""
    rows = []
    for pred, targ in zip(scores_per_query, targets_per_query):
        batch_preds, batch_targets = [float(x) for x in pred], [float(x) for x in targ]
        top1 = 1 if batch_targets.index(max(batch_targets)) == batch_preds.index(max(batch_preds)) else 0
        idx1 = [a for a, _ in sorted(enumerate(batch_targets), key=lambda x: x[1], reverse=True)]
        idx2 = [a for a, _ in sorted(enumerate(batch_preds), key=lambda x: x[1], reverse=True)]
        length = round(len(batch_preds) * ratio)
        if length == 0:
            length = 1
        num = sum(1 for i in range(length) if idx2[i] in idx1[:length])
        t1 = 1 if batch_targets.index(max(batch_targets)) in idx2[:length] else 0
        rows.append((top1, num / length, t1))
    m = np.asarray(rows, np.float64)
    return float(m[:, 0].mean()), float(m[:, 1].mean()), float(m[:, 2].mean()), m


def cal_ndcg(preds: torch.Tensor, targets: torch.Tensor, n: int):
    """train/eval.py:309-325 (cal_NDCG): linear gains, log2 discount, both truncated at n."""
    targets, preds = targets[:n], preds[:n]
    idcg = torch.sum(targets / torch.log2(torch.arange(targets.size(0), dtype=torch.float32) + 2))
    dcg_n = torch.sum(preds / torch.log2(torch.arange(preds.size(0), dtype=torch.float32) + 2))
    return dcg_n / idcg


def calculate_ndcg_from_scores(scores_per_query, targets_per_query, ndcg_cut=0.5, means=None, stds=None):
    """calculate_ndcg, is_order branch (train/eval.py:329-457) on given per-query scores (the first column of a 2-D
    output, :390-396, after the optional de-standardisation :379-388).  Returns (NDCG_mean, KL_mean, per-query rows
    [NDCG, KL]).  Rank-derived gains `length + 1 - order`, truncated at ceil(length * cut) positions of the TARGET
    order (:421-425); KL(softmax(targets) || softmax(preds)) with un-shifted exponentials (:401-404).  Ties are broken
    by position (a stable sort); torch.sort / argsort in the reference leave that open."""
    import math
    rows = []
    for pred, targ in zip(scores_per_query, targets_per_query):
        bp = torch.as_tensor(np.asarray(pred), dtype=torch.float32)
        bt = torch.as_tensor(np.asarray(targ), dtype=torch.float32)
        if means is not None:
            bp = (bp * stds) + means
        P = torch.exp(bt) / torch.sum(torch.exp(bt))
        Qd = torch.exp(bp) / torch.sum(torch.exp(bp))
        kl = torch.sum(P * torch.log(P / Qd))
        _, idx = torch.sort(bt, descending=True, stable=True)
        sorted_preds = bp[idx]
        pred_order = torch.argsort(torch.argsort(sorted_preds, descending=True, stable=True), stable=True) + 1
        length = bt.size(0)
        true_order = torch.arange(length, dtype=torch.float32) + 1
        nd = cal_ndcg(float(length + 1) - pred_order.float(), float(length + 1) - true_order, math.ceil(length * ndcg_cut))
        rows.append((float(nd), float(kl)))
    m = np.asarray(rows, np.float64)
    return float(m[:, 0].mean()), float(m[:, 1].mean()), m


def params_from_numpy(w: Dict[str, np.ndarray], requires_grad=False) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in w.items():
        t = torch.tensor(np.asarray(v), dtype=torch.float32)
        if requires_grad and not k.endswith("cached_zero_vector"):
            t.requires_grad_(True)
        out[k] = t
    return out
