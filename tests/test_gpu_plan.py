"""The whole-model step plan (rr_reaction_forward / rr_reaction_backward, csrc/plan.hip) against the per-op path of
reactranker_amd/functions.py: same kernels in the same order with the same dropout streams, so scores, loss and every
parameter gradient must be BIT-IDENTICAL - in all three reactant modes (plain, de-duplicated in eval mode, shared prefix
in train mode), with and without biases / add_features, for every head layout and depths 1..6."""
import numpy as np
import pytest
import torch

from reactranker_amd import featurization, synth
from reactranker_amd import functions as Fn
from reactranker_amd import loss as RL
from oracle import ref_cpu as O
from tests.test_gpu_model import make_model

pytestmark = pytest.mark.gpu


def _run(model, rb, pb, qb, seed, plan, loss="mle"):
    Fn.StepPlan.enabled = plan
    try:
        model.zero_grad()
        model.dropout_seed = seed
        add = qb.add_features if model.ffn.hidden_size > model.diff_encoder.hidden_size else None
        out = model(rb, pb, gpu=0, add_features=add)
        tg = torch.tensor(qb.targets)
        if loss == "mle":
            l = RL.MLEloss()(out if out.dim() == 1 else out[:, 0], qb.scope, tg, 0)
        else:
            l = (out * torch.linspace(0.5, 1.5, out.numel()).cuda().view_as(out)).sum()
        l.sum().backward()
        return out.detach().clone(), l.detach().clone(), {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None}
    finally:
        Fn.StepPlan.enabled = True


def _same(a, b):
    assert torch.equal(a[0], b[0]), float((a[0] - b[0]).abs().max())
    assert torch.equal(a[1], b[1])
    assert a[2].keys() == b[2].keys()
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), (k, float((a[2][k] - b[2][k]).abs().max()))


CASES = [
    # hidden, depth, diff_depth, ffn_depth, bias, task_num, last, task_type, F, dropout, train
    (32, 3, 3, 3, True, 1, "with_softplus", None, 1, 0.2, True),        # shared prefix + per-copy dropout
    (32, 3, 3, 3, True, 1, "with_softplus", None, 1, 0.0, False),       # eval: de-duplicated reactants
    (64, 2, 2, 2, False, 1, "no_softplus", None, 0, 0.1, True),         # no biases, no add_features, depth 2
    (32, 1, 0, 1, True, 1, "no_softplus", None, 1, 0.1, True),          # depth 1 / diff depth 0 / single FFN layer
    (32, 6, 6, 3, True, 2, "no_softplus", "evidential_ranking", 1, 0.1, True),
    (32, 3, 2, 3, True, 4, "with_softplus", None, 1, 0.0, False),       # evidential 4-parameter head
    (300, 3, 3, 3, True, 1, "with_softplus", None, 1, 0.1, True),       # the headline shape
    (600, 2, 2, 3, True, 2, "with_softplus", None, 1, 0.1, True),       # N = 600: two column blocks per row block
    (32, 11, 12, 2, True, 1, "no_softplus", None, 1, 0.1, True),        # depth >= 10: more than 8 per-iteration dZ buffers
    (32, 16, 16, 2, True, 1, "no_softplus", None, 0, 0.0, True),        # the deepest model a plan takes (MAXD)
]


@pytest.mark.parametrize("H,d,dd,fd,bias,tn,last,tt,F,p,train", CASES)
@pytest.mark.parametrize("dedup", ["auto", False])
def test_plan_is_bit_identical_to_the_per_op_path(H, d, dd, fd, bias, tn, last, tt, F, p, train, dedup):
    cfg = dict(hidden_size=H, mpnn_depth=d, mpnn_diff_depth=dd, ffn_depth=fd, use_bias=bias, task_num=tn, ffn_last_layer=last,
               task_type=tt, add_features_dim=F)
    w = synth.seeded_weights(O.model_shapes(H, d, dd, fd, tn, F, bias), 5)
    model = make_model(cfg, w, dropout=p)
    model = model.train() if train else model.eval()
    model.dedup_reactants = dedup
    qb = synth.make_queries(17, 4, [7, 3, 9, 5], atoms_lo=5, atoms_hi=14)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    loss = "mle" if tn == 1 else "lin"
    a = _run(model, rb, pb, qb, 4242, plan=False, loss=loss)
    b = _run(model, rb, pb, qb, 4242, plan=True, loss=loss)
    _same(a, b)
    _same(b, _run(model, rb, pb, qb, 4242, plan=True, loss=loss))          # run-to-run
    if p > 0:
        c = _run(model, rb, pb, qb, 4243, plan=True, loss=loss)             # another dropout stream: different numbers
        assert not torch.equal(b[0], c[0])


@pytest.mark.parametrize("d", [4, 6, 8])
def test_plan_sum_of_the_per_copy_gradients_on_the_gather_equals_the_pre_summed_form(d, monkeypatch):
    """Shared-prefix backward at depth >= 4: the per-copy layers' dZ ride on the gather over the copies
    (rr_gather_sum_multi_f32); RR_NO_GATHER_MULTI=1 pre-sums them with rr_axpby_f32 as before.  Same additions, same order."""
    cfg = dict(hidden_size=64, mpnn_depth=d, mpnn_diff_depth=2, ffn_depth=2, use_bias=True, task_num=1,
               ffn_last_layer="with_softplus", task_type=None, add_features_dim=1)
    w = synth.seeded_weights(O.model_shapes(64, d, 2, 2, 1, 1, True), 5)
    model = make_model(cfg, w, dropout=0.1).train()
    qb = synth.make_queries(11, 3, [6, 4, 8], atoms_lo=5, atoms_hi=12)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    a = _run(model, rb, pb, qb, 7, plan=True)
    monkeypatch.setenv("RR_NO_GATHER_MULTI", "1")
    _same(a, _run(model, rb, pb, qb, 7, plan=True))
    _same(a, _run(model, rb, pb, qb, 7, plan=False))


@pytest.mark.parametrize("p,dedup", [(0.1, "auto"), (0.0, False)])
def test_plan_last_weight_gradient_on_the_chain_equals_the_side_stream_placement(p, dedup, monkeypatch):
    """The last weight gradient of rr_reaction_backward runs on the caller's stream behind the side stream's work (round 5);
    RR_NO_TAIL_JOIN=1 puts it back on the side stream.  Same kernels, same order of accumulation: every gradient bit for bit
    (shared-prefix reactant pass with dropout, and the plain two-pass form)."""
    cfg = dict(hidden_size=64, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, task_num=1,
               ffn_last_layer="with_softplus", task_type=None, add_features_dim=1)
    w = synth.seeded_weights(O.model_shapes(64, 3, 3, 3, 1, 1, True), 5)
    model = make_model(cfg, w, dropout=p).train()
    model.dedup_reactants = dedup
    qb = synth.make_queries(13, 3, [6, 4, 8], atoms_lo=5, atoms_hi=12)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    a = _run(model, rb, pb, qb, 7, plan=True)
    monkeypatch.setenv("RR_NO_TAIL_JOIN", "1")
    _same(a, _run(model, rb, pb, qb, 7, plan=True))


@pytest.mark.parametrize("p,dedup,d", [(0.1, "auto", 3), (0.0, False, 3), (0.1, "auto", 5)])
def test_plan_weight_gradient_in_front_of_its_layers_gemm_equals_the_order_behind_it(p, dedup, d, monkeypatch):
    """RR_PLAN_WGRAD_EARLY (functions.WgradOrder): same kernels, same order on the weight-gradient stream - every gradient bit
    for bit in both orders; and the measuring mode settles on one of them without changing a bit either."""
    cfg = dict(hidden_size=64, mpnn_depth=d, mpnn_diff_depth=d, ffn_depth=3, use_bias=True, task_num=1,
               ffn_last_layer="with_softplus", task_type=None, add_features_dim=1)
    w = synth.seeded_weights(O.model_shapes(64, d, d, 3, 1, 1, True), 5)
    model = make_model(cfg, w, dropout=p).train()
    model.dedup_reactants = dedup
    qb = synth.make_queries(13, 3, [6, 4, 8], atoms_lo=5, atoms_hi=12)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    monkeypatch.setattr(Fn.WgradOrder, "mode", "late")
    a = _run(model, rb, pb, qb, 7, plan=True)
    monkeypatch.setattr(Fn.WgradOrder, "mode", "early")
    _same(a, _run(model, rb, pb, qb, 7, plan=True))
    monkeypatch.setattr(Fn.WgradOrder, "mode", "auto")
    Fn.WgradOrder.reset()
    for _ in range(Fn.WgradOrder.warm + 2 * Fn.WgradOrder.samples + 2):
        _same(a, _run(model, rb, pb, qb, 7, plan=True))
        torch.cuda.synchronize()
    assert Fn.WgradOrder.settled() and len(Fn.WgradOrder.choices()) == 1
    Fn.WgradOrder.reset()


def test_plan_without_side_and_aux_streams_and_no_grad_forward():
    cfg = dict(hidden_size=64, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, task_num=1,
               ffn_last_layer="with_softplus", task_type=None, add_features_dim=1)
    w = synth.seeded_weights(O.model_shapes(64, 3, 3, 3, 1, 1, True), 5)
    model = make_model(cfg, w, dropout=0.1).train()
    qb = synth.make_queries(3, 3, [6, 4, 8], atoms_lo=5, atoms_hi=12)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    a = _run(model, rb, pb, qb, 7, plan=True)
    Fn.SideStream.enabled = Fn.AuxStream.enabled = False
    try:
        b = _run(model, rb, pb, qb, 7, plan=True)
    finally:
        Fn.SideStream.enabled = Fn.AuxStream.enabled = True
    _same(a, b)
    model.eval()
    with torch.no_grad():
        o1 = model(rb, pb, gpu=0, add_features=qb.add_features)
        Fn.StepPlan.enabled = False
        try:
            o2 = model(rb, pb, gpu=0, add_features=qb.add_features)
        finally:
            Fn.StepPlan.enabled = True
    assert torch.equal(o1, o2)


def test_plan_rejects_a_too_small_workspace_before_launching_anything():
    import ctypes as C
    from reactranker_amd import _lib
    cfg = dict(hidden_size=32, mpnn_depth=2, mpnn_diff_depth=2, ffn_depth=2, use_bias=True, task_num=1,
               ffn_last_layer="no_softplus", task_type=None, add_features_dim=0)
    w = synth.seeded_weights(O.model_shapes(32, 2, 2, 2, 1, 0, True), 1)
    model = make_model(cfg, w).eval()
    qb = synth.make_queries(1, 2, [3, 4], atoms_lo=5, atoms_hi=8)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    model.dedup_reactants = False
    st = dict(r=rb.device_graph(0), p_graph=pb.device_graph(0), dedup=None, prefix=None, H=32, depth=2, diff_depth=2, p=0.0,
              seed=0, feat=None, F=0, head=0, squeeze=True)
    params = model.flat_params()
    out = torch.empty(pb.n_mols, 1).cuda()
    M, S, keep = Fn.StepPlan.build(st, params, out)
    need = int(_lib.lib().rr_reaction_workspace_bytes(C.byref(M), C.byref(S)))
    assert need > 0
    FL = _lib.RR_PLAN_F16X2_GEMM                     # the layout rr_reaction_workspace_bytes sizes for: the largest (it holds the magnitude slots)
    ws = torch.empty(need, dtype=torch.uint8).cuda()
    S.workspace, S.workspace_bytes = C.c_void_p(ws.data_ptr()), need // 20                   # far below the forward's share
    assert _lib.lib().rr_reaction_forward(C.byref(M), C.byref(S), FL, _lib.stream()) == -5     # RR_ERR_WORKSPACE
    S.workspace_bytes = need
    assert _lib.lib().rr_reaction_forward(C.byref(M), C.byref(S), FL, _lib.stream()) == 0
    S.workspace_bytes = need - 4096                                                           # forward fits, backward does not
    G = _lib.Grads()
    gr = [torch.empty_like(q) for q in params]
    for gi, wi in enumerate([0, 2, 4, 6, 8, 10, 12, 14]):
        G.w[gi], G.b[gi] = _lib.ptr(gr[wi]), _lib.ptr(gr[wi + 1])
    dout = torch.ones_like(out)
    assert _lib.lib().rr_reaction_backward(C.byref(M), C.byref(S), _lib.ptr(dout), C.byref(G), FL, _lib.stream()) == -5
    torch.cuda.synchronize()
