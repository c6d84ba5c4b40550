"""rr_ffn_chain_f32 (csrc/ffn.hip, ABI revision 8): the FFN head (models/base_model.py:32-60) and its input-gradient chain as ONE
launch each.  Held bit for bit (torch.equal) against the same layers issued one by one through rr_linear_f32 - the per-layer
path every other parity test of the repository holds against the oracle and the reference's vectors - for the shapes the
configs use and for ragged ones: rows not a multiple of 16, widths 32 ... 600, 2 ... 4 layers, one and two outputs, with and
without biases, dropout on and off.  The step plans use the chain; tests/test_gpu_plan.py compares them with the per-op mirror
(per-layer launches) on top of this."""
import ctypes as C

import pytest
import torch

from reactranker_amd import _lib
from reactranker_amd import functions as Fn
from reactranker_amd._lib import check, lib, ptr, stream

pytestmark = pytest.mark.gpu
dev = "cuda"


def _layers(widths, bias, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = []
    for k, n in zip(widths[:-1], widths[1:]):
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
        b = (torch.randn(n, generator=g) * 0.3).to(dev) if bias else None
        out.append(Fn.LinW(w, b, big=False))
    return out


def _chain_forward(x, K0, layers, p, seed):
    M = x.shape[0]
    a = _lib.FfnChainArgs()
    a.M, a.n_stages, a.x, a.ldx, a.drop_p, a.mask_scale = M, len(layers), ptr(x), x.stride(0), p, 1.0
    hs, keep = [], []
    k = K0
    for li, L in enumerate(layers):
        g = a.stage[li]
        n = L.w.shape[0]
        wp = L.pk(k)
        keep.append(wp)
        g.w, g.ldw, g.bias, g.n_out, g.n_in = ptr(wp), wp.stride(0), ptr(L.b), n, k
        if li < len(layers) - 1:
            o = torch.full((M, (n + 3) // 4 * 4), float("nan"), device=dev)
            g.relu, g.dropout, g.drop_seed = 1, 1, Fn._site_seed(seed, 4000 + li)
        else:
            o = torch.full((M, n), float("nan"), device=dev)
            g.rowdot = 1
        g.out, g.ld_out = ptr(o), o.stride(0)
        hs.append(o)
        k = n
    st = lib().rr_ffn_chain_f32(C.byref(a), stream())
    return st, hs


@pytest.mark.parametrize("M,widths,bias,p", [
    (4096, [301, 300, 300, 1], True, 0.1),        # configs[1..3]: H = 300, F = 1, ffn_depth 3
    (2048, [301, 300, 300, 1], True, 0.0),
    (4096, [601, 600, 600, 2], True, 0.1),        # configs[4]: H = 600, two outputs (evidential head)
    (37, [33, 32, 32, 1], False, 0.2),            # one partial row tile, no biases
    (1000, [300, 300, 1], True, 0.1),             # two layers, F = 0
    (515, [129, 64, 128, 96, 4], True, 0.3),      # four layers of different widths, four outputs
    (16400, [301, 300, 300, 1], True, 0.1),       # more rows than the per-layer path's small-M geometry (ranknet: 256 x 64)
])
def test_forward_chain_equals_the_layers_issued_one_by_one(M, widths, bias, p):
    torch.manual_seed(M)
    K0 = widths[0]
    x = torch.randn(M, (K0 + 3) // 4 * 4, device=dev)
    layers = _layers(widths, bias, 3)
    ref_out, (ref_hs, ref_raw) = Fn.ffn_forward(x[:, :K0], layers, p, 91, 0)
    st, hs = _chain_forward(x, K0, layers, p, 91)
    check(st, "rr_ffn_chain_f32")
    for li in range(len(layers) - 1):
        n = widths[li + 1]
        assert torch.equal(hs[li][:, :n], ref_hs[li + 1][:, :n]), (li, float((hs[li][:, :n] - ref_hs[li + 1][:, :n]).abs().max()))
    assert torch.equal(hs[-1], ref_raw), float((hs[-1] - ref_raw).abs().max())


@pytest.mark.parametrize("M,widths,bias,p", [
    (4096, [301, 300, 300, 1], True, 0.1),
    (4096, [601, 600, 600, 2], True, 0.1),
    (37, [33, 32, 32, 1], False, 0.2),
    (515, [129, 64, 128, 96, 4], True, 0.3),
    (16400, [301, 300, 300, 1], True, 0.0),
])
def test_backward_chain_equals_the_layers_issued_one_by_one(M, widths, bias, p):
    torch.manual_seed(M + 1)
    K0 = widths[0]
    nl = len(widths) - 1
    x = torch.randn(M, (K0 + 3) // 4 * 4, device=dev)
    layers = _layers(widths, bias, 5)
    _, saved = Fn.ffn_forward(x[:, :K0], layers, p, 17, 0)
    hs, raw = saved
    d = torch.randn(M, widths[-1], device=dev)
    dx_cols = K0 - 1 if nl > 1 else None            # the readout columns only (the appended feature column has no gradient)
    old = Fn.SideStream.enabled
    Fn.SideStream.enabled = False
    try:
        ref_dx, _ = Fn.ffn_backward(layers, p, 0, saved, d, need_dx=True, dx_cols=dx_cols)
    finally:
        Fn.SideStream.enabled = old
    ks = 1.0 / (1.0 - p)
    a = _lib.FfnChainArgs()
    a.M, a.n_stages, a.x, a.ldx, a.drop_p, a.mask_scale = M, nl, ptr(d), d.stride(0), 0.0, ks
    outs, keep = [], []
    for j in range(nl):
        li = nl - 1 - j
        L = layers[li]
        nin = L.w.shape[1] if (li > 0 or dx_cols is None) else dx_cols
        wt = L.pk_t(0, nin)
        keep.append(wt)
        g = a.stage[j]
        g.w, g.ldw, g.n_out, g.n_in = ptr(wt), wt.stride(0), nin, L.w.shape[0]
        o = torch.full((M, (nin + 3) // 4 * 4), float("nan"), device=dev)
        g.out, g.ld_out = ptr(o), o.stride(0)
        if li > 0:
            g.post_mask, g.ld_mask = ptr(hs[li]), hs[li].stride(0)
        outs.append(o)
    check(lib().rr_ffn_chain_f32(C.byref(a), stream()), "rr_ffn_chain_f32")
    n0 = ref_dx.shape[1]
    assert torch.equal(outs[-1][:, :n0], ref_dx), float((outs[-1][:, :n0] - ref_dx).abs().max())


def test_shapes_the_chain_does_not_take_are_refused_not_mangled():
    x = torch.randn(64, 304, device=dev)
    layers = _layers([301, 298, 1], True, 1)        # a hidden width that is not a multiple of 4
    st, _ = _chain_forward(x, 301, layers, 0.0, 1)
    assert st == -4                                  # RR_ERR_UNSUPPORTED: the caller issues the layers one by one
    layers = _layers([301, 300, 12], True, 1)       # more than 8 outputs for the row-dot stage
    st, _ = _chain_forward(x, 301, layers, 0.0, 1)
    assert st == -4


def test_plan_with_and_without_the_chain_is_the_same_step():
    """RR_PLAN_NO_FFN_CHAIN: scores, loss and every gradient of a training step are the same bits either way"""
    from oracle import ref_cpu as O
    from reactranker_amd import featurization, synth
    from tests.test_gpu_model import make_model
    from tests.test_gpu_plan import _run, _same
    for H, tn, tt, last in ((300, 1, None, "with_softplus"), (64, 2, "evidential_ranking", "no_softplus")):
        cfg = dict(hidden_size=H, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, task_num=tn, ffn_last_layer=last,
                   task_type=tt, add_features_dim=1)
        w = synth.seeded_weights(O.model_shapes(H, 3, 3, 3, tn, 1, True), 5)
        model = make_model(cfg, w, dropout=0.1).train()
        qb = synth.make_queries(17, 6, [7, 3, 9, 5, 64, 2], atoms_lo=5, atoms_hi=14)
        rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
        loss = "mle" if tn == 1 else "lin"
        try:
            Fn.FfnChain.enabled = True
            a = _run(model, rb, pb, qb, 4242, plan=True, loss=loss)
            Fn.FfnChain.enabled = False
            b = _run(model, rb, pb, qb, 4242, plan=True, loss=loss)
        finally:
            Fn.FfnChain.enabled = True
        _same(a, b)
