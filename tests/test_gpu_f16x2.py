"""The two-f16-term GEMM path (rr_linear_args.w_packed = 3, rr_wgrad_args.split = 2, RR_PLAN_F16X2_GEMM): every operand is
scaled by a power of two from a bound of its tensor's largest magnitude and written as two f16 terms (22 significant bits),
three v_mfma_f32_16x16x32_f16 per k-step instead of six bf16 ones.  Checked here against f64 on operands of very different
magnitudes (the scaling must keep the terms inside f16's 5 exponent bits), for every kernel geometry and operand form, plus
the magnitude outputs that spare the next GEMM a pass over its operand (c_amax_out, dz_amax_out, rr_gather_epi.amax_out).
Model-level parity of this path against the oracle: tests/conftest.py runs the model / trainer / headline modules in both modes."""
import ctypes as C

import pytest
import torch

from reactranker_amd import _lib
from reactranker_amd import functions as Fn
from reactranker_amd._lib import PackDesc, check, lib, ptr, stream

pytestmark = pytest.mark.gpu
dev = "cuda"


def _pack(w, transpose, rows, c0, k1, k2, kind):
    dst = torch.empty(int(lib().rr_split_weight_bytes(rows, k1, k2)), dtype=torch.uint8, device=dev)
    d = (PackDesc * 1)()
    d[0].src, d[0].ld_src, d[0].transpose, d[0].rows, d[0].c0, d[0].k1, d[0].k2 = w.data_ptr(), w.stride(0), transpose, rows, c0, k1, k2
    d[0].dst, d[0].split = dst.data_ptr(), kind
    check(lib().rr_pack_weights_f32(d, 1, stream()), "pack")
    if kind == 2:
        dst._rr_f16 = True
    return dst


def _rel(out, ref, den):
    return float(((out.double() - ref).abs() / den).max())


@pytest.mark.parametrize("rows,cols,ld", [(1, 4, 4), (1000, 300, 300), (777, 83, 84), (513, 61, 64), (300, 7, 9), (70001, 300, 304)])
def test_amax_matches_torch(rows, cols, ld):
    torch.manual_seed(rows)
    x = torch.randn(rows, ld, device=dev) * torch.exp(3 * torch.randn(rows, 1, device=dev))
    x[:, cols:] = 1e30                                   # padding columns must not count
    x[rows // 2, cols // 2] = float("nan")               # NaNs are skipped
    out = torch.zeros(_lib.RR_AMAX_FLOATS, device=dev)   # a magnitude slot: 16 lanes on separate cache lines, the bound is their maximum
    check(lib().rr_amax_f32(ptr(x), rows, cols, ld, ptr(out), stream()), "amax")
    ref = torch.nan_to_num(x[:, :cols], nan=0.0).abs().max()
    assert float(out.max()) == float(ref)
    lanes = out.view(_lib.RR_AMAX_LANES, _lib.RR_AMAX_STRIDE)
    assert float(lanes[:, 1:].abs().max()) == 0.0        # only the lanes' first floats are written
    out[0] = float(ref) * 2                              # a maximum INTO the slot
    check(lib().rr_amax_f32(ptr(x), rows, cols, ld, ptr(out), stream()), "amax")
    assert float(out.max()) == float(ref) * 2


# (M, N, K1, K2): <4,4,8>, <10,10,8>, <19,5,8> (few rows), <19,19,12> one block per workgroup, persistent, <38,19,12>
SHAPES = [(999, 32, 32, 0), (5000, 128, 128, 40), (4133, 300, 300, 0), (20011, 300, 61, 300), (60000, 300, 300, 0), (9001, 600, 600, 83)]


@pytest.mark.parametrize("M,N,K1,K2", SHAPES)
@pytest.mark.parametrize("xscale,wscale", [(1.0, 1.0), (1e-7, 30.0), (3e4, 1e-3)])
def test_linear_f16x2_against_f64(M, N, K1, K2, xscale, wscale, parity_log):
    torch.manual_seed(M + N)
    K = K1 + K2
    W = torch.randn(N, K, device=dev) / K ** 0.5 * wscale
    b = torch.randn(N, device=dev) * xscale * wscale
    # rows of very different magnitude inside one tensor (gradient-like): the scale follows the largest
    x1 = torch.randn(M, (K1 + 3) // 4 * 4, device=dev) * torch.exp(2 * torch.randn(M, 1, device=dev)) * xscale
    x2 = torch.randn(M, (K2 + 3) // 4 * 4, device=dev) * xscale * 0.3 if K2 else None
    wh, wb = _pack(W, 0, N, 0, K1, K2, 2), _pack(W, 0, N, 0, K1, K2, 1)
    cam = torch.zeros(1, device=dev)
    o3 = torch.empty(M, N, device=dev)
    A = Fn.linear                                        # (the wrapper computes the operand bounds with rr_amax_f32)
    kw = dict(a1=x1, k1=K1, bias=b)
    if K2:
        kw.update(a2=x2, k2=K2)
    o2 = A(M, N, wb, **kw)
    A(M, N, wh, out=o3, **kw)
    X = torch.cat([x1[:, :K1], x2[:, :K2]], 1) if K2 else x1[:, :K1]
    ref = X.double() @ W.double().t() + b.double()
    den = X.double().abs() @ W.double().abs().t() + b.double().abs() + 1e-300
    # the two-term form's contract: 22 bits of every element within 2^-18 of its TENSOR's largest, an absolute error below
    # 2^-40 of that largest otherwise - rows far below the largest row keep fewer bits (the three-term bf16 form keeps all)
    floor = 2.0 ** -36 * float(X.abs().max()) * W.double().abs().sum(1)[None, :]
    big = (X.abs().amax(1) >= 2.0 ** -14 * X.abs().max())
    e2, e3 = _rel(o2, ref, den), _rel(o3, ref, den + 1e6 * floor)
    e3_big = _rel(o3[big], ref[big], den[big])
    parity_log(f"M {M} N {N} K {K1}+{K2} scales {xscale:g}/{wscale:g}: max err / sum|x||w|  bf16x3 {e2:.2e}  f16x2 {e3_big:.2e} "
               f"(rows within 2^-14 of the largest: {int(big.sum())} of {M}); all rows, against sum|x||w| + 2^-16 max|x| sum|w|: {e3:.2e}")
    assert e3_big <= 1e-6 and e3 <= 1e-6, (e2, e3_big, e3)


def test_linear_f16x2_operand_forms_and_magnitude_outputs(parity_log):
    """gathered operand minus gathered subtrahend with residual / ReLU / dropout; the masked dX form with its dZ side output,
    sign-bit mask and column sums; c_amax_out / dz_amax_out equal the stored tensors' largest magnitudes"""
    torch.manual_seed(3)
    H, M = 300, 30011
    nA = M // 2 + 3
    W = torch.randn(H, H, device=dev) / 17
    b = torch.randn(H, device=dev)
    am = torch.randn(nA, H, device=dev) * 40
    msg = torch.relu(torch.randn(M, H, device=dev)) * 40
    inp = torch.randn(M, H, device=dev)
    b2a = torch.randint(0, nA, (M,), device=dev, dtype=torch.int32)
    rev = torch.randint(0, M, (M,), device=dev, dtype=torch.int32)
    b2a[0] = -1
    rev[0] = -1
    wh = _pack(W, 0, H, 0, H, 0, 2)
    A = _lib.LinearArgs()
    out = torch.empty(M, H, device=dev)
    bits = torch.empty(M, int(lib().rr_mask_bits_row_bytes(H)), dtype=torch.uint8, device=dev)
    slots = torch.zeros(4, _lib.RR_AMAX_FLOATS, device=dev)
    a_am, a_msg = Fn.amax(am), Fn.amax(msg)
    A.M, A.N = M, H
    A.a1, A.lda1, A.k1, A.a1_idx = ptr(am), H, H, ptr(b2a)
    A.a1_sub, A.lda1_sub, A.a1_sub_idx = ptr(msg), H, ptr(rev)
    A.mask_scale = 1.0
    A.w, A.ldw, A.w_packed = ptr(wh), 0, 3
    A.bias, A.residual, A.ldr, A.act = ptr(b), ptr(inp), H, Fn.ACT_RELU
    A.drop_p, A.drop_seed = 0.1, 77
    A.c, A.ldc = ptr(out), H
    A.mask_bits_out = ptr(bits)
    A.a1_amax, A.a1_sub_amax = ptr(a_am), ptr(a_msg)
    A.c_amax_out = ptr(slots[0])
    check(lib().rr_linear_f32(C.byref(A), stream()), "linear")
    z = torch.zeros(1, device=dev)
    X = torch.where(b2a[:, None] >= 0, am[b2a.clamp(min=0).long()], z) - torch.where(rev[:, None] >= 0, msg[rev.clamp(min=0).long()], z)
    pre = X.double() @ W.double().t() + b.double() + inp.double()
    den = X.double().abs() @ W.double().abs().t() + 1
    ref = torch.where(out != 0, torch.relu(pre) / 0.9, torch.zeros_like(pre))
    e = _rel(out, ref, den)
    kept = float((out != 0).double().mean())
    assert e <= 1e-6 and 0.35 < kept < 0.55, (e, kept)
    assert float(slots[0].max()) == float(out.abs().max())
    # masked dX: dz = dy * (out > 0) / 0.9 from the sign bits, dX = dz W, dz stored, weighted column sums
    dy = torch.randn(M, H, device=dev) * 1e-6
    wt = _pack(W, 1, H, 0, H, 0, 2)
    dx = torch.empty(M, H, device=dev)
    dz = torch.empty(M, H, device=dev)
    cw = torch.rand(M, device=dev)
    part = torch.empty(int(lib().rr_linear_colsum_rows(M)), H, device=dev)
    a_dy = Fn.amax(dy)
    B = _lib.LinearArgs()
    B.M, B.N = M, H
    B.a1, B.lda1, B.k1 = ptr(dy), H, H
    B.a_mask, B.ld_mask, B.mask_scale, B.a_mask_bits = ptr(out), H, 1.0 / 0.9, ptr(bits)
    B.dz_out, B.ld_dz = ptr(dz), H
    B.w, B.ldw, B.w_packed = ptr(wt), 0, 3
    B.c, B.ldc = ptr(dx), H
    B.colsum_w, B.colsum_partial, B.ld_partial = ptr(cw), ptr(part), H
    B.a1_amax = ptr(a_dy)
    B.c_amax_out, B.dz_amax_out = ptr(slots[1]), ptr(slots[2])
    check(lib().rr_linear_f32(C.byref(B), stream()), "linear dX")
    dzr = torch.where(out > 0, dy * (1.0 / 0.9), torch.zeros_like(dy))
    assert torch.equal(dz, dzr)
    ref = dzr.double() @ W.double()
    den = dzr.double().abs() @ W.double().abs() + 1e-300
    e2 = _rel(dx, ref, den)
    cs = (dx.double() * cw.double()[:, None]).sum(0)
    ecs = float((part.double().sum(0) - cs).abs().max() / cs.abs().max())
    parity_log(f"gathered forward max err {e:.2e}; masked dX (|dy| ~ 1e-6) max err {e2:.2e}; column sums {ecs:.2e}")
    assert e2 <= 1e-6 and ecs <= 1e-5
    assert float(slots[1].max()) == float(dx.abs().max()) and float(slots[2].max()) == float(dz.abs().max())


@pytest.mark.parametrize("M", [9000, 70001])
@pytest.mark.parametrize("zscale", [1.0, 1e-8])
def test_wgrad_f16x2_against_f64(M, zscale, parity_log):
    torch.manual_seed(M)
    H = 300
    nA = M // 2 + 5
    am = torch.randn(nA, H, device=dev)
    msg = torch.relu(torch.randn(M, H, device=dev))
    dz = torch.randn(M, H, device=dev) * torch.exp(2 * torch.randn(M, 1, device=dev)) * zscale
    b2a = torch.randint(0, nA, (M,), device=dev, dtype=torch.int32)
    rev = torch.randint(0, M, (M,), device=dev, dtype=torch.int32)
    z = torch.zeros(1, device=dev)
    X = am[b2a.long()] - msg[rev.long()]
    cases = [("W_h gather-sub", dict(x1=am, k1=H, x1_idx=b2a, x1_sub=msg, x1_sub_idx=rev), X, dz, None, 1.0)]
    fa = torch.zeros(M, 64, device=dev)
    fa[:, :61] = (torch.rand(M, 61, device=dev) < 0.1).float()
    a2 = torch.randn(M, H, device=dev) * 100
    y = torch.relu(torch.randn(M, H, device=dev))
    cases.append(("W_o 61|300 masked", dict(x1=fa, k1=61, x2=a2, k2=H, mask=y, mask_scale=1.25), torch.cat([fa[:, :61], a2], 1),
                  torch.where(y > 0, dz * 1.25, torch.zeros_like(dz)), y, 1.25))
    for name, kw, Xr, dzr, _, _ in cases:
        K = Xr.shape[1]
        outs = []
        for f16 in (False, True):
            dw = torch.zeros(H, K, device=dev)
            db = torch.zeros(H, device=dev)
            bounds = [Fn.amax(dz), Fn.amax(kw["x1"], kw["k1"]), Fn.amax(kw["x1_sub"]) if "x1_sub" in kw else None,
                      Fn.amax(kw["x2"]) if "x2" in kw else None] if f16 else None
            Fn.wgrad(M, H, dz, dw, dbias=db, amax_of=bounds, **kw)
            outs.append((dw, db))
        ref = dzr.double().t() @ Xr.double()
        den = dzr.double().abs().t() @ Xr.double().abs() + 1e-300
        rb = dzr.double().sum(0)
        dbn = dzr.double().abs().sum(0) + 1e-300
        e = [(_rel(w, ref, den), _rel(b_, rb, dbn)) for w, b_ in outs]
        parity_log(f"M {M} {name} |dz| x {zscale:g}: dW max err bf16x3 {e[0][0]:.2e} f16x2 {e[1][0]:.2e}; dbias {e[0][1]:.2e} / {e[1][1]:.2e}")
        assert e[1][0] <= 2e-6 and e[1][1] <= 2e-6, e


def test_gather_magnitude_outputs():
    torch.manual_seed(5)
    H, nB, nA, K = 300, 20000, 9000, 4
    src = torch.randn(nB, H, device=dev)
    idx = torch.randint(-1, nB, (nA, K), device=dev, dtype=torch.int32)
    out = torch.empty(nA, H, device=dev)
    slot = torch.zeros(_lib.RR_AMAX_FLOATS, device=dev)
    check(lib().rr_gather_sum_amax_f32(ptr(src), nB, H, ptr(idx), nA, K, H, None, 0, 0, ptr(out), H, ptr(slot), stream()), "gather")
    ref = torch.where(idx[:, :, None] >= 0, src[idx.clamp(min=0).long()], torch.zeros(1, device=dev)).sum(1)
    assert torch.allclose(out, ref, atol=1e-5) and float(slot.max()) == float(out.abs().max())
    # the fused-epilogue gather: mask, scale, addends, padding-row partials
    y = torch.randn(nA, H, device=dev)
    add = torch.randn(nA, H, device=dev) * 3
    part = torch.randn(7, H, device=dev)
    e = _lib.GatherEpi()
    e.mask, e.ld_mask, e.mask_scale, e.n_adds, e.ld_add = ptr(y), H, 1.5, 1, H
    e.adds[0] = add.data_ptr()
    slot2 = torch.zeros(_lib.RR_AMAX_FLOATS, device=dev)
    e.amax_out = ptr(slot2)
    out2 = torch.empty(nA, H, device=dev)
    check(lib().rr_gather_sum_epi_f32(ptr(src), nB, H, ptr(idx), nA, K, H, ptr(part), 7, H, C.byref(e), ptr(out2), H, stream()), "gather epi")
    assert float(slot2.max()) == float(out2.abs().max())


# every plan shape of tests/test_gpu_plan.py (reactant modes, head layouts, depths 1..16, H = 32 .. 600) through the step plans
# in BOTH arithmetics: the two-term form must agree with the three-term form (which test_gpu_plan.py holds bit-identical to
# the per-op mirror and the other modules hold against the oracle) within the parity tolerances, and be run-to-run identical
from tests.test_gpu_plan import CASES as PLAN_CASES  # noqa: E402


@pytest.mark.parametrize("H,d,dd,fd,bias,tn,last,tt,F,p,train", PLAN_CASES)
def test_plans_agree_between_the_two_arithmetics(H, d, dd, fd, bias, tn, last, tt, F, p, train, parity_log):
    from oracle import ref_cpu as O
    from reactranker_amd import featurization, synth
    from tests.test_gpu_model import make_model
    from tests.test_gpu_plan import _run, _same
    cfg = dict(hidden_size=H, mpnn_depth=d, mpnn_diff_depth=dd, ffn_depth=fd, use_bias=bias, task_num=tn, ffn_last_layer=last,
               task_type=tt, add_features_dim=F)
    w = synth.seeded_weights(O.model_shapes(H, d, dd, fd, tn, F, bias), 5)
    model = make_model(cfg, w, dropout=p)
    model = model.train() if train else model.eval()
    qb = synth.make_queries(17, 4, [7, 3, 9, 5], atoms_lo=5, atoms_hi=14)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    loss = "mle" if tn == 1 else "lin"
    old = Fn.SplitGemm.f16
    try:
        Fn.SplitGemm.f16 = True
        a = _run(model, rb, pb, qb, 4242, plan=True, loss=loss)
        _same(a, _run(model, rb, pb, qb, 4242, plan=True, loss=loss))      # atomic maxima are exact: run-to-run bit-identical
        Fn.SplitGemm.f16 = False
        b = _run(model, rb, pb, qb, 4242, plan=True, loss=loss)
    finally:
        Fn.SplitGemm.f16 = old
    es = float(((a[0] - b[0]).abs() / (1 + b[0].abs())).max())
    el = float(((a[1] - b[1]).abs() / (1 + b[1].abs())).max())
    # gradients: within 5e-5 of the tensor's largest entry - except tensors whose gradient is analytically zero (hazard H5: the
    # output bias and always-active last-hidden biases under a shift-invariant ranking loss), which are rounding noise in
    # either arithmetic and only have to BE noise: below 2e-5 of the model's largest gradient entry in both
    gmax = max(float(v.abs().max()) for v in b[2].values())
    eg, noise, worst = 0.0, [], ""
    for k in b[2]:
        ta, tb = float(a[2][k].abs().max()), float(b[2][k].abs().max())
        if max(ta, tb) <= 2e-5 * gmax:
            noise.append(k)
            continue
        r = float((a[2][k] - b[2][k]).abs().max()) / tb
        if r > eg:
            eg, worst = r, f"{k} (max |g| {tb:.2e} of the model's {gmax:.2e})"
    parity_log(f"H {H} depth {d}/{dd} train {train}: two-term vs three-term plans - scores {es:.2e}, loss {el:.2e}, worst gradient / its "
               f"tensor's max {eg:.2e} at {worst}; analytically-zero gradients (noise in both, < 2e-5 of the model's largest): {noise}")
    assert es <= 1e-5 and el <= 1e-5 and eg <= 5e-5


def test_magnitude_outputs_of_the_dropout_gather_and_the_readout_adjoint():
    """rr_gather_dropout_amax_f32 / rr_segment_mean_bwd_masked_amax_f32: same results as the entry points without the slot,
    and the slot holds the stored tensor's largest magnitude"""
    torch.manual_seed(9)
    H, nU, nB, p = 300, 3000, 40000, 0.1
    z = torch.relu(torch.randn(nU, H, device=dev)) * 7
    bmap = torch.randint(0, nU, (nB,), device=dev, dtype=torch.int32)
    o1 = torch.empty(nB, H, device=dev)
    o2 = torch.empty(nB, H, device=dev)
    slot = torch.zeros(_lib.RR_AMAX_FLOATS, device=dev)
    check(lib().rr_gather_dropout_f32(ptr(z), nU, H, ptr(bmap), nB, H, p, 99, ptr(o1), H, stream()), "gather_dropout")
    check(lib().rr_gather_dropout_amax_f32(ptr(z), nU, H, ptr(bmap), nB, H, p, 99, ptr(o2), H, ptr(slot), stream()), "gather_dropout_amax")
    assert torch.equal(o1, o2) and float(slot.max()) == float(o2.abs().max())
    # readout adjoint with the mask of the layer below
    M, F = 500, 1
    sizes = torch.randint(5, 25, (M,))
    starts = torch.cumsum(sizes, 0) - sizes + 1                      # atom 0 is the padding row
    nA = int(sizes.sum()) + 1
    a_scope = torch.stack([starts, sizes], 1).to(torch.int32).to(dev).contiguous()
    atom2mol = torch.full((nA,), -1, dtype=torch.int32)
    for mi in range(M):
        atom2mol[int(starts[mi]):int(starts[mi] + sizes[mi])] = mi
    atom2mol = atom2mol.to(dev)
    dvecs = torch.randn(M, 304, device=dev) * 1e-4
    hid = torch.randn(nA, H, device=dev)
    d1 = torch.empty(nA, H, device=dev)
    d2 = torch.empty(nA, H, device=dev)
    slot2 = torch.zeros(_lib.RR_AMAX_FLOATS, device=dev)
    args = (ptr(dvecs), 304, ptr(a_scope), ptr(atom2mol), nA, H, F, p, 1234, ptr(hid), H, None, 1.0 / (1 - p))
    check(lib().rr_segment_mean_bwd_masked_f32(*args, ptr(d1), H, stream()), "segment_mean_bwd_masked")
    check(lib().rr_segment_mean_bwd_masked_amax_f32(*args, ptr(d2), H, ptr(slot2), stream()), "segment_mean_bwd_masked_amax")
    assert torch.equal(d1, d2) and float(d2.abs().max()) > 0 and float(slot2.max()) == float(d2.abs().max())


def test_a_querys_scores_in_two_different_batches(parity_log):
    """DESIGN.md section 2, H6 ii: with three bf16 terms a query's eval-mode scores are the same bits whatever else is in the
    batch (every GEMM row is computed on its own); with two f16 terms the operand scale follows the batch's largest magnitude,
    which reaches elements more than 2^-18 below it - measured here: how far the scores of the same 12 candidates move
    between a batch of 3 queries and a batch of 40 (H = 300, eval mode, step plans)."""
    from oracle import ref_cpu as O
    from reactranker_amd import featurization, synth
    from tests.test_gpu_model import make_model
    cfg = dict(hidden_size=300, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, task_num=1,
               ffn_last_layer="with_softplus", task_type=None, add_features_dim=1)
    model = make_model(cfg, synth.seeded_weights(O.model_shapes(300, 3, 3, 3, 1, 1, True), 11)).eval()
    q1 = synth.make_queries(5, 1, [12], atoms_lo=10, atoms_hi=24)
    small = synth.make_queries(6, 2, [7, 9], atoms_lo=10, atoms_hi=24)
    big = synth.make_queries(7, 39, 32, atoms_lo=10, atoms_hi=24)

    def scores(other):
        r = q1.r_specs + other.r_specs
        p = q1.p_specs + other.p_specs
        add = torch.cat([torch.tensor(q1.add_features), torch.tensor(other.add_features)]).numpy()
        rb, pb = featurization.BatchMolGraph(r, K=4), featurization.BatchMolGraph(p, K=4)
        with torch.no_grad():
            return model(rb, pb, gpu=0, add_features=add)[:12].clone()
    old = Fn.SplitGemm.f16
    try:
        Fn.SplitGemm.f16 = False
        a3, b3 = scores(small), scores(big)
        Fn.SplitGemm.f16 = True
        a2, b2 = scores(small), scores(big)
    finally:
        Fn.SplitGemm.f16 = old
    d3 = float((a3 - b3).abs().max())
    d2 = float(((a2 - b2).abs() / (1 + a2.abs())).max())
    parity_log(f"the same 12 candidates in a batch of 3 and of 40 queries: three bf16 terms max |score difference| {d3:.1e}; "
               f"two f16 terms {d2:.2e} of (1 + |score|); identical candidate order {bool(torch.equal(a2.argsort(), b2.argsort()))}")
    assert d3 == 0.0
    assert d2 <= 1e-6 and torch.equal(a2.argsort(), b2.argsort())


@pytest.mark.parametrize("p,train", [(0.1, True), (0.0, False)])
def test_two_term_plans_do_not_depend_on_the_stream_layout(p, train):
    """the magnitude slots are filled on whichever stream produces or first consumes a tensor and read on others: with the
    side / aux streams off, and with the reactant encoder's backward forked onto the aux stream (RR_PLAN_AUX_BACKWARD: both
    passes read bounds found before the fork), scores, loss and every gradient must be the same bits as by default"""
    from oracle import ref_cpu as O
    from reactranker_amd import featurization, synth
    from tests.test_gpu_model import make_model
    from tests.test_gpu_plan import _run, _same
    cfg = dict(hidden_size=64, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, task_num=1,
               ffn_last_layer="with_softplus", task_type=None, add_features_dim=1)
    model = make_model(cfg, synth.seeded_weights(O.model_shapes(64, 3, 3, 3, 1, 1, True), 5), dropout=p)
    model = model.train() if train else model.eval()
    qb = synth.make_queries(23, 5, [7, 3, 9, 5, 6], atoms_lo=5, atoms_hi=14)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    old = (Fn.SplitGemm.f16, Fn.SideStream.enabled, Fn.AuxStream.enabled, Fn.AuxStream.backward)
    try:
        Fn.SplitGemm.f16 = True
        a = _run(model, rb, pb, qb, 99, plan=True)
        Fn.SideStream.enabled = Fn.AuxStream.enabled = False
        _same(a, _run(model, rb, pb, qb, 99, plan=True))
        Fn.SideStream.enabled = Fn.AuxStream.enabled = True
        Fn.AuxStream.backward = True
        _same(a, _run(model, rb, pb, qb, 99, plan=True))
    finally:
        Fn.SplitGemm.f16, Fn.SideStream.enabled, Fn.AuxStream.enabled, Fn.AuxStream.backward = old


def test_linear_f16x2_random_shapes_against_f64(parity_log):
    """a seeded sweep over output widths 4 .. 608, one or two operand segments of odd widths, row counts 1 .. 30,000, with and
    without gather / subtract / bias / residual / ReLU: every kernel geometry (<4,4,8>, <10,10,8>, <19,5,8>, <19,19,12> one-block
    and persistent, <38,19,12>), partial last k-steps and column tiles, the weight image's trailer - against f64"""
    import numpy as np
    rng = np.random.default_rng(20260)
    worst, n_cases = 0.0, 0
    for case in range(48):
        N = int(rng.choice([4, 8, 32, 60, 64, 68, 128, 160, 164, 300, 304, 308, 600, 608]))
        K1 = int(rng.integers(1, 420))
        K2 = int(rng.choice([0, 0, int(rng.integers(1, 200))]))
        M = int(rng.choice([1, 15, 16, 17, 191, 193, 4000, 8192, 8193, int(rng.integers(9000, 30000))]))
        gather = bool(rng.integers(0, 2)) and M > 16
        sub = gather and bool(rng.integers(0, 2))
        relu, resid, bias = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        xs, ws = float(10.0 ** rng.integers(-6, 4)), float(10.0 ** rng.integers(-3, 3))
        torch.manual_seed(case)
        K = K1 + K2
        W = torch.randn(N, K, device=dev) / K ** 0.5 * ws
        nsrc = M // 2 + 3 if gather else M
        x1 = torch.randn(nsrc, (K1 + 3) // 4 * 4, device=dev) * xs
        kw = dict(a1=x1, k1=K1)
        A1 = x1[:, :K1]
        if gather:
            idx = torch.randint(0, nsrc, (M,), device=dev, dtype=torch.int32)
            idx[0] = -1
            kw["a1_idx"] = idx
            A1 = torch.where(idx[:, None] >= 0, x1[idx.clamp(min=0).long(), :K1], torch.zeros(1, device=dev))
            if sub:
                xsub = torch.randn(M, (K1 + 3) // 4 * 4, device=dev) * xs
                sidx = torch.randint(0, M, (M,), device=dev, dtype=torch.int32)
                kw.update(a1_sub=xsub, a1_sub_idx=sidx)
                A1 = A1 - xsub[sidx.long(), :K1]
        X = A1
        if K2:
            x2 = torch.randn(M, (K2 + 3) // 4 * 4, device=dev) * xs * 0.5
            kw.update(a2=x2, k2=K2)
            X = torch.cat([A1, x2[:, :K2]], 1)
        b = torch.randn(N, device=dev) * xs * ws if bias else None
        res = torch.randn(M, N, device=dev) * xs * ws if resid else None
        if bias:
            kw["bias"] = b
        if resid:
            kw["residual"] = res
        if relu:
            kw["act"] = Fn.ACT_RELU
        wh = _pack(W, 0, N, 0, K1, K2, 2)
        out = Fn.linear(M, N, wh, **kw)
        ref = X.double() @ W.double().t()
        den = X.double().abs() @ W.double().abs().t() + 2.0 ** -36 * float(X.abs().max()) * W.double().abs().sum(1)[None, :] * 1e6
        if bias:
            ref = ref + b.double()
            den = den + b.double().abs()
        if resid:
            ref = ref + res.double()
            den = den + res.double().abs()
        if relu:
            ref = torch.relu(ref)
        e = _rel(out, ref, den + 1e-300)
        worst = max(worst, e)
        n_cases += 1
        assert e <= 2e-6, (case, M, N, K1, K2, gather, sub, relu, resid, bias, xs, ws, e)
    parity_log(f"{n_cases} random shapes (N 4..608, K1 1..419, K2 0..199, M 1..30,000, gather / subtract / bias / residual / ReLU): worst max err {worst:.2e}")


def test_wgrad_f16x2_random_shapes_against_f64(parity_log):
    """the same sweep for the weight gradient (rr_wgrad_args.split = 2): N 4..600, one or two X segments of odd widths (all three
    k-block widths), 1 .. 40,000 rows, gathered / subtracted X, masked dZ; dW and dbias against f64"""
    import numpy as np
    rng = np.random.default_rng(777)
    old_min, old_f16 = Fn.SPLIT_MIN_ROWS, Fn.SplitGemm.f16
    Fn.SPLIT_MIN_ROWS, Fn.SplitGemm.f16 = 1, True
    worst = 0.0
    try:
        for case in range(32):
            N = int(rng.choice([4, 32, 64, 160, 164, 300, 600]))
            K1 = int(rng.integers(1, 400))
            K2 = int(rng.choice([0, 0, int(rng.integers(1, 160))]))
            M = int(rng.choice([1, 31, 33, 500, 8192, int(rng.integers(9000, 40000))]))
            gather = bool(rng.integers(0, 2)) and M > 40
            sub = gather and bool(rng.integers(0, 2))
            mask = bool(rng.integers(0, 2))
            zs, xs = float(10.0 ** rng.integers(-7, 2)), float(10.0 ** rng.integers(-3, 3))
            torch.manual_seed(1000 + case)
            dz = torch.randn(M, N, device=dev) * zs
            nsrc = M // 2 + 3 if gather else M
            x1 = torch.randn(nsrc, (K1 + 3) // 4 * 4, device=dev) * xs
            kw = dict(x1=x1, k1=K1)
            X1 = x1[:, :K1]
            if gather:
                idx = torch.randint(0, nsrc, (M,), device=dev, dtype=torch.int32)
                kw["x1_idx"] = idx
                X1 = x1[idx.long(), :K1]
                if sub:
                    xsub = torch.randn(M, (K1 + 3) // 4 * 4, device=dev) * xs
                    sidx = torch.randint(0, M, (M,), device=dev, dtype=torch.int32)
                    kw.update(x1_sub=xsub, x1_sub_idx=sidx)
                    X1 = X1 - xsub[sidx.long(), :K1]
            X = X1
            if K2:
                x2 = torch.randn(M, (K2 + 3) // 4 * 4, device=dev) * xs * 3
                kw.update(x2=x2, k2=K2)
                X = torch.cat([X1, x2[:, :K2]], 1)
            dzr = dz
            if mask:
                y = torch.randn(M, N, device=dev)
                kw.update(mask=y, mask_scale=1.25)
                dzr = torch.where(y > 0, dz * 1.25, torch.zeros_like(dz))
            dw = torch.zeros(N, K1 + K2, device=dev)
            db = torch.zeros(N, device=dev)
            Fn.wgrad(M, N, dz, dw, dbias=db, **kw)
            ref = dzr.double().t() @ X.double()
            # (the contract's absolute floor: 2^-40 of each operand tensor's largest magnitude per product, M products per entry)
            floor = 2.0 ** -36 * M * float(dzr.abs().max()) * max(float(X.abs().max()), 1.0)
            den = dzr.double().abs().t() @ X.double().abs() + floor + 1e-300
            rb = dzr.double().sum(0)
            dbn = dzr.double().abs().sum(0) + floor + 1e-300
            e = max(_rel(dw, ref, den), _rel(db, rb, dbn))
            worst = max(worst, e)
            assert e <= 3e-6, (case, M, N, K1, K2, gather, sub, mask, zs, xs, e)
    finally:
        Fn.SPLIT_MIN_ROWS, Fn.SplitGemm.f16 = old_min, old_f16
    parity_log(f"32 random weight-gradient shapes (N 4..600, K1 1..399, K2 0..159, M 1..40,000, gather / subtract / mask): worst max err {worst:.2e}")


def test_shared_prefix_input_gradient_bound_follows_its_last_write(parity_log):
    """Round-4 advice (high): in the shared-prefix backward d_inp_u is written by a gather and then accumulated IN PLACE twice
    (+ dz1_u, + relu'(msg0) * d_msg0_u); a magnitude slot claimed by the gather would bound the first summand only and the
    W_i weight gradient's f16 split would overflow (inf / NaN) as soon as the later summands exceed it by the 2x-4x headroom.
    Here W_h is scaled up so that they exceed it by far more: >= 8,192 distinct-reactant bonds (the weight gradient then runs
    on the split path), depth 4 (two per-copy layers: the gather path exists), train mode.  The two-term plan must stay
    finite and agree with the three-term plan (which the other modules hold against the oracle)."""
    from oracle import ref_cpu as O
    from reactranker_amd import featurization, synth
    from tests.test_gpu_model import make_model
    from tests.test_gpu_plan import _run
    H, d = 32, 4
    cfg = dict(hidden_size=H, mpnn_depth=d, mpnn_diff_depth=2, ffn_depth=2, use_bias=True, task_num=1, ffn_last_layer="no_softplus",
               task_type=None, add_features_dim=1)
    w = synth.seeded_weights(O.model_shapes(H, d, 2, 2, 1, 1, True), 5)
    w["encoder.W_h.weight"] = w["encoder.W_h.weight"] * 6.0       # every step back through W_h grows the gradient
    model = make_model(cfg, w, dropout=0.1).train()
    nq = 300
    qb = synth.make_queries(23, nq, 2, atoms_lo=14, atoms_hi=24)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    ub, _, _ = rb.unique()
    assert ub.n_bonds >= 8192, ub.n_bonds                          # the distinct reactants' W_i gradient takes the split path
    old = Fn.SplitGemm.f16
    try:
        Fn.SplitGemm.f16 = True
        a = _run(model, rb, pb, qb, 99, plan=True)
        Fn.SplitGemm.f16 = False
        b = _run(model, rb, pb, qb, 99, plan=True)
    finally:
        Fn.SplitGemm.f16 = old
    worst = 0.0
    gmax = max(float(v.abs().max()) for v in b[2].values())
    for k in b[2]:
        assert torch.isfinite(a[2][k]).all(), k
        tb = float(b[2][k].abs().max())
        if tb > 2e-5 * gmax:                                       # (analytically-zero gradients are noise in both, hazard H5)
            worst = max(worst, float((a[2][k] - b[2][k]).abs().max()) / tb)
    gi = a[2]["encoder.W_i.weight"]
    parity_log(f"shared prefix, W_h x 6, {ub.n_bonds} distinct bonds: max |dW_i| {float(gi.abs().max()):.3e}; two-term vs three-term "
               f"gradients, worst / tensor max {worst:.2e}")
    assert worst <= 5e-5
