"""The three-bf16-term GEMM path (rr_linear_args.w_packed = 2, rr_wgrad_args.split, rr_pack_desc.split).

Claim under test: the result is an f32 GEMM - every operand is represented EXACTLY by its three bf16 terms, and the six
products kept per multiply leave an error against an f64 GEMM that is not above the error of the exact-f32 MFMA chain
(v_mfma_f32_16x16x4_f32) on the same inputs.  Errors are measured relative to sum_k |a_k b_k| (the scale of a dot
product's rounding error); the north-star tolerance 1e-5 * (1 + |ref|) is asserted as well."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import helpers as Hh

from reactranker_amd import functions as Fn
from reactranker_amd import _lib
from reactranker_amd._lib import PackDesc, check, lib, ptr, stream

pytestmark = pytest.mark.gpu


def _pack_split(w, transpose, rows, c0, k1, k2):
    dst = torch.empty(int(lib().rr_split_weight_bytes(rows, k1, k2)), dtype=torch.uint8, device=w.device)
    d = (PackDesc * 1)()
    d[0].src, d[0].ld_src, d[0].transpose, d[0].rows, d[0].c0, d[0].k1, d[0].k2 = ptr(w), w.stride(0), transpose, rows, c0, k1, k2
    d[0].dst, d[0].split = ptr(dst), 1
    check(lib().rr_pack_weights_f32(d, 1, stream()), "rr_pack_weights_f32")
    return dst


def _terms_to_f64(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


@pytest.mark.parametrize("rows,k1,k2,transpose,c0", [(300, 300, 0, 0, 0), (300, 133, 300, 0, 0), (300, 147, 0, 0, 0),
                                                      (300, 300, 0, 1, 133), (64, 32, 0, 0, 0), (32, 300, 61, 0, 0),
                                                      (160, 40, 0, 1, 0), (600, 600, 0, 0, 0), (600, 133, 600, 1, 0)])
def test_pack_terms_are_exact_and_laid_out_as_documented(rows, k1, k2, transpose, c0):
    """t0 + t1 + t2 == w bit for bit (the sum is formed in f64, where it is exact), zeros in the padding, and the
    [k-step][16-column tile][term][lane][8] layout of include/reactranker_hip.h."""
    rng = np.random.default_rng(rows + k1 + 7 * k2)
    K = k1 + k2
    shape = (K + 3, c0 + rows + 5) if transpose else (rows, c0 + K + 5)      # L[r, c] = src[c, c0 + r] / src[r, c0 + c]
    w = (rng.standard_normal(shape) * np.exp(3 * rng.standard_normal(shape))).astype(np.float32)
    w.flat[::17] = 0.0
    w.flat[5::29] = 1e-30
    wd = torch.as_tensor(w).cuda()
    got = _pack_split(wd, transpose, rows, c0, k1, k2).cpu().numpy().view(np.uint16)
    nt = 4 if rows <= 64 else (10 if rows <= 160 else (19 if rows <= 304 else 38))
    t1 = (k1 + 31) // 32
    steps = t1 + (k2 + 31) // 32
    assert got.size == steps * nt * 3 * 512
    img = _terms_to_f64(got).reshape(steps, nt, 3, 64, 8)
    total = img.sum(2)                                               # [step, tile, lane, e]
    logical = np.zeros((rows, K), np.float64)
    for r in range(rows):
        logical[r] = (w[:K, c0 + r] if transpose else w[r, c0:c0 + K])
    want = np.zeros((steps, nt, 64, 8))
    for s in range(steps):
        for lane in range(64):
            for e in range(8):
                kk = (s if s < t1 else s - t1) * 32 + (lane >> 4) * 8 + e
                if s < t1:
                    col = kk if kk < k1 else -1
                else:
                    col = k1 + kk if kk < k2 else -1
                if col < 0:
                    continue
                n = np.arange(nt) * 16 + (lane & 15)
                ok = n < rows
                want[s, ok, lane, e] = logical[n[ok], col]
    assert np.array_equal(total, want)
    # each term is what the definition says: t0 = bf16(x) (round to nearest even), the remainders follow
    x = want.astype(np.float32)
    t0 = img[:, :, 0].astype(np.float32)
    r1 = (x - t0).astype(np.float32)
    assert np.all(np.abs(r1) <= np.abs(x) * 2.0 ** -8 + 1e-45)
    assert np.all(np.abs(img[:, :, 2]) <= np.abs(x) * 2.0 ** -16 + 1e-45)


def _err(out, ref, den):
    e = (out.double() - ref).abs() / den
    return float(e.max()), float(e.mean())


def _compare(o32, osp, ref, den, what, north_star=True):
    m32, a32 = _err(o32, ref, den)
    msp, asp = _err(osp, ref, den)
    Hh.record(what + " | split max err / sum|ab|", msp, 2e-6)
    Hh.record(what + " | f32-MFMA max err / sum|ab|", m32)
    assert asp <= 1.25 * a32 + 1e-12, f"{what}: mean error split {asp:.3e} vs f32 {a32:.3e}"
    assert msp <= 2.0 * m32 + 1e-12, f"{what}: max error split {msp:.3e} vs f32 {m32:.3e}"
    assert msp <= 2e-6, f"{what}: max error {msp:.3e} (relative to sum |a b|)"
    if north_star:      # (wide-range inputs cancel: there the exact-f32 chain itself is ~1e-4 off by this measure)
        tol = ((osp.double() - ref).abs() / (1 + ref.abs())).max().item()
        assert tol <= 1e-5, f"{what}: north-star tolerance {tol:.3e}"


@pytest.mark.parametrize("M", [1, 191, 192, 193, 5000])
@pytest.mark.parametrize("wide", [False, True])
def test_linear_split_error_is_not_above_the_f32_mfma_chain(M, wide):
    torch.manual_seed(M + wide)
    H = 300
    dev = "cuda"
    sc = (lambda *s: torch.exp(2.5 * torch.randn(*s, device=dev))) if wide else (lambda *s: torch.ones(*s, device=dev))
    W = torch.randn(H, H, device=dev) / 17 * sc(H, H)
    b = torch.randn(H, device=dev)
    z = torch.zeros(1, device=dev)
    # mode 2: dX = (dy * relu-mask) W with the dZ side output
    dy = torch.randn(M, H, device=dev) * sc(M, 1)
    y = torch.relu(torch.randn(M, H, device=dev))
    L = Fn.LinW(W, None)
    Fn.SplitGemm.enabled = False
    try:
        w32 = L.pk_t(0, H)
        w32f = Fn.LinW(W, b).pk(H)
    finally:
        Fn.SplitGemm.enabled = True
    wsp = L.pk_t(0, H)
    assert w32.dtype == torch.float32 and wsp.dtype == torch.uint8
    dz1, dz2 = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev)
    o32 = Fn.linear(M, H, w32, w_packed=True, a1=dy, k1=H, a_mask=y, mask_scale=1.1, dz_out=dz1)
    osp = Fn.linear(M, H, wsp, w_packed=True, a1=dy, k1=H, a_mask=y, mask_scale=1.1, dz_out=dz2)
    dzr = torch.where(y > 0, dy * 1.1, torch.zeros_like(dy))
    assert torch.equal(dz1, dzr) and torch.equal(dz2, dzr)
    _compare(o32, osp, dzr.double() @ W.double(), dzr.double().abs() @ W.double().abs() + 1e-300, "masked dX", not wide)
    # mode 1: gathered operand minus gathered operand, bias, residual, ReLU (no dropout: same zero pattern is not the point)
    nA = M // 2 + 3
    am, msg, inp = torch.randn(nA, H, device=dev) * sc(nA, 1), torch.relu(torch.randn(M, H, device=dev)), torch.randn(M, H, device=dev)
    b2a = torch.randint(-1, nA, (M,), device=dev, dtype=torch.int32)
    rev = torch.randint(-1, M, (M,), device=dev, dtype=torch.int32)
    kw = dict(a1=am, k1=H, a1_idx=b2a, a1_sub=msg, a1_sub_idx=rev, bias=b, residual=inp)
    wspf = Fn.LinW(W, b).pk(H)
    o32 = Fn.linear(M, H, w32f, w_packed=True, **kw)
    osp = Fn.linear(M, H, wspf, w_packed=True, **kw)
    A = torch.where(b2a[:, None] >= 0, am[b2a.clamp(min=0).long()], z) - torch.where(rev[:, None] >= 0, msg[rev.clamp(min=0).long()], z)
    ref = A.double() @ W.double().t() + b.double() + inp.double()
    den = A.double().abs() @ W.double().abs().t() + b.double().abs() + inp.double().abs() + 1e-300
    _compare(o32, osp, ref, den, "gathered forward", not wide)


@pytest.mark.parametrize("M", [8192, 20011])
def test_wgrad_split_error_is_not_above_the_f32_mfma_chain(M):
    torch.manual_seed(M)
    H, dev = 300, "cuda"
    z = torch.zeros(1, device=dev)
    nA = M // 2 + 5
    am, msg = torch.randn(nA, H, device=dev), torch.relu(torch.randn(M, H, device=dev))
    dz = torch.randn(M, H, device=dev) * torch.exp(2 * torch.randn(M, 1, device=dev))
    b2a = torch.randint(-1, nA, (M,), device=dev, dtype=torch.int32)
    rev = torch.randint(-1, M, (M,), device=dev, dtype=torch.int32)
    X = torch.where(b2a[:, None] >= 0, am[b2a.clamp(min=0).long()], z) - torch.where(rev[:, None] >= 0, msg[rev.clamp(min=0).long()], z)
    fa = torch.zeros(M, 136, device=dev)
    fa[:, :133] = (torch.rand(M, 133, device=dev) < 0.1).float()
    a2 = torch.randn(M, H, device=dev)
    cases = [("W_h", H, X, dict(x1=am, k1=H, x1_idx=b2a, x1_sub=msg, x1_sub_idx=rev)),
             ("W_o", 433, torch.cat([fa[:, :133], a2], 1), dict(x1=fa, k1=133, x2=a2, k2=H))]
    for name, K, Xr, kw in cases:
        got = {}
        for en in (False, True):
            Fn.SplitGemm.enabled = en
            try:
                dw, db = torch.zeros(H, K, device=dev), torch.zeros(H, device=dev)
                Fn.wgrad(M, H, dz, dw, dbias=db, **kw)
                dw2, db2 = dw.clone(), db.clone()
                Fn.wgrad(M, H, dz, dw2, dbias=db2, accumulate=True, **kw)
            finally:
                Fn.SplitGemm.enabled = True
            assert torch.equal(dw2, dw + dw) and torch.equal(db2, db + db)           # fixed-order reduction, accumulate
            got[en] = (dw, db)
        ref, den = dz.double().t() @ Xr.double(), dz.double().abs().t() @ Xr.double().abs() + 1e-300
        _compare(got[False][0], got[True][0], ref, den, name + " dW", False)          # (dz spans e^+-4: see above)
        _compare(got[False][1], got[True][1], dz.double().sum(0), dz.double().abs().sum(0) + 1e-300, name + " dbias", False)


def test_split_and_f32_paths_give_the_same_ranking_and_scores_within_tolerance():
    """Whole model, train mode: scores / loss / gradients of the two GEMM paths agree to the north-star tolerance, the
    candidate ORDER of every query is identical, and the plan honours RR_PLAN_F32_GEMM (bit-identical to the per-op
    f32 path)."""
    from reactranker_amd import featurization, synth
    from reactranker_amd import loss as RL
    from oracle import ref_cpu as O
    from tests.test_gpu_model import make_model
    cfg = dict(hidden_size=300, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, task_num=1,
               ffn_last_layer="with_softplus", task_type=None, add_features_dim=1)
    w = synth.seeded_weights(O.model_shapes(300, 3, 3, 3, 1, 1, True), 11)
    model = make_model(cfg, w, dropout=0.1).train()
    qb = synth.make_queries(5, 6, [9, 4, 12, 7, 3, 10], atoms_lo=6, atoms_hi=20)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    res = {}
    for en, plan in ((True, True), (False, True), (False, False)):
        Fn.SplitGemm.enabled, Fn.StepPlan.enabled = en, plan
        try:
            model.zero_grad()
            model.dropout_seed = 99
            out = model(rb, pb, gpu=0, add_features=qb.add_features)
            l = RL.MLEloss()(out, qb.scope, torch.tensor(qb.targets), 0)
            l.sum().backward()
            res[(en, plan)] = (out.detach().clone(), l.detach().clone(),
                               {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None})
        finally:
            Fn.SplitGemm.enabled, Fn.StepPlan.enabled = True, True
    sp, f32p, f32o = res[(True, True)], res[(False, True)], res[(False, False)]
    assert torch.equal(f32p[0], f32o[0]) and torch.equal(f32p[1], f32o[1])
    for k in f32p[2]:
        assert torch.equal(f32p[2][k], f32o[2][k]), k
    assert ((sp[0] - f32p[0]).abs() / (1 + f32p[0].abs())).max().item() <= 1e-5
    assert ((sp[1] - f32p[1]).abs() / (1 + f32p[1].abs())).max().item() <= 1e-5
    o = 0
    for n in qb.scope:
        assert torch.equal(torch.argsort(sp[0][o:o + n], stable=True), torch.argsort(f32p[0][o:o + n], stable=True))
        o += n
    for k in f32p[2]:
        s = max(1e-3, float(f32p[2][k].abs().max()))
        # two f32 evaluations with different summation orders: gradients of this step cancel to ~1e-3 of their terms, so
        # each path is ~1e-4 of max|g| away from exact arithmetic (the per-GEMM tests above bound each GEMM by itself)
        assert ((sp[2][k] - f32p[2][k]).abs().max() / s).item() <= 5e-4, k


@pytest.mark.parametrize("M,H", [(1, 300), (193, 300), (5000, 300), (777, 600), (500, 64)])
def test_sign_bit_masks_round_trip_through_the_split_gemm(M, H):
    """mask_bits_out of a forward GEMM is the sign of what it stored, in the documented layout, and a dX GEMM that reads it
    (a_mask_bits) equals the one that reads the f32 activation bit for bit - dZ side output included."""
    torch.manual_seed(M + H)
    dev = "cuda"
    W = torch.randn(H, H, device=dev) / 17
    b = torch.randn(H, device=dev)
    x = torch.randn(M, H, device=dev)
    L = Fn.LinW(W, b)
    rowb = int(lib().rr_mask_bits_row_bytes(H))
    bits = torch.zeros(M, rowb, dtype=torch.uint8, device=dev)
    y = Fn.linear(M, H, L.pk(H), w_packed=True, a1=x, k1=H, bias=b, act=Fn.ACT_RELU, drop_p=0.2, seed=77, mask_bits_out=bits)
    got = bits.cpu().numpy()
    pos = (y > 0).cpu().numpy()
    n = np.arange(H)
    blk, t, h, e = n // 304, (n % 304) // 16, ((n % 304) % 16) // 8, n % 8
    dec = (got[:, blk * 40 + h * 20 + t] >> e) & 1
    assert np.array_equal(dec.astype(bool), pos)
    dy = torch.randn(M, H, device=dev)
    dz1, dz2 = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev)
    cw = torch.rand(M, device=dev)
    o1, p1 = Fn.linear(M, H, L.pk_t(0, H), w_packed=True, a1=dy, k1=H, a_mask=y, mask_scale=1.25, dz_out=dz1, colsum_w=cw)
    o2, p2 = Fn.linear(M, H, L.pk_t(0, H), w_packed=True, a1=dy, k1=H, a_mask_bits=bits, mask_scale=1.25, dz_out=dz2, colsum_w=cw)
    assert torch.equal(o1, o2) and torch.equal(dz1, dz2) and torch.equal(p1, p2)
