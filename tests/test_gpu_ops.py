"""GPU parity of the individual HIP ops (called through the C-ABI) against plain fp32 math on the
CPU.  Tolerance for floating point: |got - ref| <= 1e-5 * (1 + |ref|) (north star: 1e-5 fp32);
index / mask results are bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests import helpers as Hh

from reactranker_amd import functions as Fn
from reactranker_amd import _lib
from oracle import dropout_ref

pytestmark = pytest.mark.gpu
TOL = 1e-5


def dev(a):
    return torch.as_tensor(a).cuda()


def close(got, ref, tol=TOL, what=""):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = ref.detach().cpu().double().numpy() if torch.is_tensor(ref) else np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    if got.size == 0:
        return
    err = np.max(np.abs(got - ref) / (1 + np.abs(ref)))
    Hh.record(what, err, tol)
    assert err <= tol, f"{what}: err {err:.3e} > {tol}"


@pytest.mark.parametrize("H,ld", [(300, 300), (32, 32), (30, 30), (83, 84), (600, 600)])
def test_gather_sum_and_diff(H, ld):
    rng = np.random.default_rng(H)
    n_src, n_out, K = 517, 301, 4
    src = np.zeros((n_src, ld), np.float32)
    src[:, :H] = rng.standard_normal((n_src, H))
    idx = rng.integers(-1, n_src, size=(n_out, K)).astype(np.int32)
    idx[0] = 0
    ref = np.where(idx[..., None] >= 0, src[np.maximum(idx, 0)][..., :H], 0).astype(np.float32).sum(1)
    got = Fn.gather_sum(dev(src), dev(idx), H)
    close(got, ref, what="gather_sum")
    ia = rng.integers(-1, n_src, size=n_out).astype(np.int32)
    im = rng.integers(-1, n_src, size=n_out).astype(np.int32)
    ref = np.where(ia[:, None] >= 0, src[np.maximum(ia, 0)][:, :H], 0) - np.where(im[:, None] >= 0, src[np.maximum(im, 0)][:, :H], 0)
    got = Fn.gather_diff(dev(src), dev(ia), dev(src), dev(im), H)
    assert np.array_equal(got.cpu().numpy(), ref.astype(np.float32))          # one subtraction: bit-exact
    # K = 1, 1-D index (the backward use)
    got = Fn.gather_sum(dev(src), dev(ia), H)
    assert np.array_equal(got.cpu().numpy(), np.where(ia[:, None] >= 0, src[np.maximum(ia, 0)][:, :H], 0).astype(np.float32))


def test_gather_sum_is_deterministic_and_linear():
    rng = np.random.default_rng(1)
    src = dev(rng.standard_normal((70000, 300)).astype(np.float32))
    idx = dev(rng.integers(0, 70000, size=(40000, 4)).astype(np.int32))
    a = Fn.gather_sum(src, idx, 300)
    b = Fn.gather_sum(src, idx, 300)
    assert torch.equal(a, b)
    close(Fn.gather_sum(src * 2, idx, 300), a * 2, what="linearity")


@pytest.mark.parametrize("n", [1, 255, 256, 1000, 70001])
def test_weighted_colsum(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, 300)).astype(np.float32)
    w = rng.integers(0, 5, size=n).astype(np.float32)
    out = torch.full((300,), 2.0, device="cuda")
    Fn.weighted_colsum(dev(x), dev(w), 300, out, accumulate=True)
    ref = 2.0 + (w[:, None].astype(np.float64) * x).sum(0)
    # fp32 summation bound: a few ulps of the column's L1 norm (the sums themselves can be ~0)
    l1 = (w[:, None] * np.abs(x)).sum(0).astype(np.float64)
    assert np.all(np.abs(out.cpu().numpy() - ref) <= 1e-6 * l1 + 1e-5), "weighted colsum"
    out2 = torch.empty(300, device="cuda")
    Fn.weighted_colsum(dev(x), None, 300, out2, accumulate=False)
    assert np.all(np.abs(out2.cpu().numpy() - x.astype(np.float64).sum(0)) <= 1e-6 * np.abs(x).sum(0) + 1e-5)
    out3 = torch.empty(300, device="cuda")
    Fn.weighted_colsum(dev(x), None, 300, out3, accumulate=False)
    assert torch.equal(out2, out3)                       # fixed-order reduction: run-to-run identical


def _ref_linear(a, w, bias, res, act):
    y = a.double() @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    if res is not None:
        y = y + res.double()
    pre = y.clone()
    if act:
        y = torch.relu(y)
    return y, pre


@pytest.mark.parametrize("M", [1, 63, 64, 65, 1000])
@pytest.mark.parametrize("N,K,ldk", [(300, 300, 300), (300, 83, 84), (32, 32, 32), (1, 300, 300), (2, 32, 32),
                                     (100, 301, 301), (600, 600, 600), (301, 64, 64), (160, 48, 48)])
def test_linear_plain(M, N, K, ldk):
    torch.manual_seed(M * 1000 + N + K)
    a = torch.zeros(M, ldk)
    a[:, :K] = torch.randn(M, K)
    w = torch.randn(N, K) / K ** 0.5
    bias = torch.randn(N)
    res = torch.randn(M, N)
    ref, pre = _ref_linear(a[:, :K], w, bias, res, True)
    pre_out = torch.empty(M, N, device="cuda")
    got = Fn.linear(M, N, w.cuda(), a1=a.cuda(), k1=K, bias=bias.cuda(), residual=res.cuda(), act=Fn.ACT_RELU,
                    c_pre=pre_out)
    close(got, ref, what="linear relu")
    close(pre_out, pre, what="linear pre")
    got = Fn.linear(M, N, w.cuda(), a1=a.cuda(), k1=K)
    close(got, a[:, :K].double() @ w.double().t(), what="linear bare")


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("M,N,K", [(517, 300, 300), (64, 300, 300), (3, 32, 32), (130, 600, 300)])
def test_linear_residual_gathered_by_index(M, N, K, packed):
    """residual rows picked through residual_idx (-1 = no residual for that row), with bias + ReLU, on the
    straight-line (packed W) and the generic kernel."""
    torch.manual_seed(M + N)
    nR = 41
    a = torch.randn(M, K)
    w = torch.randn(N, K) / K ** 0.5
    b = torch.randn(N)
    res = torch.randn(nR, N)
    ridx = torch.randint(-1, nR, (M,), dtype=torch.int32)
    r = torch.where((ridx >= 0)[:, None], res[ridx.clamp(min=0).long()], torch.zeros(1, N))
    ref = torch.relu(a.double() @ w.double().t() + b.double() + r.double())
    if packed:
        W = Fn.LinW(w.cuda(), b.cuda())
        got = Fn.linear(M, N, W.pk(K), w_packed=True, a1=a.cuda(), k1=K, bias=W.b, residual=res.cuda(),
                        residual_idx=ridx.cuda(), act=Fn.ACT_RELU)
    else:
        got = Fn.linear(M, N, w.cuda(), a1=a.cuda(), k1=K, bias=b.cuda(), residual=res.cuda(), residual_idx=ridx.cuda(),
                        act=Fn.ACT_RELU)
    close(got, ref, what="residual_idx")


@pytest.mark.parametrize("H", [300, 32, 30])
def test_linear_concat_gather_mask(H):
    torch.manual_seed(H)
    nA, nB = 211, 397
    f_atoms = torch.zeros(nA, 64); f_atoms[:, :61] = torch.rand(nA, 61)
    a_msg = torch.randn(nA, H)
    w = torch.randn(H, 61 + H) / 10
    b = torch.randn(H)
    ref = torch.relu(torch.cat([f_atoms[:, :61], a_msg], 1).double() @ w.double().t() + b.double())
    got = Fn.linear(nA, H, w.cuda(), a1=f_atoms.cuda(), k1=61, a2=a_msg.cuda(), k2=H, bias=b.cuda(), act=Fn.ACT_RELU)
    close(got, ref, what="concat")
    # gather + subtract prologue
    msg = torch.randn(nB, H)
    b2a = torch.randint(0, nA, (nB,), dtype=torch.int32)
    b2r = torch.randint(0, nB, (nB,), dtype=torch.int32)
    wh = torch.randn(H, H) / H ** 0.5
    inp = torch.randn(nB, H)
    m_in = a_msg[b2a.long()] - msg[b2r.long()]
    ref = torch.relu(inp.double() + m_in.double() @ wh.double().t())
    got = Fn.linear(nB, H, wh.cuda(), a1=a_msg.cuda(), k1=H, a1_idx=b2a.cuda(), a1_sub=msg.cuda(),
                    a1_sub_idx=b2r.cuda(), residual=inp.cuda(), act=Fn.ACT_RELU)
    close(got, ref, what="gather-sub")
    # plain subtract (diff = p - r) with a second segment
    p_h, r_h = torch.randn(nA, H), torch.randn(nA, H)
    w2 = torch.randn(H, 2 * H) / H ** 0.5
    ref = torch.cat([p_h - r_h, a_msg], 1).double() @ w2.double().t()
    got = Fn.linear(nA, H, w2.cuda(), a1=p_h.cuda(), k1=H, a1_sub=r_h.cuda(), a2=a_msg.cuda(), k2=H)
    close(got, ref, what="sub+concat")
    # relu-backward mask prologue
    dy, y = torch.randn(nB, H), torch.relu(torch.randn(nB, H))
    ref = (dy * (y > 0) * 1.25).double() @ wh.double().t()
    got = Fn.linear(nB, H, wh.cuda(), a1=dy.cuda(), k1=H, a_mask=y.cuda(), mask_scale=1.25)
    close(got, ref, what="mask")
    # in-place residual accumulate
    c = torch.randn(nB, H)
    cc = c.cuda()
    Fn.linear(nB, H, wh.cuda(), a1=dy.cuda(), k1=H, residual=cc, out=cc)
    close(cc, c.double() + dy.double() @ wh.double().t(), what="inplace")


def test_linear_dropout_matches_stream():
    torch.manual_seed(0)
    M, N, K, p, seed = 130, 300, 64, 0.2, 0x1234567890ABCDEF
    a, w = torch.randn(M, K), torch.randn(N, K) / 8
    ref = torch.relu(a.double() @ w.double().t())
    keep = dropout_ref.keep_mask(seed, np.arange(M * N, dtype=np.uint64), p).reshape(M, N)
    ref = torch.where(torch.from_numpy(keep), ref / (1 - np.float32(p)).astype(np.float64), torch.zeros_like(ref))
    got = Fn.linear(M, N, w.cuda(), a1=a.cuda(), k1=K, act=Fn.ACT_RELU, drop_p=p, seed=seed)
    close(got, ref, what="dropout epilogue")
    x = torch.randn(1000)
    got = Fn.dropout(x.cuda(), p, seed)
    keep = dropout_ref.keep_mask(seed, np.arange(1000, dtype=np.uint64), p)
    assert np.array_equal((got.cpu().numpy() != 0), keep & (x.numpy() != 0))


@pytest.mark.parametrize("M", [1, 16, 100, 5000, 70000])
@pytest.mark.parametrize("N,k1,k2", [(300, 300, 0), (300, 83, 0), (300, 61, 300), (32, 32, 83), (1, 300, 0), (2, 33, 0),
                                     (300, 300, 83)])
def test_wgrad(M, N, k1, k2):
    if M == 70000 and N < 300:
        pytest.skip("large-M only for the hot shapes")
    torch.manual_seed(M + N + k1 + k2)
    ld1 = (k1 + 3) // 4 * 4
    x1 = torch.zeros(M, ld1); x1[:, :k1] = torch.randn(M, k1)
    x2 = torch.randn(M, k2) if k2 else None
    dy = torch.randn(M, N)
    y = torch.relu(torch.randn(M, N))
    dz = (dy * (y > 0) * 0.5).double()
    X = x1[:, :k1].double() if x2 is None else torch.cat([x1[:, :k1], x2], 1).double()
    ref_w, ref_b = dz.t() @ X, dz.sum(0)
    dw = torch.empty(N, k1 + k2, device="cuda"); db = torch.empty(N, device="cuda")
    Fn.wgrad(M, N, dy.cuda(), dw, dbias=db, mask=y.cuda(), mask_scale=0.5, x1=x1.cuda(), k1=k1,
             x2=None if x2 is None else x2.cuda(), k2=k2)
    s = max(1.0, float(ref_w.abs().max()))
    close(dw / s, ref_w / s, tol=2e-5, what="dw")
    close(db / s, ref_b / s, tol=2e-5, what="db")
    Fn.wgrad(M, N, dy.cuda(), dw, dbias=db, mask=y.cuda(), mask_scale=0.5, x1=x1.cuda(), k1=k1,
             x2=None if x2 is None else x2.cuda(), k2=k2, accumulate=True)
    close(dw / s, 2 * ref_w / s, tol=4e-5, what="dw accumulate")


@pytest.mark.parametrize("gather_a,gather_s,mask", [(True, True, False), (True, True, True), (False, False, True),
                                                     (False, True, False), (True, False, False)])
@pytest.mark.parametrize("nB", [290, 17, 4099])
def test_wgrad_gather_sub_operand(gather_a, gather_s, mask, nB):
    """x = a[ia] - s[is] in the operand loader: direct / gathered sources, negative indices read as zero rows,
    row counts that are not a multiple of the 16-row tile."""
    torch.manual_seed(3 + nB)
    nA, H = 150, 300
    src_a = torch.randn(nA if gather_a else nB, H)
    src_s = torch.randn(nB + 7 if gather_s else nB, H)
    dz, y = torch.randn(nB, H), torch.relu(torch.randn(nB, H))
    ia = torch.randint(-1, nA, (nB,), dtype=torch.int32) if gather_a else None
    isub = torch.randint(-1, nB + 7, (nB,), dtype=torch.int32) if gather_s else None

    def rows(src, idx):
        if idx is None:
            return src
        return torch.where((idx >= 0)[:, None], src[idx.clamp(min=0).long()], torch.zeros(1, H))
    X = (rows(src_a, ia) - rows(src_s, isub)).double()
    dzz = (dz * (y > 0) * 1.25 if mask else dz).double()
    dw = torch.empty(H, H, device="cuda"); db = torch.empty(H, device="cuda")
    Fn.wgrad(nB, H, dz.cuda(), dw, dbias=db, mask=y.cuda() if mask else None, mask_scale=1.25, x1=src_a.cuda(), k1=H,
             x1_idx=None if ia is None else ia.cuda(), x1_sub=src_s.cuda(), x1_sub_idx=None if isub is None else isub.cuda())
    ref = dzz.t() @ X
    s = float(ref.abs().max())
    close(dw / s, ref / s, tol=2e-5, what="dw gather-sub")
    close(db / s, dzz.sum(0) / s, tol=2e-5, what="db gather-sub")


def test_relu_bwd_axpby_head_segment():
    torch.manual_seed(5)
    dy, y, acc = torch.randn(1001, 30), torch.relu(torch.randn(1001, 30)), torch.randn(1001, 30)
    a = acc.cuda()
    dz = Fn.relu_bwd(dy.cuda(), y.cuda(), -1.5, acc=a)
    ref = dy * (y > 0) * -1.5
    assert torch.equal(dz.cpu(), ref)
    close(a, acc + ref)
    close(Fn.axpby(2.0, dy.cuda(), -1.0, y.cuda()), 2 * dy - y)
    for head, N in ((0, 1), (1, 1), (2, 1), (3, 2), (4, 2), (5, 2), (6, 4), (3, 4), (6, 8)):
        raw = torch.randn(50, N) * 5
        raw[0, 0] = 25.0                                  # softplus threshold branch
        r = raw.clone().requires_grad_(True)
        sp = torch.nn.Softplus()
        mv = 1e-6
        if head == 0: ref = r
        elif head == 1: ref = sp(r)
        elif head == 2: ref = sp(r) + 1
        elif head == 3:
            s_, u = torch.split(r, N // 2, dim=1); ref = torch.stack((s_, sp(u) + mv), dim=2).view(r.size())
        elif head == 4:
            s_, u = torch.split(r, N // 2, dim=1); ref = torch.stack((s_, sp(u)), dim=2).view(r.size())
        elif head == 5:
            s_, u = torch.split(r, N // 2, dim=1); ref = torch.stack((sp(s_) + mv, sp(u) + mv), dim=2).view(r.size())
        else:
            m_, l_, a_, b_ = torch.split(r, N // 4, dim=1)
            ref = torch.stack((m_, sp(l_) + mv, sp(a_) + mv + 1, sp(b_) + mv), dim=2).view(r.size())
        got = Fn.head_fwd(raw.cuda(), head)
        close(got, ref, what=f"head {head}")
        go = torch.randn(50, N)
        ref.backward(go)
        close(Fn.head_bwd(go.cuda(), raw.cuda(), head), r.grad, what=f"head bwd {head}")


@pytest.mark.parametrize("M,N,K", [(1, 300, 300), (63, 300, 300), (64, 32, 32), (4099, 300, 300), (1000, 600, 300),
                                   (70001, 300, 300)])
def test_linear_weighted_colsum_side_output_and_padrow_gather(M, N, K):
    """rr_linear_args.colsum_partial: per-row-block partial sums of w[m] * C[m, :] next to the GEMM output, consumed by
    rr_gather_sum_padrow_f32 as output row 0 (the padding row's adjoint, features/featurization.py:286)."""
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g)
    y = torch.relu(torch.randn(M, K, generator=g))                 # ReLU mask source (MODE 2, the backward's usual caller)
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    rw = torch.randint(0, 5, (M,), generator=g).float()            # npad-like weights
    W = Fn.LinW(w.cuda(), None)
    dz = torch.empty(M, K).cuda()
    out, part = Fn.linear(M, N, W.pk(K), w_packed=True, a1=a.cuda(), k1=K, a_mask=y.cuda(), mask_scale=1.25, dz_out=dz,
                          colsum_w=rw.cuda())
    az = (a * (y > 0) * 1.25).double()
    ref = az @ w.double().t()
    close(out, ref.float(), tol=2e-5, what="C")
    assert part.shape[0] == (M + 63) // 64 == int(_lib.lib().rr_linear_colsum_rows(M))
    want = (rw.double()[:, None] * ref).sum(0)
    got = part[:, :N].double().sum(0).cpu()
    scale = float((rw.double()[:, None] * ref.abs()).sum(0).max()) + 1e-30      # size of the summed terms
    assert float((got - want).abs().max()) <= 2e-6 * scale
    # plain kernels (no mask) write it too; and the gather's row 0 = ordered sum of the partial rows
    out2, part2 = Fn.linear(M, N, W.pk(K), w_packed=True, a1=dz, k1=K, colsum_w=rw.cuda())
    assert torch.equal(out2, out) and torch.equal(part2, part)
    n_out, Kt = 77, 3
    idx = torch.randint(-1, M, (n_out, Kt), generator=g).to(torch.int32)
    idx[0] = -1
    gs = Fn.gather_sum(out, idx.cuda(), N, row0_partial=part)
    plain = Fn.gather_sum(out, idx.cuda(), N)
    assert torch.equal(gs[1:], plain[1:]) and float(plain[0].abs().max()) == 0.0
    s = torch.zeros(N, dtype=torch.float32).cuda()
    ref0 = part[:, :N].double().sum(0)
    assert float((gs[0].double() - ref0).abs().max()) <= 2e-6 * scale
    assert torch.equal(Fn.gather_sum(out, idx.cuda(), N, row0_partial=part)[0], gs[0])     # fixed order: bit-stable


@pytest.mark.parametrize("H", [300, 30])
@pytest.mark.parametrize("K", [1, 3, 64])
def test_gather_sum_masked_equals_relu_backward_then_gather(H, K):
    """The fused pass of the shared-prefix backward: bit-identical to rr_relu_bwd_f32 followed by rr_gather_sum_f32."""
    rng = np.random.default_rng(H + K)
    n_src, n_out = 1543, 97
    src = dev(rng.standard_normal((n_src, H)).astype(np.float32))
    mask = dev(np.maximum(rng.standard_normal((n_src, H)), 0).astype(np.float32))
    idx = rng.integers(-1, n_src, size=(n_out, K)).astype(np.int32)
    idx[0] = -1
    got = Fn.gather_sum_masked(src, mask, 1.0 / 0.9, dev(idx), H)
    ref = Fn.gather_sum(Fn.relu_bwd(src, mask, 1.0 / 0.9), dev(idx), H)
    assert torch.equal(got, ref)
    want = np.where(idx[..., None] >= 0, np.where(mask.cpu().numpy() > 0, src.cpu().numpy() * np.float32(1.0 / 0.9), 0)[np.maximum(idx, 0)], 0).sum(1)
    close(got, want, what="masked gather")


@pytest.mark.parametrize("H", [300, 600, 30])
@pytest.mark.parametrize("K", [1, 2, 5, 12, 64])
@pytest.mark.parametrize("n_srcs", [1, 2, 3, 4, 6])
def test_gather_sum_of_a_sum_equals_the_axpby_chain_then_gather(H, K, n_srcs):
    """rr_gather_sum_multi_f32 (shared-prefix backward at depth >= 4: the per-copy layers' dZ summed over the layers and over
    the copies in one pass) against the sequence it replaces - rr_axpby_f32 passes forming the inner sum, then rr_gather_sum_f32:
    torch.equal.  More than RR_MAX_GATHER_SRCS addends, and H = 30 (rows that are not 16-byte chunks), take the pre-summed route."""
    rng = np.random.default_rng(1000 * H + 10 * K + n_srcs)
    n_src, n_out = 1543, 97
    srcs = [dev((rng.standard_normal((n_src, H)) * 0.05).astype(np.float32)) for _ in range(n_srcs)]
    idx = rng.integers(-1, n_src, size=(n_out, K)).astype(np.int32)
    idx[0] = -1
    inner = srcs[0]
    for t in srcs[1:]:
        inner = Fn.axpby(1.0, inner, 1.0, t)
    ref = Fn.gather_sum(inner, dev(idx), H)
    got = Fn.gather_sum_multi(srcs, dev(idx), H)
    assert torch.equal(got, ref)
    assert float(got[0].abs().max()) == 0.0
    os.environ["RR_NO_GATHER_MULTI"] = "1"
    try:
        assert torch.equal(Fn.gather_sum_multi(srcs, dev(idx), H), ref)
    finally:
        del os.environ["RR_NO_GATHER_MULTI"]
    want = sum(np.where(idx[..., None] >= 0, t.cpu().numpy().astype(np.float64)[np.maximum(idx, 0)], 0).sum(1) for t in srcs)
    close(got, want, what="gather of a sum")


def test_gather_sum_multi_rejects_bad_arguments():
    src = dev(np.zeros((8, 30), np.float32))
    idx = dev(np.zeros((4, 2), np.int32))
    out = torch.empty(4, 30, device="cuda")
    arr = (C.c_void_p * 5)(*([src.data_ptr()] * 5))
    L = _lib.lib()
    assert L.rr_gather_sum_multi_f32(arr, 2, 8, 30, Fn.ptr(idx), 4, 2, 30, Fn.ptr(out), 30, None) == -2      # RR_ERR_ALIGN
    assert L.rr_gather_sum_multi_f32(arr, 5, 8, 32, Fn.ptr(idx), 4, 2, 32, Fn.ptr(out), 32, None) == -1
    assert L.rr_gather_sum_multi_f32(arr, 0, 8, 32, Fn.ptr(idx), 4, 2, 32, Fn.ptr(out), 32, None) == -1


@pytest.mark.parametrize("H,p", [(300, 0.1), (300, 0.0), (64, 0.35), (600, 0.2)])
@pytest.mark.parametrize("copies", [1, 5, 64])
def test_gather_sum_with_derived_dropout_mask_equals_the_materialised_mask(H, p, copies):
    """rr_gather_sum_dropmask_f32 (shared-prefix backward: the copies' masks re-derived from the dropout stream and the small
    pre-dropout tensor) against rr_gather_sum_masked_f32 on the copies rr_gather_dropout_f32 actually produced: torch.equal,
    incl. destinations with fewer copies than the table is wide and one with none."""
    rng = np.random.default_rng(H + copies)
    n_u, seed = 41, 0x1234ABCD5678
    n_full = n_u * copies + 1
    y = dev(np.maximum(rng.standard_normal((n_u, H)), 0).astype(np.float32))                # relu output of the shared layer
    owner = rng.integers(0, n_u, size=n_full).astype(np.int32)                               # full row -> distinct row
    owner[owner == 7] = 8                                                                    # row 7 has no copies
    full = Fn.gather_dropout(y, dev(owner), H, p, seed)                                      # what the forward materialises
    table = np.full((n_u, max(1, int(np.bincount(owner, minlength=n_u).max()))), -1, np.int32)
    fill = np.zeros(n_u, np.int64)
    for j, u in enumerate(owner):
        table[u, fill[u]] = j
        fill[u] += 1
    src = dev(rng.standard_normal((n_full, H)).astype(np.float32))
    ks = 1.0 / (1.0 - p)
    ref = Fn.gather_sum_masked(src, full, ks, dev(table), H)
    got = Fn.gather_sum_dropmask(src, y, ks, dev(table), H, p, seed)
    assert torch.equal(got, ref)
    assert float(got[7].abs().max()) == 0.0
    if p > 0:
        assert not torch.equal(got, Fn.gather_sum_dropmask(src, y, ks, dev(table), H, p, seed + 1))   # the stream matters


@pytest.mark.parametrize("H,F,p", [(300, 1, 0.1), (300, 0, 0.0), (32, 1, 0.3), (30, 1, 0.2)])
def test_segment_mean_backward_is_the_adjoint_with_the_forward_dropout_stream(H, F, p):
    """Readout (models/mpn.py:224-238): dx[a] = dout[mol(a)] / size * keep / (1 - p), keep taken from the SAME counter-based
    stream element (m * (H + F) + c) the forward used - the 4-column kernel (H % 4 == 0) and the scalar one."""
    from types import SimpleNamespace
    rng = np.random.default_rng(H + F)
    sizes = rng.integers(1, 9, size=37)
    M, nA = len(sizes), int(sizes.sum()) + 1                         # atom 0 is the padding atom (no molecule)
    starts = np.concatenate([[1], 1 + np.cumsum(sizes)[:-1]])
    a_scope = np.stack([starts, sizes], 1).astype(np.int32)
    atom2mol = np.full(nA, -1, np.int32)
    for m, (s0, n) in enumerate(a_scope):
        atom2mol[s0:s0 + n] = m
    g = SimpleNamespace(a_scope=dev(a_scope), atom2mol=dev(atom2mol), M=M, nA=nA)
    W = H + F
    dout = rng.standard_normal((M, W)).astype(np.float32)
    seed = 0x1234567890ABCDEF
    got = Fn.segment_mean_bwd(dev(dout), g, H, F, p, seed)
    keep = dropout_ref.keep_mask(seed, (np.arange(M)[:, None] * W + np.arange(H)[None, :]).astype(np.uint64), p) if p > 0 \
        else np.ones((M, H), bool)
    per_mol = (dout[:, :H] / sizes[:, None].astype(np.float32)).astype(np.float32)
    per_mol = np.where(keep, per_mol * np.float32(1.0 / (1.0 - p)), 0).astype(np.float32) if p > 0 else per_mol
    want = np.zeros((nA, H), np.float32)
    want[atom2mol >= 0] = per_mol[atom2mol[atom2mol >= 0]]
    close(got, want, tol=1e-6, what="segment_mean_bwd")
    x = rng.standard_normal((nA, H)).astype(np.float32)
    feat = rng.standard_normal((M, max(F, 1))).astype(np.float32)[:, :F]
    fwd = Fn.segment_mean_fwd(dev(x), g, H, dev(feat) if F else None, F, p, seed)
    lhs = float((fwd[:, :H].double().cpu() * torch.as_tensor(dout[:, :H]).double()).sum())
    rhs = float((torch.as_tensor(x).double() * got.double().cpu()).sum())
    assert abs(lhs - rhs) <= 1e-5 * (1 + abs(lhs))                  # <fwd(x), dout> == <x, bwd(dout)>


@pytest.mark.parametrize("H", [300, 600, 32])
@pytest.mark.parametrize("K,n_adds,padrow", [(1, 0, True), (3, 2, True), (4, 5, True), (4, 1, False), (3, 7, True), (2, 15, False)])
def test_gather_sum_epilogue_equals_the_separate_kernel_sequence(H, K, n_adds, padrow):
    """rr_gather_sum_epi_f32 (ABI revision 4): out = mask(sum_k src[idx]) * scale + sum(adds) in ONE pass, bit-identical
    to rr_gather_sum(_padrow)_f32 followed by rr_relu_bwd_sum_f32 - with the mask read as the f32 activation and as the
    sign-bit image a split GEMM wrote for it (models/mpn.py:89-97 backward)."""
    rng = np.random.default_rng(H + 10 * K + n_adds)
    n_src, n_out = 1543, 977
    src = dev(rng.standard_normal((n_src, H)).astype(np.float32))
    idx = rng.integers(-1, n_src, size=(n_out, K)).astype(np.int32)
    idx[5] = -1
    adds = [dev(rng.standard_normal((n_out, H)).astype(np.float32)) for _ in range(n_adds)]
    part = dev(rng.standard_normal((29, (H + 3) // 4 * 4)).astype(np.float32)) if padrow else None
    # the mask: the output y of a split GEMM (ReLU + dropout), with its sign-bit image
    W = torch.randn(H, H, device="cuda") / 17
    x = torch.randn(n_out, H, device="cuda")
    y = Fn.linear(n_out, H, Fn.LinW(W, None).pk(H), w_packed=True, a1=x, k1=H, act=Fn.ACT_RELU, drop_p=0.2, seed=9, want_bits=True)
    assert getattr(y, "_rr_bits", None) is not None
    y_plain = y.clone()                                              # same values, no sign-bit image attached
    g = Fn.gather_sum(src, dev(idx), H, row0_partial=part)
    want = Fn.relu_bwd_sum(g, y_plain, 1.25, adds)
    got_bits = Fn.gather_sum(src, dev(idx), H, row0_partial=part, mask=y, mask_scale=1.25, adds=adds)
    got_f32 = Fn.gather_sum(src, dev(idx), H, row0_partial=part, mask=y_plain, mask_scale=1.25, adds=adds)
    assert torch.equal(got_bits, want) and torch.equal(got_f32, want)
    # adds only / mask only
    if n_adds:
        assert torch.equal(Fn.gather_sum(src, dev(idx), H, row0_partial=part, adds=adds),
                           Fn.relu_bwd_sum(g, torch.ones_like(g), 1.0, adds))
    assert torch.equal(Fn.gather_sum(src, dev(idx), H, row0_partial=part, mask=y, mask_scale=1.25), Fn.relu_bwd(g, y_plain, 1.25))
    ref = np.where(y.cpu().numpy() > 0, g.cpu().numpy() * np.float32(1.25), 0)
    for t in adds:
        ref = ref + t.cpu().numpy()
    close(got_bits, ref, what="gather epilogue")


def test_segment_mean_backward_with_the_fused_mask():
    """rr_segment_mean_bwd_masked_f32 == rr_segment_mean_bwd_f32 followed by rr_relu_bwd_f32, f32 mask and sign bits."""
    from types import SimpleNamespace
    rng = np.random.default_rng(11)
    H, F, p = 300, 1, 0.1
    sizes = rng.integers(1, 9, size=211)
    M, nA = len(sizes), int(sizes.sum()) + 1
    starts = np.concatenate([[1], 1 + np.cumsum(sizes)[:-1]])
    a_scope = np.stack([starts, sizes], 1).astype(np.int32)
    atom2mol = np.full(nA, -1, np.int32)
    for m, (s0, n) in enumerate(a_scope):
        atom2mol[s0:s0 + n] = m
    g = SimpleNamespace(a_scope=dev(a_scope), atom2mol=dev(atom2mol), M=M, nA=nA)
    dout = dev(rng.standard_normal((M, H + F)).astype(np.float32))
    W = torch.randn(H, H, device="cuda") / 17
    hid = Fn.linear(nA, H, Fn.LinW(W, None).pk(H), w_packed=True, a1=torch.randn(nA, H, device="cuda"), k1=H, act=Fn.ACT_RELU,
                    drop_p=p, seed=3, want_bits=True)
    plain = hid.clone()
    want = Fn.relu_bwd(Fn.segment_mean_bwd(dout, g, H, F, p, 77), plain, 1.0 / (1.0 - p))
    assert torch.equal(Fn.segment_mean_bwd(dout, g, H, F, p, 77, mask=hid, mask_scale=1.0 / (1.0 - p)), want)
    assert torch.equal(Fn.segment_mean_bwd(dout, g, H, F, p, 77, mask=plain, mask_scale=1.0 / (1.0 - p)), want)


def test_hip_adam_is_torch_adam_in_one_launch():
    """train_utils.HipAdam (rr_adam_step_f32) against torch.optim.Adam on the same gradients: parameters and moments over
    five steps with a changing learning rate (NoamLR writes param_groups[0]['lr']), a parameter that gets no gradient in
    some steps (skipped, its step count stays behind), weight decay, and the state_dict layout torch's Adam has."""
    from reactranker_amd.train_utils import HipAdam
    torch.manual_seed(0)
    shapes = [(300, 83), (300,), (300, 300), (1, 300), (7,), (2049,)]
    for wd in (0.0, 0.01):
        P1 = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
        P2 = [torch.nn.Parameter(p.detach().clone()) for p in P1]
        o1 = HipAdam([{"params": P1, "lr": 1e-4, "weight_decay": wd}])
        o2 = torch.optim.Adam([{"params": P2, "lr": 1e-4, "weight_decay": wd}])
        for step in range(5):
            lr = 1e-4 * (1 + step)
            o1.param_groups[0]["lr"] = o2.param_groups[0]["lr"] = lr
            for i, (a, b) in enumerate(zip(P1, P2)):
                if i == 4 and step in (1, 3):
                    a.grad = b.grad = None                               # skipped like torch skips it
                    continue
                g = torch.randn_like(a) * (10.0 ** (step - 3))
                a.grad, b.grad = g.clone(), g.clone()
            o1.step()
            o2.step()
            for i, (a, b) in enumerate(zip(P1, P2)):
                scale = float(b.detach().abs().max())
                err = float((a.detach() - b.detach()).abs().max())
                Hh.record(f"HipAdam vs torch.optim.Adam, parameters (wd {wd})", err / scale, 1e-6)
                assert err <= 1e-6 * scale, (wd, step, i, err)
                for key in ("exp_avg", "exp_avg_sq"):
                    m1, m2 = o1.state[a][key], o2.state[b][key]
                    assert float((m1 - m2).abs().max()) <= 1e-6 * float(m2.abs().max()) + 1e-30, (key, step, i)
                assert int(o1.state[a]["step"]) == int(o2.state[b]["step"])
        sd1, sd2 = o1.state_dict(), o2.state_dict()
        assert sd1["state"].keys() == sd2["state"].keys()
        assert all(set(sd1["state"][k]) == {"step", "exp_avg", "exp_avg_sq"} for k in sd1["state"])
    # Adam's first update moves every entry with a gradient above eps by lr, whatever the gradient's size
    p = torch.nn.Parameter(torch.zeros(1000, device="cuda"))
    o = HipAdam([p], lr=1e-3)
    p.grad = torch.full_like(p, 3e-5)
    o.step()
    assert torch.allclose(p.detach(), torch.full_like(p, -1e-3), rtol=1e-3, atol=0)
