"""The kernels and the path the headline bench line actually times, pinned to an external reference.

rr_linear_f32 sends split GEMMs with M > 8192 rows to the 12-wave geometry linear_split_kernel<19,19,MODE,12> (two-step
operand prefetch, counted vmcnt waits, sign-bit masks, 192-row blocks) and H = 600 to <38,19,MODE,12>; everything below
8192 rows runs <19,5,MODE,8>.  The tests of tests/test_gpu_split.py stay below that switch, so here every MODE of the
12-wave geometry is compared with an f64 GEMM at the row counts of a BASELINE configs[2] step (71,425 atoms / 138,881
bonds) and right above the switch, and the whole model is compared with the fp64 oracle in TRAIN mode (dropout 0.1,
step plan, shared reactant prefix) on a batch whose atom and bond counts exceed 8192.
Reference layers: models/mpn.py:84-105, 199-219; models/base_model.py:150-171."""
import numpy as np
import pytest
import torch

from reactranker_amd import functions as Fn
from reactranker_amd import featurization, synth
from reactranker_amd import loss as RL
from reactranker_amd._lib import lib
from oracle import ref_cpu as O
from tests.test_gpu_model import _masks_for, close, make_model
from tests import helpers as Hh
from tests.test_gpu_split import _compare

pytestmark = pytest.mark.gpu


def _packs(W, b):
    """(f32-MFMA packs, split packs) of W (forward layout) and W^T (the dX layout)."""
    H = W.shape[0]
    Fn.SplitGemm.enabled = False
    try:
        f32 = (Fn.LinW(W, b).pk(W.shape[1]), Fn.LinW(W, None).pk_t(0, W.shape[1]))
    finally:
        Fn.SplitGemm.enabled = True
    sp = (Fn.LinW(W, b).pk(W.shape[1]), Fn.LinW(W, None).pk_t(0, W.shape[1]))
    assert f32[0].dtype == torch.float32 and sp[0].dtype == torch.uint8 and H % 4 == 0
    return f32, sp


@pytest.mark.parametrize("M,H", [(8193, 300), (71425, 300), (138881, 300), (20000, 600)])
def test_twelve_wave_split_geometry_against_f64(M, H):
    """Modes 0 / 1 / 2 / 3 of the geometry the bench times: error against an f64 GEMM not above the exact-f32 MFMA
    chain's (same bounds as tests/test_gpu_split.py), the dZ side output exact, and MODE 3 (sign bits) == MODE 2 (f32
    mask) bit for bit in output, dZ and the weighted column-sum partials."""
    torch.manual_seed(M + H)
    dev = "cuda"
    W = torch.randn(H, H, device=dev) / 17
    b = torch.randn(H, device=dev)
    (w32f, w32t), (wspf, wspt) = _packs(W, b)
    z = torch.zeros(1, device=dev)
    Wd = W.double()

    # ---- MODE 0: plain operand, bias + residual + ReLU + sign-bit side output (a forward layer)
    x = torch.randn(M, H, device=dev)
    res = torch.randn(M, H, device=dev)
    rowb = int(lib().rr_mask_bits_row_bytes(H))
    bits = torch.zeros(M, rowb, dtype=torch.uint8, device=dev)
    kw = dict(a1=x, k1=H, bias=b, residual=res)
    o32 = Fn.linear(M, H, w32f, w_packed=True, **kw)
    osp = Fn.linear(M, H, wspf, w_packed=True, **kw)
    ref = x.double() @ Wd.t() + b.double() + res.double()
    den = x.double().abs() @ Wd.abs().t() + b.double().abs() + res.double().abs() + 1e-300
    _compare(o32, osp, ref, den, f"mode 0 M={M} H={H}")
    y = Fn.linear(M, H, wspf, w_packed=True, act=Fn.ACT_RELU, drop_p=0.1, seed=5, mask_bits_out=bits, **kw)
    del o32, osp, ref, den

    # ---- MODE 1: gathered operand minus gathered operand (the W_h layer of models/mpn.py:91-95)
    nA = M // 2 + 3
    am = torch.randn(nA, H, device=dev)
    b2a = torch.randint(-1, nA, (M,), device=dev, dtype=torch.int32)
    rev = torch.randint(-1, M, (M,), device=dev, dtype=torch.int32)
    kw = dict(a1=am, k1=H, a1_idx=b2a, a1_sub=y, a1_sub_idx=rev, bias=b, residual=res)
    o32 = Fn.linear(M, H, w32f, w_packed=True, **kw)
    osp = Fn.linear(M, H, wspf, w_packed=True, **kw)
    A = torch.where(b2a[:, None] >= 0, am[b2a.clamp(min=0).long()], z) - torch.where(rev[:, None] >= 0, y[rev.clamp(min=0).long()], z)
    ref = A.double() @ Wd.t() + b.double() + res.double()
    den = A.double().abs() @ Wd.abs().t() + b.double().abs() + res.double().abs() + 1e-300
    _compare(o32, osp, ref, den, f"mode 1 M={M} H={H}")
    del o32, osp, ref, den, A, am

    # ---- MODE 2 and MODE 3: dX = (dy * mask) W with the dZ side output and the weighted column sums
    dy = torch.randn(M, H, device=dev)
    cw = torch.rand(M, device=dev)
    dz32, dz2, dz3 = (torch.empty(M, H, device=dev) for _ in range(3))
    o32, p32 = Fn.linear(M, H, w32t, w_packed=True, a1=dy, k1=H, a_mask=y, mask_scale=1.1, dz_out=dz32, colsum_w=cw)
    o2, p2 = Fn.linear(M, H, wspt, w_packed=True, a1=dy, k1=H, a_mask=y, a_mask_bits=None, mask_scale=1.1, dz_out=dz2, colsum_w=cw)
    o3, p3 = Fn.linear(M, H, wspt, w_packed=True, a1=dy, k1=H, a_mask_bits=bits, mask_scale=1.1, dz_out=dz3, colsum_w=cw)
    dzr = torch.where(y > 0, dy * 1.1, torch.zeros_like(dy))
    assert torch.equal(dz32, dzr) and torch.equal(dz2, dzr) and torch.equal(dz3, dzr)
    assert torch.equal(o2, o3) and torch.equal(p2, p3)                   # sign bits == f32 mask, bit for bit
    ref = dzr.double() @ Wd
    den = dzr.double().abs() @ Wd.abs() + 1e-300
    _compare(o32, o3, ref, den, f"mode 3 M={M} H={H}")
    # the column-sum side output: partial rows sum to sum_m cw[m] * C[m, :] (of the values the kernel stored)
    want = (o3.double() * cw.double()[:, None]).sum(0)
    got = p3.double().sum(0)[:H]
    scale = (o3.double().abs() * cw.double()[:, None]).sum(0) + 1e-300
    assert float(((got - want).abs() / scale).max()) < 1e-6


def _rel_err(got, ref):
    got, ref = got.detach().cpu().double().reshape(-1), ref.detach().cpu().double().reshape(-1)
    return float(((got - ref).abs() / (1 + ref.abs())).max())


def _hip_gates(model, rb, masks):
    """The ReLU gates the HIP step just took, keyed like the oracle's layers, from the activations its workspace keeps
    (rr_reaction_saved_f32): a stored output is > 0 exactly where the gate was open AND dropout kept the element, which is
    all the backward ever uses.  The shared reactant prefix stores relu(W_i f_bonds) per DISTINCT bond: expanded by bmap."""
    from reactranker_amd import _lib
    d, dd, nf = model.encoder.depth, model.diff_encoder.depth, len(model.ffn.linears())
    S = Fn.StepPlan.saved
    gates = {}

    def put(key, t):
        assert t is not None, key
        gates[key] = (t > 0).cpu()
    for tag, MSG, Hh in (("r", _lib.RR_SAVED_R_MSG, _lib.RR_SAVED_R_H), ("p", _lib.RR_SAVED_P_MSG, _lib.RR_SAVED_P_H)):
        m0 = S(MSG, 0)
        if m0 is None:                                   # RR_STEP_PREFIX (reactant side)
            bmap, _ = rb.unique_bonds()
            m0 = S(_lib.RR_SAVED_R_MSG0_U)[torch.from_numpy(bmap).long().cuda()]
        put(f"{tag}.enc.in", m0)
        for it in range(d - 1):
            put(f"{tag}.enc.{it}", S(MSG, it + 1))
        put(f"{tag}.enc.out", S(Hh))
    put("diff.in", S(_lib.RR_SAVED_D_MSG, 0))
    for it in range(dd - 1):
        put(f"diff.{it}", S(_lib.RR_SAVED_D_MSG, it + 1))
    put("diff.out", S(_lib.RR_SAVED_D_HID))
    for li in range(1, nf):
        put(f"ffn.{li}", S(_lib.RR_SAVED_FFN_H, li))
    return gates


def _train_step_vs_fp64_oracle(cfg, Q, Cn, loss_kind, p, atoms_lo, atoms_hi, seed, log):
    """One training step (train mode, dropout p, step plan, shared reactant prefix) against the fp64 oracle with
    identical dropout masks.  Two finite-precision evaluations of a ReLU network differ in KIND, not only in rounding,
    wherever a pre-activation lies within their error of zero: the gate opens in one and stays shut in the other, and each
    such flip moves one row of a weight gradient (and, through dX, everything below it) by a non-rounding amount.  That
    is separated from rounding here instead of being given an allowance:
      1. every gate where the HIP step and the fp64 oracle disagree (among the elements dropout kept) must have an fp64
         pre-activation within 1e-5 of zero - nothing else may flip;
      2. with the oracle's gates set to the ones the HIP step took (oracle `gates=`), every parameter gradient must meet
         the tight bound - 5e-5 of the tensor's largest entry, or 3 x the fp32 oracle's own distance to fp64 under the
         same gates - with no outlier allowance;
      3. scores and loss within 1e-5 (1 + |ref|), or 3 x the fp32 oracle's distance where that is larger.
    The measured numbers (flips per layer, errors with natural and with dictated gates) go to the parity log."""
    H, d, dd = cfg["hidden_size"], cfg["mpnn_depth"], cfg["mpnn_diff_depth"]
    shapes = O.model_shapes(H, d, dd, cfg["ffn_depth"], cfg["task_num"], cfg["add_features_dim"], cfg["use_bias"])
    w = synth.seeded_weights(shapes, seed)
    model = make_model(cfg, w, dropout=p).train()
    model.dropout_seed = 0xABCDEF0123
    qb = synth.make_queries(seed + 1, Q, Cn, atoms_lo=atoms_lo, atoms_hi=atoms_hi)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    assert rb.n_atoms > 8192 and rb.n_bonds > 8192, (rb.n_atoms, rb.n_bonds)   # the 12-wave geometry for every encoder GEMM
    M = len(qb.p_specs)
    masks = _masks_for(model, model.dropout_seed, rb, pb, M, cfg["add_features_dim"], p)
    scope, targets = qb.scope, torch.tensor(qb.targets)
    assert Fn.StepPlan.enabled and Fn.SplitGemm.enabled and model.dedup_reactants
    Fn.StepPlan.keep_last = True
    try:
        out = model(rb, pb, gpu=0, add_features=qb.add_features)
        hip_gates = _hip_gates(model, rb, masks)
    finally:
        Fn.StepPlan.keep_last, Fn.StepPlan.last = False, None
    head = O.resolve_task_type(cfg["task_num"], cfg["ffn_last_layer"], cfg["task_type"])
    mc = dict(depth=d, diff_depth=dd, ffn_depth=cfg["ffn_depth"], task_type=head, dropout=p)

    # fp64 oracle with the same keep-masks = ground truth; the fp32 oracle's distance to it = the noise floor of fp32
    def run_oracle(dt, gates=None, trace=None):
        P = {k: v.detach().to(dt).requires_grad_(v.requires_grad) for k, v in O.params_from_numpy(w, requires_grad=True).items()}

        def gt(specs):
            g = O.graph_tensors(O.pack_batch(specs, K=4))
            g["f_atoms"], g["f_bonds"] = g["f_atoms"].to(dt), g["f_bonds"].to(dt)
            return g
        mk = {k: v.to(dt) for k, v in masks.items()}
        ref = O.reaction_forward(P, mc, gt(qb.r_specs), gt(qb.p_specs), torch.tensor(qb.add_features).to(dt), masks=mk,
                                 gates=gates, trace=trace)
        t = targets.to(dt)
        l = O.listmle_loss(ref, scope, t) if loss_kind == "mle" else O.evidential_ranking_loss(ref, scope, t)
        names = [k for k in P if P[k].requires_grad]
        g = torch.autograd.grad(l.sum(), [P[k] for k in names], allow_unused=True)
        return ref.detach(), l.detach(), {k: (torch.zeros_like(P[k]) if gi is None else gi) for k, gi in zip(names, g)}
    trace = {}
    ref64, l64, g64 = run_oracle(torch.float64, trace=trace)
    ref32, l32, g32 = run_oracle(torch.float32)

    # ---- 1. which gates differ, and how close to zero their fp64 pre-activation is
    n_flip, worst_z, n_gates = 0, 0.0, 0
    for key, hg in hip_gates.items():
        z = trace[key]
        kept = masks[key].bool() if key in masks else torch.ones_like(hg)
        flips = kept & (hg != (z > 0))
        n_gates += int(kept.sum())
        if bool(flips.any()):
            zf = float(z[flips].abs().max())
            n_flip += int(flips.sum())
            worst_z = max(worst_z, zf)
            log(f"gates {key}: {int(flips.sum())} of {int(kept.sum())} kept gates differ from the fp64 oracle's; largest |fp64 pre-activation| "
                f"among them {zf:.2e}")
            assert zf <= 1e-5, f"{key}: a gate flipped whose fp64 pre-activation is {zf:.3e} from zero"
    log(f"gates: {n_flip} of {n_gates} differ in total; largest |fp64 pre-activation| at a flipped gate {worst_z:.2e}")

    # ---- 3. scores and loss
    e_s, e_s32 = _rel_err(out, ref64), _rel_err(ref32, ref64)
    if loss_kind == "mle":
        l = RL.MLEloss()(out, scope, targets, 0)
    else:
        l = RL.evidential_ranking()(out, scope, targets, None, None, None, 0)
    e_l, e_l32 = _rel_err(l, l64), _rel_err(l32, l64)
    log(f"scores: |err vs fp64| / (1+|ref|) = {e_s:.2e} (fp32 oracle: {e_s32:.2e}); loss: {e_l:.2e} (fp32 oracle: {e_l32:.2e})")
    assert e_s <= max(1e-5, 3.0 * e_s32), ("train-mode scores vs fp64 oracle", e_s, e_s32)
    assert e_l <= max(1e-5, 3.0 * e_l32), ("train-mode loss vs fp64 oracle", e_l, e_l32)

    # ---- 2. gradients: against the fp64 oracle evaluated with the gates the HIP step took
    l.sum().backward()
    got = dict(model.named_parameters())
    fg = {k: v.double() for k, v in hip_gates.items()}
    _, _, g64h = run_oracle(torch.float64, gates=fg)
    _, _, g32h = run_oracle(torch.float32, gates={k: v.float() for k, v in hip_gates.items()})
    for k, gd in g64h.items():
        g = got[k].grad
        g = torch.zeros_like(got[k]) if g is None else g
        gh = g.detach().cpu().double()
        scale = float(gd.abs().max())
        err_nat = float((gh - g64[k]).abs().max())
        err = float((gh - gd).abs().max())
        noise = float((g32h[k].double() - gd).abs().max())
        bound = max(5e-5 * scale + 1e-6, 3.0 * noise)
        if scale < 1e-12:                       # analytically zero (the ranking losses cannot see the output bias)
            log(f"grad {k}: analytically zero (max|g64| {scale:.1e}); max|err| {err:.2e} absolute (fp32 oracle: {noise:.2e})")
        else:
            log(f"grad {k}: max|err| / max|g| = {err / scale:.2e} with the HIP gates dictated to the fp64 oracle "
                f"({err_nat / scale:.2e} against its own gates; fp32 oracle under the same gates: {noise / scale:.2e})")
        assert err <= bound, f"grad {k}: |err vs fp64 (same gates)| {err:.3e} > {bound:.3e} (fp32 oracle noise {noise:.3e}, scale {scale:.3e})"
    # the step really went through the plan with the shared reactant prefix and dropout acted
    model.eval()
    out_eval = model(rb, pb, gpu=0, add_features=qb.add_features)
    assert float((out_eval.detach() - out.detach()).abs().max()) > 1e-4


def test_train_mode_plan_path_above_8192_rows_against_fp64_oracle_h300(parity_log):
    """8 queries x 64 candidates (about 9.2k atoms / 17k bonds per side): train mode, dropout 0.1, step plan, shared
    reactant prefix - scores, ListMLE loss and EVERY parameter gradient against the fp64 oracle with identical masks."""
    cfg = dict(hidden_size=300, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, task_num=1,
               ffn_last_layer="with_softplus", task_type=None, add_features_dim=1)
    _train_step_vs_fp64_oracle(cfg, 8, 64, "mle", 0.1, 16, 24, seed=501, log=parity_log)


def test_train_mode_plan_path_above_8192_rows_against_fp64_oracle_h600_d6(parity_log):
    """BASELINE configs[4] shape (hidden 600, depth 6, evidential_ranking head) at one size above 8192 rows."""
    cfg = dict(hidden_size=600, mpnn_depth=6, mpnn_diff_depth=6, ffn_depth=3, use_bias=True, task_num=2,
               ffn_last_layer="no_softplus", task_type="evidential_ranking", add_features_dim=1)
    _train_step_vs_fp64_oracle(cfg, 8, 64, "evidential", 0.1, 16, 24, seed=601, log=parity_log)


def test_split_path_nonfinite_and_huge_operands_behave_as_documented(gemm_mode):
    """the three-term split keeps every finite f32 operand exactly while bf16(x) is finite, i.e.
    |x| < 3.3962e38 (the midpoint between the largest bf16, 3.3895e38, and 2^128); what differs in kind from the f32
    chain is pinned here - an infinite operand gives NaN (inf - inf in the remainder) where the f32 MFMA gives +-inf, a
    finite |x| at or above that midpoint rounds its first term to inf (so its output row becomes NaN as well), NaN stays
    NaN, and rows without such operands are unaffected."""
    torch.manual_seed(3)
    dev, M, H = "cuda", 9000, 300
    W = torch.randn(H, H, device=dev) / 17
    (w32f, _), (wspf, _) = _packs(W, None)
    x = torch.randn(M, H, device=dev)
    x[10, 7] = float("inf")
    x[11, 8] = float("-inf")
    x[12, 9] = float("nan")
    x[13, 5] = 3.40e38                                   # finite, but bf16(x) = inf (above the rounding midpoint 3.3962e38)
    x[14, 5] = 3.39e38                                   # finite, bf16(x) = the largest bf16: still split exactly
    x[15, 6] = 1e-40                                     # subnormal: flushed or kept, it is far below the result's ulp
    o32 = Fn.linear(M, H, w32f, w_packed=True, a1=x, k1=H)
    osp = Fn.linear(M, H, wspf, w_packed=True, a1=x, k1=H)
    if gemm_mode == "f16x2":
        # two f16 terms: the operand scale follows the TENSOR's largest magnitude, so an infinite (or > 2^110) element leaves
        # no usable scale and every output is NaN - loudly, not a row of plausible zeros; a NaN element (which the magnitude
        # pass skips) poisons its own row only, like everywhere else
        assert torch.isnan(osp).all()
        x[10, 7] = x[11, 8] = 1.0
        x[13, 5] = x[14, 5] = 2.0
        osp = Fn.linear(M, H, wspf, w_packed=True, a1=x, k1=H)
        clean = torch.ones(M, dtype=torch.bool, device=dev)
        clean[12] = False
        assert torch.isnan(osp[12]).all() and torch.isfinite(osp[clean]).all()
        ref = x[clean].double() @ W.double().t()
        den = x[clean].double().abs() @ W.double().abs().t() + 1e-300
        assert float(((osp[clean].double() - ref).abs() / den).max()) < 2e-6
        return
    assert torch.isinf(o32[10]).all() and torch.isinf(o32[11]).all()            # the f32 chain: inf * w = +-inf
    assert torch.isnan(osp[10]).all() and torch.isnan(osp[11]).all()            # the split path: NaN
    assert torch.isnan(o32[12]).all() and torch.isnan(osp[12]).all()
    assert torch.isfinite(o32[13]).all() and torch.isnan(osp[13]).all()
    clean = torch.ones(M, dtype=torch.bool, device=dev)
    clean[10:14] = False
    assert torch.isfinite(osp[clean]).all()
    ref = x[clean].double() @ W.double().t()
    den = x[clean].double().abs() @ W.double().abs().t() + 1e-300
    assert float(((osp[clean].double() - ref).abs() / den).max()) < 2e-6
    assert float(((osp[14].double() - x[14].double() @ W.double().t()).abs() / (x[14].double().abs() @ W.double().abs().t())).max()) < 2e-6


@pytest.mark.parametrize("M", [49153, 71425, 138881])
def test_persistent_split_gemm_is_bit_identical_to_one_block_per_workgroup(M):
    """linear_split_kernel<19,19,0,12,0> runs persistent when there are more 192-row blocks than CUs (one workgroup per CU
    walks blocks b, b + #CUs, ...; the k-loop's operand / weight pipeline continues across the block boundary, the next
    block's step-0 chunks travel through LDS).  Same products in the same order per element: every output - C, the second
    pre-activation output, the sign bits, the weighted column-sum partials - must equal the one-block-per-workgroup launch
    (RR_NO_PERSIST) bit for bit, for the plain dX form, a forward form with bias / ReLU / dropout / side outputs, and the
    two-segment operand of W_o; and stay within the f64 bounds of the other geometry tests."""
    import os
    torch.manual_seed(M)
    dev, H = "cuda", 300
    W = torch.randn(H, H, device=dev) / 17
    b = torch.randn(H, device=dev)
    x = torch.randn(M, H, device=dev)
    cw = torch.rand(M, device=dev)
    rowb = int(lib().rr_mask_bits_row_bytes(H))
    Wo = torch.randn(H, 61 + H, device=dev) / 19
    fa = torch.randn(M, 64, device=dev)[:, :61]
    small = torch.randn(5000, H, device=dev)
    gidx = torch.randint(-1, 5000, (M,), device=dev, dtype=torch.int32)

    def run():
        L = Fn.LinW(W, b)
        out = {}
        o, p = Fn.linear(M, H, L.pk_t(0, H), w_packed=True, a1=x, k1=H, colsum_w=cw)               # dX: plain + column sums
        out["dx"], out["dx_colsum"] = o, p
        bits = torch.zeros(M, rowb, dtype=torch.uint8, device=dev)
        pre = torch.empty(M, H, device=dev)
        out["fwd"] = Fn.linear(M, H, L.pk(H), w_packed=True, a1=x, k1=H, bias=b, act=Fn.ACT_RELU, drop_p=0.1, seed=9,
                               c_pre=pre, mask_bits_out=bits)
        out["fwd_pre"], out["fwd_bits"] = pre, bits
        Lo = Fn.LinW(Wo, b)
        out["wo"] = Fn.linear(M, H, Lo.pk(61, H), w_packed=True, a1=fa, k1=61, a2=x, k2=H, bias=b, act=Fn.ACT_RELU)   # 2 + 10 k-steps
        out["gathered"] = Fn.linear(M, H, L.pk(H), w_packed=True, a1=small, k1=H, a1_idx=gidx, bias=b)   # rows through an index, -1 = no row
        torch.cuda.synchronize()
        return out
    assert "RR_NO_PERSIST" not in os.environ
    got = run()
    os.environ["RR_NO_PERSIST"] = "1"
    try:
        ref = run()
    finally:
        del os.environ["RR_NO_PERSIST"]
    for k in got:
        assert torch.equal(got[k], ref[k]), k
    r64 = x.double() @ W.double()
    den = x.double().abs() @ W.double().abs() + 1e-300
    err = float(((got["dx"].double() - r64).abs() / den).max())
    Hh.record(f"persistent dX GEMM M={M}: max err / sum|ab|", err, 2e-6)
    assert err <= 2e-6
    z = torch.zeros(1, device=dev)
    A = torch.where(gidx[:, None] >= 0, small[gidx.clamp(min=0).long()], z)
    rg = A.double() @ W.double().t() + b.double()
    dg = A.double().abs() @ W.double().abs().t() + b.double().abs() + 1e-300
    assert float(((got["gathered"].double() - rg).abs() / dg).max()) <= 2e-6
