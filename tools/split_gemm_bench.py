"""Split (three bf16 terms) GEMM next to the exact-f32 MFMA GEMM on the same inputs: error against f64 and time for the
three operand forms (masked dX + dZ side output + column sums, gathered forward, two-segment forward).
Usage (GPU box): python tools/split_gemm_bench.py [quick]"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import functions as Fn
from reactranker_amd._lib import lib, PackDesc, check, ptr, stream
dev = "cuda"
torch.manual_seed(0)

def pack_split(w, transpose, rows, c0, k1, k2):
    nb = int(lib().rr_split_weight_bytes(rows, k1, k2))
    dst = torch.empty(nb, dtype=torch.uint8, device=dev)
    d = (PackDesc * 1)()
    d[0].src, d[0].ld_src, d[0].transpose, d[0].rows, d[0].c0, d[0].k1, d[0].k2 = w.data_ptr(), w.stride(0), transpose, rows, c0, k1, k2
    d[0].dst, d[0].split = dst.data_ptr(), 1
    check(lib().rr_pack_weights_f32(d, 1, stream()), "pack")
    return dst

def f32_pack(fn):
    """the exact-f32 MFMA layout of a weight (LinW packs the bf16-term images by default)"""
    Fn.SplitGemm.enabled = False
    try:
        return fn()
    finally:
        Fn.SplitGemm.enabled = True

def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def err(out, ref, den):
    e = (out.double() - ref).abs() / den
    return f"max {e.max().item():.2e} mean {e.mean().item():.2e}"

H = 300
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
for M in ((1000, 4133) if quick else (4133, 138881, 71425)):
    W = torch.randn(H, H, device=dev) / 17
    b = torch.randn(H, device=dev)
    # ---- mode 2: dX = (dy * relu mask) W, dz side output, colsum
    dy = torch.randn(M, H, device=dev); y = torch.relu(torch.randn(M, H, device=dev)); cw = torch.rand(M, device=dev)
    Wt = Fn.LinW(W, None)
    wp = f32_pack(lambda: Wt.pk_t(0, H)); ws = pack_split(W, 1, H, 0, H, 0)
    o1 = torch.empty(M, H, device=dev); o2 = torch.empty(M, H, device=dev); dz1 = torch.empty(M, H, device=dev); dz2 = torch.empty(M, H, device=dev)
    _, p1 = Fn.linear(M, H, wp, w_packed=1, a1=dy, k1=H, a_mask=y, mask_scale=1.1, out=o1, dz_out=dz1, colsum_w=cw)
    _, p2 = Fn.linear(M, H, ws, w_packed=2, ldw=0, a1=dy, k1=H, a_mask=y, mask_scale=1.1, out=o2, dz_out=dz2, colsum_w=cw)
    dzr = torch.where(y > 0, dy * 1.1, torch.zeros_like(dy))
    ref = dzr.double() @ W.double(); den = dzr.double().abs() @ W.double().abs() + 1e-30
    print(f"M {M} mode2  f32: {err(o1, ref, den)}   split: {err(o2, ref, den)}   dz equal {torch.equal(dz1, dz2)}  colsum maxdiff {(p1.sum(0) - p2.sum(0)).abs().max().item():.2e} / {p1.sum(0).abs().max().item():.2e}", flush=True)
    if not quick:
        u1 = t(lambda: Fn.linear(M, H, wp, w_packed=1, a1=dy, k1=H, a_mask=y, mask_scale=1.1, out=o1, dz_out=dz1, colsum_w=cw))
        u2 = t(lambda: Fn.linear(M, H, ws, w_packed=2, ldw=0, a1=dy, k1=H, a_mask=y, mask_scale=1.1, out=o2, dz_out=dz2, colsum_w=cw))
        print(f"      time f32 {u1:.1f} us   split {u2:.1f} us   ({2.0*M*H*H/u2/1e6:.1f} TF f32-equivalent)", flush=True)
    # ---- mode 1: gathered A minus gathered sub, bias, residual, relu, dropout
    nA = M // 2 + 3
    am = torch.randn(nA, H, device=dev); msg = torch.relu(torch.randn(M, H, device=dev)); inp = torch.randn(M, H, device=dev)
    b2a = torch.randint(0, nA, (M,), device=dev, dtype=torch.int32); rev = torch.randint(0, M, (M,), device=dev, dtype=torch.int32)
    b2a[0] = -1; rev[0] = -1
    Wl = Fn.LinW(W, b)
    wp = f32_pack(lambda: Wl.pk(H)); ws = pack_split(W, 0, H, 0, H, 0)
    kw = dict(a1=am, k1=H, a1_idx=b2a, a1_sub=msg, a1_sub_idx=rev, bias=b, residual=inp, act=Fn.ACT_RELU, drop_p=0.1, seed=1234)
    o1 = Fn.linear(M, H, wp, w_packed=1, **kw); o2 = Fn.linear(M, H, ws, w_packed=2, ldw=0, **kw)
    z = torch.zeros(1, device=dev)
    A = torch.where(b2a[:, None] >= 0, am[b2a.clamp(min=0).long()], z) - torch.where(rev[:, None] >= 0, msg[rev.clamp(min=0).long()], z)
    pre = A.double() @ W.double().t() + b.double() + inp.double(); den = A.double().abs() @ W.double().abs().t() + 1
    keep = (o1 != 0) | (o2 != 0)
    ref = torch.where(keep, torch.relu(pre) / 0.9, torch.zeros_like(pre))
    print(f"M {M} mode1  f32: {err(o1, ref, den)}   split: {err(o2, ref, den)}   zero pattern differs at {((o1 == 0) != (o2 == 0)).sum().item()} of {o1.numel()}", flush=True)
    if not quick:
        u1 = t(lambda: Fn.linear(M, H, wp, w_packed=1, **kw)); u2 = t(lambda: Fn.linear(M, H, ws, w_packed=2, ldw=0, **kw))
        print(f"      time f32 {u1:.1f} us   split {u2:.1f} us", flush=True)
    # ---- mode 0: two segments (133 | 300), c_pre
    Wo = torch.randn(H, 133 + H, device=dev) / 20
    fa = torch.zeros(M, 136, device=dev); fa[:, :133] = (torch.rand(M, 133, device=dev) < 0.1).float(); a2 = torch.randn(M, H, device=dev)
    Wl = Fn.LinW(Wo, b)
    wp = f32_pack(lambda: Wl.pk(133, H)); ws = pack_split(Wo, 0, H, 0, 133, H)
    pre1 = torch.empty(M, H, device=dev); pre2 = torch.empty(M, H, device=dev)
    kw = dict(a1=fa, k1=133, a2=a2, k2=H, bias=b, act=Fn.ACT_RELU)
    o1 = Fn.linear(M, H, wp, w_packed=1, c_pre=pre1, **kw); o2 = Fn.linear(M, H, ws, w_packed=2, ldw=0, c_pre=pre2, **kw)
    A = torch.cat([fa[:, :133], a2], 1)
    ref = A.double() @ Wo.double().t() + b.double(); den = A.double().abs() @ Wo.double().abs().t() + 1
    print(f"M {M} mode0  f32: {err(pre1, ref, den)}   split: {err(pre2, ref, den)}   relu out: {err(o2, torch.relu(ref), den)}", flush=True)
    if not quick:
        u1 = t(lambda: Fn.linear(M, H, wp, w_packed=1, **kw)); u2 = t(lambda: Fn.linear(M, H, ws, w_packed=2, ldw=0, **kw))
        print(f"      time f32 {u1:.1f} us   split {u2:.1f} us", flush=True)
