"""Two-term f16 split GEMM (w_packed = 3: 3 MFMAs per k-step) next to the three-term bf16 one (w_packed = 2: 6 MFMAs) and the
exact-f32 MFMA GEMM on the same inputs: error against f64 and time, for the plain, masked-dX and gathered forward forms.
The operand bounds (magnitude slots) are found once outside the timing loops, as the step plans get them from the producing kernels.
Usage (GPU box): python tools/f16_gemm_bench.py        (RR_LIB_PATH=build/variants/lib_X.so times a variant build)"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import functions as Fn
from reactranker_amd._lib import lib, PackDesc, check, ptr, stream
dev = "cuda"
torch.manual_seed(0)

def pack_split(w, transpose, rows, c0, k1, k2, kind):
    nb = int(lib().rr_split_weight_bytes(rows, k1, k2))
    dst = torch.empty(nb, dtype=torch.uint8, device=dev)
    d = (PackDesc * 1)()
    d[0].src, d[0].ld_src, d[0].transpose, d[0].rows, d[0].c0, d[0].k1, d[0].k2 = w.data_ptr(), w.stride(0), transpose, rows, c0, k1, k2
    d[0].dst, d[0].split = dst.data_ptr(), kind
    check(lib().rr_pack_weights_f32(d, 1, stream()), "pack")
    if kind == 2:
        dst._rr_f16 = True
    return dst

def t(fn, n=30):
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
for M in (71425, 138881):
    W = torch.randn(H, H, device=dev) / 17
    b = torch.randn(H, device=dev)
    x = torch.randn(M, H, device=dev)
    wb, wh = pack_split(W, 0, H, 0, H, 0, 1), pack_split(W, 0, H, 0, H, 0, 2)
    Fn.SplitGemm.enabled = False
    wf = Fn.LinW(W, None).pk(H)
    Fn.SplitGemm.enabled = True
    ref = x.double() @ W.double().t(); den = x.double().abs() @ W.double().abs().t() + 1e-30
    o = [torch.empty(M, H, device=dev) for _ in range(3)]
    am = [Fn.amax(x), None, None]                       # operand bounds, found once (the step plans get them from the producers)
    Fn.linear(M, H, wf, w_packed=1, a1=x, k1=H, out=o[0]); Fn.linear(M, H, wb, a1=x, k1=H, out=o[1]); Fn.linear(M, H, wh, a1=x, k1=H, out=o[2], amax_of=am)
    print(f"M {M} plain   f32: {err(o[0], ref, den)}   bf16x3: {err(o[1], ref, den)}   f16x2: {err(o[2], ref, den)}", flush=True)
    u = [t(lambda: Fn.linear(M, H, wf, w_packed=1, a1=x, k1=H, out=o[0])), t(lambda: Fn.linear(M, H, wb, a1=x, k1=H, out=o[1])),
         t(lambda: Fn.linear(M, H, wh, a1=x, k1=H, out=o[2], amax_of=am))]
    print(f"      time f32 {u[0]:.1f}   bf16x3 {u[1]:.1f}   f16x2 {u[2]:.1f} us", flush=True)
    # masked dX with dz side output and column sums
    dy = torch.randn(M, H, device=dev); y = torch.relu(torch.randn(M, H, device=dev)); cw = torch.rand(M, device=dev)
    wbt, wht = pack_split(W, 1, H, 0, H, 0, 1), pack_split(W, 1, H, 0, H, 0, 2)
    dz = torch.empty(M, H, device=dev)
    dzr = torch.where(y > 0, dy * 1.1, torch.zeros_like(dy))
    ref = dzr.double() @ W.double(); den = dzr.double().abs() @ W.double().abs() + 1e-30
    kw = dict(a1=dy, k1=H, a_mask=y, mask_scale=1.1, dz_out=dz, colsum_w=cw)
    am = [Fn.amax(dy), None, None]
    Fn.linear(M, H, wbt, out=o[1], **kw); Fn.linear(M, H, wht, out=o[2], amax_of=am, **kw)
    print(f"M {M} masked  bf16x3: {err(o[1], ref, den)}   f16x2: {err(o[2], ref, den)}", flush=True)
    u = [t(lambda: Fn.linear(M, H, wbt, out=o[1], **kw)), t(lambda: Fn.linear(M, H, wht, out=o[2], amax_of=am, **kw))]
    print(f"      time bf16x3 {u[0]:.1f}   f16x2 {u[1]:.1f} us", flush=True)
    # gathered forward: A = am[b2a] - msg[rev], bias, residual, relu
    nA = M // 2 + 3
    am_ = torch.randn(nA, H, device=dev); msg = torch.relu(torch.randn(M, H, device=dev)); inp = torch.randn(M, H, device=dev)
    b2a = torch.randint(0, nA, (M,), device=dev, dtype=torch.int32); rev = torch.randint(0, M, (M,), device=dev, dtype=torch.int32)
    kw = dict(a1=am_, k1=H, a1_idx=b2a, a1_sub=msg, a1_sub_idx=rev, bias=b, residual=inp, act=Fn.ACT_RELU)
    A = am_[b2a.long()] - msg[rev.long()]
    ref = torch.relu(A.double() @ W.double().t() + b.double() + inp.double()); den = A.double().abs() @ W.double().abs().t() + 1
    am = [Fn.amax(am_), Fn.amax(msg), None]
    Fn.linear(M, H, wb, out=o[1], **kw); Fn.linear(M, H, wh, out=o[2], amax_of=am, **kw)
    print(f"M {M} gather  bf16x3: {err(o[1], ref, den)}   f16x2: {err(o[2], ref, den)}", flush=True)
    u = [t(lambda: Fn.linear(M, H, wb, out=o[1], **kw)), t(lambda: Fn.linear(M, H, wh, out=o[2], amax_of=am, **kw))]
    print(f"      time bf16x3 {u[0]:.1f}   f16x2 {u[1]:.1f} us", flush=True)
