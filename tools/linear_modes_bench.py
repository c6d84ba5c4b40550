"""Isolated timings of the split GEMM launches a BASELINE configs[2] training step makes (M = 138,881 bond rows / 71,425
atom rows, H = 300), for same-box A/B of kernel variants: RR_LIB_PATH=build/variants/lib_X.so python tools/linear_modes_bench.py
Prints one line per launch form: microseconds per launch (median of 5 x 20 back-to-back launches)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reactranker_amd import functions as Fn
from reactranker_amd._lib import lib
dev = "cuda"
torch.manual_seed(0)
H = int(os.environ.get("RR_BENCH_H", "300"))      # 600: BASELINE configs[4]


def t(fn, n=20, reps=5):
    for _ in range(5):
        fn()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(out)


res = {}
for M in (138881, 71425):
    W = torch.randn(H, H, device=dev) / 17
    b = torch.randn(H, device=dev)
    L = Fn.LinW(W, b)
    wf, wt = L.pk(H), L.pk_t(0, H)
    x = torch.randn(M, H, device=dev)
    res_ = torch.randn(M, H, device=dev)
    cw = torch.rand(M, device=dev)
    out = torch.empty(M, H, device=dev)
    dz = torch.empty(M, H, device=dev)
    bits = torch.zeros(M, int(lib().rr_mask_bits_row_bytes(H)), dtype=torch.uint8, device=dev)
    y = Fn.linear(M, H, wf, w_packed=True, a1=x, k1=H, bias=b, residual=res_, act=Fn.ACT_RELU, drop_p=0.1, seed=3, mask_bits_out=bits)
    nA = M // 2 + 3
    am = torch.randn(nA, H, device=dev)
    # molecule-local gather indices, like the packer's tables (rows of one molecule are neighbours in memory)
    base = torch.arange(M, device=dev)
    b2a = ((base // 34) * 17 + torch.randint(0, 17, (M,), device=dev)).clamp(max=nA - 1).to(torch.int32)
    rev = ((base // 34) * 34 + torch.randint(0, 34, (M,), device=dev)).clamp(max=M - 1).to(torch.int32)
    forms = {
        "dX plain + colsum (mode 0)": lambda: Fn.linear(M, H, wt, w_packed=True, a1=x, k1=H, out=out, colsum_w=cw),
        "dX sign-bit mask + dZ + colsum (mode 3)": lambda: Fn.linear(M, H, wt, w_packed=True, a1=x, k1=H, a_mask_bits=bits, mask_scale=1.1, out=out, dz_out=dz, colsum_w=cw),
        "fwd gathered - gathered, bias, residual, relu, dropout, bits (mode 1)": lambda: Fn.linear(
            M, H, wf, w_packed=True, a1=am, k1=H, a1_idx=b2a, a1_sub=y, a1_sub_idx=rev, bias=b, residual=res_, act=Fn.ACT_RELU,
            drop_p=0.1, seed=5, out=out, mask_bits_out=bits),
        "fwd plain, bias, residual, relu, dropout, bits (mode 0)": lambda: Fn.linear(
            M, H, wf, w_packed=True, a1=x, k1=H, bias=b, residual=res_, act=Fn.ACT_RELU, drop_p=0.1, seed=5, out=out, mask_bits_out=bits),
    }
    # the gather in front of the forward W_h GEMM: per atom (what the step does: a_message, then the GEMM gathers a_message[b2a] -
    # message[rev] in its loader, mode 1) against per bond (the GEMM's operand materialised: bond-to-bond table, K = 3; the GEMM is mode 0)
    a2b = ((torch.arange(nA, device=dev)[:, None] // 17) * 34 + torch.randint(0, 34, (nA, 4), device=dev)).clamp(max=M - 1).to(torch.int32)
    b2b = ((base[:, None] // 34) * 34 + torch.randint(0, 34, (M, 3), device=dev)).clamp(max=M - 1).to(torch.int32)
    forms["gather per atom (K 4): a_message"] = lambda: Fn.gather_sum(y, a2b, H)
    forms["gather per bond (K 3): the W_h GEMM's operand"] = lambda: Fn.gather_sum(y, b2b, H)
    if M == 138881:      # W_i-like: K = 83 (ld 84), two outputs (pre-activation + relu), sign bits
        fb = torch.randn(M, 84, device=dev); fb[:, 83] = 0
        Wi = Fn.LinW(torch.randn(H, 83, device=dev) / 9, b)
        pre = torch.empty(M, H, device=dev)
        forms["fwd W_i: K 83, bias, relu, c_pre, bits (mode 0)"] = lambda: Fn.linear(
            M, H, Wi.pk(83), w_packed=True, a1=fb, k1=83, bias=b, act=Fn.ACT_RELU, out=out, c_pre=pre, mask_bits_out=bits)
    else:                # W_o-like: two segments (61 | 300), bias, relu, dropout, sign bits, no residual
        fa = torch.randn(M, 64, device=dev); fa[:, 61:] = 0
        Wo = Fn.LinW(torch.randn(H, 361, device=dev) / 19, b)
        forms["fwd W_o: K 61 | 300, bias, relu, dropout, bits (mode 0)"] = lambda: Fn.linear(
            M, H, Wo.pk(61, H), w_packed=True, a1=fa, k1=61, a2=x, k2=H, bias=b, act=Fn.ACT_RELU, drop_p=0.1, seed=7, out=out, mask_bits_out=bits)
    for name, fn in forms.items():
        us = t(fn)
        res[(M, name)] = us
        print(f"M {M:6d}  {us:7.1f} us  {name}", flush=True)
print("sum", round(sum(res.values()), 1))
