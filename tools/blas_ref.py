"""Measurement only: what the vendor fp32 GEMM reaches on the hot shapes (ceiling check, not used by the product)."""
import torch
nB, H = 138881, 300
a = torch.randn(nB, H, device="cuda"); w = torch.randn(H, H, device="cuda") / 17
a6 = torch.randn(nB, 2 * H, device="cuda"); w6 = torch.randn(H, 2 * H, device="cuda")
dy = torch.randn(nB, H, device="cuda")
def t(fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * nB * H * H
for name, fn, f in (("mm NT k300", lambda: torch.mm(a, w.t()), fl), ("mm NT k600", lambda: torch.mm(a6, w6.t()), 2 * fl),
                    ("mm TN wgrad", lambda: torch.mm(dy.t(), a), fl), ("addmm", lambda: torch.addmm(dy, a, w.t()), fl)):
    us = t(fn)
    print(f"{name:14s} {us:8.1f} us  {f / us / 1e6:6.1f} TF")
