import sys, os, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import functions as Fn
H = 300; dev = "cuda"
W = Fn.LinW(torch.randn(H, H, device=dev) / 17, None)
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (49152*2, 49152*5//2, 133120, 138881, 144000, 49152*3, 49152*3+64*64, 49152*4, 138881, 49152*3):
    dy = torch.randn(M, H, device=dev); y = torch.relu(torch.randn(M, H, device=dev)); out = torch.empty(M, H, device=dev)
    us = t(lambda: Fn.linear(M, H, W.pk_t(0, H), w_packed=True, a1=dy, k1=H, a_mask=y, mask_scale=1.1, out=out))
    wgs = (M + 63) // 64
    print(f"M {M:7d}  wgs {wgs:5d}  rounds {wgs/768:5.2f}  {us:7.1f} us  {2.0*M*H*H/us/1e6:6.1f} TF   us per 1000 rows {us/M*1e3:6.3f}")
