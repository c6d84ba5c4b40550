import sys, os, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import functions as Fn
nB, H = 138881, 300
dev = "cuda"
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * nB * H * H
for name, mk in (("random", lambda *s: torch.randn(*s, device=dev)), ("zeros", lambda *s: torch.zeros(*s, device=dev)),
                 ("random", lambda *s: torch.randn(*s, device=dev))):
    msg = mk(nB, H); W = Fn.LinW(mk(H, H) / 17, mk(H)); out = torch.empty(nB, H, device=dev)
    dy = mk(nB, H); y = torch.relu(mk(nB, H)) if name == "random" else torch.ones(nB, H, device=dev)
    dw = torch.empty(H, H, device=dev); db = torch.empty(H, device=dev)
    us = t(lambda: Fn.linear(nB, H, W.pk(H), w_packed=True, a1=msg, k1=H, out=out))
    us2 = t(lambda: Fn.linear(nB, H, W.pk_t(0, H), w_packed=True, a1=dy, k1=H, a_mask=y, mask_scale=1.1, out=out))
    us3 = t(lambda: Fn.wgrad(nB, H, dy, dw, dbias=db, x1=msg, k1=H))
    print(f"{name}: lin_m0 {us:.1f} us {fl/us/1e6:.1f} TF | lin_m2 {us2:.1f} us {fl/us2/1e6:.1f} TF | wgrad {us3:.1f} us {fl/us3/1e6:.1f} TF")
