import torch, time, sys, os
sys.path.insert(0, os.getcwd())
from reactranker_amd import functions as Fn
x = torch.randn(4096, 300, device="cuda"); y = torch.randn(4096, 300, device="cuda"); o = torch.empty_like(x)
big = torch.randn(138881, 300, device="cuda"); big2 = torch.empty_like(big)
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("axpby 4096x300 (4.9 MB x3) back-to-back: %.1f us per launch" % t(lambda: Fn.axpby(1.0, x, 1.0, y, out=o)))
tiny = torch.randn(64, device="cuda"); tiny2 = torch.randn(64, device="cuda"); to = torch.empty_like(tiny)
print("axpby 64 floats back-to-back: %.1f us per launch" % t(lambda: Fn.axpby(1.0, tiny.view(1, 64), 1.0, tiny2.view(1, 64), out=to.view(1, 64))))
print("torch add tiny: %.1f us" % t(lambda: torch.add(tiny, tiny2, out=to)))
print("axpby 138881x300 (167 MB x3): %.1f us" % t(lambda: Fn.axpby(1.0, big, 1.0, big, out=big2), 50))
# alternate big write then tiny kernel: boundary cost after a kernel that dirtied a lot of L2
def alt():
    Fn.axpby(1.0, big, 1.0, big, out=big2)
    Fn.axpby(1.0, tiny.view(1, 64), 1.0, tiny2.view(1, 64), out=to.view(1, 64))
print("big + tiny pair: %.1f us" % t(alt, 50))
