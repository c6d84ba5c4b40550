"""Isolated timing of rr_ffn_chain_f32 (forward chain) against the per-layer launches: us per launch, back-to-back.
Usage: python tools/ffn_chain_bench.py"""
import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import functions as Fn
from tests.test_gpu_ffn import _chain_forward, _layers
dev = "cuda"


def t(fn, n=50, reps=5):
    for _ in range(10):
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


for M, widths in ((4096, [301, 300, 300, 1]), (2048, [301, 300, 300, 1]), (4096, [301, 300, 1]), (4096, [301, 300, 300, 300, 1]),
                  (4096, [601, 600, 600, 2]), (16384, [301, 300, 300, 1])):
    K0 = widths[0]
    x = torch.randn(M, (K0 + 3) // 4 * 4, device=dev)
    layers = _layers(widths, True, 3)
    for L in layers:
        L.pk(L.w.shape[1])
    us_chain = t(lambda: _chain_forward(x, K0, layers, 0.1, 91))
    us_layers = t(lambda: Fn.ffn_forward(x[:, :K0], layers, 0.1, 91, 0))
    print(f"M {M:6d} widths {widths}: chain {us_chain:6.1f} us   per-layer launches {us_layers:6.1f} us", flush=True)
