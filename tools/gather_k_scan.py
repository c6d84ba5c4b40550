"""What bounds the bond <- bonds gather: time against the number of source rows per destination (K = 1..4, every entry used, sources
inside the destination's molecule) next to a plain copy of the same tensor.  If a launch costs the copy's time plus a fixed amount per
extra source, the extra reads (L2 hits: the rows are shared by neighbouring destinations) are what it pays for.
Usage: python tools/gather_k_scan.py [H]"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reactranker_amd import functions as Fn
dev = "cuda"
torch.manual_seed(0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 300
NB = 138881


def t(fn, n=30, reps=5):
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


src = torch.randn(NB, H, device=dev)
out = torch.empty(NB, H, device=dev)
mb = 2 * NB * H * 4 / 1e6
us = t(lambda: out.copy_(src))
print(f"torch copy                         {us:6.1f} us  {mb / us:5.2f} TB/s")
us = t(lambda: Fn.axpby(1.0, src, 0.0, None, out=out))
print(f"rr_axpby (alpha a, one source)     {us:6.1f} us  {mb / us:5.2f} TB/s")
base = torch.arange(NB, device=dev)
for span in (34, 1):
    for K in (1, 2, 3, 4):
        if span == 1:
            idx = base[:, None].repeat(1, K)               # every source = the destination's own row (K reads of one line set)
        else:
            idx = ((base[:, None] // span) * span + torch.randint(0, span, (NB, K), device=dev)).clamp(max=NB - 1)
        idx = idx.to(torch.int32).contiguous()
        us = t(lambda: Fn.gather_sum(src, idx, H, out=out))
        print(f"gather K {K}, sources within {span:2d} rows  {us:6.1f} us  {(mb + NB * K * 4 / 1e6) / us:5.2f} TB/s algorithmic")
