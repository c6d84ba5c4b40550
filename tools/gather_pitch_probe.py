"""Stand-alone timing of the step's two gather shapes with activation rows of 1200 bytes (pitch 300 floats: a row straddles
128-byte lines) against 1280 bytes (pitch 320: no row straddles a line, +6.7 % bytes): DESIGN.md section 9 item 4.
    python tools/gather_pitch_probe.py"""
import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import functions as Fn
dev = "cuda"
torch.manual_seed(0)
H, NB, NA = 300, 138881, 71425


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


def local_idx(n_out, n_src, K, per_mol_out, per_mol_src, fill):
    base = torch.arange(n_out, device=dev) // per_mol_out
    idx = (base[:, None] * per_mol_src + torch.randint(0, per_mol_src, (n_out, K), device=dev)).clamp(max=n_src - 1)
    idx[torch.rand(n_out, K, device=dev) > fill] = -1
    return idx.to(torch.int32)


b2b = local_idx(NB, NB, 3, 34, 34, 0.7)
a2b = local_idx(NA, NB, 4, 17, 34, 0.5)
for pitch in (300, 320, 304):
    def buf(n):
        return torch.randn(n, pitch, device=dev)[:, :H]
    bond, out_b, out_a = buf(NB), buf(NB), buf(NA)
    adds = [buf(NB), buf(NB)]
    forms = {
        "atom <- bonds (a2b, K 4), plain": (lambda: Fn.gather_sum(bond, a2b, H, out=out_a), 4 * (NB * H + NA * H + 4 * NA)),
        "bond <- bonds (b2b_t, K 3), plain": (lambda: Fn.gather_sum(bond, b2b, H, out=out_b), 4 * (2 * NB * H + 3 * NB)),
        "bond <- bonds (b2b_t, K 3), 2 addends": (lambda: Fn.gather_sum(bond, b2b, H, out=out_b, adds=adds), 4 * (4 * NB * H + 3 * NB)),
    }
    for name, (fn, by) in forms.items():
        us = t(fn)
        print(f"pitch {pitch * 4:5d} B  {name:42s} {us:7.1f} us   {by / us / 1e6:5.2f} TB/s of algorithmic bytes", flush=True)
