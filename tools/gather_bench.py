"""Isolated timings of the gather launches a BASELINE configs[2] training step makes (139k bond rows / 71k atom rows, H = 300,
molecule-local index tables like the packer's), for same-box A/B of kernel variants:
    RR_LIB_PATH=build/variants/lib_X.so python tools/gather_bench.py
Prints microseconds per launch (median of 5 x 30 back-to-back launches) and algorithmic TB/s."""
import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import functions as Fn
from reactranker_amd._lib import lib
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
    """[n_out, K] int32: sources inside the destination's own molecule (rows of a molecule are neighbours), `fill` of
    the K entries used on average, the rest -1."""
    base = torch.arange(n_out, device=dev) // per_mol_out
    idx = (base[:, None] * per_mol_src + torch.randint(0, per_mol_src, (n_out, K), device=dev)).clamp(max=n_src - 1)
    drop = torch.rand(n_out, K, device=dev) > fill
    idx[drop] = -1
    return idx.to(torch.int32)


bond = torch.randn(NB, H, device=dev)
atom = torch.randn(NA, H, device=dev)
y = torch.relu(torch.randn(NB, H, device=dev))
bits = torch.zeros(NB, int(lib().rr_mask_bits_row_bytes(H)), dtype=torch.uint8, device=dev)
Fn.linear(NB, H, Fn.LinW(torch.eye(H, device=dev), None).pk(H), w_packed=True, a1=y, k1=H, out=torch.empty_like(y), mask_bits_out=bits)
y._rr_bits = bits
adds = [torch.randn(NB, H, device=dev) for _ in range(2)]
b2b = local_idx(NB, NB, 3, 34, 34, 0.7)          # bond-to-bond table: K - 1 = 3 entries, ~2 used
a2b = local_idx(NA, NB, 4, 17, 34, 0.5)          # incoming bonds of an atom
a2a = local_idx(NA, NA, 4, 17, 17, 0.5)
out_b, out_a = torch.empty(NB, H, device=dev), torch.empty(NA, H, device=dev)
forms = {
    "bond <- bonds (b2b_t, K 3), sign-bit mask": (lambda: Fn.gather_sum(bond, b2b, H, out=out_b, mask=y, mask_scale=1.1), 4 * (2 * NB * H + 3 * NB) + NB * 40),
    "bond <- bonds (b2b_t, K 3), sign-bit mask + 2 addends": (lambda: Fn.gather_sum(bond, b2b, H, out=out_b, mask=y, mask_scale=1.1, adds=adds), 4 * (4 * NB * H + 3 * NB) + NB * 40),
    "bond <- bonds (b2b_t, K 3), plain": (lambda: Fn.gather_sum(bond, b2b, H, out=out_b), 4 * (2 * NB * H + 3 * NB)),
    "atom <- bonds (a2b, K 4)": (lambda: Fn.gather_sum(bond, a2b, H, out=out_a), 4 * (NB * H + NA * H + 4 * NA)),
    "atom <- atoms (a2a, K 4)": (lambda: Fn.gather_sum(atom, a2a, H, out=out_a), 4 * (2 * NA * H + 4 * NA)),
    "atom <- atoms (a2a, K 4), f32 mask": (lambda: Fn.gather_sum(atom, a2a, H, out=out_a, mask=y[:NA], mask_scale=1.1), 4 * (3 * NA * H + 4 * NA)),
}
# readout and the shared-prefix kernels
class _G:
    pass


g = _G()
g.M, g.nA = 4096, NA
sizes = torch.full((g.M,), NA // g.M, dtype=torch.int32)
sizes[: (NA - 1) - int(sizes.sum())] += 1
starts = 1 + torch.cumsum(sizes, 0) - sizes
g.a_scope = torch.stack([starts, sizes], 1).to(torch.int32).contiguous().to(dev)
a2m = torch.full((NA,), -1, dtype=torch.int32)
a2m[1:1 + int(sizes.sum())] = torch.repeat_interleave(torch.arange(g.M, dtype=torch.int32), sizes.long())
g.atom2mol = a2m.to(dev)
feat = torch.rand(g.M, 1, device=dev)
dvec = torch.randn(g.M, 304, device=dev)[:, :301]
ya = y[:NA]
ya._rr_bits = bits[:NA]
copies = (torch.arange(NB, device=dev) // 34 // 64 * 34 + torch.arange(NB, device=dev) % 34).clamp(max=2200).to(torch.int32)
small = torch.randn(2201, H, device=dev)
forms.update({
    "readout: segment mean over ~17 atoms + feature, dropout": (lambda: Fn.segment_mean_fwd(atom, g, H, feat, 1, 0.1, 5), 4 * (NA * H + g.M * 301)),
    "readout adjoint, sign-bit mask": (lambda: Fn.segment_mean_bwd(dvec, g, H, 1, 0.1, 5, mask=ya, mask_scale=1.1), 4 * (NA * H + g.M * 301) + NA * 40),
    "bond <- distinct bond (1 source), dropout": (lambda: Fn.gather_dropout(small, copies, H, 0.1, 7), 4 * (NB * H + NB)),
})
tot = 0.0
for name, (fn, nbytes) in forms.items():
    us = t(fn)
    tot += us
    print(f"{us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s  {name}", flush=True)
print("sum", round(tot, 1))
