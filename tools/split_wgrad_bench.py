"""Split weight-gradient GEMM next to the f32-MFMA one: error against f64 and time for the W_h / W_o / W_i / diff-W_h forms.
Usage (GPU box): python tools/split_wgrad_bench.py [quick]"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import functions as Fn
dev = "cuda"; torch.manual_seed(0); H = 300
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def err(o, ref, den):
    e = (o.double() - ref).abs() / den
    return f"max {e.max().item():.2e} mean {e.mean().item():.2e}"
def both(M, N, K, **kw):
    outs = []
    for en in (False, True):
        Fn.SplitGemm.enabled = en
        dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
        Fn.wgrad(M, N, kw.pop("dy") if False else kw["dy"], dw, dbias=db, **{k: v for k, v in kw.items() if k != "dy"})
        outs.append((dw, db))
    return outs
def run(name, M, N, K, ref_x, dz_ref, **kw):
    (w1, b1), (w2, b2) = both(M, N, K, **kw)
    ref = dz_ref.double().t() @ ref_x.double(); den = dz_ref.double().abs().t() @ ref_x.double().abs() + 1e-30
    rb = dz_ref.double().sum(0); db = dz_ref.double().abs().sum(0) + 1e-30
    print(f"M {M} {name}: dW f32 {err(w1, ref, den)} | split {err(w2, ref, den)}   dbias f32 {err(b1, rb, db)} | split {err(b2, rb, db)}", flush=True)
    if not quick:
        us = []
        for en in (False, True):
            Fn.SplitGemm.enabled = en
            dw = torch.zeros(N, K, device=dev); db_ = torch.zeros(N, device=dev)
            us.append(t(lambda: Fn.wgrad(M, N, kw["dy"], dw, dbias=db_, **{k: v for k, v in kw.items() if k != "dy"})))
        Fn.SplitGemm.enabled = True                      # two f16 terms (rr_wgrad_args.split = 2), bounds computed once outside the loop
        am = [Fn.amax(kw[k], kw.get(c)) if kw.get(k) is not None else None for k, c in (("dy", None), ("x1", "k1"), ("x1_sub", "k1"), ("x2", "k2"))]
        dw = torch.zeros(N, K, device=dev); db_ = torch.zeros(N, device=dev)
        Fn.wgrad(M, N, kw["dy"], dw, dbias=db_, amax_of=am, **{k: v for k, v in kw.items() if k != "dy"})
        u3 = t(lambda: Fn.wgrad(M, N, kw["dy"], dw, dbias=db_, amax_of=am, **{k: v for k, v in kw.items() if k != "dy"}))
        print(f"      time f32 {us[0]:.1f} us   split {us[1]:.1f} us   f16x2 {u3:.1f} us   (f16x2 error: dW {err(dw, ref, den)}, dbias {err(db_, rb, db)})", flush=True)
Fn.SPLIT_MIN_ROWS = 1
for M in ((777, 9000) if quick else (9000, 138881, 71425)):
    z = torch.zeros(1, device=dev)
    # W_h: x = a_msg[b2a] - msg[rev], dz given
    nA = M // 2 + 5
    am = torch.randn(nA, H, device=dev); msg = torch.relu(torch.randn(M, H, device=dev)); dz = torch.randn(M, H, device=dev) * torch.exp(2 * torch.randn(M, 1, device=dev))
    b2a = torch.randint(0, nA, (M,), device=dev, dtype=torch.int32); rev = torch.randint(0, M, (M,), device=dev, dtype=torch.int32); b2a[0] = -1; rev[0] = -1
    X = torch.where(b2a[:, None] >= 0, am[b2a.clamp(min=0).long()], z) - torch.where(rev[:, None] >= 0, msg[rev.clamp(min=0).long()], z)
    run("W_h (gather-sub, K 300)", M, H, H, X, dz, dy=dz, x1=am, k1=H, x1_idx=b2a, x1_sub=msg, x1_sub_idx=rev)
    # W_o: [f_atoms 133 | a_msg 300], masked dy
    fa = torch.zeros(M, 136, device=dev); fa[:, :133] = (torch.rand(M, 133, device=dev) < 0.1).float(); a2 = torch.randn(M, H, device=dev)
    dy = torch.randn(M, H, device=dev); y = torch.relu(torch.randn(M, H, device=dev))
    dzr = torch.where(y > 0, dy * 1.25, torch.zeros_like(dy))
    run("W_o (133|300, mask)", M, H, 433, torch.cat([fa[:, :133], a2], 1), dzr, dy=dy, mask=y, mask_scale=1.25, x1=fa, k1=133, x2=a2, k2=H)
    # W_i: f_bonds 147
    fb = torch.zeros(M, 148, device=dev); fb[:, :147] = torch.randn(M, 147, device=dev)
    run("W_i (147)", M, H, 147, fb[:, :147], dz, dy=dz, x1=fb, k1=147)
    # diff W_h: [a_msg 300 | fb_sum 147]
    run("dW_h (300|147)", M, H, 447, torch.cat([a2, fb[:, :147]], 1), dz, dy=dz, x1=a2, k1=H, x2=fb, k2=147)
