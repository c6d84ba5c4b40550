"""A/B harness: time rr_linear_f32 / rr_linear_wgrad_f32 on the hot shapes for several builds of the
library (variants compiled with different -D flags).  Usage: gemm_bench.py lib1.so lib2.so ..."""
import ctypes as C, subprocess, sys, os, json
CODE = r'''
import sys, os, json, torch, ctypes as C
sys.path.insert(0, os.getcwd())
from reactranker_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from reactranker_amd import functions as Fn
torch.manual_seed(0)
nA, nB, H = 71425, 138881, 300
dev = "cuda"
a_msg = torch.randn(nA, H, device=dev); msg = torch.randn(nB, H, device=dev); inp = torch.randn(nB, H, device=dev)
b2a = torch.randint(0, nA, (nB,), dtype=torch.int32, device=dev); b2r = torch.randint(0, nB, (nB,), dtype=torch.int32, device=dev)
# locality like a real batch: neighbours are near
b2a = (torch.arange(nB, device=dev) * nA // nB).to(torch.int32); b2r = (torch.arange(nB, device=dev) ^ 1).clamp(max=nB - 1).to(torch.int32)
W = Fn.LinW(torch.randn(H, H, device=dev) / 17, torch.randn(H, device=dev))
dy = torch.randn(nB, H, device=dev); y = torch.relu(torch.randn(nB, H, device=dev))
dw = torch.empty(H, H, device=dev); db = torch.empty(H, device=dev)
out = torch.empty(nB, H, device=dev)
def t(fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * nB * H * H
res = {}
res["lin_m1"] = t(lambda: Fn.linear(nB, H, W.pk(H), w_packed=True, a1=a_msg, k1=H, a1_idx=b2a, a1_sub=msg, a1_sub_idx=b2r, bias=W.b, residual=inp, act=1, drop_p=0.1, seed=5, out=out))
res["m1_bare"] = t(lambda: Fn.linear(nB, H, W.pk(H), w_packed=True, a1=a_msg, k1=H, a1_idx=b2a, a1_sub=msg, a1_sub_idx=b2r, out=out))
res["m1_noidx"] = t(lambda: Fn.linear(nB, H, W.pk(H), w_packed=True, a1=msg, k1=H, a1_sub=inp, out=out))
res["m0_idx"] = t(lambda: Fn.linear(nB, H, W.pk(H), w_packed=True, a1=a_msg, k1=H, a1_idx=b2a, out=out))
res["m0_idxrev"] = t(lambda: Fn.linear(nB, H, W.pk(H), w_packed=True, a1=msg, k1=H, a1_idx=b2r, out=out))
res["lin_m0"] = t(lambda: Fn.linear(nB, H, W.pk(H), w_packed=True, a1=msg, k1=H, out=out))
msg2 = torch.randn(nB, 2 * H, device=dev)
W2 = Fn.LinW(torch.randn(H, 2 * H, device=dev) / 24, None)
res["lin_m0_k600"] = t(lambda: Fn.linear(nB, H, W2.pk(2 * H), w_packed=True, a1=msg2, k1=2 * H, out=out))
res["lin_m0_drop"] = t(lambda: Fn.linear(nB, H, W.pk(H), w_packed=True, a1=msg, k1=H, act=1, drop_p=0.1, seed=5, out=out))
res["lin_m0_res"] = t(lambda: Fn.linear(nB, H, W.pk(H), w_packed=True, a1=msg, k1=H, bias=W.b, residual=inp, act=1, out=out))
res["lin_m2"] = t(lambda: Fn.linear(nB, H, W.pk_t(0, H), w_packed=True, a1=dy, k1=H, a_mask=y, mask_scale=1.1, out=out))
res["wg_ms"] = t(lambda: Fn.wgrad(nB, H, dy, dw, dbias=db, mask=y, mask_scale=1.1, x1=a_msg, k1=H, x1_idx=b2a, x1_sub=msg, x1_sub_idx=b2r))
res["wg_plain"] = t(lambda: Fn.wgrad(nB, H, dy, dw, dbias=db, x1=msg, k1=H))
print(json.dumps({k: [round(v, 1), round(fl / v / 1e6, 1)] for k, v in res.items()}))
'''
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CODE, os.path.abspath(lib)], capture_output=True, text=True, timeout=300)
    last = (r.stdout.strip().splitlines() or ["(no output)"])[-1]
    print(os.path.basename(lib), "rc", r.returncode, last, flush=True)
    if r.returncode != 0:
        print(r.stderr[-600:], flush=True)
        sys.exit(1)
