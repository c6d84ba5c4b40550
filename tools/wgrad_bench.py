"""A/B harness for rr_linear_wgrad_f32 on the step's real shapes, several builds, alternating processes.
Usage: wgrad_bench.py lib1.so lib2.so ...   (each lib is run twice, interleaved)"""
import subprocess, sys, os, json
CODE = r'''
import sys, os, json, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from reactranker_amd import functions as Fn
torch.manual_seed(0)
nA, nB, H = 71425, 138881, 300
dev = "cuda"
a_msg = torch.randn(nA, H, device=dev); msg = torch.randn(nB, H, device=dev)
b2a = (torch.arange(nB, device=dev) * nA // nB).to(torch.int32); b2r = (torch.arange(nB, device=dev) ^ 1).clamp(max=nB - 1).to(torch.int32)
dy = torch.randn(nB, H, device=dev); y = torch.relu(torch.randn(nB, H, device=dev))
fb = torch.randn(nB, 84, device=dev); fa = torch.randn(nA, 64, device=dev); fbs = torch.randn(nA, 84, device=dev)
def t(fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = {}
def W(n, k): return torch.empty(n, k, device=dev), torch.empty(n, device=dev)
dw, db = W(H, H)
res["bond_ms5"] = (t(lambda: Fn.wgrad(nB, H, dy, dw, dbias=db, mask=y, mask_scale=1.1, x1=a_msg, k1=H, x1_idx=b2a, x1_sub=msg, x1_sub_idx=b2r)), 2.0 * nB * H * (H + 1))
dw2, db2 = W(H, H + 83)
res["atom_m4"] = (t(lambda: Fn.wgrad(nA, H, dy[:nA], dw2, dbias=db2, mask=y[:nA], mask_scale=1.1, x1=a_msg, k1=H, x2=fbs, k2=83)), 2.0 * nA * H * (H + 84))
dw3, db3 = W(H, 61 + H)
res["atom_wo4"] = (t(lambda: Fn.wgrad(nA, H, dy[:nA], dw3, dbias=db3, mask=y[:nA], mask_scale=1.1, x1=fa, k1=61, x2=a_msg, k2=H)), 2.0 * nA * H * (61 + H + 1))
dw4, db4 = W(H, 83)
res["bond_wi3"] = (t(lambda: Fn.wgrad(nB, H, dy, dw4, dbias=db4, x1=fb, k1=83)), 2.0 * nB * H * 84)
dw5, db5 = W(H, 2 * H)
res["atom_s5"] = (t(lambda: Fn.wgrad(nA, H, dy[:nA], dw5, dbias=db5, mask=y[:nA], mask_scale=1.1, x1=a_msg, k1=H, x1_sub=msg[:nA], x2=msg[nB - nA:nB], k2=H)), 2.0 * nA * H * (2 * H + 1))
res["bond_s5"] = (t(lambda: Fn.wgrad(nB, H, dy, dw, dbias=db, x1=a_msg, k1=H, x1_idx=b2a, x1_sub=msg, x1_sub_idx=b2r)), 2.0 * nB * H * (H + 1))
res["bond_m5"] = (t(lambda: Fn.wgrad(nB, H, dy, dw, dbias=db, mask=y, mask_scale=1.1, x1=msg, k1=H)), 2.0 * nB * H * (H + 1))
res["bond_plain5"] = (t(lambda: Fn.wgrad(nB, H, dy, dw, dbias=db, x1=msg, k1=H)), 2.0 * nB * H * (H + 1))
print(json.dumps({k: [round(v, 1), round(fl / v / 1e6, 1)] for k, (v, fl) in res.items()}))
'''
libs = sys.argv[1:]
for rep in range(2):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", CODE, os.path.abspath(lib)], capture_output=True, text=True, timeout=300)
        last = (r.stdout.strip().splitlines() or ["(no output)"])[-1]
        print(f"{os.path.basename(lib):24s}", last if r.returncode == 0 else r.stderr[-500:], flush=True)
