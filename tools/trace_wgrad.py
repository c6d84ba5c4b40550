"""Per-workgroup phase timeline of wgrad_fast_kernel (build with -DRR_TRACE).  Usage:
RR_LIB_PATH=build/variants/lib_trace.so python tools/trace_wgrad.py [bonds|atoms|wi]"""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import _lib, functions as Fn
mode = sys.argv[1] if len(sys.argv) > 1 else "bonds"
torch.manual_seed(0)
nA, nB, H = 71425, 138881, 300
dev = "cuda"
a_msg = torch.randn(nA, H, device=dev); msg = torch.randn(nB, H, device=dev)
b2a = (torch.arange(nB, device=dev) * nA // nB).to(torch.int32); b2r = (torch.arange(nB, device=dev) ^ 1).clamp(max=nB - 1).to(torch.int32)
dyB = torch.randn(nB, H, device=dev); yB = torch.relu(torch.randn(nB, H, device=dev))
dyA = torch.randn(nA, H, device=dev); yA = torch.relu(torch.randn(nA, H, device=dev))
fa = torch.randn(nA, 64, device=dev); fb = torch.randn(nB, 84, device=dev)
def run():
    if mode == "bonds":      # W_h: mask + gather/sub, K = 300
        dw = torch.empty(H, H, device=dev); db = torch.empty(H, device=dev)
        Fn.wgrad(nB, H, dyB, dw, dbias=db, mask=yB, mask_scale=1.1, x1=a_msg, k1=H, x1_idx=b2a, x1_sub=msg, x1_sub_idx=b2r)
    elif mode == "atoms":    # W_o: mask, K = 61 + 300
        dw = torch.empty(H, 361, device=dev); db = torch.empty(H, device=dev)
        Fn.wgrad(nA, H, dyA, dw, dbias=db, mask=yA, mask_scale=1.1, x1=fa, k1=61, x2=a_msg, k2=H)
    else:                    # W_i: K = 83
        dw = torch.empty(H, 83, device=dev); db = torch.empty(H, device=dev)
        Fn.wgrad(nB, H, dyB, dw, dbias=db, x1=fb, k1=83)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"mode {mode}: back-to-back x10 (kernel + reduce): {e0.elapsed_time(e1) * 100:.1f} us per call")
nwg = 4096
buf = torch.zeros(nwg * 8 + 8 * 48 * 8, dtype=torch.int64, device=dev)
fn = C.CDLL(_lib.LIB_PATH).rr_debug_set_trace
fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
assert fn(buf.data_ptr()) == 0
run(); torch.cuda.synchronize()
raw = buf.cpu().numpy()
loop = raw[nwg * 8:].reshape(8, 48, 8)          # (block*4 + wave, iteration, stamp) in shader cycles (-DRR_TRACE_LOOP)
if loop[0, 10, 0] > 0:
    for w in range(8):
        L = loop[w, 8:40].astype(np.float64)
        d = lambda a, b: np.median(L[:, b] - L[:, a])
        per = np.median(L[1:, 0] - L[:-1, 0])
        print(f"  block {w // 4} wave {w % 4}: cycles/iteration {per:7.0f} | issue {d(0, 1):6.0f} | ds_read+MFMA {d(1, 2):6.0f} | "
              f"wait loads {d(2, 3):6.0f} | commit {d(3, 4):6.0f} | barrier {d(4, 5):6.0f}")
t = raw[:nwg * 8].reshape(nwg, 8)
t = t[t[:, 3] > 0]
t0 = t[:, 0].min()
st, pro, kl, ep = (t[:, 0] - t0) / 100.0, (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t[:, 1]) / 100.0, (t[:, 3] - t[:, 2]) / 100.0
end = (t[:, 3] - t0) / 100.0
clk = (t[:, 6] - t[:, 5]) / np.maximum(1, (t[:, 2] - t[:, 1])) * 100.0
print(f"  {len(t)} workgroups, span {end.max():.1f} us, shader clock in the loop p50 {np.percentile(clk, 50):.0f} MHz")
for name, v in (("start", st), ("prologue", pro), ("m-loop", kl), ("epilogue", ep), ("end", end)):
    q = np.percentile(v, [0, 10, 50, 90, 100])
    print(f"  {name:9s} min {q[0]:8.2f} p10 {q[1]:8.2f} p50 {q[2]:8.2f} p90 {q[3]:8.2f} max {q[4]:8.2f} us")
