"""Per-workgroup phase timeline of linear_fast_kernel (build with -DRR_TRACE): entry / prologue done /
k-loop done / epilogue done stamps (100 MHz realtime counter) + HW_ID per workgroup.
Usage: RR_LIB_PATH=build/variants/lib_trace.so python tools/trace_linear.py [mode]"""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import _lib, functions as Fn
mode = sys.argv[1] if len(sys.argv) > 1 else "m0"
torch.manual_seed(0)
nA, nB, H = 71425, 138881, 300
dev = "cuda"
a_msg = torch.randn(nA, H, device=dev); msg = torch.randn(nB, H, device=dev); inp = torch.randn(nB, H, device=dev)
b2a = (torch.arange(nB, device=dev) * nA // nB).to(torch.int32); b2r = (torch.arange(nB, device=dev) ^ 1).clamp(max=nB - 1).to(torch.int32)
W = Fn.LinW(torch.randn(H, H, device=dev) / 17, torch.randn(H, device=dev))
out = torch.empty(nB, H, device=dev)
nwg = (nB + 63) // 64
buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
L = _lib.lib()
def run():
    if mode == "m0":
        Fn.linear(nB, H, W.pk(H), w_packed=True, a1=msg, k1=H, out=out)
    elif mode == "m1_bare":
        Fn.linear(nB, H, W.pk(H), w_packed=True, a1=a_msg, k1=H, a1_idx=b2a, a1_sub=msg, a1_sub_idx=b2r, out=out)
    elif mode == "m1_noidx":
        Fn.linear(nB, H, W.pk(H), w_packed=True, a1=msg, k1=H, a1_sub=inp, out=out)
    elif mode == "m0_res":
        Fn.linear(nB, H, W.pk(H), w_packed=True, a1=msg, k1=H, bias=W.b, residual=inp, act=1, out=out)
    else:
        Fn.linear(nB, H, W.pk(H), w_packed=True, a1=a_msg, k1=H, a1_idx=b2a, a1_sub=msg, a1_sub_idx=b2r, bias=W.b, residual=inp, act=1, drop_p=0.1, seed=5, out=out)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"back-to-back x10: {e0.elapsed_time(e1) * 100:.1f} us per launch")
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print(f"single launch: {e0.elapsed_time(e1) * 1000:.1f} us")
L._handle if False else None
fn = C.CDLL(_lib.LIB_PATH).rr_debug_set_trace
fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
assert fn(buf.data_ptr()) == 0
run(); torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nwg, 8)
t0 = t[:, 0].min()
st, pro, kl, ep = (t[:, 0] - t0) / 100.0, (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t[:, 1]) / 100.0, (t[:, 3] - t[:, 2]) / 100.0
end = (t[:, 3] - t0) / 100.0
print(f"mode {mode}: {nwg} workgroups, kernel span {end.max():.1f} us")
clk = (t[:, 6] - t[:, 5]) / np.maximum(1, (t[:, 2] - t[:, 1])) * 100.0      # shader cycles per us -> MHz
q = np.percentile(clk, [0, 10, 50, 90, 100])
print(f"  shader clock during the k-loop (s_memtime / s_memrealtime): min {q[0]:.0f} p10 {q[1]:.0f} p50 {q[2]:.0f} p90 {q[3]:.0f} max {q[4]:.0f} MHz")
order = np.argsort(st)
for name, v in (("start", st), ("prologue", pro), ("k-loop", kl), ("epilogue", ep), ("end", end)):
    q = np.percentile(v, [0, 10, 50, 90, 100])
    print(f"  {name:9s} min {q[0]:8.2f} p10 {q[1]:8.2f} p50 {q[2]:8.2f} p90 {q[3]:8.2f} max {q[4]:8.2f} us")
# rounds: cluster start times
hist, edges = np.histogram(st, bins=40)
print("  start-time histogram (us):", " ".join(f"{int(e)}:{h}" for h, e in zip(hist, edges) if h))
hw = t[:, 4] & 0xFFFFFFFF; xcc = t[:, 4] >> 32
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
key = xcc * 1000 + se * 100 + sh * 20 + cu
print("  distinct (xcc,se,sh,cu):", len(np.unique(key)), "blocks/unit min/max:", np.bincount(np.unique(key, return_inverse=True)[1]).min(), np.bincount(np.unique(key, return_inverse=True)[1]).max())
for b in (0, 1, 8, 255, 256, 767, 768, 1500, nwg - 1):
    print(f"  wg {b:5d} xcc {xcc[b]} se {se[b]} cu {cu[b]} start {st[b]:7.2f} pro {pro[b]:6.2f} k {kl[b]:7.2f} epi {ep[b]:6.2f}")
