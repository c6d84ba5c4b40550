"""Per-workgroup phase timeline of linear_split_kernel<19,19,MODE,12> (build with -DRR_TRACE): entry / prologue done /
k-loop done / epilogue done stamps (100 MHz realtime counter), shader clock inside the k-loop, HW_ID per workgroup.
Usage: RR_LIB_PATH=build/variants/lib_trace.so python tools/trace_split.py [m0|m0res|m1|m3] [rows]"""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import _lib, functions as Fn
mode = sys.argv[1] if len(sys.argv) > 1 else "m0"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 138881
torch.manual_seed(0)
H, dev = 300, "cuda"
nA = M // 2 + 3
x = torch.randn(M, H, device=dev); res = torch.randn(M, H, device=dev); am = torch.randn(nA, H, device=dev)
cw = torch.rand(M, device=dev)
base = torch.arange(M, device=dev)
b2a = ((base // 34) * 17 + torch.randint(0, 17, (M,), device=dev)).clamp(max=nA - 1).to(torch.int32)
rev = ((base // 34) * 34 + torch.randint(0, 34, (M,), device=dev)).clamp(max=M - 1).to(torch.int32)
W = Fn.LinW(torch.randn(H, H, device=dev) / 17, torch.randn(H, device=dev))
out = torch.empty(M, H, device=dev); dz = torch.empty(M, H, device=dev)
bits = torch.zeros(M, int(_lib.lib().rr_mask_bits_row_bytes(H)), dtype=torch.uint8, device=dev)
y = Fn.linear(M, H, W.pk(H), w_packed=True, a1=x, k1=H, bias=W.b, residual=res, act=1, drop_p=0.1, seed=3, mask_bits_out=bits)
nwg = (M + 191) // 192
buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
def run():
    if mode == "m0":
        Fn.linear(M, H, W.pk_t(0, H), w_packed=True, a1=x, k1=H, out=out, colsum_w=cw)
    elif mode == "m0res":
        Fn.linear(M, H, W.pk(H), w_packed=True, a1=x, k1=H, bias=W.b, residual=res, act=1, drop_p=0.1, seed=5, out=out, mask_bits_out=bits)
    elif mode == "m1":
        Fn.linear(M, H, W.pk(H), w_packed=True, a1=am, k1=H, a1_idx=b2a, a1_sub=y, a1_sub_idx=rev, bias=W.b, residual=res, act=1,
                  drop_p=0.1, seed=5, out=out, mask_bits_out=bits)
    else:
        Fn.linear(M, H, W.pk_t(0, H), w_packed=True, a1=x, k1=H, a_mask_bits=bits, mask_scale=1.1, out=out, dz_out=dz, colsum_w=cw)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"mode {mode} M {M}: back-to-back x10: {e0.elapsed_time(e1) * 100:.1f} us per launch")
fn = C.CDLL(_lib.LIB_PATH).rr_debug_set_trace
fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
assert fn(buf.data_ptr()) == 0
for _ in range(3): run()          # stamps of the last launch survive (warm, back to back)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nwg, 8)
t0 = t[:, 0].min()
st, pro, kl, ep = (t[:, 0] - t0) / 100.0, (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t[:, 1]) / 100.0, (t[:, 3] - t[:, 2]) / 100.0
end = (t[:, 3] - t0) / 100.0
print(f"  {nwg} workgroups, kernel span {end.max():.1f} us")
clk = (t[:, 6] - t[:, 5]) / np.maximum(1, (t[:, 2] - t[:, 1])) * 100.0
q = np.percentile(clk, [0, 10, 50, 90, 100])
print(f"  shader clock during the k-loop: min {q[0]:.0f} p10 {q[1]:.0f} p50 {q[2]:.0f} p90 {q[3]:.0f} max {q[4]:.0f} MHz")
for name, v in (("start", st), ("prologue", pro), ("k-loop", kl), ("epilogue", ep), ("end", end)):
    q = np.percentile(v, [0, 10, 50, 90, 100])
    print(f"  {name:9s} min {q[0]:8.2f} p10 {q[1]:8.2f} p50 {q[2]:8.2f} p90 {q[3]:8.2f} max {q[4]:8.2f} us")
for rnd, sel in (("round 1", st < 5), ("later", st >= 5)):
    if sel.any():
        print(f"  {rnd}: n {int(sel.sum())} prologue p50 {np.median(pro[sel]):.2f} k-loop p50 {np.median(kl[sel]):.2f} epilogue p50 {np.median(ep[sel]):.2f}")
hist, edges = np.histogram(st, bins=30)
print("  start-time histogram (us):", " ".join(f"{int(e)}:{h}" for h, e in zip(hist, edges) if h))
kcyc = (t[:, 6] - t[:, 5])
print(f"  k-loop shader cycles p50 {np.median(kcyc):.0f} (MFMA floor {10 * 3 * 114 * 16} at 3 waves/SIMD)")
