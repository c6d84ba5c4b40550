import sys, os, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import functions as Fn
sys.path.insert(0, os.path.join(os.getcwd(), "tools", "scratch"))
dev = "cuda"; H = 300
from reactranker_amd._lib import lib, PackDesc, check, ptr, stream
def pack_split(w, transpose, rows, c0, k1, k2):
    nb = int(lib().rr_split_weight_bytes(rows, k1, k2))
    dst = torch.empty(nb, dtype=torch.uint8, device=dev)
    d = (PackDesc * 1)()
    d[0].src, d[0].ld_src, d[0].transpose, d[0].rows, d[0].c0, d[0].k1, d[0].k2 = w.data_ptr(), w.stride(0), transpose, rows, c0, k1, k2
    d[0].dst, d[0].split = dst.data_ptr(), 1
    check(lib().rr_pack_weights_f32(d, 1, stream()), "pack")
    return dst
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
W = torch.randn(H, H, device=dev) / 17
ws = pack_split(W, 1, H, 0, H, 0)
for M in (192 * 256 * 3, 138881):
    dy = torch.randn(M, H, device=dev); y = torch.relu(torch.randn(M, H, device=dev)); o = torch.empty(M, H, device=dev); dz = torch.empty(M, H, device=dev)
    for dbg, name in ((0, "full"), (1, "no mfma"), (2, "no w dma"), (4, "no x loads"), (3, "no mfma, no dma"), (7, "nothing but epilogue+prologue"), (6, "mfma only")):
        os.environ["RR_SPLIT_DBG"] = str(dbg)
        u0 = t(lambda: Fn.linear(M, H, ws, w_packed=2, ldw=0, a1=dy, k1=H, out=o))
        u2 = t(lambda: Fn.linear(M, H, ws, w_packed=2, ldw=0, a1=dy, k1=H, a_mask=y, mask_scale=1.1, out=o, dz_out=dz))
        print(f"M {M} dbg {dbg} ({name}): mode0 {u0:.1f} us   mode2+dz {u2:.1f} us", flush=True)
