import numpy as np
rng = np.random.default_rng(0)
def trunc_bf16(x):
    u = x.view(np.uint32) & np.uint32(0xFFFF0000)
    return u.view(np.float32)
def rne_bf16(x):
    u = x.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)
def split3(x, f):
    a0 = f(x); r1 = (x - a0).astype(np.float32); a1 = f(r1); r2 = (r1 - a1).astype(np.float32); a2 = f(r2)
    return a0, a1, a2
M, K, N = 512, 300, 304
for scale_desc, A, B in [("normal", rng.standard_normal((M, K)).astype(np.float32), (rng.standard_normal((K, N)) / 17).astype(np.float32)),
                         ("relu-sparse/wide", (np.maximum(rng.standard_normal((M, K)), 0) * np.exp(3 * rng.standard_normal((M, K)))).astype(np.float32), (rng.standard_normal((K, N)) / 17).astype(np.float32))]:
    ref = A.astype(np.float64) @ B.astype(np.float64)
    den = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    # f32 chain (k sequential, fma ~ f32 round each step)
    acc = np.zeros((M, N), np.float32)
    for k in range(K):
        acc = (acc.astype(np.float64) + A[:, k:k+1].astype(np.float64) * B[k:k+1, :].astype(np.float64)).astype(np.float32)
    e32 = np.abs(acc - ref) / den
    for name, f in (("trunc", trunc_bf16), ("rne", rne_bf16)):
        As = split3(A, f); Bs = split3(B, f)
        chk = np.abs((As[0].astype(np.float64) + As[1] + As[2]) - A).max()
        for nprod, pairs in ((3, [(0,0),(0,1),(1,0)]), (6, [(0,0),(0,1),(1,0),(0,2),(1,1),(2,0)]), (9, [(i,j) for i in range(3) for j in range(3)])):
            # accumulate per k32 block: exact products summed in f64 within the block, rounded to f32 at block end (per product kind)
            acc = np.zeros((M, N), np.float32)
            for k0 in range(0, K, 32):
                # small terms first
                for (i, j) in sorted(pairs, key=lambda p: -(p[0] + p[1])):
                    blk = As[i][:, k0:k0+32].astype(np.float64) @ Bs[j][k0:k0+32, :].astype(np.float64)
                    acc = (acc.astype(np.float64) + blk).astype(np.float32)
            e = np.abs(acc - ref) / den
            print(f"{scale_desc:18s} {name:5s} x{nprod}: split exact err {chk:.1e}  rel-to-sum|ab| max {e.max():.2e} mean {e.mean():.2e}   (f32 chain: max {e32.max():.2e} mean {e32.mean():.2e})")
