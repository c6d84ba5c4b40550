"""The arithmetic of the three-bf16-term GEMM path, restated in numpy (no GPU): the operand split is exact, and the six
products the kernels keep (accumulated in f32 per 32-deep k-step, smallest first) leave an error against the f64 product
that is not above the error of an f32 fmaf chain - which is what v_mfma_f32_16x16x4_f32 computes.  The GPU kernels are
checked against the same claim on the device in tests/test_gpu_split.py; this file pins the reasoning itself."""
import numpy as np
import pytest


def bf16_rne(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    t0 = bf16_rne(x)
    r1 = (x - t0).astype(np.float32)
    t1 = bf16_rne(r1)
    t2 = (r1 - t1).astype(np.float32)
    return t0, t1, t2


def is_bf16(x):
    return np.all((x.view(np.uint32) & 0xFFFF) == 0)


@pytest.mark.parametrize("seed,scale", [(0, 0.0), (1, 3.0), (2, 8.0)])
def test_three_terms_are_bf16_and_sum_exactly(seed, scale):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(200000) * np.exp(scale * rng.standard_normal(200000))).astype(np.float32)
    x[::7] = 0.0
    t0, t1, t2 = split3(x)
    assert is_bf16(t0) and is_bf16(t1) and is_bf16(t2)            # the second remainder needs no rounding
    assert np.array_equal(t0.astype(np.float64) + t1.astype(np.float64) + t2.astype(np.float64), x.astype(np.float64))
    nz = x != 0
    assert np.all(np.abs(t1[nz]) <= np.abs(x[nz]) * 2.0 ** -8) and np.all(np.abs(t2[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


@pytest.mark.parametrize("wide", [False, True])
def test_six_products_are_not_worse_than_an_f32_fma_chain(wide):
    rng = np.random.default_rng(5 + wide)
    M, K, N = 96, 300, 80
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((K, N)) / 17
    if wide:
        A = np.maximum(A, 0) * np.exp(3 * rng.standard_normal((M, K)))
    A, B = A.astype(np.float32), B.astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    den = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64) + 1e-300
    chain = np.zeros((M, N), np.float32)
    for k in range(K):                                             # fmaf chain: one rounding per k
        chain = (chain.astype(np.float64) + A[:, k:k + 1].astype(np.float64) * B[k:k + 1].astype(np.float64)).astype(np.float32)
    As, Bs = split3(A), split3(B)
    pairs = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]       # the kernels' order: smallest terms first
    acc = np.zeros((M, N), np.float32)
    for k0 in range(0, K, 32):                                     # one MFMA = 32 k: exact products, one f32 rounding
        for i, j in pairs:
            blk = As[i][:, k0:k0 + 32].astype(np.float64) @ Bs[j][k0:k0 + 32].astype(np.float64)
            acc = (acc.astype(np.float64) + blk).astype(np.float32)
    e_chain = np.abs(chain - ref) / den
    e_split = np.abs(acc - ref) / den
    assert e_split.mean() <= e_chain.mean() and e_split.max() <= 1.5 * e_chain.max()
    # dropping the three second-order products instead (a 2-term split) would NOT do: two orders of magnitude worse
    acc3 = np.zeros((M, N), np.float32)
    for k0 in range(0, K, 32):
        for i, j in [(1, 0), (0, 1), (0, 0)]:
            blk = As[i][:, k0:k0 + 32].astype(np.float64) @ Bs[j][k0:k0 + 32].astype(np.float64)
            acc3 = (acc3.astype(np.float64) + blk).astype(np.float32)
    assert (np.abs(acc3 - ref) / den).mean() > 10 * e_chain.mean()
