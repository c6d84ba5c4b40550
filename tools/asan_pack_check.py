"""Drive the host half of the C-ABI (the graph packer) built with AddressSanitizer + UBSan (`make -C reactranker_amd/csrc
asan`).  Run with the sanitizer runtime preloaded:
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_pack_check.py
Exercises rr_pack_sizes / rr_pack_graphs / rr_derive_tables / rr_derive_bond_tables on ragged, empty, wide-K and
single-atom batches and on the rejected inputs; any out-of-bounds access aborts the process with an ASan report."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "reactranker_amd", "csrc", "libreactranker_pack_asan.so")
lib = C.CDLL(LIB)
P = lambda a: C.c_void_p(a.ctypes.data) if a is not None else None     # noqa: E731


def pack(arrs, K_override):
    mol_atoms, mol_bonds, fa, fb, b2a_l, b2r_l, a2b_off, a2b_l = arrs
    M = len(mol_atoms)
    nA, nB, K = C.c_int64(), C.c_int64(), C.c_int32()
    st = lib.rr_pack_sizes(P(mol_atoms), P(mol_bonds), C.c_int64(M), P(a2b_off), C.c_int(K_override), C.byref(nA),
                           C.byref(nB), C.byref(K))
    if st != 0:
        return st, None
    nA, nB, K = nA.value, nB.value, K.value
    o = dict(f_atoms=np.empty((nA, 64), np.float32), f_bonds=np.empty((nB, 84), np.float32), a2b=np.empty((nA, K), np.int32),
             b2a=np.empty(nB, np.int32), b2revb=np.empty(nB, np.int32), a2a=np.empty((nA, K), np.int32),
             a_scope=np.empty((M, 2), np.int32), a2b_rev_t=np.empty((nA, K), np.int32), b2t=np.empty(nB, np.int32),
             a2a_t=np.empty((nA, K), np.int32), npad=np.empty(nA, np.float32), atom2mol=np.empty(nA, np.int32))
    st = lib.rr_pack_graphs(P(mol_atoms), P(mol_bonds), C.c_int64(M), P(fa), C.c_int(61), P(fb), C.c_int(83), P(b2a_l),
                            P(b2r_l), P(a2b_off), P(a2b_l), C.c_int(K), P(o["f_atoms"]), C.c_int64(64), P(o["f_bonds"]),
                            C.c_int64(84), P(o["a2b"]), P(o["b2a"]), P(o["b2revb"]), P(o["a2a"]), P(o["a_scope"]),
                            P(o["a2b_rev_t"]), P(o["b2t"]), P(o["a2a_t"]), P(o["npad"]), P(o["atom2mol"]))
    if st != 0:
        return st, None
    Kb = max(1, K - 1)
    o["b2b_t"], o["npad_b"] = np.empty((nB, Kb), np.int32), np.empty(nB, np.float32)
    st = lib.rr_derive_bond_tables(P(o["a2b_rev_t"]), P(o["b2t"]), P(o["b2revb"]), P(o["b2a"]), P(o["npad"]), C.c_int64(nA),
                                   C.c_int64(nB), C.c_int(K), C.c_int(Kb), P(o["b2b_t"]), P(o["npad_b"]))
    # the tables alone, from the padded arrays (the path a reference BatchMolGraph takes)
    d = dict(a2a=np.empty((nA, K), np.int32), a2b_rev_t=np.empty((nA, K), np.int32), b2t=np.empty(nB, np.int32),
             a2a_t=np.empty((nA, K), np.int32), npad=np.empty(nA, np.float32), atom2mol=np.empty(nA, np.int32))
    st2 = lib.rr_derive_tables(P(o["a2b"]), P(o["b2a"]), P(o["b2revb"]), C.c_int64(nA), C.c_int64(nB), C.c_int(K),
                               P(o["a_scope"]), C.c_int64(M), P(d["a2a"]), P(d["a2b_rev_t"]), P(d["b2t"]), P(d["a2a_t"]),
                               P(d["npad"]), P(d["atom2mol"]))
    assert st2 == 0 and all(np.array_equal(d[k], o[k]) for k in d)
    return st, o


def main():
    from reactranker_amd import synth
    from reactranker_amd.featurization import _concat_specs
    n = 0
    for seed, scope, K in ((0, [3, 1, 4], 0), (1, [1], 0), (2, [2, 2], 6), (3, [5, 7, 2, 9], 4)):
        qb = synth.make_queries(seed, len(scope), scope, atoms_lo=3, atoms_hi=14)
        for specs in (qb.r_specs, qb.p_specs):
            arrs = tuple(np.ascontiguousarray(a) for a in _concat_specs(specs))
            st, o = pack(arrs, K)
            assert st == 0, st
            assert o["a2b"].max() < o["b2a"].shape[0] and o["b2a"].max() < o["a2b"].shape[0]
            st_small, _ = pack(arrs, 1)                              # pad width below the largest in-degree: rejected, no write
            assert st_small != 0 or o["a2b"].shape[1] == 1
            n += 1
    empty = (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 61), np.float32), np.zeros((0, 83), np.float32),
             np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(1, np.int64), np.zeros(0, np.int32))
    st, o = pack(empty, 0)
    assert st == 0 and o["a2b"].shape == (1, 1) and o["b2a"].shape == (1,)
    # corrupt local indices must be rejected, not followed
    qb = synth.make_queries(9, 1, [2], atoms_lo=4, atoms_hi=6)
    arrs = [np.ascontiguousarray(a).copy() for a in _concat_specs(qb.p_specs)]
    arrs[4][0] = 10 ** 6                                             # b2a_local out of range
    st, _ = pack(tuple(arrs), 0)
    assert st != 0
    print(f"asan_pack_check: {n + 2} batches packed under AddressSanitizer/UBSan, no report")


if __name__ == "__main__":
    main()
