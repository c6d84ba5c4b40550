"""Property tests (hypothesis) of the host half on random ragged batches: the packer's tables are adjoint-consistent for
any pad width, subset() of whole queries equals packing those queries directly (what a rank's shard is), unique() maps
reconstruct the full reactant batch, and a shard file returns exactly what went in."""
import os
import tempfile

import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

from reactranker_amd import featurization, shards, synth

SET = settings(max_examples=12, deadline=None, suppress_health_check=[HealthCheck.too_slow])


@SET
@given(seed=st.integers(0, 10 ** 6), scope=st.lists(st.integers(1, 6), min_size=1, max_size=5), extra_k=st.integers(0, 3))
def test_tables_are_adjoint_consistent_for_any_batch_and_pad_width(seed, scope, extra_k):
    qb = synth.make_queries(seed, len(scope), scope, atoms_lo=3, atoms_hi=11)
    base = featurization.BatchMolGraph(qb.p_specs)
    b = featurization.BatchMolGraph(qb.p_specs, K=base.max_num_bonds + extra_k)
    h = b._host
    nA, nB, K = h["nA"], h["nB"], h["K"]
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((nB, 3))
    # adjoint of m_in[b] = sum_k msg[a2b[b2a[b], k]] - msg[b2revb[b]] by brute-force scatter vs the bond-to-bond table
    want = np.zeros((nB, 3))
    d_amsg = np.zeros((nA, 3))
    np.add.at(d_amsg, h["b2a"], d)
    np.add.at(want, h["a2b"].reshape(-1), np.repeat(d_amsg, K, axis=0))
    np.add.at(want, h["b2revb"], -d)
    t = h["b2b_t"]
    got = np.where(t[..., None] >= 0, d[np.maximum(t, 0)], 0.0).sum(1)
    got[0] += (h["npad_b"][:, None] * d).sum(0)
    assert np.allclose(got, want)
    assert (h["npad"] >= 0).all() and h["npad"][0] == K and (h["a2b"] >= 0).all() and h["a2b"].max() < nB


@SET
@given(seed=st.integers(0, 10 ** 6), scope=st.lists(st.integers(1, 5), min_size=2, max_size=5), cut=st.integers(1, 4))
def test_subset_of_whole_queries_equals_packing_them_directly(seed, scope, cut):
    qb = synth.make_queries(seed, len(scope), scope, atoms_lo=3, atoms_hi=10)
    cut = min(cut, len(scope) - 1)
    m0 = sum(scope[:cut])
    for specs in (qb.r_specs, qb.p_specs):
        full = featurization.BatchMolGraph(specs, K=5)
        sub = full.subset(np.arange(m0, len(specs)))
        ref = featurization.BatchMolGraph(specs[m0:], K=5)
        for k in featurization._PACK_KEYS + ("b2b_t", "npad_b"):
            assert np.array_equal(sub._host[k], ref._host[k]), k
    rb = featurization.BatchMolGraph(qb.r_specs, K=5)
    ub, amap, amap_t = rb.unique()
    assert ub.n_mols == len(scope)
    assert np.array_equal(ub._host["f_atoms"][amap], rb._host["f_atoms"])       # the copies ARE the distinct reactants' rows
    bmap, bmap_t = rb.unique_bonds()
    assert np.array_equal(ub._host["f_bonds"][bmap], rb._host["f_bonds"])
    for tt, mp in ((amap_t, amap), (bmap_t, bmap)):                            # transposed maps list every copy exactly once
        rows = np.sort(tt[tt >= 0])
        assert np.array_equal(rows, np.arange(mp.shape[0])) and all((mp[tt[u][tt[u] >= 0]] == u).all() for u in range(tt.shape[0]))


@SET
@given(seed=st.integers(0, 10 ** 6), n_steps=st.integers(1, 3), scope=st.lists(st.integers(1, 4), min_size=1, max_size=4))
def test_shard_roundtrip_any_steps(seed, n_steps, scope):
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.rrshard")
        packed = []
        with shards.ShardWriter(path) as w:
            for i in range(n_steps):
                qb = synth.make_queries(seed + i, len(scope), scope, atoms_lo=3, atoms_hi=9)
                rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
                w.add_step(rb, pb, qb.scope, qb.targets, qb.add_features)
                packed.append((qb, rb, pb))
        r = shards.ShardSet([path])
        assert len(r) == n_steps
        for i, (qb, rb, pb) in enumerate(packed):
            h = r.host_step(i)
            assert np.array_equal(h["p.a2b"], pb._host["a2b"]) and np.array_equal(h["r.b2b_t"], rb._host["b2b_t"])
            assert np.array_equal(h["targets"], qb.targets) and list(h["scope"]) == list(qb.scope)
            assert r.meta(i)["M"] == pb.n_mols and r.blob(i).shape[0] % shards.BLOB_ALIGN == 0
