"""Host side of the packed-step pipeline: the shard file round-trips the packer's arrays,
BatchMolGraph.subset()/unique() work from packed arrays alone, and a batch reloaded with load_packed() keeps its
molecule identities (so reactant de-duplication works on it)."""
import numpy as np
import pytest

from reactranker_amd import featurization, shards, synth


def _steps(n, scope=(4, 2, 5), K=4):
    out = []
    for i in range(n):
        qb = synth.make_queries(100 + i, len(scope), list(scope), atoms_lo=5, atoms_hi=10)
        out.append((qb, featurization.BatchMolGraph(qb.r_specs, K=K), featurization.BatchMolGraph(qb.p_specs, K=K)))
    return out


def test_shard_file_roundtrips_every_array(tmp_path):
    path = str(tmp_path / "train.rrshard")
    steps = _steps(4)
    with shards.ShardWriter(path) as w:
        for qb, rb, pb in steps:
            w.add_step(rb, pb, qb.scope, qb.targets, qb.add_features)
    r = shards.ShardReader(path)
    assert len(r) == 4 and r.atom_fdim == 61 and r.bond_fdim == 22
    for i, (qb, rb, pb) in enumerate(steps):
        m, h = r.meta(i), r.host_step(i)
        assert m["offset"] % shards.BLOB_ALIGN == 0 and m["M"] == pb.n_mols and m["Q"] == len(qb.scope)
        assert m["nA"] == pb.n_atoms and m["nB"] == pb.n_bonds and m["K"] == 4 and m["has_unique"]
        for k in shards._TABLES + ("f_atoms",):
            assert np.array_equal(h["p." + k], pb._host[k]), k
        assert np.array_equal(h["p.fbond"][:, :22], pb._host["f_bonds"][:, 61:83]) and not h["p.fbond"][:, 22:].any()
        assert "r.f_atoms" not in h                                 # repeated reactants: tables only
        ub, amap, amap_t = rb.unique()
        bmap, bmap_t = rb.unique_bonds()
        for k in shards._TABLES:
            assert np.array_equal(h["r." + k], rb._host[k]), k
            assert np.array_equal(h["u." + k], ub._host[k]), k
        assert np.array_equal(h["amap"], amap) and np.array_equal(h["amap_t"], amap_t)
        assert np.array_equal(h["bmap"], bmap) and np.array_equal(h["bmap_t"], bmap_t)
        # the reactant features ARE gathers of the distinct reactants' rows (what the device side rebuilds)
        assert np.array_equal(h["u.f_atoms"][amap], rb._host["f_atoms"])
        assert np.array_equal(h["u.fbond"][bmap][:, :22], rb._host["f_bonds"][:, 61:83])
        assert np.array_equal(h["scope"], np.asarray(qb.scope)) and np.array_equal(h["targets"], qb.targets)
        assert np.array_equal(h["add"], qb.add_features)
        for k, a in h.items():                                      # every array 256-byte aligned inside its blob
            assert (a.__array_interface__["data"][0] - r.blob(i).__array_interface__["data"][0]) % shards.ARRAY_ALIGN == 0, k


def test_shard_without_repeated_reactants_keeps_reactant_features(tmp_path):
    qb = synth.make_queries(3, 3, [1, 1, 1], atoms_lo=5, atoms_hi=8)    # one candidate per query: nothing repeats
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    path = str(tmp_path / "s.rrshard")
    with shards.ShardWriter(path) as w:
        w.add_step(rb, pb, qb.scope, qb.targets, None)
    h = shards.ShardReader(path).host_step(0)
    assert not shards.ShardReader(path).meta(0)["has_unique"] and "u.a2b" not in h and "add" not in h
    assert np.array_equal(h["r.f_atoms"], rb._host["f_atoms"])


def test_writer_rejects_mismatched_sides(tmp_path):
    qb = synth.make_queries(3, 2, [2, 3], atoms_lo=5, atoms_hi=8)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=5)
    with shards.ShardWriter(str(tmp_path / "x.rrshard")) as w:
        with pytest.raises(ValueError):
            w.add_step(rb, pb, qb.scope, qb.targets)


def test_subset_and_unique_from_packed_arrays():
    qb = synth.make_queries(8, 3, [4, 1, 3], atoms_lo=4, atoms_hi=9)
    b = featurization.BatchMolGraph(qb.r_specs, K=5)
    distinct = []
    for s in qb.r_specs:
        if not any(s is t for t in distinct):
            distinct.append(s)
    ub, amap, amap_t = b.unique()
    direct = featurization.BatchMolGraph(distinct, K=5)
    for k in featurization._PACK_KEYS + ("b2b_t", "npad_b"):
        assert np.array_equal(ub._host[k], direct._host[k]), k
    assert list(b.molecule_ids()) == [0, 0, 0, 0, 1, 2, 2, 2]
    # a rank's block of whole queries (queries 1..2) as a subset, against packing those molecules directly
    sub = b.subset(np.arange(4, 8))
    ref = featurization.BatchMolGraph(qb.r_specs[4:8], K=5)
    for k in featurization._PACK_KEYS:
        assert np.array_equal(sub._host[k], ref._host[k]), k
    assert sub.b_scope == ref.b_scope and sub.a_scope == ref.a_scope


def test_load_packed_keeps_molecule_identities_for_dedup(tmp_path):
    """a batch from load_packed() must drive the model's reactant de-duplication (unique / unique_bonds)."""
    qb = synth.make_queries(5, 3, [3, 2, 4], atoms_lo=5, atoms_hi=9)
    b = featurization.BatchMolGraph(qb.r_specs, K=4)
    path = str(tmp_path / "b.npz")
    featurization.save_packed(b, path)
    l = featurization.load_packed(path)
    (ub0, amap0, amap_t0), (ub1, amap1, amap_t1) = b.unique(), l.unique()
    assert ub1.n_mols == 3 and np.array_equal(amap0, amap1) and np.array_equal(amap_t0, amap_t1)
    for k in featurization._PACK_KEYS:
        assert np.array_equal(ub0._host[k], ub1._host[k]), k
    for x, y in zip(b.unique_bonds(), l.unique_bonds()):
        assert np.array_equal(x, y)


def test_largest_steps_come_first_for_the_allocator_warm_up(tmp_path):
    """ShardSet.largest(k): the steps with the most atoms + bonds (what a streamed epoch should run first, bench.py)"""
    from reactranker_amd import featurization, shards, synth
    paths = []
    sizes = []
    for f, cands in enumerate(([3, 9], [5, 2, 7])):
        p = str(tmp_path / f"s{f}.rrshard")
        with shards.ShardWriter(p) as w:
            for i, c in enumerate(cands):
                qb = synth.make_queries(10 * f + i, 2, [c, c + 1], atoms_lo=5, atoms_hi=9)
                rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
                w.add_step(rb, pb, qb.scope, qb.targets, qb.add_features)
        paths.append(p)
    rd = shards.ShardSet(paths)
    size = [rd.meta(i)["nA"] + rd.meta(i)["nB"] for i in range(len(rd))]
    top = rd.largest(2)
    assert len(top) == 2 and size[top[0]] == max(size) and size[top[1]] == sorted(size)[-2]
    assert rd.largest(0) == [] and len(rd.largest(99)) == len(rd)
