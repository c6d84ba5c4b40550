"""CPU-only checks of the host side: the native graph packer against the reference's own
BatchMolGraph arrays (golden), the C-ABI library's exported symbols against include/*.h, the
dropout stream against its numpy restatement, and the module surface (state_dict keys)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from reactranker_amd import _lib, featurization, synth
from reactranker_amd.base_model import build_model
from oracle import ref_cpu as O
from oracle import dropout_ref
from tests import helpers as Hh

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "reactranker_hip.h")).read()
    declared = set(re.findall(r"\b(rr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rr_dropout_keep"}            # mentioned in a comment only
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"symbols declared in the header but not exported: {missing}"
    assert set(_lib.EXPORTED_SYMBOLS) == declared, (set(_lib.EXPORTED_SYMBOLS) ^ declared)
    assert _lib.lib().rr_version() == _lib.ABI_VERSION
    assert _lib.lib().rr_strerror(-1).decode().startswith("invalid argument")


def test_ctypes_struct_layout_matches_compiled_header():
    # rr_linear_args / rr_wgrad_args are passed by pointer; a size mismatch would corrupt fields silently
    sl, sw = ctypes.c_size_t(), ctypes.c_size_t()
    _lib.lib().rr_abi_struct_sizes(ctypes.byref(sl), ctypes.byref(sw))
    assert sl.value == ctypes.sizeof(_lib.LinearArgs)
    assert sw.value == ctypes.sizeof(_lib.WgradArgs)


@pytest.mark.parametrize("path", Hh.model_case_files(), ids=lambda p: p.split("model_")[-1][:-4])
def test_native_packer_matches_reference_batchmolgraph(path):
    d, cfg = Hh.load_case(path)
    qb = Hh.case_queries(cfg)
    for prefix, specs in (("r_", qb.r_specs), ("p_", qb.p_specs)):
        b = featurization.BatchMolGraph(specs)
        f_atoms, f_bonds, a2b, b2a, b2revb, a_scope, b_scope = b.get_components()
        assert np.array_equal(f_atoms.numpy(), d[prefix + "f_atoms"])
        assert np.array_equal(f_bonds.numpy(), d[prefix + "f_bonds"])
        assert a2b.dtype == torch.int64 and np.array_equal(a2b.numpy(), d[prefix + "a2b"])
        assert np.array_equal(b2a.numpy(), d[prefix + "b2a"])
        assert np.array_equal(b2revb.numpy(), d[prefix + "b2revb"])
        assert np.array_equal(b.get_a2a().numpy(), d[prefix + "a2a"])
        assert np.array_equal(np.asarray(a_scope, np.int32).reshape(-1, 2), d[prefix + "a_scope"])
        assert np.array_equal(np.asarray(b_scope, np.int32).reshape(-1, 2), d[prefix + "b_scope"])
        assert b.max_num_bonds == int(d["K_" + prefix[0]])
        # duck-typed (python list) molecules go through the other concat path and must agree
        b2 = featurization.BatchMolGraph([synth.ListMolGraph(s) for s in specs])
        for k in ("f_atoms", "f_bonds", "a2b", "b2a", "b2revb", "a2a", "a_scope", "a2b_rev_t", "b2t", "a2a_t", "npad",
                  "atom2mol"):
            assert np.array_equal(b._host[k], b2._host[k]), k


def test_backward_tables_are_the_transposes():
    qb = synth.make_queries(5, 3, [3, 4, 2], atoms_lo=4, atoms_hi=9)
    b = featurization.BatchMolGraph(qb.p_specs, K=6)          # wider pad than needed (global-K case)
    h = b._host
    nA, nB, K = h["nA"], h["nB"], h["K"]
    assert K == 6
    rng = np.random.default_rng(0)
    msg = rng.standard_normal((nB, 5))
    # forward gather-sum over a2b (pad -> row 0), adjoint via tables
    a_msg = msg[h["a2b"]].sum(1)
    d_a = rng.standard_normal((nA, 5))
    want = np.zeros_like(msg)
    np.add.at(want, h["a2b"].reshape(-1), np.repeat(d_a, K, axis=0))
    got = np.where(h["b2t"][:, None] >= 0, d_a[np.maximum(h["b2t"], 0)], 0.0)
    got[0] += (h["npad"][:, None] * d_a).sum(0)
    assert np.allclose(got, want)
    # bond message m_in[b] = a_msg[b2a[b]] - msg[b2revb[b]] ; adjoint wrt a_msg via a2b_rev_t
    d_min = rng.standard_normal((nB, 5))
    want_a = np.zeros((nA, 5))
    np.add.at(want_a, h["b2a"], d_min)
    t = h["a2b_rev_t"]
    got_a = np.where(t[..., None] >= 0, d_min[np.maximum(t, 0)], 0.0).sum(1)
    assert np.allclose(got_a, want_a)
    want_m = np.zeros((nB, 5))
    np.add.at(want_m, h["b2revb"], -d_min)
    assert np.allclose(-d_min[h["b2revb"]], want_m)
    # a2a neighbour sums: adjoint via a2a_t + pad row
    x = rng.standard_normal((nA, 5))
    want_x = np.zeros_like(x)
    np.add.at(want_x, h["a2a"].reshape(-1), np.repeat(d_a, K, axis=0))
    t = h["a2a_t"]
    got_x = np.where(t[..., None] >= 0, d_a[np.maximum(t, 0)], 0.0).sum(1)
    got_x[0] += (h["npad"][:, None] * d_a).sum(0)
    assert np.allclose(got_x, want_x)
    assert h["atom2mol"][0] == -1 and (np.bincount(h["atom2mol"][1:]) == h["a_scope"][:, 1]).all()
    assert a_msg.shape == (nA, 5)


@pytest.mark.parametrize("K", [None, 6])
def test_bond_to_bond_table_is_the_adjoint_of_the_bond_message(K):
    """b2b_t / npad_b (rr_derive_bond_tables): d message = adjoint of
    m_in[b] = (sum_k msg[a2b[b2a[b], k]]) - msg[b2revb[b]]  (models/mpn.py:89-92), pad row included."""
    qb = synth.make_queries(11, 4, [3, 1, 4, 2], atoms_lo=4, atoms_hi=10)
    b = featurization.BatchMolGraph(qb.p_specs, K=K)
    h = b._host
    nA, nB, Kk = h["nA"], h["nB"], h["K"]
    assert h["b2b_t"].shape == (nB, max(1, Kk - 1)) and h["npad_b"].shape == (nB,)
    rng = np.random.default_rng(1)
    d_min = rng.standard_normal((nB, 7))
    # brute-force adjoint by scattering through the forward's own index arrays
    want = np.zeros((nB, 7))
    d_amsg = np.zeros((nA, 7))
    np.add.at(d_amsg, h["b2a"], d_min)                                  # a_msg[b2a[b]] term
    np.add.at(want, h["a2b"].reshape(-1), np.repeat(d_amsg, Kk, axis=0))  # a_msg[a] = sum_k msg[a2b[a,k]]
    np.add.at(want, h["b2revb"], -d_min)                                # - msg[b2revb[b]]
    t = h["b2b_t"]
    got = np.where(t[..., None] >= 0, d_min[np.maximum(t, 0)], 0.0).sum(1)
    got[0] += (h["npad_b"][:, None] * d_min).sum(0)
    assert np.allclose(got, want)
    assert (t[0] == -1).all() and h["npad_b"][0] == Kk - 1


def test_pack_rejects_too_small_k_and_handles_empty():
    qb = synth.make_queries(1, 1, [2])
    with pytest.raises(RuntimeError):
        featurization.BatchMolGraph(qb.p_specs, K=1)
    b = featurization.BatchMolGraph([])
    assert b.n_atoms == 1 and b.n_bonds == 1 and b.max_num_bonds == 1 and b.a_scope == []
    # molecule without bonds: K = max(1, 0)
    lone = synth.MolSpec(1, np.zeros((1, 61), np.float32), np.zeros((0, 2), np.int32), np.zeros((0, 22), np.float32))
    b = featurization.BatchMolGraph([lone])
    assert b.max_num_bonds == 1 and b.n_atoms == 2 and b.n_bonds == 1


def test_dropout_stream_host_matches_numpy_restatement():
    L = _lib.lib()
    rng = np.random.default_rng(3)
    for p in (0.0, 0.1, 0.5, 0.9):
        for seed in (0, 1, 0xDEADBEEFCAFEF00D):
            idx = np.concatenate([np.arange(64), rng.integers(0, 2 ** 40, size=64)]).astype(np.uint64)
            ref = dropout_ref.keep_mask(seed, idx, p)
            got = np.array([L.rr_dropout_keep_host(seed, int(i), p) for i in idx], bool)
            assert np.array_equal(ref, got)
    big = dropout_ref.keep_mask(7, np.arange(200000, dtype=np.uint64), 0.1)
    assert abs(big.mean() - 0.9) < 5e-3


def test_dropout_stream_statistics():
    """The two-level stream (one hash per aligned group of 4 elements + one multiply per element) must still look
    like independent Bernoulli draws: keep rate per lane, no correlation inside a group, across groups or seeds."""
    n = 1 << 20
    idx = np.arange(n, dtype=np.uint64)
    for p in (0.1, 0.2, 0.5):
        k = dropout_ref.keep_mask(12345, idx, p).astype(np.float64)
        sd = (p * (1 - p) / (n / 4)) ** 0.5
        for lane in range(4):                                          # each lane of the group on its own
            assert abs(k[lane::4].mean() - (1 - p)) < 5 * sd
        g = k.reshape(-1, 4) - (1 - p)
        for a in range(4):
            for b in range(a + 1, 4):                                  # lanes of one group are uncorrelated
                corr = (g[:, a] * g[:, b]).mean() / (p * (1 - p))
                assert abs(corr) < 5 / (n / 4) ** 0.5
        corr = ((k[:-4] - (1 - p)) * (k[4:] - (1 - p))).mean() / (p * (1 - p))   # neighbouring groups
        assert abs(corr) < 5 / n ** 0.5
        k2 = dropout_ref.keep_mask(12346, idx, p).astype(np.float64)             # neighbouring seeds
        corr = ((k - (1 - p)) * (k2 - (1 - p))).mean() / (p * (1 - p))
        assert abs(corr) < 5 / n ** 0.5


def test_module_surface_matches_reference_state_dict():
    cfgs = [dict(hidden_size=300, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, task_num=1,
                 add_features_dim=1),
            dict(hidden_size=32, mpnn_depth=1, mpnn_diff_depth=0, ffn_depth=1, use_bias=False, task_num=2,
                 add_features_dim=0)]
    for c in cfgs:
        m = build_model(dropout=0.1, ffn_last_layer="with_softplus", **c)
        want = O.model_shapes(c["hidden_size"], c["mpnn_depth"], c["mpnn_diff_depth"], c["ffn_depth"], c["task_num"],
                              c["add_features_dim"], c["use_bias"])
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert got == want
    m = build_model(hidden_size=300, task_num=1, add_features_dim=1, ffn_last_layer="with_softplus")
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 791101


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from reactranker_amd.loss import MLEloss
    with pytest.raises(RuntimeError):
        MLEloss()(torch.zeros(3), [3], torch.zeros(3), None)


def test_packed_format_roundtrip(tmp_path):
    qb = synth.make_queries(1, 2, [3, 4])
    b = featurization.BatchMolGraph(qb.p_specs, K=4)
    path = str(tmp_path / "batch.npz")
    featurization.save_packed(b, path)
    c = featurization.load_packed(path)
    for k in featurization._PACK_KEYS:
        assert np.array_equal(b._host[k], c._host[k]), k
    assert c.a_scope == b.a_scope and c.b_scope == b.b_scope and c.max_num_bonds == 4 and c.n_mols == b.n_mols
    assert torch.equal(c.get_components()[0], b.get_components()[0])


def test_unique_maps_are_consistent():
    qb = synth.make_queries(3, 3, [4, 2, 5], atoms_lo=4, atoms_hi=7)
    b = featurization.BatchMolGraph(qb.r_specs, K=4)
    ub, amap, amap_t = b.unique()
    bmap, bmap_t = b.unique_bonds()
    assert ub.n_mols == 3 and ub.max_num_bonds == 4
    assert np.array_equal(b._host["f_atoms"], ub._host["f_atoms"][amap])
    assert np.array_equal(b._host["f_bonds"], ub._host["f_bonds"][bmap])
    assert np.array_equal(amap[b._host["b2a"]], ub._host["b2a"][bmap])
    assert np.array_equal(bmap[b._host["b2revb"]], ub._host["b2revb"][bmap])
    assert np.array_equal(bmap[b._host["a2b"]], ub._host["a2b"][amap])
    for mp, tt in ((amap, amap_t), (bmap, bmap_t)):
        for u in range(tt.shape[0]):
            rows = tt[u][tt[u] >= 0]
            assert np.array_equal(np.sort(rows), np.flatnonzero(mp == u))


def test_train_utils_match_reference_scheduler(golden_dir):
    """NoamLR / build_optimizer / build_lr_scheduler against the learning rates the reference classes produced."""
    import json
    from reactranker_amd import train_utils as TU
    z = np.load(os.path.join(golden_dir, "train_utils.npz"))
    i = 0
    while f"c{i}.cfg" in z:
        c = json.loads(str(z[f"c{i}.cfg"]))
        net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 1))
        opt = TU.build_optimizer(net)
        g = opt.param_groups[0]
        assert g["weight_decay"] == 0 and g["betas"] == (0.9, 0.999) and g["eps"] == 1e-8
        sch = TU.build_lr_scheduler(opt, **c)
        assert sch.warmup_steps == int(z[f"c{i}.warmup_steps"]) and sch.total_steps == int(z[f"c{i}.total_steps"])
        want = z[f"c{i}.lrs"]
        got = [opt.param_groups[0]["lr"]]
        for _ in range(len(want) - 1):
            sch.step()
            got.append(opt.param_groups[0]["lr"])
        assert np.allclose(got, want, rtol=1e-12, atol=0), (i, np.abs(np.asarray(got) - want).max())
        sch.step(current_step=7)
        assert np.isclose(opt.param_groups[0]["lr"], float(z[f"c{i}.jump7"]), rtol=1e-12)
        assert TU.param_count(net) == int(z[f"c{i}.param_count"])
        i += 1
    assert i == 3


def test_target_standardisation_matches_the_reference_trainers(golden_dir):
    """tests/golden/standardize.npz holds what the reference's own train() (train_listwise.py:66-122) and run_train()
    (run_train_pairwise.py:36-45) wrote into their DataFrames (tools/make_golden.py:gen_standardize)."""
    import json
    from reactranker_amd.run_train_pairwise import standardize_pairwise
    from reactranker_amd.train_listwise import standardize_targets
    d = np.load(f"{golden_dir}/standardize.npz")
    seen = set()
    for i in range(int(d["n_listwise"])):
        c = json.loads(str(d[f"l{i}.cfg"]))
        tr, va, mean, std = standardize_targets(d["train_raw"], d["val_raw"], c["target_name"], c["normalize_target"],
                                                c["save_metric"])
        assert np.allclose(tr, d[f"l{i}.train"], rtol=1e-12, atol=1e-12), c
        assert np.allclose(va, d[f"l{i}.val"], rtol=1e-12, atol=1e-12), c
        assert abs(mean - float(d["mean"])) < 1e-12 and abs(std - float(d["std"])) < 1e-12
        seen.add((c["target_name"], str(c["normalize_target"])))
    assert len(seen) == 12                                            # 3 target kinds x 4 normalisation modes
    for j in range(2):
        tr, va, _, _ = standardize_pairwise(d["train_raw"], d["val_raw"], str(d[f"p{j}.target_name"]))
        assert np.allclose(tr, d[f"p{j}.train"], rtol=1e-12) and np.allclose(va, d[f"p{j}.val"], rtol=1e-12)


def test_packer_under_address_sanitizer():
    """no GPU sanitizer exists on this pool, so the HOST half of the C-ABI (the graph packer) is
    built with g++ -fsanitize=address,undefined (`make asan`) and driven on ragged / empty / wide-K / corrupt batches
    in a child process that preloads the sanitizer runtime."""
    import shutil
    import subprocess
    import sys
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip("no libasan runtime")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "reactranker_amd", "csrc"), "asan"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "asan_pack_check.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "no report" in r.stdout


def test_smiles_featuriser_layout_matches_the_reference_molgraph(golden_dir):
    """reactranker_amd.rdkit_features against what the REFERENCE's MolGraph (featurization.py:135-210) produced for the
    same molecule descriptions, both served through tests/fake_rdkit.py (tools/make_golden.py gen_featurizer): every
    one-hot block with its unknown slot, atom order by (repeated / zero) map number, bond numbering, f_bonds = atom ||
    bond features, and the same arrays after the native packer."""
    import json
    from tests import fake_rdkit
    from reactranker_amd import rdkit_features as RF
    g = np.load(os.path.join(golden_dir, "featurizer.npz"))
    descs = json.loads(str(g["descriptions"]))
    assert descs == json.loads(json.dumps(fake_rdkit.descriptions())), "fixture and tests/fake_rdkit.py disagree: regenerate"
    chem = fake_rdkit.chem_namespace(descs)
    specs = []
    for name in descs:
        for reaction, tag in ((True, "rxn"), (False, "plain")):
            spec = RF.spec_from_smiles(name, reaction, chem=chem)
            k = f"{name}.{tag}"
            assert spec.n_atoms == g[k + ".f_atoms"].shape[0] and spec.n_bonds == g[k + ".b2a"].shape[0]
            np.testing.assert_array_equal(spec.f_atoms, g[k + ".f_atoms"])
            f_bonds, b2a, b2revb, a2b = spec.directed()
            np.testing.assert_array_equal(f_bonds, g[k + ".f_bonds"])
            np.testing.assert_array_equal(b2a, g[k + ".b2a"])
            np.testing.assert_array_equal(b2revb, g[k + ".b2revb"])
            for i, row in enumerate(a2b):
                want = g[k + ".a2b"][i]
                assert row == [int(v) for v in want[want >= 0]]
            if reaction:
                specs.append(spec)
    # ... and through MolGraph / the native packer: the batch's feature rows are the molecules' rows in order
    mg = [featurization.MolGraph.from_spec(s) for s in specs]
    bg = featurization.BatchMolGraph(mg)
    fa = bg.f_atoms.numpy()
    off = 1
    for s in specs:
        np.testing.assert_array_equal(fa[off:off + s.n_atoms], s.f_atoms)
        off += s.n_atoms
    assert off == bg.n_atoms
    with pytest.raises(ValueError):
        RF.spec_from_smiles("not in the table", True, chem=chem)


def test_molgraph_from_smiles_says_what_is_missing_without_rdkit():
    from reactranker_amd import rdkit_features as RF
    if RF.available():
        pytest.skip("RDKit is installed here")
    with pytest.raises(RuntimeError, match="RDKit"):
        featurization.MolGraph("CCO")
    with pytest.raises(RuntimeError, match="RDKit"):
        featurization.mol2graph(["CCO", "CC"])


def test_round4_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument checks run before any launch, so they can be exercised on the CPU: the ABI-revision-6 entry points return
    RR_ERR_ARG (-1) for null pointers / out-of-range parameters instead of faulting."""
    l = _lib.lib()
    assert l.rr_adam_step_f32(None, 1, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, None) == -1          # null descriptor array
    arr = (_lib.AdamTensor * 1)()
    p = ctypes.cast(arr, ctypes.c_void_p)
    assert l.rr_adam_step_f32(p, 1, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, None) == -1             # step counts from 1
    assert l.rr_adam_step_f32(p, 1, 1, 1e-3, 1.0, 0.999, 1e-8, 0.0, None) == -1             # beta1 < 1
    assert l.rr_adam_step_f32(p, _lib.RR_MAX_ADAM + 1, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, None) == -1
    assert l.rr_adam_step_f32(p, 1, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, None) == 0              # a tensor without gradient: nothing to do
    # rr_ranking_metrics_f32: ratio and cut are fractions
    one = ctypes.c_void_p(16)
    assert l.rr_ranking_metrics_f32(one, 1, one, one, 1, 4, 1.5, 0.5, one, one, None) == -1
    assert l.rr_ranking_metrics_f32(one, 1, one, one, 1, 4, 0.25, -0.1, one, one, None) == -1
    assert l.rr_ranking_metrics_f32(one, 1, one, one, 0, 4, 0.25, 0.5, one, one, None) == 0   # no queries: nothing launched
    assert l.rr_ranking_metrics_f32(one, 1, one, one, 1, 8193, 0.25, 0.5, one, one, None) == -4   # RR_ERR_UNSUPPORTED: list too long
    # rr_reaction_saved_f32: needs model / step / output slots
    assert l.rr_reaction_saved_f32(None, None, 0, 0, 0, None, None, None) == -1
    assert l.rr_comm_backend() in (0, 1, 2)


def test_revision7_entry_points_reject_bad_arguments_without_a_gpu():
    """The two-f16-term GEMMs' entry points check their arguments before any launch (ABI revision 7): a w_packed = 3 GEMM
    without its operand bounds, magnitude outputs asked of a kernel that has none, a null magnitude slot."""
    l = _lib.lib()
    one = ctypes.cast(ctypes.c_void_p(256), _lib.c_f32p)
    assert l.rr_amax_f32(None, 4, 4, 4, one, None) == -1 and l.rr_amax_f32(one, 4, 4, 4, None, None) == -1
    assert l.rr_amax_f32(one, 4, 8, 4, one, None) == -1                     # ld < cols
    assert l.rr_amax_f32(one, 0, 4, 4, one, None) == 0                      # no rows: nothing launched
    A = _lib.LinearArgs()
    A.M, A.N, A.k1, A.lda1, A.ldc, A.ldw = 64, 32, 32, 32, 32, 0
    A.a1, A.w, A.c = one, one, one
    A.mask_scale = 1.0
    A.w_packed = 3
    assert l.rr_linear_f32(ctypes.byref(A), None) == -1                     # two f16 terms need a1_amax
    A.w_packed = 2
    A.c_amax_out = one
    assert l.rr_linear_f32(ctypes.byref(A), None) == -1                     # magnitude outputs: w_packed = 3 only
    A.c_amax_out = None
    A.dz_amax_out, A.w_packed = one, 3
    A.a1_amax = one
    assert l.rr_linear_f32(ctypes.byref(A), None) == -1                     # dz_amax_out without dz_out
    W = _lib.WgradArgs()
    W.M, W.N, W.k1, W.ld_dy, W.ldx1, W.ld_dw = 64, 32, 32, 32, 32, 32
    W.dy, W.x1, W.dw, W.workspace, W.workspace_bytes = one, one, one, 256, 1 << 30
    W.mask_scale, W.split = 1.0, 2
    assert l.rr_linear_wgrad_f32(ctypes.byref(W), None) == -1               # split = 2 needs dy_amax / x1_amax
    W.split = 3
    assert l.rr_linear_wgrad_f32(ctypes.byref(W), None) == -1
    assert _lib.RR_AMAX_FLOATS == 512 and _lib.RR_PLAN_F16X2_GEMM == 32
